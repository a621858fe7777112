#!/bin/bash
# Runs on the GPU box: the parity evidence of round 4 on the final kernel sources -- the 500-seed configuration sweep (with the
# per-joint 1e-4 assertion), the long-run error study, the exact re-solve guard over 12 seeds with and without LDS poisoning,
# the launch-order soak -- into gpurun_out/r4_evidence/, stamped with the kernel source hash.
# Locally afterwards: cp gpurun_out/r4_evidence/* profiles/
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r4_evidence
mkdir -p $OUT
HASH=$(python3 -c "import bench; print(bench.source_hash())")
{ echo "# kernel sources $HASH; RG_SWEEP_SEEDS=500 python -m pytest tests/test_gpu_parity.py -k randomised_configurations -q";
  RG_SWEEP_SEEDS=500 RG_SWEEP_VERBOSE=1 timeout 3000 python3 -m pytest tests/test_gpu_parity.py -k randomised_configurations -q -s -p no:cacheprovider 2>&1 | grep -v "^$" | tail -520; } > $OUT/r4_sweep500.log
{ echo "# kernel sources $HASH; python tests/studies/worst_errors.py all"; timeout 2400 python3 tests/studies/worst_errors.py all 2>&1 | grep -v amdgpu.ids; } > $OUT/r4_worst_errors.txt
{ echo "# kernel sources $HASH"
  echo "## RG_GUARD_SEEDS=12 pytest -k exact_resolve_kernel"
  RG_GUARD_SEEDS=12 timeout 1500 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k exact_resolve_kernel -p no:cacheprovider 2>&1 | tail -3
  echo "## tests/studies/launch_order_soak.py 120 2"
  timeout 1500 python3 tests/studies/launch_order_soak.py 120 2 2>&1 | grep -v amdgpu.ids | tail -6
} > $OUT/r4_soak.txt 2>&1
tail -3 $OUT/r4_sweep500.log; cat $OUT/r4_worst_errors.txt; cat $OUT/r4_soak.txt
