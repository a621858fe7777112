#!/bin/bash
# Runs on the GPU box: soak of the shipped binary -- the multi-body exact kernel guard over 12 seeds, the launch-order soak,
# and 1000 further seeds of the configuration sweep (the test's seeds 0..1499; 0..499 are profiles/r3_sweep500.log).
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r3_soak.txt
HASH=$(python3 -c "import bench; print(bench.source_hash())")
{ echo "# kernel sources $HASH"
  echo "## RG_GUARD_SEEDS=12 pytest -k multi_body_exact"
  RG_GUARD_SEEDS=12 timeout 1200 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k multi_body_exact -p no:cacheprovider 2>&1 | tail -3
  echo "## tests/studies/launch_order_soak.py 120 2"
  timeout 1500 python3 tests/studies/launch_order_soak.py 120 2 2>&1 | grep -v amdgpu.ids | tail -6
  echo "## RG_SWEEP_SEEDS=1500 pytest -k randomised_configurations"
  RG_SWEEP_SEEDS=1500 timeout 3000 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k randomised_configurations -p no:cacheprovider 2>&1 | tail -4
} > $OUT 2>&1
cat $OUT
