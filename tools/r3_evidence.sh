#!/bin/bash
# Runs on the GPU box: the parity evidence the round-2 review asked to see committed -- the 500-seed configuration sweep,
# the long-run error study and the steps/s-vs-tolerance table -- into gpurun_out/r3_evidence/, stamped with the kernel
# source hash.  Locally afterwards: cp gpurun_out/r3_evidence/* profiles/  (names r3_sweep500.log, r3_worst_errors.txt, r3_tolerance_table.md)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r3_evidence
mkdir -p $OUT
HASH=$(python3 -c "import bench; print(bench.source_hash())")
{ echo "# kernel sources $HASH; RG_SWEEP_SEEDS=500 python -m pytest tests/test_gpu_parity.py -k randomised_configurations -q"; 
  RG_SWEEP_SEEDS=500 RG_SWEEP_VERBOSE=1 timeout 3000 python3 -m pytest tests/test_gpu_parity.py -k randomised_configurations -q -s -p no:cacheprovider 2>&1 | grep -v "^$" | tail -520; } > $OUT/r3_sweep500.log
{ echo "# kernel sources $HASH; python tests/studies/worst_errors.py all"; timeout 2400 python3 tests/studies/worst_errors.py all 2>&1; } > $OUT/r3_worst_errors.txt
{ echo "kernel sources $HASH; python tests/studies/tolerance_table.py"; echo; timeout 2400 python3 tests/studies/tolerance_table.py 2>&1; } > $OUT/r3_tolerance_table.md
tail -3 $OUT/r3_sweep500.log; cat $OUT/r3_worst_errors.txt; cat $OUT/r3_tolerance_table.md
