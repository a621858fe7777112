#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
HDR=$(python3 tools/evidence_guard.py) || { echo "$HDR"; exit 1; }
mkdir -p gpurun_out/r5_final
for g in 1 2; do
  { echo "# $HDR; python -m pytest tests -m gpu -q --lane-grid $g"; timeout 1500 python3 -m pytest tests -m gpu -q -p no:cacheprovider --lane-grid $g 2>&1 | tail -6; } > gpurun_out/r5_final/r5_gpu_tests_grid$g.log
done
tail -2 gpurun_out/r5_final/*.log
bash tools/r5_sweep_n.sh 2000 6000 | tail -3
