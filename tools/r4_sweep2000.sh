#!/bin/bash
# Runs on the GPU box: the seeded configuration sweep over 2000 seeds (per-joint 1e-4 assertion in every case).
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
HASH=$(python3 -c "import bench; print(bench.source_hash())")
{ echo "# kernel sources $HASH; RG_SWEEP_SEEDS=2000 python -m pytest tests/test_gpu_parity.py -k randomised_configurations -q"
  RG_SWEEP_SEEDS=2000 timeout 3000 python3 -m pytest tests/test_gpu_parity.py -k randomised_configurations -q -p no:cacheprovider 2>&1 | tail -4; } > gpurun_out/r4_sweep2000.txt
cat gpurun_out/r4_sweep2000.txt
