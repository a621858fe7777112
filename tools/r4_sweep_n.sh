#!/bin/bash
# Runs on the GPU box: the seeded configuration sweep over N seeds (default 4000; per-joint 1e-4 assertion in every case).
# -> gpurun_out/r4_sweep<N>.txt   (N = 4000 takes 46 minutes)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
N=${1:-4000}
mkdir -p gpurun_out
HASH=$(python3 -c "import bench; print(bench.source_hash())")
{ echo "# kernel sources $HASH; RG_SWEEP_SEEDS=$N python -m pytest tests/test_gpu_parity.py -k randomised_configurations -q"
  RG_SWEEP_SEEDS=$N timeout 3300 python3 -m pytest tests/test_gpu_parity.py -k randomised_configurations -q -p no:cacheprovider 2>&1 | tail -4; } > gpurun_out/r4_sweep$N.txt
cat gpurun_out/r4_sweep$N.txt
