#!/bin/bash
# Runs on the GPU box (tools/stamp_commit.py first, HERE): the seeded configuration sweep over N seeds, seeds OFFSET.. (per-joint
# 1e-4 assertion, lane grid drawn per seed).  -> gpurun_out/r6_sweep<N>_from<OFFSET>.txt
cd ${GRAFT_REPO_ROOT:-$(pwd)}
N=${1:-4000}; OFF=${2:-0}
mkdir -p gpurun_out
HDR=$(python3 tools/evidence_guard.py) || { echo "$HDR"; exit 1; }
{ echo "# $HDR; RG_SWEEP_SEEDS=$N RG_SWEEP_OFFSET=$OFF python -m pytest tests/test_gpu_parity.py -k randomised_configurations -q"
  RG_SWEEP_SEEDS=$N RG_SWEEP_OFFSET=$OFF timeout 3400 python3 -m pytest tests/test_gpu_parity.py -k randomised_configurations -q -p no:cacheprovider 2>&1 | tail -4; } > gpurun_out/r6_sweep${N}_from${OFF}.txt
cat gpurun_out/r6_sweep${N}_from${OFF}.txt
