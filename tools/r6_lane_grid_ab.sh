#!/bin/bash
# Runs on the GPU box: A/B of the two lane grids of the default plan at horizon 10 (rg_mpc_config.lane_grid) over batch sizes.
# -> gpurun_out/r6_lane_grid_ab.txt
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r6_lane_grid_ab.txt
mkdir -p gpurun_out
{ python3 tools/evidence_guard.py || true
  for B in ${AB_BATCHES:-1 16 64 256 512 768 1024 1536}; do
    for extra in "" "--fixed-cmd"; do
      for g in 1 2; do
        timeout 600 python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-extras --batch $B --lane-grid $g $extra 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch %5d %-11s lane_grid $g  %8.3f M steps/s  tick %.1f us  %s' % ($B, '$extra', d['value']/1e6, d['ms_per_step']*1e3, d['roofline']['kernel_ms']))"
      done
    done
  done
} > $OUT 2>&1
cat $OUT
