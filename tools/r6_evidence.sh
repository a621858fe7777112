#!/bin/bash
# Runs on the GPU box (tools/stamp_commit.py first, HERE): the parity evidence of round 6 on a COMMIT's kernel sources -- the
# GPU suite on both lane grids, the seeded configuration sweep (N seeds, default 2000; lane grid drawn per seed), the
# long-run error study, the exact re-solve guard with and without LDS poisoning, the launch-order soak, the launch schedules
# (make stamp build).  Every file starts with bench.evidence_header().  -> gpurun_out/r6_evidence/; then cp into profiles/.
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r6_evidence
mkdir -p $OUT
HDR=$(python3 tools/evidence_guard.py) || { echo "$HDR"; exit 1; }
N=${1:-2000}
for g in 1 2; do
  { echo "# $HDR; python -m pytest tests -m gpu -q --lane-grid $g"; timeout 1500 python3 -m pytest tests -m gpu -q -p no:cacheprovider --lane-grid $g 2>&1 | tail -6; } > $OUT/r6_gpu_tests_grid$g.log
done
{ echo "# $HDR; RG_SWEEP_SEEDS=$N python -m pytest tests/test_gpu_parity.py -k randomised_configurations -q"
  RG_SWEEP_SEEDS=$N timeout 3300 python3 -m pytest tests/test_gpu_parity.py -k randomised_configurations -q -p no:cacheprovider 2>&1 | tail -4; } > $OUT/r6_sweep$N.txt
{ echo "# $HDR; python tests/studies/worst_errors.py all"; timeout 2400 python3 tests/studies/worst_errors.py all 2>&1 | grep -v amdgpu.ids; } > $OUT/r6_worst_errors.txt
{ echo "# $HDR"
  echo "## RG_GUARD_SEEDS=12 pytest -k exact_resolve_kernel"
  RG_GUARD_SEEDS=12 timeout 1500 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k exact_resolve_kernel -p no:cacheprovider 2>&1 | tail -3
  echo "## tests/studies/launch_order_soak.py 120 2"
  timeout 1500 python3 tests/studies/launch_order_soak.py 120 2 2>&1 | grep -v amdgpu.ids | tail -6
} > $OUT/r6_soak.txt 2>&1
{ echo "# $HDR; tests/studies/fused_launch_schedule.py (make stamp build)"
  python3 tests/studies/fused_launch_schedule.py 4096 2>&1 | grep -v amdgpu.ids | cut -c1-600
  python3 tests/studies/fused_launch_schedule.py 1024 '{"fixed_cmd": true}' '{"fixed_cmd": true, "lane_grid": 1}' 2>&1 | grep -v amdgpu.ids | cut -c1-600
  python3 tests/studies/fused_launch_schedule.py 1 '{"lane_grid": 2}' '{"lane_grid": 1}' 2>&1 | grep -v amdgpu.ids | cut -c1-600
  python3 tests/studies/fused_launch_schedule.py 4096 '{"horizon": 20}' 2>&1 | grep -v amdgpu.ids | cut -c1-600; } > $OUT/r6_launch_schedule.txt
bash tools/r6_lane_grid_ab.sh > /dev/null 2>&1; cp gpurun_out/r6_lane_grid_ab.txt $OUT/
tail -3 $OUT/r6_gpu_tests_grid1.log $OUT/r6_gpu_tests_grid2.log $OUT/r6_sweep$N.txt; cat $OUT/r6_worst_errors.txt; cat $OUT/r6_soak.txt
