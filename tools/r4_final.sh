#!/bin/bash
# Runs on the GPU box after profiles/ holds the summaries of the final kernel sources: the bench lines carrying the
# profile-derived fields, the exact-everywhere plan's profiles, the plan table, the parity evidence, the per-robot launch
# schedules (make stamp build), and the whole GPU test suite.
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
TAG=r4 bash tools/run_bench_lines.sh > gpurun_out/r4_bench_lines.log 2>&1
tools/collect_profiles.sh r4_s1 stats --solver 1 > gpurun_out/collect_s1.log 2>&1
tools/collect_profiles.sh r4_config2_s1 stats --solver 1 --batch 1024 --fixed-cmd > gpurun_out/collect_config2_s1.log 2>&1
bash tools/r4_plan_table.sh > gpurun_out/r4_plan_table.log 2>&1
{ echo "# tests/studies/fused_launch_schedule.py (make stamp build), kernel sources $(python3 -c 'import bench; print(bench.source_hash())')"
  python3 tests/studies/fused_launch_schedule.py 4096 2>&1 | grep -v amdgpu.ids
  python3 tests/studies/fused_launch_schedule.py 1024 '{"fixed_cmd": true}' 2>&1 | grep -v amdgpu.ids
  python3 tests/studies/fused_launch_schedule.py 1 2>&1 | grep -v amdgpu.ids; } > gpurun_out/r4_launch_schedule.txt
python3 -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/r4_gpu_tests.log 2>&1
bash tools/r4_evidence.sh > gpurun_out/r4_evidence.log 2>&1
tail -3 gpurun_out/r4_gpu_tests.log; cat gpurun_out/r4_bench_lines.log | head -12; cat gpurun_out/r4_plan_table.md
