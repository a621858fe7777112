#!/bin/bash
# Runs on the GPU box (tools/stamp_commit.py first, HERE): the short form of the round's evidence -- the GPU suite, the bench
# lines of the headline / config 2 / batch 1 with their rocprofv3 kernel stats (no counter passes), the launch schedule of the
# headline (make stamp build).  -> gpurun_out/ ; then tools/summarize_profiles.py per tag and cp into profiles/.
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
HDR=$(python3 tools/evidence_guard.py) || { echo "$HDR"; exit 1; }
{ echo "# $HDR; python -m pytest tests -m gpu -q"; timeout 1500 python3 -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -6; } > gpurun_out/r6_gpu_tests.log
run() {   # key mode args...
  local key=$1 mode=$2; shift; shift
  local tag=r6; [ "$key" != "headline" ] && tag=r6_$key
  python3 bench.py --steps 20 --warmup 5 "$@" > gpurun_out/bench_$key.json 2> gpurun_out/bench_$key.err
  tools/collect_profiles.sh $tag $mode "$@" > gpurun_out/collect_$key.log 2>&1
  echo "$key: $(python3 -c "import json;d=json.loads(open('gpurun_out/bench_$key.json').read());c=d['config'];print(round(d['value']/1e6,3),'M steps/s', round(d['ms_per_step'],4),'ms', d['roofline']['kernel_ms'], 'sustained', c.get('sustained'), 'dropin', c.get('dropin_get_action_latency_us'))")"
}
run headline ${1:-stats}
run config2 stats --batch 1024 --fixed-cmd
run b1 stats --batch 1
{ echo "# $HDR; tests/studies/fused_launch_schedule.py (make stamp build)"
  python3 tests/studies/fused_launch_schedule.py 4096 2>&1 | grep -v amdgpu.ids | cut -c1-600; } > gpurun_out/r6_launch_schedule.txt
tail -3 gpurun_out/r6_gpu_tests.log
