#!/bin/bash
# Runs on the GPU box (tools/stamp_commit.py first, HERE): the bench line of every BASELINE workload (with cpu_baseline and the
# extras: sustained rate, audit cost, drop-in latency at batch 1) and the rocprofv3 summaries of the same commands.
# -> gpurun_out/bench_<key>.json, gpurun_out/prof_r6*/.  Then locally: for t in r6 r6_config2 ...; do python tools/summarize_profiles.py $t; done
set -u
TAG=r6
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p gpurun_out
python3 tools/evidence_guard.py || exit 1
run() {   # key mode args...
  local key=$1 mode=$2; shift; shift
  local tag=$TAG; [ "$key" != "headline" ] && tag=${TAG}_$key
  python3 bench.py --steps 20 --warmup 5 "$@" > gpurun_out/bench_$key.json 2> gpurun_out/bench_$key.err
  tools/collect_profiles.sh $tag $mode "$@" > gpurun_out/collect_$key.log 2>&1
  echo "$key: $(python3 -c "import json;d=json.loads(open('gpurun_out/bench_$key.json').read());c=d['config'];print(round(d['value']/1e6,3),'M steps/s', round(d['ms_per_step'],4),'ms', d['roofline']['kernel_ms'], 'sustained', c.get('sustained'), 'audit cost %', c.get('audit_cost_pct'), 'dropin', c.get('dropin_get_action_latency_us'), 'cpu', d.get('cpu_baseline',{}).get('value'))")"
}
run headline full
run config2 full --batch 1024 --fixed-cmd
run b1 full --batch 1
run h20 full --horizon 20
run config5 full --horizon 20 --random-schedule
run b32768 stats --batch 32768
run kin1 stats --kin-mode 1
run config2_grid1 stats --batch 1024 --fixed-cmd --lane-grid 1
