#!/bin/bash
# Runs on the GPU box (via gpurun): the bench line of every BASELINE workload (with its cpu_baseline at the same batch and
# the B0 / B1 variants) and the rocprofv3 summaries of the same commands -- kernel stats and the separate PMC passes for
# every workload.  Output: gpurun_out/bench_<key>.json, gpurun_out/prof_<tag>/.
# Then, locally: for t in r3 ${TAG}_config2 ...; do python tools/summarize_profiles.py $t; done; then tools/run_bench_lines.sh
set -u
TAG=${TAG:-r4}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p gpurun_out
run() {   # key mode args...
  local key=$1 mode=$2; shift; shift
  local tag=$TAG; [ "$key" != "headline" ] && tag=${TAG}_$key
  python3 bench.py --steps 20 --warmup 5 "$@" > gpurun_out/bench_$key.json 2> gpurun_out/bench_$key.err
  tools/collect_profiles.sh $tag $mode "$@" > gpurun_out/collect_$key.log 2>&1
  echo "$key: $(python3 -c "import json;d=json.loads(open('gpurun_out/bench_$key.json').read());print(round(d['value']/1e6,3),'M steps/s', round(d['ms_per_step'],4),'ms', d['roofline']['kernel_ms'], 'cpu', d.get('cpu_baseline',{}).get('value'))")"
}
run headline full
run config2 full --batch 1024 --fixed-cmd
run b1 full --batch 1
run h20 full --horizon 20
run config5 full --horizon 20 --random-schedule
run config5_cap600 stats --horizon 20 --random-schedule --cap 600
run b32768 full --batch 32768
run kin1 full --kin-mode 1
