#!/bin/bash
# Runs on the GPU box (via gpurun) AFTER profiles/ holds the summaries of the current kernel sources: the bench line of every
# BASELINE workload, now carrying the profile-derived fields (roofline.traffic, roofline.issue_view).  -> gpurun_out/bench_<key>.json
set -u
TAG=${TAG:-r6}
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
python3 tools/evidence_guard.py || exit 1
run() { local key=$1; shift; python3 bench.py --steps 20 --warmup 5 "$@" > gpurun_out/bench_$key.json 2> gpurun_out/bench_$key.err; echo "$key: $(python3 -c "import json;d=json.loads(open('gpurun_out/bench_$key.json').read());print(round(d['value']/1e6,3),'M steps/s', d['roofline']['kernel_ms'], 'traffic', d['roofline']['traffic'])")"; }
run headline
run config2 --batch 1024 --fixed-cmd
run b1 --batch 1
run h20 --horizon 20
run config5 --horizon 20 --random-schedule
run b32768 --batch 32768
run kin1 --kin-mode 1
run config2_grid1 --batch 1024 --fixed-cmd --lane-grid 1
{ echo "# $(python3 tools/evidence_guard.py); tools/vec_env_bench.py 20"; python3 tools/vec_env_bench.py 20; } > gpurun_out/${TAG}_vec_env_host.txt 2> gpurun_out/${TAG}_vec_env_host.err; cat gpurun_out/${TAG}_vec_env_host.txt
