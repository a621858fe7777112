#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/ab2
run() { name=$1; shift; timeout 300 python bench.py --no-cpu-baseline --no-extras "$@" > gpurun_out/ab2/$name.json 2> gpurun_out/ab2/$name.err; }
for rep in 1 2; do
run s20_a8_$rep --steps 20 --warmup 5
run s20_a0_$rep --steps 20 --warmup 5 --audit-k 0
run s20_a8_noev_$rep --steps 20 --warmup 5 --no-kernel-events
run s20_a0_noev_$rep --steps 20 --warmup 5 --audit-k 0 --no-kernel-events
run s100_a8_$rep --steps 100 --warmup 20
run s100_a0_$rep --steps 100 --warmup 20 --audit-k 0
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/ab2/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d["value"]/1e6,3), "M", round(d["ms_per_step"],4), d["roofline"]["kernel_ms"], d["config"]["admm_iterations"]["iters_mean"])
    except Exception as e:
        print(f, "ERR", e)
PY
