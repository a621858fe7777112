#!/bin/bash
# round-3 first GPU pass: GPU test suite, then A/B of the audit lane's cost on the headline workload
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > gpurun_out/r3_tests.log
for rep in 1 2; do
  for k in 8 0; do
    timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extras --no-kernel-events --audit-k $k > gpurun_out/r3_ab_audit${k}_$rep.json 2> gpurun_out/r3_ab_audit${k}_$rep.err
  done
done
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3_bench_default.json 2> gpurun_out/r3_bench_default.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3_ab_audit*.json"))+["gpurun_out/r3_bench_default.json"]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d["value"]/1e6,3), "M", round(d["ms_per_step"],4), d["config"]["audit"], d["roofline"]["kernel_ms"])
    except Exception as e:
        print(f, "ERR", e)
PY
tail -15 gpurun_out/r3_tests.log
