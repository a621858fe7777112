#!/bin/bash
# kernel timeline of a short bench run (both streams): gpurun_out/r3_trace/
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3_trace
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 $ROOT/bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-extras --no-kernel-events "$@" > $OUT/log.txt 2>&1
f=$(find $OUT/t -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
t0=None
sel=[r for r in rows if "rg_" in r["Kernel_Name"]]
# last 40 kernels
base=int(sel[-60]["Start_Timestamp"])
for r in sel[-60:]:
    n=r["Kernel_Name"].split("(")[0][:40]
    print(f'{(int(r["Start_Timestamp"])-base)/1e3:9.1f} {(int(r["End_Timestamp"])-base)/1e3:9.1f} dur {(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:7.1f} q{r.get("Queue_Id","?")} grid {r.get("Grid_Size","?")} {n}')
PY
