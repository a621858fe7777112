#!/bin/bash
# Runs on the GPU box: the three solver plans side by side -- 2 (ADMM bodies + exact re-solve), 3 (hybrid, the default) and
# 1 (an exact body for every robot) -- at batch 1024 (fixed command), 4096 and 32768: steps/s over 200 ticks, and each plan's
# worst torque error against the oracle on the trot study (4096 robots x 50 ticks).  -> gpurun_out/r4_plan_table.md
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r4_plan_table.md
HASH=$(python3 -c "import bench; print(bench.source_hash())")
{ echo "kernel sources $HASH; tools/r4_plan_table.sh"; echo
  echo "| plan | batch 1024 fixed command | batch 4096 (headline) | batch 32768 | worst error per robot / per joint (trot study, 205 k robot-ticks) |"
  echo "|---|---|---|---|---|"
  for s in 2 3 1; do
    row="| $s |"
    for args in "--batch 1024 --fixed-cmd" "" "--batch 32768"; do
      v=$(timeout 600 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras --no-kernel-events --solver $s $args 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f M' % (d['value']/1e6))")
      row="$row $v |"
    done
    e=$(SOLVER=$s timeout 1200 python3 tests/studies/worst_errors.py trot 2>/dev/null | grep "^trot " | python3 -c "import sys,re; l=sys.stdin.read(); m=re.search(r'max (\S+)\s+per-joint max (\S+)', l); print(m.group(1), '/', m.group(2))")
    echo "$row $e |"
  done
} > $OUT
cat $OUT
