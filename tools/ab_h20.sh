#!/bin/bash
# Runs on the GPU box: A/B of two builds of the library (RG_MPC_LIB) on the workloads that run the schedule body (horizon 20,
# contact schedules), alternating, 3 repetitions each.  Usage: tools/ab_h20.sh <lib A> <lib B>  -> gpurun_out/ab_h20.txt
cd ${GRAFT_REPO_ROOT:-$(pwd)}
A=$1; B=$2; OUT=gpurun_out/ab_h20.txt
mkdir -p gpurun_out
{ for args in "--horizon 20" "--horizon 20 --random-schedule" "--random-schedule" "--horizon 20 --batch 1024"; do
    for rep in 1 2 3; do
      for lib in $A $B; do
        RG_MPC_LIB=$PWD/$lib timeout 600 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras $args 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-34s %-40s %8.3f M  tick %.1f us  %s' % ('$args', '$lib', d['value']/1e6, d['ms_per_step']*1e3, d['roofline']['kernel_ms']))"
      done
    done
  done
} > $OUT 2>&1
cat $OUT
