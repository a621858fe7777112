#!/bin/bash
# Same-box A/B of the ADMM warm start (previous-tick iterate) against a cold start every tick: 200 timed ticks per run,
# two repetitions, at the headline batch, batch 32768 and horizon 20.  -> gpurun_out/r3_warm_ab.txt
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r3_warm_ab.txt
echo "# kernel sources $(python3 -c 'import bench; print(bench.source_hash())'); bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras --no-kernel-events [--cold-start]" > $OUT
run() { name=$1; shift
  for rep in 1 2; do for mode in warm cold; do
    extra=""; [ $mode = cold ] && extra="--cold-start"
    timeout 600 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras --no-kernel-events "$@" $extra 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$name $mode rep$rep: %.3f M steps/s  %.4f ms/tick  mean iterations %.1f  max %d  exact re-solves %d  audit max_rel %.2e over_tol %d' % (d['value']/1e6, d['ms_per_step'], d['config']['admm_iterations']['iters_mean'], d['config']['admm_iterations']['iters_max'], d['config']['admm_iterations']['retried_exact'], d['config']['audit']['audit_max_rel'], d['config']['audit']['audit_over_tol']))" >> $OUT
  done; done; }
run "batch 4096 H=10"
run "batch 32768 H=10" --batch 32768
run "batch 4096 H=20" --horizon 20
run "batch 1024 H=10 fixed cmd" --batch 1024 --fixed-cmd
cat $OUT
