#!/bin/bash
# Runs on the GPU box: GPU test suite, then 200-tick bench values of the main workloads.  -> gpurun_out/r5_quick.txt
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r5_quick.txt
mkdir -p gpurun_out
{ echo "kernel sources $(python3 -c 'import bench; print(bench.source_hash())')"
  if [ "${QUICK_TESTS:-1}" = 1 ]; then timeout 1500 python3 -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider ${QUICK_K:+-k "$QUICK_K"} 2>&1 | tail -15; fi
  for args in "" "--batch 1024 --fixed-cmd" "--batch 1" "--batch 32768" "--horizon 20" "--horizon 20 --random-schedule" ${QUICK_EXTRA:+"$QUICK_EXTRA"}; do
    for rep in 1 2; do
      timeout 600 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras $args 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-32s %8.3f M  ms/step %.4f  %s  mean work %.1f' % ('$args', d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['admm_iterations']['iters_mean']))"
    done
  done
} > $OUT 2>&1
cat $OUT
