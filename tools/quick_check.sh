#!/bin/bash
# Runs on the GPU box: a quick look at a kernel change -- the exact-solver parity tests, then 200-tick bench values of the
# workloads the change could move (QUICK_SET=h20: the horizon-20 / scheduled-contact workloads; QUICK_K: pytest -k expression).  -> gpurun_out/quick_check.txt
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/quick_check.txt
mkdir -p gpurun_out
{ echo "kernel sources $(python3 -c 'import bench; print(bench.source_hash())')"
  timeout 1500 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x --tb=short -p no:cacheprovider -k "${QUICK_K:-exact or hybrid or overflow or strict or golden}" 2>&1 | tail -5
  if [ "${QUICK_SET:-h10}" = h20 ]; then set -- "--horizon 20" "--horizon 20 --random-schedule" "--horizon 20 --random-schedule --cap 600" "--random-schedule"; else set -- "" "--batch 1024 --fixed-cmd" "--batch 1" "--kin-mode 1" "--batch 32768" "--solver 1"; fi
  for args in "$@" ${QUICK_EXTRA:+"$QUICK_EXTRA"}; do
    for rep in 1 2; do
      timeout 600 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras $args 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-28s %8.3f M  ms/step %.4f  %s  mean work %.1f' % ('$args', d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['admm_iterations']['iters_mean']))"
    done
  done
} > $OUT 2>&1
cat $OUT
