#!/bin/bash
# Runs on the GPU box: the GPU suite, then A/B of robot_gym_amd/csrc/librg_mpc_old.so against librg_mpc.so (RG_MPC_LIB) on the
# main workloads, alternating, 2 repetitions each.  -> gpurun_out/ab_front.txt
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
{ timeout 1200 python3 -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider ${AB_K:+-k "$AB_K"} 2>&1 | tail -5
  for args in "" "--batch 1" "--batch 1024 --fixed-cmd" "--horizon 20 --random-schedule"; do
    for rep in 1 2; do
      for lib in robot_gym_amd/csrc/librg_mpc_old.so robot_gym_amd/csrc/librg_mpc.so; do
        RG_MPC_LIB=$PWD/$lib timeout 600 python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-extras $args 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-32s %-44s %8.3f M  tick %.1f us  %s' % ('$args', '$lib', d['value']/1e6, d['ms_per_step']*1e3, d['roofline']['kernel_ms']))"
      done
    done
  done
} > gpurun_out/ab_front.txt 2>&1
cat gpurun_out/ab_front.txt
