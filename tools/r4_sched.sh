cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_sched; mkdir -p $O
for s in 1.0 0.7 0.5 0.35; do
  timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extras --horizon 20 --rho-sched $s > $O/h20_$s.json 2> $O/h20_$s.err
  timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extras --horizon 20 --random-schedule --rho-sched $s > $O/c5_$s.json 2> $O/c5_$s.err
done
timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extras --horizon 20 --tol 1e-6 > $O/h20_tol6.json 2> $O/h20_tol6.err
timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extras --horizon 20 --random-schedule --tol 1e-6 > $O/c5_tol6.json 2> $O/c5_tol6.err
