cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_rho34; mkdir -p $O
for s in 1.0 0.8 0.6 0.5 0.4 0.3; do
  timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras --rho34 $s > $O/s$s.json 2> $O/s$s.err
  timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras --rho34 $s --tol 1e-7 > $O/t7s$s.json 2> $O/t7s$s.err
done
for a in 40 60; do
  timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras --rho34 0.5 --accel $a > $O/a$a.json 2> $O/a$a.err
done
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras --rho34 0.5 --batch 1024 --fixed-cmd > $O/c2.json 2> $O/c2.err
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras --rho34 0.5 --audit-k 0 > $O/noaudit.json 2> $O/noaudit.err
