cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_accel; mkdir -p $O
for a in 0 40 50 60 70 80 100; do
  timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-extras --accel $a > $O/a$a.json 2> $O/a$a.err
done
for s in 0.4 0.6 0.7; do
  timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-extras --rho34 $s > $O/s$s.json 2> $O/s$s.err
done
timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-extras --audit-k 0 > $O/noaudit.json 2> $O/noaudit.err
timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-extras --switch 100 > $O/sw100.json 2> $O/sw100.err
