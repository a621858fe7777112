cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_relax; mkdir -p $O
for r in 1.6 1.7 1.8 1.85 1.9 1.95; do
  timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-extras --relax $r > $O/r$r.json 2> $O/r$r.err
done
for c in 3 4 6; do
  timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-extras --check $c > $O/c$c.json 2> $O/c$c.err
done
