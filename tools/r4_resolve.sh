cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_resolve; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=line -k "resolve or collinear or lookahead or unbalanced or audit or persistently or strict or exact_solver" 2>&1 | grep -E "Error|assert|^/|FAILED|passed|failed" | cut -c1-400 | tail -30 > $O/tests.log
for tag in "headline:" "kin1:--kin-mode 1" "kin0chain:--chain-geometry" "s1:--solver 1"; do
  name=${tag%%:*}; args=${tag#*:}
  timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras $args > $O/$name.json 2> $O/$name.err
done
