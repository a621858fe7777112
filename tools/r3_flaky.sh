#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for i in 1 2 3 4 5 6 7 8; do
  RG_SWEEP_SEEDS=${1:-24} timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "randomised_configurations" -p no:cacheprovider $2 > /tmp/flaky_$i.log 2>&1
  if grep -q "failed" /tmp/flaky_$i.log; then echo "RUN $i FAILED"; grep -E "^E  |Error|assert|FAILED" /tmp/flaky_$i.log | cut -c1-900 | head -30; else echo "run $i ok: $(tail -1 /tmp/flaky_$i.log)"; fi
done
