#!/bin/bash
# seeds 0..8 in one process, many repetitions, with and without the audit lane
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for over in '{}' '{"audit_k": 0}'; do
  fails=0
  for i in $(seq 1 ${1:-12}); do
    RG_SWEEP_OVER="$over" RG_SWEEP_SEEDS=9 timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "randomised_configurations" -p no:cacheprovider > /tmp/f.log 2>&1
    if grep -q "failed" /tmp/f.log; then fails=$((fails+1)); grep -E "^FAILED" /tmp/f.log | cut -c1-120; fi
  done
  echo "over=$over: $fails failures of ${1:-12}"
done
