#!/bin/bash
# host-array corruption hunt (tests/studies/host_corruption_hunt.py) in fresh processes; usage: r3_flaky5.sh N 'json overrides' ...
cd ${GRAFT_REPO_ROOT:-$(pwd)}
N=${1:-40}; shift
[ $# -eq 0 ] && set -- '{}'
for over in "$@"; do
  n=0
  for i in $(seq 1 $N); do
    out=$(RG_SWEEP_OVER="$over" timeout 120 python tests/studies/host_corruption_hunt.py 2>&1 | grep "CORRUPTION" | cut -c1-300)
    if [ -n "$out" ]; then echo "$out"; n=$((n+1)); fi
  done
  echo "over=$over: fresh processes with a host-array corruption: $n of $N"
done
