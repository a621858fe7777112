#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
n=0
for i in $(seq 1 ${1:-40}); do
  out=$(timeout 120 python tests/studies/order_dependence_hunt.py 1 0 1 2 3 4 5 6 7 8 2>&1 | grep -A5 "PARITY MISS" | cut -c1-1500)
  if [ -n "$out" ]; then echo "$out"; n=$((n+1)); fi
done
echo "fresh processes with a parity miss: $n of ${1:-40}"
