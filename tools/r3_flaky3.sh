#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
n=0
for i in $(seq 1 ${1:-40}); do
  timeout 120 python tests/studies/order_dependence_hunt.py 1 0 1 2 3 4 5 6 7 8 9 10 11 2>&1 | grep "PARITY MISS" | cut -c1-1200 && n=$((n+1))
done
echo "fresh processes with a parity miss: $n of ${1:-40}"
