#!/bin/bash
# the round-2 tree (scratch/r2tree, git worktree of fa1573d): does the rare seed-8 parity miss exist there too?
cd ${GRAFT_REPO_ROOT:-$(pwd)}/scratch/r2tree
n=0
for i in $(seq 1 ${1:-60}); do
  out=$(timeout 120 python hunt.py 2>&1 | grep "PARITY MISS")
  if [ -n "$out" ]; then echo "$out"; n=$((n+1)); fi
done
echo "round-2 tree: fresh processes with a parity miss: $n of ${1:-60}"
