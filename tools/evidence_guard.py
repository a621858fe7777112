#!/usr/bin/env python3
"""Run on the GPU box at the top of an evidence script: prints bench.evidence_header() and exits non-zero when the kernel
sources of the snapshot are not those of a commit (no stamp, stale stamp, or dirty without RG_ALLOW_DIRTY=1)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
commit, dirty = bench.git_head()
print(bench.evidence_header())
if commit is None:
    sys.exit("no commit known for these kernel sources: run tools/stamp_commit.py before gpurun")
if dirty and os.environ.get("RG_ALLOW_DIRTY") != "1":
    sys.exit("kernel sources differ from the commit: evidence refused")
