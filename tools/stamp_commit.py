#!/usr/bin/env python3
"""Run HERE (where .git is) before a gpurun call that collects evidence: writes .rg_source_commit (git-ignored, travels with
the snapshot) = {commit, dirty, source_hash} of the compiled kernel sources.  The evidence scripts on the GPU box put
bench.evidence_header() into every file they write and REFUSE to run when the sources are uncommitted (RG_ALLOW_DIRTY=1
overrides, and the header then says so)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
if os.path.exists(bench.STAMP_FILE):
    os.remove(bench.STAMP_FILE)
commit, dirty = bench.git_head()
json.dump({"commit": commit, "dirty": bool(dirty), "source_hash": bench.source_hash()}, open(bench.STAMP_FILE, "w"))
print(open(bench.STAMP_FILE).read())
if dirty and os.environ.get("RG_ALLOW_DIRTY") != "1":
    sys.exit("compiled kernel sources differ from HEAD: commit first (or RG_ALLOW_DIRTY=1)")
