"""Evidence under profiles/ names the commit it was measured on.

Every round-5 file (profiles/r5_*) carries `kernel sources <hash> commit <sha>` (bench.evidence_header(): text files in their
first line, JSON files as source_hash / commit or inside the bench line), and the hash IS the hash of the compiled kernel
sources at that commit -- recomputed here from `git show <sha>:<path>`.  (Round 4's sweeps were stamped with hashes that
matched no commit: the files had been produced from a working tree with uncommitted edits.)  CSV tables have no header of
their own: they belong to the *_traffic.json of the same tag.  Skipped where there is no git history (the GPU box).

Round 6 hashes the sources WITHOUT comments and white space (bench.strip_c_comments): a stale sentence in a header can then
be fixed without orphaning the evidence measured on that code.  Round-5 files keep the raw-byte hash they were stamped with."""
import glob
import json
import os
import re
import subprocess

import pytest

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROFILES = os.path.join(ROOT, "profiles")
ROUNDS = [("r5", True), ("r6", False)]   # (file prefix, hashed over raw bytes)


# stand-alone studies (their own binaries, no library build behind them): described in LAB_NOTES.md, not tied to a library commit
UNSTAMPED = {"r6_sweep_bench_study.txt", "r6_isa_identity_friction_rows.txt", "r6_config5_resolve_census.txt", "r6_accel_tail.txt"}


def _git(*args):
    return subprocess.run(["git"] + list(args), cwd=ROOT, capture_output=True, text=True, timeout=60)


def _stamp_of(path):
    """(source hash, commit) a profiles/ file says it was measured on, or None."""
    if path.endswith(".json"):
        try:
            d = json.load(open(path))
        except ValueError:
            d = json.loads(open(path).read().strip().splitlines()[-1])
        if "source_hash" in d:
            return d.get("source_hash"), d.get("commit")
        c = d.get("config", {})
        return c.get("kernel_sources"), c.get("commit")
    m = re.search(r"kernel sources ([0-9a-f]{16}) commit ([0-9a-f]{40})( \+uncommitted changes)?", open(path, errors="replace").read(2000))
    return (m.group(1), m.group(2) + (m.group(3) or "")) if m else None


def test_compiled_sources_are_what_the_makefile_compiles():
    srcs = list(bench.compiled_sources())
    assert "robot_gym_amd/csrc/rg_mpc.hip" in srcs and "include/rg_mpc.h" in srcs and "robot_gym_amd/csrc/rg_qp_exact_kernel.inc" in srcs
    assert all(s.endswith((".hip", ".inc", ".h")) for s in srcs) and len(srcs) >= 8
    # a stray file next to the sources is not part of the library and not part of the hash
    stray = os.path.join(ROOT, "robot_gym_amd", "csrc", "zz_stray_experiment.inc")
    h0 = bench.source_hash()
    open(stray, "w").write("// not included by anything\n")
    try:
        assert bench.source_hash() == h0
    finally:
        os.remove(stray)


def test_the_source_hash_ignores_comments_and_white_space_only():
    code = b'int a = 1; // note "x\n/* block\n */ const char *s = "a // kept /* kept */";   char c = \'"\';\n'
    assert bench.strip_c_comments(code) == b'int a = 1; const char *s = "a // kept /* kept */"; char c = \'"\';'
    srcs = bench.compiled_sources()
    h0 = bench.source_hash()

    def edited(mutate):
        def read(rel):
            return mutate(rel, srcs[rel]) if rel in srcs else None
        return bench.source_hash(read)
    hdr = "include/rg_mpc.h"
    assert edited(lambda rel, d: d + b"\n// a trailing remark\n" if rel == hdr else d) == h0
    assert edited(lambda rel, d: d.replace(b"\n", b"\n\n  ") if rel == hdr else d) == h0
    assert edited(lambda rel, d: d + b"\nstatic int rg_new_symbol;\n" if rel == hdr else d) != h0


@pytest.mark.parametrize("ROUND,raw", ROUNDS)
def test_round_evidence_names_a_commit_whose_sources_it_hashes(ROUND, raw):
    if _git("rev-parse", "HEAD").returncode != 0:
        pytest.skip("no git history here")
    files = sorted(f for f in glob.glob(os.path.join(PROFILES, f"{ROUND}_*")) if not f.endswith(".csv") and os.path.basename(f) not in UNSTAMPED)
    if not files:
        pytest.skip(f"no profiles/{ROUND}_* yet")
    seen = {}
    for f in files:
        st = _stamp_of(f)
        assert st and st[0] and st[1], f"{os.path.basename(f)} does not say which kernel sources / commit it was measured on"
        src, commit = st
        assert "uncommitted" not in commit, f"{os.path.basename(f)} was measured on uncommitted kernel sources"
        if commit not in seen:
            assert _git("cat-file", "-e", commit + "^{commit}").returncode == 0, f"{os.path.basename(f)}: commit {commit} is not in this history"
            seen[commit] = bench.source_hash_at(commit, raw)
        assert seen[commit] == src, f"{os.path.basename(f)}: stamped {src}, the compiled sources of commit {commit[:12]} hash to {seen[commit]}"
    for csv in glob.glob(os.path.join(PROFILES, f"{ROUND}*_kernel_stats.csv")) + glob.glob(os.path.join(PROFILES, f"{ROUND}*_pmc_per_launch.csv")):
        tag = os.path.basename(csv).replace("_kernel_stats.csv", "").replace("_pmc_per_launch.csv", "")
        assert os.path.exists(os.path.join(PROFILES, f"{tag}_traffic.json")), f"{os.path.basename(csv)} has no {tag}_traffic.json naming its commit"


def test_evidence_header_and_stamp_fallback(tmp_path, monkeypatch):
    """bench.git_head(): the commit from git where there is a history; on the GPU box (no .git) from the stamp file
    tools/stamp_commit.py wrote -- honoured only while its source hash is the tree's."""
    if _git("rev-parse", "HEAD").returncode != 0:
        pytest.skip("no git history here")
    head = _git("rev-parse", "HEAD").stdout.strip()
    commit, dirty = bench.git_head()
    assert commit == head and dirty in (False, True)
    hdr = bench.evidence_header()
    assert hdr.startswith(f"kernel sources {bench.source_hash()} commit {head}")
    # the fallback: pretend there is no git, with a valid and then a stale stamp
    import subprocess
    real_run = subprocess.run

    def no_git(cmd, *a, **k):
        if cmd and cmd[0] == "git":
            raise OSError("no git here")
        return real_run(cmd, *a, **k)
    monkeypatch.setattr(subprocess, "run", no_git)
    stamp = tmp_path / "stamp.json"
    monkeypatch.setattr(bench, "STAMP_FILE", str(stamp))
    assert bench.git_head() == (None, None)
    json.dump({"commit": "ab" * 20, "dirty": False, "source_hash": bench.source_hash()}, open(stamp, "w"))
    assert bench.git_head() == ("ab" * 20, False)
    json.dump({"commit": "ab" * 20, "dirty": False, "source_hash": "0" * 16}, open(stamp, "w"))
    assert bench.git_head() == (None, None)
