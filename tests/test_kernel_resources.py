"""The compiler's answer for every kernel is pinned (no GPU needed: hipcc cross-compiles gfx950).

Every QP kernel sits at 252-256 VGPRs and depends on register allocation the source can only nudge (opaque lane indices,
re-read kernel arguments, one work item per workgroup).  One device-only compile of rg_mpc.hip (tools/kernel_report.py)
gives, per kernel, VGPR / AGPR / scratch bytes / occupancy and the number of scratch operations INSIDE solver loops; each
must be no worse than the committed table profiles/r6_resource_usage.txt, which is regenerated (tools/resource_usage.py
profiles/r6_resource_usage.txt) whenever a kernel changes on purpose.  A ROCm bump or an edit that tips a solver loop into
scratch then fails here, not in a bench three rounds later."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_report  # noqa: E402

TABLE = os.path.join(ROOT, "profiles", "r6_resource_usage.txt")


@pytest.fixture(scope="module")
def compiled():
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc here")
    return {r["name"]: r for r in kernel_report.report()}


def test_every_kernel_is_no_worse_than_the_committed_table(compiled):
    want = kernel_report.parse_table(open(TABLE).read())
    assert want, "profiles/r6_resource_usage.txt has no kernel rows"
    assert set(compiled) == set(want), f"kernel set changed: new {sorted(set(compiled) - set(want))}, gone {sorted(set(want) - set(compiled))} -- regenerate the table"
    worse = []
    for name, w in want.items():
        g = compiled[name]
        for key in ("vgpr", "agpr", "scratch", "loop_scratch"):
            if g[key] > w[key]:
                worse.append(f"{name}: {key} {w[key]} -> {g[key]}")
        if g["waves"] < w["waves"]:
            worse.append(f"{name}: waves/SIMD {w['waves']} -> {g['waves']}")
    assert not worse, "compiler resources got worse than profiles/r6_resource_usage.txt:\n  " + "\n  ".join(worse)


def test_the_default_plans_kernels_have_no_spill_in_a_solver_loop(compiled):
    """The kernels every shipped configuration runs (uniform friction): the headline launch and its re-solve, the horizon-20
    launch, the batch-of-a-few 256-lane grid.  (The schedule kernel at horizon 10 and the per-leg-friction instantiations carry
    2-3 reloads per ADMM iteration: pinned by the table above, not claimed clean.)"""
    for name in ("rg_qp_fused_kernel<10,2,1,0,0>", "rg_qp_fused_kernel<10,2,1,1,0>", "rg_qp_fused_kernel<20,2,1,0,0>", "rg_qp_sched_kernel<20,2,0>",
                 "rg_qp_resolve_kernel<10,0,0>", "rg_qp_sched_retry_kernel<20,0>", "rg_front_kernel"):
        assert compiled[name]["loop_scratch"] == 0, f"{name}: scratch operations inside a solver loop"
    # two waves per SIMD is what the launch plans assume for the QP launches (2048 robots resident at horizon 10)
    for name, r in compiled.items():
        if name.startswith(("rg_qp_fused_kernel", "rg_qp_sched_kernel<")):
            assert r["waves"] >= 2 and r["vgpr"] + r["agpr"] <= 256, name
