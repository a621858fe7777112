"""Study (GPU + oracle): what the ADMM stopping tolerance costs and buys.  For admm_tol in {1e-6 (default), 3e-7, 1e-7}:
worst per-robot and per-joint torque error against the exact oracle on the trot workload (4096 robots x 30 ticks) and the
k3lso / on-device-kinematics workload (2048 x 20), mean ADMM iterations, and the headline bench's steps/s with that tolerance
(bench.py as a child process).  Prints a markdown table; tools/r3_evidence.sh stores it under profiles/.
Per robot: max_j |dtau_j| / max(max_j |tau_j|, 1 N m) -- the parity bar (1e-4).  Per joint: |dtau_j| / max(|tau_j|, 1 N m)."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from oracle import oracle as O                   # noqa: E402
from robot_gym_amd.core.config import MPCConfig  # noqa: E402
from robot_gym_amd import synthetic              # noqa: E402
from tests import helpers                        # noqa: E402

cases = [("trot", "ghost", {}, 4096, 30, 0), ("k3lso kin_mode 1", "k3lso", dict(kin_mode=1), 2048, 20, 3)]
oracle_runs = {}
rows = []
for tol in (1e-6, 3e-7, 1e-7):
    worst_robot = worst_joint = 0.0
    iters = []
    audit_max = 0.0
    for name, robot, kw, B, ticks, seed in cases:
        cfg = MPCConfig.for_robot(robot, admm_tol=tol, **kw)
        state, cmd, t_off = synthetic.make_states(B, cfg, seed=seed)
        if name not in oracle_runs:
            oracle_runs[name] = helpers.run_oracle(O, cfg, state, cmd, t_off, ticks=ticks, jitter=0.1)
        gpu = helpers.run_gpu(cfg, state, cmd, t_off, ticks=ticks, jitter=0.1, poison=False)
        for g, o in zip(gpu, oracle_runs[name]):
            m = helpers.compare_tick(g, o)
            worst_robot, worst_joint = max(worst_robot, m["tau_rel_max"]), max(worst_joint, m["tau_rel_elem_max"])
        iters.append(gpu[-1]["solver_stats"]["iters_mean"])
        audit_max = max(audit_max, gpu[-1]["audit"]["audit_max_rel_elem"])
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "100", "--warmup", "20", "--no-cpu-baseline", "--no-extras",
                          "--no-kernel-events", "--tol", str(tol)], capture_output=True, text=True, cwd=ROOT)
    line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    rows.append((tol, worst_robot, worst_joint, audit_max, iters, d["value"], d["config"]["admm_iterations"]["iters_mean"]))
print("| admm_tol | worst per-robot error | worst per-joint error | audit lane: worst per-joint | mean iterations (trot / k3lso) | headline steps/s | bench mean iterations |")
print("|---|---|---|---|---|---|---|")
for tol, wr, wj, am, it, v, bi in rows:
    print(f"| {tol:g} | {wr:.2e} | {wj:.2e} | {am:.2e} | {it[0]:.1f} / {it[1]:.1f} | {v / 1e6:.2f} M | {bi:.1f} |")
