#!/usr/bin/env python3
"""Audit-lane soak: seconds of back-to-back ticks per workload with the audit lane at its maximum rate (audit_k = 16), inputs
different every tick (bench.make_input_ring), then the lane's counters -- how many CONVERGED ADMM solves were re-solved by the
exact bodies on the side stream, how many were off by more than audit_tol (the 1e-4 torque bar), and the largest per-robot /
per-joint error seen.  The parity tests compare whole batches with the oracle for tens of ticks; this is the long-run check
of the one part of the path whose exit is a tuned rule.  Usage: audit_soak.py [seconds per workload]   (GPU)"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from robot_gym_amd import synthetic  # noqa: E402
from robot_gym_amd.controllers.mpc.batched import BatchedMPCController  # noqa: E402
from robot_gym_amd.core.config import MPCConfig  # noqa: E402

SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
WORKLOADS = [
    ("headline: batch 4096, hybrid", dict(), 4096, dict()),
    ("every robot on ADMM (solver 2), batch 4096", dict(solver=2), 4096, dict()),
    ("config 2: batch 1024, fixed command", dict(), 1024, dict(fixed_cmd=(0.3, 0.0, 0.0))),
    ("k3lso, chain geometry (kin_mode 1), batch 4096", dict(kin_mode=1, robot="k3lso"), 4096, dict(chain_geom=True)),
    ("horizon 20, hybrid, batch 4096", dict(horizon=20), 4096, dict()),
    ("config 5: horizon 20, random schedule, batch 4096", dict(horizon=20, contact_lookahead=1), 4096, dict(schedule=True)),
    ("horizon 10, random schedule, batch 4096", dict(contact_lookahead=1), 4096, dict(schedule=True)),
]


def main():
    device = torch.device("cuda:0")
    print("#", bench.evidence_header() + f"; tests/studies/audit_soak.py {SECONDS:g}")
    total = dict(audited=0, over=0, ticks=0)
    for name, over, B, ring_kw in WORKLOADS:
        over = dict(over)
        robot = over.pop("robot", "ghost")
        cfg = MPCConfig.for_robot(robot, audit_k=16, **over)
        gait = synthetic.random_gaits(B, cfg, seed=5) if ring_kw.get("schedule") else None
        state, cmd, t_off, slabs = bench.make_input_ring(cfg, B, 11, device, bench.RING, 0.1, gait=gait, **ring_kw)
        ctl = BatchedMPCController(B, cfg, device=device)
        if gait is not None:
            ctl.set_gait(**gait)
        ctl.reset_at(-t_off)
        ctl.update_controller_params(torch.from_numpy(cmd.T.copy()).to(device))
        k, t0 = 0, time.perf_counter()
        worst_fail = 0
        while time.perf_counter() - t0 < SECONDS:
            for _ in range(500):
                ctl.get_action(0.01 * k, slabs[k % len(slabs)])
                k += 1
            torch.cuda.synchronize()
            worst_fail = max(worst_fail, ctl.solver_stats()["failures"])
        el = time.perf_counter() - t0
        a = ctl.audit_stats()
        print(json.dumps({"workload": name, "ticks": k, "robot_ticks": k * B, "seconds": round(el, 1), "steps_per_s_with_audit_k16": round(k * B / el),
                          "audited": a["audited"], "over_tol": a["audit_over_tol"], "max_rel_per_robot": a["audit_max_rel"],
                          "max_rel_per_joint": a["audit_max_rel_elem"], "exact_failures": a["audit_exact_failures"], "dropped": a["audit_dropped"],
                          "solver_failures_max_per_tick": worst_fail}))
        sys.stdout.flush()
        total["audited"] += a["audited"]; total["over"] += a["audit_over_tol"]; total["ticks"] += k * B
        ctl.close()
    print(json.dumps({"total_robot_ticks": total["ticks"], "total_audited": total["audited"], "total_over_tol": total["over"]}))
    return 0 if total["over"] == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
