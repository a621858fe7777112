"""Study (GPU + oracle): hunt for results that depend on what ran on the device before.  Replays sweep configurations in
one process over and over (the oracle results are computed once per configuration) and reports every tick whose commands
differ from the first time the same configuration ran -- with the robots, their stance-leg counts and iteration counts.
Usage: python tests/studies/order_dependence_hunt.py [rounds] [seeds...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O                   # noqa: E402
from tests import helpers                        # noqa: E402
from tests.test_gpu_parity import _sweep_case    # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
seeds = [int(s) for s in sys.argv[2:]] or list(range(12))
ref, orc = {}, {}
bad = 0
for r in range(rounds):
    for s in seeds:
        cfg, B, over, kw = _sweep_case(s)
        if s not in orc:
            orc[s] = helpers.run_oracle(O, cfg, **kw)
        gpu = helpers.run_gpu(cfg, **kw)
        acts = [g["action"] for g in gpu]
        if s not in ref:
            ref[s] = acts
        for k, (a, a0, g, o) in enumerate(zip(acts, ref[s], gpu, orc[s])):
            m = helpers.compare_tick(g, o)
            if m["tau_rel_max"] > 1e-4:
                w = m["tau_rel_argmax"]
                print(f"PARITY MISS round {r} seed {s} tick {k}: robot {w} of {B}: stance legs {g['stance_legs'][w]} iterations {g['iters'][w]} desired {g['desired_state'][w].tolist()} "
                      f"leg_state {g['leg_state'][w].tolist()} tau_rel {m['tau_rel_max']:.3e}; iterations of all robots {g['iters'].tolist()}", flush=True)
                print("   grf gpu   ", np.round(g["grf"][w].astype(np.float64), 4).tolist())
                print("   grf oracle", np.round(np.asarray(o["grf"][w], dtype=np.float64), 4).tolist())
                print("   foot_target gpu", np.round(g["foot_target"][w], 5).tolist(), "v_body", g["v_body"][w].tolist())
                print("   v_body oracle", np.asarray(o["v_body"][w]).tolist(), "audit", gpu[-1]["audit"], "oracle kkt", np.asarray(o["kkt"][w]).tolist())
                dv = np.abs(g["v_body"].astype(np.float64) - np.asarray(o["v_body"])).max(1)
                print("   robots whose filtered body velocity differs from the oracle's by > 1e-6:", np.nonzero(dv > 1e-6)[0].tolist(), "max", dv.max())
                st0 = helpers.perturb(kw["state"], k, kw["jitter"])
                print("   robot inputs: v_world", st0["v_world"][:, w].tolist(), "quat", st0["quat"][:, w].tolist(), "rpy", st0["rpy"][:, w].tolist(), "window", cfg.window, "t_off", float(kw["t_off"][w]))
                import dataclasses
                for label, c2 in (("exact GPU solver (RG_SOLVER_ACTIVE_SET)", dataclasses.replace(cfg, solver=1)), ("audit_k = 0", dataclasses.replace(cfg, audit_k=0)),
                                  ("admm_accel = 0", dataclasses.replace(cfg, admm_accel=0)), ("admm_tol 1e-8", dataclasses.replace(cfg, admm_tol=1e-8))):
                    try:
                        g2 = helpers.run_gpu(c2, **kw)[k]
                        print(f"   {label}: tau_rel {helpers.compare_tick(g2, o)['tau_rel_max']:.3e} iterations {g2['iters'][w]} grf {np.round(g2['grf'][w].astype(np.float64), 4).tolist()}", flush=True)
                    except Exception as e:
                        print(f"   {label}: {type(e).__name__}: {e}")
                again = helpers.run_gpu(cfg, **kw)[k]
                m2 = helpers.compare_tick(again, o)
                print(f"   same configuration again in this process: tau_rel {m2['tau_rel_max']:.3e}; grf of the robot {np.round(again['grf'][w].astype(np.float64), 4).tolist()}", flush=True)
            if not np.array_equal(a, a0) or m["tau_rel_max"] > 1e-4:
                bad += 1
                d = np.abs(a.astype(np.float64) - a0.astype(np.float64)).reshape(B, -1).max(1)
                rob = np.nonzero(d > 0)[0]
                print(f"round {r} seed {s} tick {k}: differs from first run on robots {rob.tolist()[:8]} max |d action| {d.max():.3e}; vs oracle tau_rel {m['tau_rel_max']:.2e} "
                      f"(robot {m['tau_rel_argmax']}); desired {g['desired_state'][rob[:4]].tolist() if len(rob) else []} stats {g['solver_stats']}", flush=True)
print(f"{rounds} rounds x seeds {seeds}: {bad} deviating ticks")
