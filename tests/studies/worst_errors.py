"""Study (GPU + oracle): torque-error statistics of the default configuration over many robot-ticks, per workload.
Overrides for experiments: TOL (admm_tol), EXTRAP (admm_extrap), ACCEL (admm_accel), SOLVER (solver plan); unset = the library defaults.
Usage: python tests/studies/worst_errors.py [all|h10|h20|trot]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from oracle import oracle as O                   # noqa: E402
from robot_gym_amd.core.config import MPCConfig  # noqa: E402
from robot_gym_amd import synthetic              # noqa: E402
from tests import helpers                        # noqa: E402

over = {}
for env, key, conv in (("TOL", "admm_tol", float), ("EXTRAP", "admm_extrap", float), ("ACCEL", "admm_accel", int), ("SOLVER", "solver", int)):
    if os.environ.get(env) is not None:
        over[key] = conv(os.environ[env])
which = sys.argv[1] if len(sys.argv) > 1 else "all"
cases = []
if which == "trot":
    cases += [("trot", "ghost", {}, 4096, 50, 0, False)]
if which in ("all", "h10"):
    cases += [("trot", "ghost", {}, 4096, 50, 0, False), ("trot-k3lso-kin1", "k3lso", dict(kin_mode=1), 2048, 40, 3, False),
              ("walk", "ghost", dict(duty_factor=(0.75,) * 4, init_phase=(0.0, 0.5, 0.25, 0.75), init_state=(1, 1, 1, 1)), 1024, 30, 7, False)]
if which in ("all", "h20"):
    cases += [("trot-h20", "ghost", dict(horizon=20), 1024, 12, 1, False), ("config5", "ghost", dict(horizon=20, contact_lookahead=1, admm_iters=600), 1024, 10, 2, True)]
for name, robot, kw, B, ticks, seed, sched in cases:
    cfg = MPCConfig.for_robot(robot, **dict(kw, **over))
    state, cmd, t_off = synthetic.make_states(B, cfg, seed=seed)
    gait = synthetic.random_gaits(B, cfg, seed=seed) if sched else None
    sched_fn = (lambda k, t_rel: synthetic.contact_schedule(cfg, t_rel, gait, dropout=0.1, seed=seed, tick=k)) if sched else None
    orc = helpers.run_oracle(O, cfg, state, cmd, t_off, ticks=ticks, jitter=0.1, gait=gait, sched_fn=sched_fn)
    gpu = helpers.run_gpu(cfg, state, cmd, t_off, ticks=ticks, jitter=0.1, poison=False, gait=gait, sched_fn=sched_fn)
    w = [helpers.compare_tick(g, o) for g, o in zip(gpu, orc)]
    allerr = np.concatenate([np.abs(g["action"].reshape(B, 12, 5)[:, :, 4].astype(np.float64) - o["action"].reshape(B, 12, 5)[:, :, 4].astype(np.float64)).max(1)
                             / np.maximum(np.abs(o["action"].reshape(B, 12, 5)[:, :, 4]).max(1), 1.0) for g, o in zip(gpu, orc)])
    print(f"{name:16s} robot-ticks {allerr.size:7d}  p50 %.1e p99 %.1e p99.9 %.1e max %.1e" % tuple(np.percentile(allerr, [50, 99, 99.9, 100])),
          " per-joint max %.1e  grf %.1e  mean iterations %.1f  exact re-solves %d  failures %d" % (max(m["tau_rel_elem_max"] for m in w), max(m["grf_rel_max"] for m in w),
          gpu[-1]["solver_stats"]["iters_mean"], sum(g["solver_stats"]["retried_exact"] for g in gpu), sum(g["solver_stats"]["failures"] for g in gpu)), over,
          "audit", {k: (v if isinstance(v, int) else float("%.2e" % v)) for k, v in gpu[-1]["audit"].items()})
