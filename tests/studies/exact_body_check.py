"""GPU check of the solver plans against the oracle on a few gaits (trot, walk, unbalanced one-leg phases, pace), per tick:
worst per-robot and per-joint torque error, solver statistics.  python tests/studies/exact_body_check.py [batch] [ticks]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as O   # noqa: E402
from robot_gym_amd import synthetic   # noqa: E402
from robot_gym_amd.core.config import MPCConfig   # noqa: E402
from tests import helpers   # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 12
GAITS = {
    "trot": dict(),
    "walk": dict(duty_factor=(0.8,) * 4, init_phase=(0.0, 0.5, 0.25, 0.75), init_state=(1, 1, 1, 1), stance_duration=(0.4,) * 4),
    "unbalanced": dict(duty_factor=(0.55,) * 4, init_phase=(0.0, 0.3, 0.55, 0.8), init_state=(1, 1, 1, 1)),
    "pace": dict(init_phase=(0.9, 0.0, 0.9, 0.0), init_state=(0, 1, 0, 1)),
}
for solver in (3, 1, 2):
    for name, over in GAITS.items():
        for warm in ((1, 0) if solver == 3 else (1,)):
            cfg = MPCConfig.for_robot("ghost", solver=solver, warm_start=warm, **over)
            state, cmd, t_off = synthetic.make_states(B, cfg, seed=3)
            orc = helpers.run_oracle(O, cfg, state, cmd, t_off, ticks=ticks, jitter=0.1)
            gpu = helpers.run_gpu(cfg, state, cmd, t_off, ticks=ticks, jitter=0.1)
            worst, worst_e, fails, retried, its = 0.0, 0.0, 0, 0, []
            for g, o in zip(gpu, orc):
                m = helpers.compare_tick(g, o)
                assert m["leg_state_mismatch"] == 0, m
                worst, worst_e = max(worst, m["tau_rel_max"]), max(worst_e, m["tau_rel_elem_max"])
                fails += g["solver_stats"]["failures"]
                retried += g["solver_stats"]["retried_exact"]
                its.append(g["solver_stats"]["iters_mean"])
            print(f"solver {solver} warm {warm} {name:11s} bins {gpu[-1]['bins']} tau_rel {worst:.2e} per-joint {worst_e:.2e} failures {fails} retried {retried} iters first {its[0]:.1f} last {its[-1]:.1f} max {gpu[-1]['solver_stats']['iters_max']}", flush=True)
