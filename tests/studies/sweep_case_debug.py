"""Per-robot view of one case of the configuration sweep (tests/test_gpu_parity.py::_sweep_case): error, stance legs, solver
iterations per robot and tick.  python tests/studies/sweep_case_debug.py <seed> ['{"admm_tol": 1e-6}']"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 2:
    os.environ["RG_SWEEP_OVER"] = sys.argv[2]
from oracle import oracle as O   # noqa: E402
from tests import helpers   # noqa: E402
from tests.test_gpu_parity import _sweep_case   # noqa: E402

seed = int(sys.argv[1])
cfg, B, over, kw = _sweep_case(seed)
print(seed, cfg.robot, B, over)
orc = helpers.run_oracle(O, cfg, **kw)
gpu = helpers.run_gpu(cfg, **kw)
for k, (g, o) in enumerate(zip(gpu, orc)):
    a_g = g["action"].reshape(B, 12, 5)[:, :, 4].astype(np.float64)
    a_o = o["action"].reshape(B, 12, 5)[:, :, 4].astype(np.float64)
    err = np.abs(a_g - a_o).max(1) / np.maximum(np.abs(a_o).max(1), 1.0)
    err_j = (np.abs(a_g - a_o) / np.maximum(np.abs(a_o), 1.0)).max(1)   # per joint: the strict reading the tests assert
    qerr = np.abs(g["action"].reshape(B, 12, 5)[:, :, 0].astype(np.float64) - o["action"].reshape(B, 12, 5)[:, :, 0]).max(1)
    bad = np.where((err_j > 1e-4) | (qerr > 1e-5))[0]
    print(f"tick {k}: stats {g['solver_stats']} bins {g['bins']} worst per robot {err.max():.2e} per joint {err_j.max():.2e} bad robots {len(bad)}")
    for b in (bad[:12] if len(bad) else np.argsort(-err_j)[:3]):
        print(f"    robot {b}: err {err[b]:.2e} per joint {err_j[b]:.2e} q-err {qerr[b]:.1e} stance legs {g['stance_legs'][b]} iters {g['iters'][b]} desired {o['desired'][b]} oracle qp_iters {o['qp_iters'][b]}")

# per-leg view of the worst robot of the last tick: forces and body velocity on both sides
if len(sys.argv) > 3:
    rb = int(sys.argv[3])
    for k, (g, o) in enumerate(zip(gpu, orc)):
        gg, go = g["grf"][rb].astype(np.float64), o["grf"][rb]
        print(f"tick {k} robot {rb}: v_body gpu {g.get('v_body', np.zeros((B, 3)))[rb]} oracle {o['v_body'][rb]}")
        print(f"    grf gpu    {np.array2string(gg, precision=5)}\n    grf oracle {np.array2string(go, precision=5)}\n    diff       {np.array2string(gg - go, precision=2)}")
