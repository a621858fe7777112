"""Study (GPU): what a two-rho race would buy the four-leg ADMM robots.  The bench workload is run once per value of
admm_rho34_scale (same inputs, cold start every tick so that the runs see the same QPs); per tick: the largest iteration count
among the four-leg robots for each value alone and for the per-robot minimum over pairs of values.
    python tests/studies/rho_race_potential.py [batch]"""
import itertools
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench   # noqa: E402
from robot_gym_amd.controllers.mpc.batched import BatchedMPCController   # noqa: E402
from robot_gym_amd.core.config import MPCConfig   # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 0
device = torch.device("cuda", 0)
scales = (0.25, 0.35, 0.5, 0.7, 1.0, 1.4)
its = {}
for s in scales:
    cfg = MPCConfig.for_robot("ghost", admm_rho34_scale=s, warm_start=warm)
    state, cmd, t_off, slabs = bench.make_input_ring(cfg, B, 0, device, 50, 0.1)
    ctl = BatchedMPCController(B, cfg, device=device, extra_outputs=False)
    ctl.reset_at(-t_off)
    ctl.update_controller_params(torch.from_numpy(cmd.T.copy()).to(device))
    rows = []
    for k in range(60):
        ctl.get_action(0.01 * k, slabs[k % 50])
        torch.cuda.synchronize()
        if k >= 30:
            it, nc = ctl._handle.last_iterations(B, ctl._stream())
            rows.append(np.where(nc == 4, it, -1))
    its[s] = np.array(rows)
    ctl.close()
m4 = its[scales[0]] >= 0
print(f"batch {B} warm_start {warm}: four-leg robots per tick {m4.sum(1).mean():.0f}")
for s in scales:
    x = np.where(m4, its[s], 0)
    print(f"scale {s:4.2f}: mean {its[s][m4].mean():5.1f} p99 {np.percentile(its[s][m4], 99):4.0f} per-tick max: mean {x.max(1).mean():5.1f} worst {x.max()}")
for a, b in itertools.combinations(scales, 2):
    x = np.where(m4, np.minimum(its[a], its[b]), 0)
    print(f"race {a:4.2f} / {b:4.2f}: mean {np.minimum(its[a], its[b])[m4].mean():5.1f} per-tick max: mean {x.max(1).mean():5.1f} worst {x.max()}")
