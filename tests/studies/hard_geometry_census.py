"""Study (GPU): who needs the exact re-solve launch on the chain-geometry workload (bench.py --kin-mode 1) under the default
plan -- per tick: robots re-solved / sent straight to the exact solver, their stance-leg count, their iteration counts."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench   # noqa: E402
from robot_gym_amd.controllers.mpc.batched import BatchedMPCController   # noqa: E402
from robot_gym_amd.core.config import MPCConfig   # noqa: E402

B = 4096
device = torch.device("cuda", 0)
cfg = MPCConfig.for_robot("ghost", kin_mode=1)
state, cmd, t_off, slabs = bench.make_input_ring(cfg, B, 0, device, 50, 0.1)
ctl = BatchedMPCController(B, cfg, device=device, extra_outputs=False)
ctl.reset_at(-t_off)
ctl.update_controller_params(torch.from_numpy(cmd.T.copy()).to(device))
for k in range(40):
    ctl.get_action(0.01 * k, slabs[k % 50])
    torch.cuda.synchronize()
    st = ctl.solver_stats()
    n, launches = ctl._handle.last_direct_count(ctl._stream())
    it, nc = ctl._handle.last_iterations(B, ctl._stream())
    big = np.argsort(-it)[:4]
    print(f"tick {k:2d}: retried {st['retried_exact']} direct {n} concurrent launches so far {launches} failures {st['failures']} | largest iteration counts: " + ", ".join(f"robot {b} nc{nc[b]} it {it[b]}" for b in big))
ctl.close()
