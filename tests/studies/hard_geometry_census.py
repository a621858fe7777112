"""Study (GPU): who needs the exact re-solve on the chain-geometry workload (bench.py --kin-mode 1 / --chain-geometry), where the
re-solve launch averages 170 us per tick?  Per tick: robots handed to the exact solver, their stance-leg counts, how many of
them were re-solved in the tick before, and the ADMM iteration tail."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench                                                   # noqa: E402
from robot_gym_amd.core.config import MPCConfig                # noqa: E402
from robot_gym_amd.controllers.mpc.batched import BatchedMPCController   # noqa: E402

B = 4096
cfg = MPCConfig.for_robot("ghost", kin_mode=1)
dev = torch.device("cuda", 0)
state, cmd, t_off, slabs = bench.make_input_ring(cfg, B, 0, dev, 50, 0.1)
ctl = BatchedMPCController(B, cfg, device=dev, extra_outputs=False)
ctl.reset_at(-t_off)
ctl.update_controller_params(torch.from_numpy(cmd.T.copy()).to(dev))
prev = set()
cap = cfg.admm_iters
for k in range(40):
    ctl.get_action(0.01 * k, slabs[k % 50])
    torch.cuda.synchronize()
    it, nc = ctl._handle.last_iterations(B, ctl._stream())
    st = ctl.solver_stats()
    hard = set(np.nonzero(it >= cap)[0].tolist())
    ndirect, nlaunch = ctl._handle.last_direct_count(ctl._stream())
    print(f"tick {k:2d}: exact re-solves {st['retried_exact']:3d}  stance legs of those {np.bincount(nc[list(hard)], minlength=5).tolist() if hard else []}  also re-solved last tick {len(hard & prev):3d}  "
          f"iterations p50 {int(np.percentile(it[nc > 0], 50))} p99 {int(np.percentile(it[nc > 0], 99))} p99.9 {int(np.percentile(it[nc > 0], 99.9))} robots >= 300: {(it >= 300).sum()}  sent straight to the exact solver {ndirect} (concurrent launches so far {nlaunch})", flush=True)
    prev = hard
ctl.close()
