"""Study (GPU): how many robots take the direct route (front kernel -> exact solver) and the exact re-solve per tick on a bench
workload, and how expensive they are.  python tests/studies/direct_route_census.py [batch] ['{"kin_mode": 1}']"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np   # noqa: E402
import torch   # noqa: E402
import bench   # noqa: E402
from robot_gym_amd.core.config import MPCConfig   # noqa: E402
from robot_gym_amd.controllers.mpc.batched import BatchedMPCController   # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
over = json.loads(sys.argv[2]) if len(sys.argv) > 2 else {"kin_mode": 1}
torch.cuda.set_device(0)
device = torch.device("cuda", 0)
cfg = MPCConfig.for_robot("ghost", **over)
state, cmd, t_off, slabs = bench.make_input_ring(cfg, B, 0, device, 50, 0.1, None, None, False, bool(over.get("kin_mode")))
ctl = BatchedMPCController(B, cfg, device=device)
ctl.reset_at(-t_off)
ctl.update_controller_params(torch.from_numpy(cmd.T.copy()).to(device))
prev_hard = None
for k in range(60):
    ctl.get_action(0.01 * k, slabs[k % 50])
    torch.cuda.synchronize()
    if k >= 45:
        nd, launches = ctl._handle.last_direct_count(ctl._stream())
        st = ctl.solver_stats()
        it, nc = ctl._handle.last_iterations(B, ctl._stream())
        big = np.sort(it[nc == 2])[-8:]
        print(f"tick {k}: direct-route robots {nd}, concurrent direct launches so far {launches}, exact re-solves {st['retried_exact']}, "
              f"two-leg work: p50 {np.median(it[nc == 2]):.0f} p99 {np.percentile(it[nc == 2], 99):.0f} largest {big.tolist()}; robots with work > 40: {(it[nc == 2] > 40).sum()}")
ctl.close()
