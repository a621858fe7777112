import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from robot_gym_amd.core.config import MPCConfig
from robot_gym_amd.controllers.mpc.batched import BatchedMPCController
import bench
B = int(sys.argv[1]); jit = float(sys.argv[2]); fixed = (0.3, 0, 0) if len(sys.argv) > 3 and sys.argv[3] == "fixed" else None
import os
cfg = MPCConfig.for_robot("ghost", admm_rho2=float(os.environ.get("RHO2","5e-4")), admm_switch=int(os.environ.get("SW","150")), admm_iters=int(os.environ.get("CAP","300")))
dev = torch.device("cuda:0")
state, cmd, t_off, slabs = bench.make_input_ring(cfg, B, 0, dev, 50, jit, fixed, None, False)
ctl = BatchedMPCController(B, cfg, device=dev, extra_outputs=False)
ctl.reset_at(-t_off); ctl.update_controller_params(torch.from_numpy(cmd.T.copy()).to(dev))
tot = 0
for k in range(150):
    ctl.get_action(0.01 * k, slabs[k % 50])
    st = ctl.solver_stats()
    if st["retried_exact"]:
        it, nc = ctl._handle.last_iterations(B, ctl._stream())
        idx = np.argsort(it)[-st["retried_exact"]:]
        tot += st["retried_exact"]
        if os.environ.get("V"): print("tick", k, "retried", st["retried_exact"], "robots", [(int(b), int(nc[b]), int(it[b])) for b in idx])
import time
torch.cuda.synchronize(); t0=time.perf_counter()
for k in range(100): ctl.get_action(0.01*(150+k), slabs[k % 50])
torch.cuda.synchronize(); el=time.perf_counter()-t0
print(os.environ.get("RHO2"), os.environ.get("SW"), os.environ.get("CAP"), "total retried", tot, "in 150 ticks; Msteps/s", round(B*100/el/1e6,2), ctl.solver_stats()["iters_mean"])
