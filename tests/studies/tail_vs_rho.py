"""Study (GPU): would the slow robots of the headline workload be faster with another ADMM rho?  Several controllers, identical
inputs, each with its own rho (and its own warm start); for the robots above the p95 / p99 of their stance-leg class at the
default rho: their iteration counts under the other values."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench                                                   # noqa: E402
from robot_gym_amd.core.config import MPCConfig                # noqa: E402
from robot_gym_amd.controllers.mpc.batched import BatchedMPCController   # noqa: E402

B, TICKS = 4096, 60
RHOS = [1e-4, 5e-5, 7e-5, 1.5e-4, 2e-4, 3e-4]
dev = torch.device("cuda", 0)
base = MPCConfig.for_robot("ghost")
state, cmd, t_off, slabs = bench.make_input_ring(base, B, 0, dev, 50, 0.1)
ctls = []
for rho in RHOS:
    cfg = MPCConfig.for_robot("ghost", admm_rho=rho, admm_rho2=5 * rho)
    ctl = BatchedMPCController(B, cfg, device=dev, extra_outputs=False)
    ctl.reset_at(-t_off)
    ctl.update_controller_params(torch.from_numpy(cmd.T.copy()).to(dev))
    ctls.append(ctl)
its = np.zeros((len(RHOS), TICKS, B), dtype=np.int64)
ncs = np.zeros((TICKS, B), dtype=np.int64)
for k in range(TICKS):
    for i, ctl in enumerate(ctls):
        ctl.get_action(0.01 * k, slabs[k % 50])
        torch.cuda.synchronize()
        it, nc = ctl._handle.last_iterations(B, ctl._stream())
        its[i, k] = it
        if i == 0:
            ncs[k] = nc
for ctl in ctls:
    ctl.close()
K0 = 30
for legs in (2, 4):
    print(f"--- {legs} stance legs")
    sel = ncs[K0:] == legs
    for i, rho in enumerate(RHOS):
        v = its[i, K0:][sel]
        print(f"rho {rho:.1e}: mean {v.mean():6.1f}  p50 {np.percentile(v, 50):4.0f}  p95 {np.percentile(v, 95):4.0f}  p99 {np.percentile(v, 99):4.0f}  p99.9 {np.percentile(v, 99.9):4.0f}  max {v.max():4d}   mean of per-tick max {np.mean([its[i, k][ncs[k] == legs].max() for k in range(K0, TICKS)]):6.1f}")
    for q in (95, 99):
        rows = []
        for k in range(K0, TICKS):
            m = ncs[k] == legs
            thr = np.percentile(its[0, k][m], q)
            slow = m & (its[0, k] >= thr)
            rows.append(its[:, k][:, slow])
        allv = np.concatenate(rows, axis=1)
        best = allv.min(axis=0)
        print(f"robots at or above p{q} at the default rho ({allv.shape[1]} robot-ticks): mean iterations per rho " + "  ".join(f"{r:.0e}: {allv[i].mean():.0f}" for i, r in enumerate(RHOS)) +
              f"   best of all per robot: {best.mean():.0f}   which rho is best: " + str(np.bincount(allv.argmin(axis=0), minlength=len(RHOS)).tolist()))
