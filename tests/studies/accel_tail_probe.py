"""Study (GPU): how do the four-leg robots' iteration tails move with admm_accel (the first iteration at which a vote may
extrapolate)?  Round 6: the robots that need >= 90 iterations at 80 need ~70 at 50-60, the class p99 falls 90 -> 70 -- and the per-tick
MAXIMUM does not move (other robots become the slowest: an early jump backfires for some), which is what the launch waits for."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from robot_gym_amd.core.config import MPCConfig
from robot_gym_amd.controllers.mpc.batched import BatchedMPCController
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
TICKS = 70
dev = torch.device("cuda", 0)
res = {}
for accel in (80, 60, 50, 40, 30):
    cfg = MPCConfig.for_robot("ghost", admm_accel=accel)
    state, cmd, t_off, slabs = bench.make_input_ring(cfg, B, 0, dev, 50, 0.1)
    ctl = BatchedMPCController(B, cfg, device=dev, extra_outputs=False)
    ctl.reset_at(-t_off)
    ctl.update_controller_params(torch.from_numpy(cmd.T.copy()).to(dev))
    its, ncs = [], []
    for k in range(TICKS):
        ctl.get_action(0.01 * k, slabs[k % 50]); torch.cuda.synchronize()
        it, nc = ctl._handle.last_iterations(B, ctl._stream())
        its.append(it.copy()); ncs.append(nc.copy())
    a = ctl.audit_stats()
    ctl.close()
    res[accel] = (np.array(its)[30:], np.array(ncs)[30:], a)
base_it, base_nc, _ = res[80]
slow = (base_nc == 4) & (base_it >= 90)
for accel, (it, nc, a) in res.items():
    v = it[nc == 4]
    pm = np.where(nc == 4, it, 0).max(axis=1)
    same = slow & (nc == 4)
    print(f"accel {accel}: four-leg mean {v.mean():.1f} p99 {np.percentile(v, 99):.0f} | per-tick max mean {pm.mean():.0f} worst {pm.max()} | robots slow (>= 90) at accel 80: n {same.sum()} mean there {base_it[same].mean():.0f} -> here {it[same].mean():.0f} | audit over_tol {a['audit_over_tol']} of {a['audited']} max_rel_elem {a['audit_max_rel_elem']:.1e}")
