"""Study (GPU): is the iteration tail of the headline workload made of the same robots tick after tick?  The fused launch at
batch 4096 ends with (longest first-round job) + (cheapest job); the longest jobs are four-leg robots with ~100 iterations.
If the slow robots of one tick are the slow robots of the next, a per-robot memory (another rho, an earlier stage switch)
could shorten the tail; if they are not, it cannot.  Per stance-leg count: iteration percentiles, and for the robots above
the class's p95 / p99 of a tick, where they stood in the tick before (same contact set only)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench                                                   # noqa: E402
from robot_gym_amd.core.config import MPCConfig                # noqa: E402
from robot_gym_amd.controllers.mpc.batched import BatchedMPCController   # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
TICKS = 80
cfg = MPCConfig.for_robot("ghost")
dev = torch.device("cuda", 0)
state, cmd, t_off, slabs = bench.make_input_ring(cfg, B, 0, dev, 50, 0.1)
ctl = BatchedMPCController(B, cfg, device=dev, extra_outputs=False)
ctl.reset_at(-t_off)
ctl.update_controller_params(torch.from_numpy(cmd.T.copy()).to(dev))
its, ncs = [], []
for k in range(TICKS):
    ctl.get_action(0.01 * k, slabs[k % 50])
    torch.cuda.synchronize()
    it, nc = ctl._handle.last_iterations(B, ctl._stream())
    its.append(it.copy()); ncs.append(nc.copy())
ctl.close()
its, ncs = np.array(its), np.array(ncs)
for legs in (2, 4):
    sel = ncs[30:] == legs
    v = its[30:][sel]
    print(f"{legs} stance legs: {sel.sum() / (TICKS - 30):.0f} robots per tick, iterations mean {v.mean():.1f} p50 {np.percentile(v, 50):.0f} p90 {np.percentile(v, 90):.0f} "
          f"p99 {np.percentile(v, 99):.0f} p99.9 {np.percentile(v, 99.9):.0f} max {v.max()}")
    for q in (95, 99):
        stay, tot, prev_pct = 0, 0, []
        for k in range(31, TICKS):
            same = (ncs[k] == legs) & (ncs[k - 1] == legs)
            if same.sum() < 50:
                continue
            thr_now = np.percentile(its[k][ncs[k] == legs], q)
            thr_prev = np.percentile(its[k - 1][ncs[k - 1] == legs], q)
            slow_now = same & (its[k] >= thr_now)
            tot += slow_now.sum()
            stay += (slow_now & (its[k - 1] >= thr_prev)).sum()
            ranks = np.searchsorted(np.sort(its[k - 1][ncs[k - 1] == legs]), its[k - 1][slow_now]) / max(1, (ncs[k - 1] == legs).sum())
            prev_pct += ranks.tolist()
        print(f"   robots at or above the tick's p{q}: {tot} robot-ticks, {100.0 * stay / max(tot, 1):.0f} % were also above p{q} in the tick before "
              f"(chance: {100 - q} %); their median percentile in the tick before: {100 * np.median(prev_pct):.0f}")
    # correlation of consecutive iteration counts
    xs, ys = [], []
    for k in range(31, TICKS):
        same = (ncs[k] == legs) & (ncs[k - 1] == legs)
        xs += its[k - 1][same].tolist(); ys += its[k][same].tolist()
    print(f"   correlation of a robot's iteration count with its count one tick earlier: {np.corrcoef(xs, ys)[0, 1]:.2f}")
print("max over robots per tick (ticks 30..):", its[30:].max(axis=1).tolist())
