"""Study (GPU): how often BASELINE configs[4] hands robots to the exact re-solve launch in steady state, and what those ticks cost
(per-tick solver statistics, the host waits for every tick).  Round 6: a quarter of the ticks, 1-3 robots, +260 us on those ticks."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, bench
from robot_gym_amd.core.config import MPCConfig
from robot_gym_amd.controllers.mpc.batched import BatchedMPCController
from robot_gym_amd import synthetic
B = 4096
device = torch.device("cuda", 0)
cfg = MPCConfig.for_robot("ghost", horizon=20, contact_lookahead=1)
gait = synthetic.random_gaits(B, cfg, seed=0)
state, cmd, t_off, slabs = bench.make_input_ring(cfg, B, 0, device, 50, 0.1, None, gait, True)
ctl = BatchedMPCController(B, cfg, device=device, extra_outputs=False)
ctl.set_gait(**gait)
ctl.reset_at(-t_off)
ctl.update_controller_params(torch.from_numpy(cmd.T.copy()).to(device))
rows = []
for k in range(120):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctl.get_action(0.01 * k, slabs[k % 50])
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    s = ctl.solver_stats()
    rows.append((k, round(dt * 1e6), s["retried_exact"], s["iters_max"], s["failures"]))
print("tick us retried iters_max failures")
for r in rows[:12] + rows[40:120:4]:
    print(*r)
ret = np.array([r[2] for r in rows[20:]]); us = np.array([r[1] for r in rows[20:]])
print("steady: mean tick", us.mean(), "ticks with retries", int((ret > 0).sum()), "of", len(ret), "mean retried", ret.mean(), "tick us with/without", us[ret > 0].mean() if (ret > 0).any() else None, us[ret == 0].mean() if (ret == 0).any() else None)
