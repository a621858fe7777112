import sys, numpy as np, torch, time
sys.path.insert(0, '/root/repo')
from robot_gym_amd.core.config import MPCConfig
from robot_gym_amd.controllers.mpc.batched import BatchedMPCController
from robot_gym_amd import synthetic
import bench
H = int(sys.argv[1]); solver = int(sys.argv[2]); cap = int(sys.argv[3]); rho = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-4
rho2 = float(sys.argv[5]) if len(sys.argv) > 5 else 0.0; sw = int(sys.argv[6]) if len(sys.argv) > 6 else 100
cfg = MPCConfig.for_robot("ghost", horizon=H, contact_lookahead=1, solver=solver, admm_iters=cap, admm_rho=rho, admm_rho2=rho2, admm_switch=sw)
B = 4096
dev = torch.device("cuda:0")
gait = synthetic.random_gaits(B, cfg, seed=0)
state, cmd, t_off, slabs = bench.make_input_ring(cfg, B, 0, dev, 8, 0.1, None, gait, True)
ctl = BatchedMPCController(B, cfg, device=dev, extra_outputs=False)
ctl.set_gait(**gait); ctl.reset_at(-t_off); ctl.update_controller_params(torch.from_numpy(cmd.T.copy()).to(dev))
for k in range(24):
    ctl.get_action(0.01 * k, slabs[k % 8])
torch.cuda.synchronize()
it, nc = ctl._handle.last_iterations(B, ctl._stream())
st = ctl.solver_stats()
it = it[nc > 0]
print("H", H, "solver", solver, "cap", cap, "rho", rho, "rho2", rho2, "switch", sw, st)
print("percentiles 50/90/95/99/max:", [int(np.percentile(it, p)) for p in (50, 90, 95, 99, 100)], "frac > 300:", float((it > 300).mean()), "frac>=cap", float((it >= cap).mean()))
t0 = time.perf_counter()
for k in range(20):
    ctl.get_action(0.01 * (24 + k), slabs[k % 8])
torch.cuda.synchronize()
print("ms/tick", (time.perf_counter() - t0) / 20 * 1e3)
