"""Study (GPU, needs a library built with per-robot wall_clock64 stamps written into the optional grf output -- see DESIGN.md
section 5 -- as scratch/librg_mpc_stamp.so): the actual schedule of the fused launch on the 2048 wave slots.  Outcome: every
slot gets exactly two jobs; the launch ends at (longest first-round job) + (cheapest job), 112 + 58 us, against 134 us of
mean slot load."""

import sys, os
os.environ["RG_MPC_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "librg_mpc_stamp.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, json
import bench
from robot_gym_amd.core.config import MPCConfig
from robot_gym_amd.controllers.mpc.batched import BatchedMPCController
B = 4096
torch.cuda.set_device(0); device = torch.device("cuda", 0)
over = json.loads(os.environ.get("RG_OVER", "{}"))
cfg = MPCConfig.for_robot("ghost", horizon=10, **over)
state, cmd, t_off, slabs = bench.make_input_ring(cfg, B, 0, device, 50, 0.1)
ctl = BatchedMPCController(B, cfg, device=device, extra_outputs=True)
ctl.reset_at(-t_off); ctl.update_controller_params(torch.from_numpy(cmd.T.copy()).to(device))
recs = []
for k in range(60):
    ctl.get_action(0.01 * k, slabs[k % 50]); torch.cuda.synchronize()
    if k >= 40:
        g = ctl.extra["grf"].cpu().numpy().reshape(B, 12).astype(np.float64)
        it, nc = ctl._handle.last_iterations(B, ctl._stream())
        recs.append(np.column_stack([g[:, :6], it, nc]))
np.save("gpurun_out/stamp4.npy", np.array(recs))
r = np.array(recs)
for k in range(3):
    t0, t1 = r[k, :, 0], r[k, :, 1]
    base = t0.min(); t0 = (t0 - base) % (1 << 24); t1 = (t1 - base) % (1 << 24)
    print(f"tick {k}: kernel span {t1.max() * 0.01:.1f} us; first-round starts p50 {np.percentile(t0, 25) * 0.01:.1f}; last start {t0.max() * 0.01:.1f}; mean dur {np.mean(t1 - t0) * 0.01:.1f} us; sum dur/2048 {np.sum(t1 - t0) * 0.01 / 2048:.1f}")
