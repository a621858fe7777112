"""Study (GPU; needs the measurement build `make -C robot_gym_amd/csrc stamp`): the actual schedule of the QP launch on the
2048 wave slots, from per-robot start / end clock stamps (100 MHz).  Per solver plan: kernel span, mean slot load, and the
job durations by stance-leg count and solver work (iterations).
    python tests/studies/fused_launch_schedule.py [batch] ['{"solver": 2}' ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RG_MPC_LIB"] = os.path.join(ROOT, "robot_gym_amd", "csrc", "librg_mpc_stamp.so")
sys.path.insert(0, ROOT)
import numpy as np   # noqa: E402
import torch   # noqa: E402
import bench   # noqa: E402
from robot_gym_amd.core.config import MPCConfig   # noqa: E402
from robot_gym_amd.controllers.mpc.batched import BatchedMPCController   # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
overs = [json.loads(a) for a in sys.argv[2:]] or [{}]
torch.cuda.set_device(0)
device = torch.device("cuda", 0)
for over in overs:
    fixed = over.pop("fixed_cmd", None)
    rand = over.pop("random_schedule", None)
    cfg = MPCConfig.for_robot("ghost", **{"horizon": 10, **over})
    gait = None
    if rand:   # BASELINE config 5: per-robot duty factors and a caller-supplied contact schedule
        from robot_gym_amd import synthetic
        gait = synthetic.random_gaits(B, cfg, seed=0)
    state, cmd, t_off, slabs = bench.make_input_ring(cfg, B, 0, device, 50, 0.1, (0.3, 0.0, 0.0) if fixed else None, gait, bool(rand))
    ctl = BatchedMPCController(B, cfg, device=device, extra_outputs=True)
    if gait is not None:
        ctl.set_gait(**gait)
    ctl.reset_at(-t_off)
    ctl.update_controller_params(torch.from_numpy(cmd.T.copy()).to(device))
    recs = []
    for k in range(60):
        ctl.get_action(0.01 * k, slabs[k % 50])
        torch.cuda.synchronize()
        if k >= 40:
            g = ctl.extra["grf"].cpu().numpy().reshape(B, 12).astype(np.float64)
            it, nc = ctl._handle.last_iterations(B, ctl._stream())
            recs.append(np.column_stack([g[:, :2], it, nc, g[:, 2:8]]))
    r = np.array(recs)
    print(f"== {over} fixed_cmd={bool(fixed)} batch {B}")
    spans = []
    for k in range(r.shape[0]):
        t0, t1 = r[k, :, 0], r[k, :, 1]
        base = t0.min()
        t0, t1 = (t0 - base) % (1 << 24), (t1 - base) % (1 << 24)
        spans.append((t1.max() * 0.01, np.sum(t1 - t0) * 0.01 / (2048 if cfg.horizon == 10 else 512), np.percentile(t0, 99) * 0.01))
    sp = np.array(spans)
    print(f"kernel span {sp[:, 0].mean():.1f} us (max {sp[:, 0].max():.1f}); mean slot load {sp[:, 1].mean():.1f} us; p99 job start {sp[:, 2].mean():.1f} us")
    dur = ((r[:, :, 1] - r[:, :, 0]) % (1 << 24)) * 0.01
    start = r[:, :, 0]
    for nc in (1, 2, 3, 4):
        m = r[:, :, 3] == nc
        if not m.any():
            continue
        d, it = dur[m], r[:, :, 2][m]
        line = f"  nc={nc}: n/tick {m.sum() / r.shape[0]:.0f} dur mean {d.mean():.1f} p50 {np.median(d):.1f} p90 {np.percentile(d, 90):.1f} p99 {np.percentile(d, 99):.1f} max {d.max():.1f} | iters mean {it.mean():.1f} max {it.max():.0f}"
        if len(d) > 100:   # duration against solver work
            A = np.column_stack([np.ones(len(it)), it])
            coef = np.linalg.lstsq(A, d, rcond=None)[0]
            line += f" | dur ~ {coef[0]:.1f} + {coef[1]:.2f} * iters"
        print(line)
        # phases of the body (stamp_phase): time from the job's start to the end of each phase
        ph = ((r[:, :, 4:8] - r[:, :, 0:1]) % (1 << 24))[m] * 0.01
        ok = (r[:, :, 4:8][m] > 0).all(1)
        if ok.sum() > 10:
            e = ph[ok].mean(0)
            print(f"      phases (mean us): record+tables {e[0]:.1f} | tile build {e[1] - e[0]:.1f} | sweep {e[2] - e[1]:.1f} | solve {e[3] - e[2]:.1f} | outputs {d[ok].mean() - e[3]:.1f}   (n {ok.sum()})")
        for lo, hi in ((0, 0), (1, 2), (3, 5), (6, 10), (11, 20), (21, 40), (41, 80), (81, 1000)):
            mm = (it >= lo) & (it <= hi)
            if mm.sum() >= 3:
                print(f"      iters {lo:3d}..{hi:4d}: n {mm.sum():6d}  dur mean {d[mm].mean():6.1f} max {d[mm].max():6.1f}")
    # solver work of robots that run their body from a cold start (another stance-leg count than in the tick before: no stored
    # iterate / working set) against robots continuing on the same body
    for nc in (2, 4):
        cur, prev = r[1:, :, 3], r[:-1, :, 3]
        m_w, m_c = (cur == nc) & (prev == nc), (cur == nc) & (prev != nc)
        itn, dn = r[1:, :, 2], dur[1:]
        for name, mm in (("same body as the tick before", m_w), ("body changed (cold start)", m_c)):
            if mm.sum() > 5:
                print(f"  nc={nc} {name:32s}: n/tick {mm.sum() / (r.shape[0] - 1):6.0f}  work mean {itn[mm].mean():6.1f} p90 {np.percentile(itn[mm], 90):5.0f} p99 {np.percentile(itn[mm], 99):5.0f} max {itn[mm].max():4.0f} | dur mean {dn[mm].mean():6.1f} p99 {np.percentile(dn[mm], 99):6.1f} max {dn[mm].max():6.1f}")
    # who ends the launch: the latest-finishing jobs of the last ticks (start, duration, stance legs, solver work now and in the
    # tick before -- what the front kernel predicted the cost class from)
    for k in range(r.shape[0] - 3, r.shape[0]):
        t0, t1 = r[k, :, 0], r[k, :, 1]
        base = t0.min()
        t0, t1 = ((t0 - base) % (1 << 24)) * 0.01, ((t1 - base) % (1 << 24)) * 0.01
        order = np.argsort(-t1)[:8]
        print(f"  tick {k}: span {t1.max():.1f}; last finishers: " + "; ".join(f"nc{int(r[k, b, 3])} start {t0[b]:.0f} dur {t1[b] - t0[b]:.0f} it {int(r[k, b, 2])} (prev {int(r[k - 1, b, 2])}, prev nc{int(r[k - 1, b, 3])})" for b in order))
        busy = np.array([((t0 <= t) & (t1 > t)).sum() for t in np.arange(0, t1.max(), 10.0)])
        print("      jobs in flight every 10 us:", busy.tolist())
    ctl.close()
