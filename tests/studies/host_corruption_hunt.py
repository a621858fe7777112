"""Study: something writes a zero into a host input array of the sweep (seed 8, quat[3, 51]) in ~15 % of fresh processes.
Replays seeds 0..8 and checks the host arrays of the current seed after every stage (oracle run, controller creation, every
GPU tick, controller close) to find the stage at which the array changes."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O                   # noqa: E402
from tests import helpers                        # noqa: E402
from tests.test_gpu_parity import _sweep_case    # noqa: E402
from robot_gym_amd import synthetic              # noqa: E402
from robot_gym_amd.controllers.mpc.batched import BatchedMPCController   # noqa: E402


def check(tag, state, before):
    for k, v in state.items():
        if not np.array_equal(v, before[k]):
            idx = np.argwhere(v != before[k])
            print(f"CORRUPTION after {tag}: state[{k}] (addr {v.ctypes.data:#x}, {v.nbytes} B) changed at {idx.tolist()[:6]}: {before[k][tuple(idx[0])]} -> {v[tuple(idx[0])]}", flush=True)
            before[k] = v.copy()
            return True
    return False


for s in range(9):
    cfg, B, over, kw = _sweep_case(s)
    state, cmd, t_off, gait, sched_fn = kw["state"], kw["cmd"], kw["t_off"], kw["gait"], kw["sched_fn"]
    before = {k: v.copy() for k, v in state.items()}
    helpers.run_oracle(O, cfg, **kw)
    check(f"seed {s} oracle", state, before)
    ctl = BatchedMPCController(B, cfg, device="cuda:0")
    check(f"seed {s} create", state, before)
    if gait is not None:
        ctl.set_gait(**gait)
    ctl.reset_at(-t_off)
    ctl.update_controller_params(torch.from_numpy(cmd.T.copy()).to("cuda:0"))
    check(f"seed {s} reset/command", state, before)
    for k in range(kw["ticks"]):
        t = k * 0.01
        st = helpers.perturb(state, k, kw["jitter"])
        contact = synthetic.gait_consistent_contacts(cfg, t + t_off, state["_flip"], gait)
        dev = {n: torch.from_numpy(np.ascontiguousarray(st[n])).to("cuda:0") for n in ("rpy", "rpy_rate", "v_world", "quat", "q", "foot_pos", "jac")}
        dev["contact"] = torch.from_numpy(contact).to("cuda:0")
        if sched_fn:
            dev["contact_sched"] = torch.from_numpy(np.ascontiguousarray(sched_fn(k, t + t_off))).to("cuda:0")
        check(f"seed {s} tick {k} uploads", state, before)
        ctl._handle.debug_poison_lds(ctl._stream())
        act = ctl.get_action(t, dev)
        torch.cuda.synchronize()
        check(f"seed {s} tick {k} step", state, before)
        a = act.cpu().numpy().copy()
        extras = {kx: v.cpu().numpy().copy() for kx, v in ctl.extra.items()}
        check(f"seed {s} tick {k} downloads", state, before)
        ctl.bin_counts(); ctl.solver_stats(); ctl._handle.last_iterations(B, ctl._stream())
        check(f"seed {s} tick {k} stats", state, before)
    ctl.audit_stats()
    check(f"seed {s} audit_stats", state, before)
    ctl.close()
    check(f"seed {s} close", state, before)
print("done")
