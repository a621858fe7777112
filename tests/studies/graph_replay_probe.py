"""Study (GPU): does replaying a tick (front, QP, re-solve launch) as a captured hipGraph shorten it?  Round 6, MI355X:
batch 4096: plain 130.2 us per tick, graph replay 130.0; batch 1: plain 34.8 us, graph replay 60.7 -- no: the launches are not
what the tick waits for, and a graph launch costs this runtime more than three kernel launches.  (Audit lane off: a capture
cannot span the side streams.)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, bench
from robot_gym_amd.core.config import MPCConfig
from robot_gym_amd.controllers.mpc.batched import BatchedMPCController
for B in (4096, 1):
    device = torch.device("cuda", 0)
    cfg = MPCConfig.for_robot("ghost", audit_k=0)
    state, cmd, t_off, slabs = bench.make_input_ring(cfg, B, 0, device, 50, 0.1, None, None, False)
    ctl = BatchedMPCController(B, cfg, device=device, extra_outputs=False)
    ctl.reset_at(-t_off)
    ctl.update_controller_params(torch.from_numpy(cmd.T.copy()).to(device))
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for k in range(30):
            ctl.get_action(0.01 * k, slabs[k % 50])
        s.synchronize()
        def timed(fn, n=300):
            s.synchronize(); t0 = time.perf_counter()
            for k in range(n): fn(k)
            s.synchronize(); return (time.perf_counter() - t0) / n * 1e6
        plain = timed(lambda k: ctl.get_action(0.3 + 0.01 * k, slabs[k % 50]))
        plain_same = timed(lambda k: ctl.get_action(0.3, slabs[0]))
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g, stream=s):
                ctl.get_action(0.3, slabs[0])
                ctl.get_action(0.31, slabs[1])   # two ticks per graph: the double-buffered counters come back to where they were
            gr = timed(lambda k: g.replay(), 150) / 2
            print(f"batch {B}: plain {plain:.1f} us/tick, plain same-args {plain_same:.1f}, graph replay {gr:.1f} us/tick")
        except Exception as e:
            print(f"batch {B}: plain {plain:.1f} us/tick; capture failed: {type(e).__name__}: {str(e)[:300]}")
