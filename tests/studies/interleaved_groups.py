"""Study (GPU): the batch split into G independent groups, each with its own handle and HIP stream, all stepped every tick --
the tails and kernel boundaries of one group's launches overlap the other groups' work.  Robots are independent, so
this is the single-GPU version of the multi-GPU sharding.  Reports controller steps/s over `ticks` ticks for G = 1, 2, 4.
    python tests/studies/interleaved_groups.py [total batch] [ticks] ['{"horizon": 20}' ...]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np   # noqa: E402
import torch   # noqa: E402
import bench   # noqa: E402
from robot_gym_amd.core.config import MPCConfig   # noqa: E402
from robot_gym_amd.controllers.mpc.batched import BatchedMPCController   # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 300
overs = [json.loads(a) for a in sys.argv[3:]] or [{}]
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
for over in overs:
    fixed = over.pop("fixed_cmd", None)
    cfg = MPCConfig.for_robot("ghost", **over)
    for G in (1, 2, 4, 1, 2):
        if B % G:
            continue
        Bg = B // G
        groups = []
        for g in range(G):
            state, cmd, t_off, slabs = bench.make_input_ring(cfg, Bg, g, dev, 50, 0.1, (0.3, 0.0, 0.0) if fixed else None)
            stream = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(stream):
                ctl = BatchedMPCController(Bg, cfg, device=dev, extra_outputs=False)
                ctl.reset_at(-t_off)
                ctl.update_controller_params(torch.from_numpy(cmd.T.copy()).to(dev))
            groups.append((ctl, slabs, stream))
        torch.cuda.synchronize()

        def run(n, k0):
            for k in range(n):
                for ctl, slabs, stream in groups:
                    with torch.cuda.stream(stream):
                        ctl.get_action(0.01 * (k0 + k), slabs[(k0 + k) % 50])
        run(30, 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(ticks, 30)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        print(f"{over} fixed_cmd={bool(fixed)} total batch {B}: {G} group(s) of {Bg}: {B * ticks / el / 1e6:8.3f} M steps/s  tick {el / ticks * 1e6:7.1f} us")
        for ctl, _, _ in groups:
            ctl.close()
