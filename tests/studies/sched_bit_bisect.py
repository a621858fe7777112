"""Study (GPU): which (step, leg) entries of a caller contact schedule a GPU-vs-oracle difference depends on.  Takes a case of
the configuration sweep, a robot and a tick; clears (or sets) one schedule bit of that robot at that tick at a time, on both
sides, and prints the per-joint torque error of the robot at the tick.
    python tests/studies/sched_bit_bisect.py <seed> <robot> <tick>"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as O   # noqa: E402
from tests import helpers   # noqa: E402
from tests.test_gpu_parity import _sweep_case   # noqa: E402

seed, rb, tk = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
cfg, B, over, kw = _sweep_case(seed)
base_fn = kw["sched_fn"]
H = cfg.horizon


def err_with(flip):
    def fn(k, t_rel):
        s = np.array(base_fn(k, t_rel), copy=True)
        if k == tk and flip is not None:
            s[flip[1], rb] = int(s[flip[1], rb]) ^ (1 << flip[0])
        return s
    kw2 = dict(kw, sched_fn=fn, ticks=tk + 1)
    o = helpers.run_oracle(O, cfg, **kw2)[tk]
    g = helpers.run_gpu(cfg, **kw2)[tk]
    a_g = g["action"].reshape(B, 12, 5)[rb, :, 4].astype(np.float64)
    a_o = o["action"].reshape(B, 12, 5)[rb, :, 4].astype(np.float64)
    return (np.abs(a_g - a_o) / np.maximum(np.abs(a_o), 1.0)).max(), np.abs(a_g - a_o).max()


s0 = np.array(base_fn(tk, 0.01 * tk + kw["t_off"]))[:, rb]
print("schedule rows (step 0 first):", [format(int(x) & ((1 << H) - 1), f"0{H}b")[::-1] for x in s0])
print("unmodified: per-joint error %.2e (abs %.2e)" % err_with(None))
for step in range(1, H):
    for leg in range(4):
        e = err_with((step, leg))
        print(f"flip step {step:2d} leg {leg} (was {(int(s0[leg]) >> step) & 1}): per-joint error {e[0]:.2e} abs {e[1]:.2e}")
