#!/usr/bin/env python3
"""Host-side throughput of MPCVecEnv (GPU box): env steps/s of the whole wrapper tick -- the envs' own step() halves around
ONE rg_mpc_step -- with the fake envs of tests/fake_envs.py (no physics: this is the wrapper's own cost, the number a real
PyBullet env's step time adds to).  In-process (blocking=True) against worker processes (blocking=False), B = 256 / 1024 /
4096, next to bench.py's PCIe-inclusive controller rate for scale.
Usage: python tools/vec_env_bench.py [ticks] > profiles/r3_vec_env_host.txt"""
import functools
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robot_gym_amd.core.config import MPCConfig          # noqa: E402
from robot_gym_amd.gym.vec_env import MPCVecEnv          # noqa: E402
from tests.fake_envs import make_fake_env                # noqa: E402


def main():
    ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    cfg = MPCConfig.for_robot("ghost")
    cores = os.cpu_count()
    print(f"# MPCVecEnv wrapper throughput, fake envs (tests/fake_envs.py: SplitGoEnv = pre_step/post_step, FakeGoEnv = two-pass replay), {ticks} ticks, host cores {cores}")
    print("| B | mode | env steps/s | ms per tick |")
    print("|---|---|---|---|")
    for B in (256, 1024, 4096):
        rng = np.random.default_rng(0)
        actions = rng.uniform(-1, 1, (B, 2)).astype(np.float32)
        for kind in ("split", "go"):
            ctors = [functools.partial(make_fake_env, kind, "ghost", 0, B, b) for b in range(B)]
            modes = [("in-process", dict(envs=None))] + [(f"{w} worker processes", dict(blocking=False, workers=w)) for w in sorted({8, min(32, cores or 8)})]
            for name, kw in modes:
                if "envs" in kw:
                    env0 = ctors[0]()
                    state = env0.simulation.robot.state      # one shared synthetic state batch: build the others on it directly
                    from robot_gym_amd.controllers.mpc.slot_controller import BatchSlotController
                    envs = [type(env0)(cfg, state, b, BatchSlotController) for b in range(B)]
                    venv = MPCVecEnv(envs, config=cfg)
                else:
                    venv = MPCVecEnv(constructors=ctors, config=cfg, **kw)
                venv.reset()
                for _ in range(3):
                    venv.step(actions)
                t0 = time.perf_counter()
                for _ in range(ticks):
                    venv.step(actions)
                el = time.perf_counter() - t0
                venv.close()
                print(f"| {B} | {kind} envs, {name} | {B * ticks / el:,.0f} | {1e3 * el / ticks:.2f} |", flush=True)


if __name__ == "__main__":
    main()
