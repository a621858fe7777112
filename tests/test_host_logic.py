"""Host-side logic that needs no GPU: sharding, the world_size-2 action all-gather over gloo,
synthetic-state generator determinism."""
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from robot_gym_amd.core.config import MPCConfig
from robot_gym_amd.core.sharding import shard_bounds, all_gather_actions
from robot_gym_amd import synthetic


def test_shard_bounds_partition_the_batch():
    for total, world in ((32768, 8), (4097, 8), (7, 8), (4096, 1), (10, 3)):
        edges = [shard_bounds(total, r, world) for r in range(world)]
        assert edges[0][0] == 0 and edges[-1][1] == total
        assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
        sizes = [hi - lo for lo, hi in edges]
        assert max(sizes) - min(sizes) <= 1
    assert shard_bounds(32768, 3, 8) == (12288, 16384)  # BASELINE config 4: 8 x 4096


def _gather_worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_bounds(64, rank, world)
    full = torch.arange(64 * 60, dtype=torch.float32).view(64, 60)
    out = all_gather_actions(full[lo:hi].clone())
    ok = torch.equal(out, full)
    # max-over-ranks timing reduction used by bench.py
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ok = ok and t.item() == float(world)
    with open(os.path.join(tmp, f"r{rank}"), "w") as f:
        f.write("ok" if ok else "bad")
    dist.destroy_process_group()


def test_all_gather_actions_world_size_2_gloo(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_gather_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert [open(tmp_path / f"r{r}").read() for r in range(2)] == ["ok", "ok"]


def test_synthetic_states_are_deterministic_and_shaped():
    cfg = MPCConfig.for_robot("ghost")
    a, ca, ta = synthetic.make_states(128, cfg, seed=3)
    b, cb, tb = synthetic.make_states(128, cfg, seed=3)
    for k in a:
        assert np.array_equal(a[k], b[k])
    assert np.array_equal(ca, cb) and np.array_equal(ta, tb)
    assert a["rpy"].shape == (3, 128) and a["jac"].shape == (36, 128) and a["q"].dtype == np.float32
    assert np.abs(a["rpy"][:2]).max() <= 0.2 + 1e-6 and np.abs(ca[0]).max() <= 0.35 + 1e-6
    c = synthetic.gait_consistent_contacts(cfg, ta, np.zeros((4, 128), dtype=bool))
    assert set(np.unique(c.sum(0))) <= {2, 4}  # trot: a diagonal pair or all four
    fixed, cf, _ = synthetic.make_states(16, cfg, seed=0, fixed_cmd=(0.3, 0.0, 0.0))
    assert np.all(cf[0] == np.float32(0.3)) and np.all(cf[1:] == 0)


def test_packed_state_views_share_one_slab():
    """PackedState: every field is a contiguous view into ONE [77, B] slab (host and device side), contact as int32."""
    import torch
    from robot_gym_amd.controllers.mpc.batched import PackedState, STATE_FIELDS
    B = 5
    ps = PackedState(B, torch.device("cpu"), pin=False)
    assert ps.host_slab.shape == (sum(c for _, c, _ in STATE_FIELDS), B) == (77, B)
    row = 0
    for name, comps, dt in STATE_FIELDS:
        h = ps.host[name]
        assert h.shape == (comps, B) and h.dtype == dt and h.is_contiguous()
        assert h.data_ptr() == ps.host_slab[row].data_ptr()
        row += comps
    ps.host["contact"][:] = torch.arange(4 * B, dtype=torch.int32).reshape(4, B)
    ps.host["rpy"][:] = 1.5
    dev = ps.upload()
    assert torch.equal(dev["contact"], ps.host["contact"]) and torch.equal(dev["rpy"], ps.host["rpy"])
    assert dev["jac"].data_ptr() == ps.dev_slab[37].data_ptr()
