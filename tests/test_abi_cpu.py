"""The C-ABI library loads and exports every symbol include/rg_mpc.h declares; struct layouts of the
ctypes binding match the header; configuration errors are reported, never thrown.  No GPU needed."""
import ctypes as C
import os
import re

import pytest
import torch

from robot_gym_amd.core import mpc_abi
from robot_gym_amd.core.config import MPCConfig

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rg_mpc.h")


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rg_mpc_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = mpc_abi.load_library()
    declared = _declared_functions()
    assert len(declared) >= 12
    for name in declared:
        assert hasattr(lib, name), f"librg_mpc.so lacks {name}"
    assert sorted(mpc_abi.EXPORTS) == declared


def test_struct_layouts_match_header():
    lib = mpc_abi.load_library()
    assert lib.rg_mpc_abi_version() == mpc_abi.ABI_VERSION == 5
    assert lib.rg_mpc_config_size() == C.sizeof(mpc_abi.CConfig)
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    bodies = {name: body for body, name in re.findall(r"typedef struct \{([^{}]*)\} (\w+);", src)}
    for struct, cls in (("rg_mpc_state_ptrs", mpc_abi.CStatePtrs), ("rg_mpc_out_ptrs", mpc_abi.COutPtrs), ("rg_mpc_config", mpc_abi.CConfig)):
        body = bodies[struct]
        names = re.findall(r"\b\*?\s*([a-z_0-9]+)(?:\[\d+\])?\s*;", body)
        assert names == [n for n, _ in cls._fields_], struct


def test_config_round_trip_and_validation():
    cfg = MPCConfig.for_robot("ghost")
    cc = mpc_abi.make_cconfig(cfg)
    assert cc.horizon == 10 and cc.window == 20 and abs(cc.mass - 190 / 9.8) < 1e-15
    assert list(cc.init_state) == [0, 1, 1, 0] and list(cc.hip)[:3] == [0.22, -0.1, 0.0]
    assert cc.jaxis[0] == 1.0 and abs(cc.jrpy[0] - 1.57079) < 1e-12  # FR hip joint of ghost.urdf
    with pytest.raises(ValueError):
        mpc_abi.make_cconfig(MPCConfig.for_robot("ghost", hip=(0.0,) * 11))


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU error path")
def test_create_fails_loudly_without_gpu():
    with pytest.raises(mpc_abi.RgMpcError) as e:
        mpc_abi.MpcHandle(MPCConfig.for_robot("ghost"), 8)
    assert e.value.status == -3 and "HIP device" in str(e.value)
    from robot_gym_amd.controllers.mpc.batched import BatchedMPCController
    with pytest.raises(RuntimeError):
        BatchedMPCController(8)


def test_create_rejects_bad_config_before_touching_the_gpu():
    lib = mpc_abi.load_library()
    # horizons: only 10 and 20 have compiled, GPU-tested solver bodies -- everything else must be refused here, never
    # accepted and then silently not solved (round-1 finding: horizons 11-19 could leave a stance-leg bin unlaunched)
    for bad in (dict(horizon=0), dict(horizon=21), dict(horizon=5), dict(horizon=12), dict(horizon=15), dict(horizon=16),
                dict(mu=(0.45, 0.45, -0.4, 0.45)), dict(mu=(0.0, 0.45, 0.45, 0.45)), dict(window=0), dict(kin_mode=2),
                dict(solver=7), dict(solver=4), dict(admm_rho34_scale=0.0), dict(admm_rho_sched_scale=-1.0), dict(audit_k=1 << 27), dict(admm_relax=2.5), dict(motor_dir=(0.5,) * 12), dict(inertia=(0.0,) * 9),
                dict(solver=1, horizon=20), dict(solver=1, contact_lookahead=1, horizon=20), dict(contact_lookahead=1, horizon=12),
                dict(reserved0=1), dict(reserved0=32), dict(reserved2=1), dict(reserved3=1), dict(conv_friction_rows=2), dict(conv_friction_rows=1, mu=(0.3, 0.45, 0.6, 0.45)), dict(lane_grid=3), dict(lane_grid=-1),
                dict(conv_alpha_doubled=2), dict(conv_feet_rotation=-1), dict(conv_com_height=2), dict(conv_first_latch=5), dict(conv_window_divide=2), dict(audit_k=-1), dict(audit_k=17), dict(audit_tol=0.0),
                dict(accel_cos2=1.5), dict(accel_rmin=0.99), dict(accel_rate_cap=1.0), dict(admm_tol=-1.0), dict(admm_check=0), dict(admm_accel=-1)):
        cc = mpc_abi.make_cconfig(MPCConfig.for_robot("ghost", **bad))
        h = C.c_void_p()
        rc = lib.rg_mpc_create(C.byref(cc), 4, 0, C.byref(h))
        assert rc == -1 and not h.value, bad
        assert lib.rg_mpc_last_error(None)
    assert lib.rg_mpc_create(None, 4, 0, C.byref(C.c_void_p())) == -1
    cc = mpc_abi.make_cconfig(MPCConfig.for_robot("ghost"))
    assert lib.rg_mpc_create(C.byref(cc), (1 << 24) + 1, 0, C.byref(C.c_void_p())) == -1   # work-list entries pack robot | legs << 24
    assert b"2^24" in lib.rg_mpc_last_error(None)


def test_product_path_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "robot_gym_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), os.path.join(dirpath, f)
                assert "libmpc_oracle" not in txt and "mpc_oracle.h" not in txt, os.path.join(dirpath, f)
