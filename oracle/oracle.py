"""ctypes binding of oracle/_build/libmpc_oracle.so (float64 C restatement).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by robot_gym_amd/.  PARITY UNPINNED for [UPSTREAM-RECALL] parts.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libmpc_oracle.so")

WINDOW_MAX = 64
d = C.c_double
i32 = C.c_int


class Config(C.Structure):
    _fields_ = [
        ("horizon", i32), ("dt_plan", d), ("mass", d), ("inertia", d * 9), ("body_height", d),
        ("weights", d * 13), ("alpha", d), ("mu", d * 4), ("fz_max_scale", d), ("fz_min_scale", d),
        ("gravity", d),
        ("stance_duration", d * 4), ("duty_factor", d * 4), ("init_phase", d * 4), ("init_state", i32 * 4),
        ("contact_phase_thresh", d), ("window", i32),
        ("foot_clearance", d), ("swing_kp", d * 3), ("max_clearance", d), ("hip", (d * 3) * 4),
        ("motor_kp", d * 12), ("motor_kd", d * 12), ("motor_dir", d * 12), ("motor_off", d * 12),
        ("jxyz", ((d * 3) * 3) * 4), ("jrpy", ((d * 3) * 3) * 4), ("jaxis", ((d * 3) * 3) * 4),
        ("toe_xyz", (d * 3) * 4), ("toe_com", (d * 3) * 4), ("base_com", d * 3),
        ("ik_iters", i32), ("ik_damping", d), ("ik_max_step", d), ("kin_mode", i32), ("contact_lookahead", i32),
        ("conv_alpha_doubled", i32), ("conv_feet_rotation", i32), ("conv_com_height", i32), ("conv_first_latch", i32), ("conv_window_divide", i32), ("conv_friction_rows", i32),
    ]


class State(C.Structure):
    _fields_ = [
        ("reset_time", d), ("need_latch", i32), ("first_update", i32), ("last_desired", i32 * 4),
        ("desired", i32 * 4), ("leg_state", i32 * 4), ("phase", d * 4),
        ("ring", (d * WINDOW_MAX) * 3), ("ring_len", i32), ("ring_head", i32), ("fsum", d * 3), ("fcorr", d * 3),
        ("v_body", d * 3), ("latched", (d * 3) * 4), ("swing_q", d * 12), ("swing_valid", i32 * 12),
    ]


class Input(C.Structure):
    _fields_ = [
        ("rpy", d * 3), ("rpy_rate", d * 3), ("v_world", d * 3), ("quat", d * 4), ("q", d * 12),
        ("foot_pos", (d * 3) * 4), ("jac", ((d * 3) * 3) * 4), ("contact", i32 * 4), ("cmd", d * 3),
        ("sched_valid", i32), ("sched", i32 * 4),
    ]


class Output(C.Structure):
    _fields_ = [
        ("action", C.c_float * 60), ("grf", d * 12), ("tau", d * 12), ("desired", i32 * 4), ("leg_state", i32 * 4),
        ("phase", d * 4), ("v_body", d * 3), ("foot_target", (d * 3) * 4), ("qp_iters", i32), ("kkt", d * 3),
    ]


INPUT_DTYPE = np.dtype([
    ("rpy", "f8", 3), ("rpy_rate", "f8", 3), ("v_world", "f8", 3), ("quat", "f8", 4), ("q", "f8", 12),
    ("foot_pos", "f8", (4, 3)), ("jac", "f8", (4, 3, 3)), ("contact", "i4", 4), ("cmd", "f8", 3),
    ("sched_valid", "i4"), ("sched", "i4", 4)], align=True)
OUTPUT_DTYPE = np.dtype([
    ("action", "f4", 60), ("grf", "f8", 12), ("tau", "f8", 12), ("desired", "i4", 4), ("leg_state", "i4", 4),
    ("phase", "f8", 4), ("v_body", "f8", 3), ("foot_target", "f8", (4, 3)), ("qp_iters", "i4"), ("kkt", "f8", 3)],
    align=True)

_lib = None


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(
            os.path.getmtime(os.path.join(_HERE, f)) for f in ("mpc_oracle.c", "mpc_oracle.h")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.orc_default_config.argtypes = [C.POINTER(Config)]
        L.orc_reset.argtypes = [C.POINTER(Config), C.POINTER(State), d, C.c_void_p]
        L.orc_step.argtypes = [C.POINTER(Config), C.POINTER(State), d, C.POINTER(Input), C.POINTER(Output)]
        L.orc_step.restype = i32
        L.orc_set_qp_mode.argtypes = [i32, i32, d, d]
        L.orc_set_qp_mode.restype = None
        L.orc_step_batch.argtypes = [C.POINTER(Config), C.c_void_p, i32, d, C.c_void_p, C.c_void_p, i32]
        L.orc_step_batch.restype = i32
        L.orc_step_batch_cfgs.argtypes = [C.c_void_p, C.c_void_p, i32, d, C.c_void_p, C.c_void_p, i32]
        L.orc_step_batch_cfgs.restype = i32
        L.orc_gait.argtypes = [C.POINTER(Config), d, C.POINTER(i32 * 4), C.POINTER(i32 * 4), C.POINTER(i32 * 4), C.POINTER(d * 4)]
        L.orc_mpc_build.argtypes = [C.POINTER(Config)] + [C.c_void_p] * 4 + [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_mpc_build.restype = i32
        L.orc_qp_solve.argtypes = [i32, C.c_void_p, C.c_void_p, C.c_void_p, d, d, C.c_void_p, C.c_void_p]
        L.orc_qp_solve.restype = i32
        L.orc_qp_solve_rows.argtypes = [i32, C.c_void_p, C.c_void_p, C.c_void_p, d, d, C.c_void_p, C.c_void_p]
        L.orc_qp_solve_rows.restype = i32
        L.orc_leg_fk.argtypes = [C.POINTER(Config), i32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_leg_ik.argtypes = [C.POINTER(Config), i32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_leg_ik.restype = i32
        L.orc_swing_trajectory.argtypes = [d, C.c_void_p, C.c_void_p, d, C.c_void_p]
        L.orc_hybrid_to_torque.argtypes = [C.c_void_p] * 4
        L.orc_force_to_torque.argtypes = [C.POINTER(Config), i32, C.c_void_p, C.c_void_p, C.c_void_p]
        assert C.sizeof(Input) == INPUT_DTYPE.itemsize, (C.sizeof(Input), INPUT_DTYPE.itemsize)
        assert C.sizeof(Output) == OUTPUT_DTYPE.itemsize, (C.sizeof(Output), OUTPUT_DTYPE.itemsize)
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def default_config():
    c = Config()
    lib().orc_default_config(C.byref(c))
    return c


def config_from_dict(cfgd):
    """Fill a Config from the plain dict produced by robot_gym_amd.core.config.MPCConfig.to_dict()."""
    c = default_config()
    for k, v in cfgd.items():
        if not hasattr(c, k):
            continue
        cur = getattr(c, k)
        if isinstance(cur, (int, float)):
            setattr(c, k, type(cur)(v))
        else:
            arr = np.asarray(v, dtype=np.float64 if "double" in type(cur).__name__ or True else None)
            flat = np.ctypeslib.as_array(cur)
            if flat.dtype.kind == "i":
                flat[...] = np.asarray(v, dtype=np.int32).reshape(flat.shape)
            else:
                flat[...] = arr.reshape(flat.shape)
    return c


class OracleBatch:
    """B independent oracle controllers (one orc_state each), stepped with OpenMP.  `gait` (optional) gives every robot
    its own gait timing: dict of [4,B] arrays stance_duration / duty_factor / init_phase (/ init_state) -- each robot then
    runs on its own copy of the config, like B separately constructed reference controllers would."""

    def __init__(self, cfg, B, t0=0.0, nthreads=0, gait=None):
        self.cfg = cfg
        self.B = B
        self.states = (State * B)()
        self.nthreads = nthreads
        self.cfgs = None
        if gait is not None:
            self.cfgs = (Config * B)()
            for b in range(B):
                C.memmove(C.byref(self.cfgs[b]), C.byref(cfg), C.sizeof(Config))
                for l in range(4):
                    self.cfgs[b].stance_duration[l] = float(gait["stance_duration"][l][b])
                    self.cfgs[b].duty_factor[l] = float(gait["duty_factor"][l][b])
                    self.cfgs[b].init_phase[l] = float(gait["init_phase"][l][b])
                    if gait.get("init_state") is not None:
                        self.cfgs[b].init_state[l] = int(gait["init_state"][l][b])
        for b in range(B):
            lib().orc_reset(C.byref(self._cfg_of(b)), C.byref(self.states[b]), t0, None)

    def _cfg_of(self, b):
        return self.cfg if self.cfgs is None else self.cfgs[b]

    def reset(self, idx, t0, foot_pos=None):
        for k, b in enumerate(idx):
            fp = None if foot_pos is None else _p(np.ascontiguousarray(foot_pos[k], dtype=np.float64))
            lib().orc_reset(C.byref(self._cfg_of(int(b))), C.byref(self.states[int(b)]), t0, fp)

    def step(self, t, inputs):
        """inputs: structured array of INPUT_DTYPE, shape [B]. Returns OUTPUT_DTYPE array."""
        assert inputs.dtype == INPUT_DTYPE and inputs.shape == (self.B,)
        inputs = np.ascontiguousarray(inputs)
        out = np.zeros(self.B, dtype=OUTPUT_DTYPE)
        if self.cfgs is not None:
            bad = lib().orc_step_batch_cfgs(C.addressof(self.cfgs), C.addressof(self.states), self.B, float(t), _p(inputs), _p(out), self.nthreads)
        else:
            bad = lib().orc_step_batch(C.byref(self.cfg), C.addressof(self.states), self.B, float(t), _p(inputs), _p(out), self.nthreads)
        if bad:
            raise RuntimeError(f"oracle QP failed for {bad} robots")
        return out


def mpc_build(cfg, rpy, omega, v_body, foot_pos, contact, cmd):
    H = cfg.horizon
    contact = np.ascontiguousarray(contact, dtype=np.int32)
    nc = int(contact.sum())
    n = 3 * nc * H
    P = np.zeros((max(n, 1), max(n, 1)))
    q = np.zeros(max(n, 1))
    legs = np.zeros(4, dtype=np.int32)
    Ad = np.zeros((13, 13))
    Bd = np.zeros((13, 12))
    f = lambda a: _p(np.ascontiguousarray(a, dtype=np.float64))
    rpy, omega, v_body, foot_pos, cmd = [np.ascontiguousarray(a, dtype=np.float64) for a in (rpy, omega, v_body, foot_pos, cmd)]
    lib().orc_mpc_build(C.byref(cfg), _p(rpy), _p(omega), _p(v_body), _p(foot_pos), _p(contact), _p(cmd), _p(P), _p(q), _p(legs), _p(Ad), _p(Bd))
    return P[:n, :n].copy(), q[:n].copy(), legs[:nc].copy(), Ad, Bd


def qp_solve(P, q, mu, fz_min, fz_max, mu_rows=None):
    """mu: one coefficient, or one per force block (3 variables each); mu_rows: instead, one per cone ROW (-fx, +fx, -fy, +fy)
    of every block (rg_mpc_config.conv_friction_rows)."""
    n = len(q)
    P = np.ascontiguousarray(P, dtype=np.float64)
    q = np.ascontiguousarray(q, dtype=np.float64)
    u = np.zeros(n)
    kkt = np.zeros(3)
    if mu_rows is not None:
        rows = np.ascontiguousarray(mu_rows, dtype=np.float64)
        assert rows.shape == (4,)
        it = lib().orc_qp_solve_rows(n, _p(P), _p(q), _p(rows), fz_min, fz_max, _p(u), _p(kkt))
        return u, it, kkt
    mu_blk = np.ascontiguousarray(np.broadcast_to(np.asarray(mu, dtype=np.float64), (n // 3,)))
    it = lib().orc_qp_solve(n, _p(P), _p(q), _p(mu_blk), fz_min, fz_max, _p(u), _p(kkt))
    return u, it, kkt


def set_qp_mode(mode=0, admm_iters=50, rho=1e-4, relax=1.8):
    """Process-wide QP solver of the oracle: 0 = exact dual active set (the parity oracle), 1 = fixed-count ADMM with dense
    linear algebra (bench.py's cpu_baseline variant B1 only)."""
    lib().orc_set_qp_mode(int(mode), int(admm_iters), float(rho), float(relax))
