"""Known-answer and property tests of the CPU oracle (solver-independent where possible).
The reference ships no tests for this path, so these are the pins listed in
SURVEY.md 8c (4)-(7).  CPU only."""
import ctypes as C
import math

import numpy as np
import pytest
import scipy.linalg

from robot_gym_amd.core.config import MPCConfig
from robot_gym_amd.controllers.mpc.kinematics import ChainKinematics
from robot_gym_amd import synthetic
from tests import helpers

HIP = np.array([[0.22, -0.1, 0], [0.22, 0.1, 0], [-0.22, -0.1, 0], [-0.22, 0.1, 0]])


def _rand_state(rng):
    rpy = np.array([rng.uniform(-.2, .2), rng.uniform(-.2, .2), rng.uniform(-3, 3)])
    om = rng.uniform(-1, 1, 3)
    v = np.array([rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(-.2, .2)])
    fp = HIP + np.stack([rng.uniform(-.1, .1, 4), rng.uniform(-.05, .05, 4), -0.42 + rng.uniform(-.03, .03, 4)], 1)
    cmd = np.array([rng.uniform(-.35, .35), rng.uniform(-.2, .2), rng.uniform(-.4, .4)])
    return rpy, om, v, fp, cmd


def test_discretisation_matches_scipy_expm_and_is_nilpotent(oracle_lib):
    O = oracle_lib
    cfg = O.default_config()
    rng = np.random.default_rng(0)
    for _ in range(5):
        rpy, om, v, fp, cmd = _rand_state(rng)
        P, q, legs, Ad, Bd = O.mpc_build(cfg, rpy, om, v, fp.ravel(), [1, 1, 1, 1], cmd)
        # rebuild continuous A, B exactly as documented and compare with scipy's Pade expm
        cp, tp = math.cos(rpy[1]), math.tan(rpy[1])
        A = np.zeros((13, 13))
        A[0:3, 6:9] = [[1 / cp, 0, 0], [0, 1, 0], [tp, 0, 1]]
        A[3:6, 9:12] = np.eye(3)
        A[11, 12] = 1
        Bc = (Bd - 0.5 * cfg.dt_plan * A @ Bd) / 1.0  # placeholder, replaced below
        M = np.zeros((25, 25))
        M[:13, :13] = A * cfg.dt_plan
        # recover B dt from Bd rows that are not touched by A (rows 6..12): Bd[6:12] = B[6:12] dt
        Bdt = np.zeros((13, 12))
        Bdt[6:12] = Bd[6:12]
        M[:13, 13:] = Bdt
        E = scipy.linalg.expm(M)
        np.testing.assert_allclose(E[:13, :13], Ad, atol=1e-14)
        np.testing.assert_allclose(E[:13, 13:], Bd, atol=1e-14)
        assert np.abs(np.linalg.matrix_power(M, 3)).max() == 0.0  # M^3 = 0 -> series terminates
        np.testing.assert_allclose(np.eye(25) + M + M @ M / 2, E, atol=1e-15)


def test_hessian_has_kronecker_structure(oracle_lib):
    """P = 2 (N (x) G_U + S (x) G_V) + alpha I  -- the closed form the HIP kernel assembles."""
    O = oracle_lib
    cfg = O.default_config()
    rng = np.random.default_rng(1)
    H, dt, m = cfg.horizon, cfg.dt_plan, cfg.mass
    w = np.array(list(cfg.weights))
    for contact in ([1, 1, 1, 1], [0, 1, 1, 0], [1, 0, 1, 1]):
        rpy, om, v, fp, cmd = _rand_state(rng)
        P, q, legs, Ad, Bd = O.mpc_build(cfg, rpy, om, v, fp.ravel(), contact, cmd)
        cols = np.concatenate([np.arange(3 * l, 3 * l + 3) for l in legs])
        U = np.vstack([Bd[6:9, cols], Bd[9:12, cols]])           # dt [Bw; E/m]
        V = (Bd[0:6, cols]) * 2.0                                 # Bd rows 0..5 = dt^2/2 [T Bw; E/m]
        GU = U.T @ np.diag(w[6:12]) @ U
        GV = V.T @ np.diag(w[0:6]) @ V
        N = np.array([[H - max(a, b) for b in range(H)] for a in range(H)], dtype=float)
        S = np.array([[sum((k - a - .5) * (k - b - .5) for k in range(max(a, b) + 1, H + 1)) for b in range(H)] for a in range(H)])
        Pk = 2 * (np.kron(N, GU) + np.kron(S, GV)) + cfg.alpha * np.eye(len(q))
        np.testing.assert_allclose(Pk, P, rtol=1e-10, atol=1e-13)


def test_qp_solution_satisfies_kkt_and_constraints(oracle_lib):
    O = oracle_lib
    cfg = O.default_config()
    mg = cfg.mass * cfg.gravity
    rng = np.random.default_rng(2)
    for contact in ([1, 1, 1, 1], [0, 1, 1, 0], [1, 0, 0, 1], [1, 1, 0, 1], [0, 0, 1, 0]):
        for _ in range(10):
            rpy, om, v, fp, cmd = _rand_state(rng)
            P, q, legs, Ad, Bd = O.mpc_build(cfg, rpy, om, v, fp.ravel(), contact, cmd)
            u, it, kkt = O.qp_solve(P, q, 0.45, 0.1 * mg, 10 * mg)
            assert it >= 0
            assert kkt[0] < 1e-9 and kkt[1] < 1e-9 and kkt[2] < 1e-7
            U = u.reshape(-1, 3)
            assert (np.abs(U[:, 0]) <= 0.45 * U[:, 2] + 1e-9).all() and (np.abs(U[:, 1]) <= 0.45 * U[:, 2] + 1e-9).all()
            assert (U[:, 2] >= 0.1 * mg - 1e-9).all() and (U[:, 2] <= 10 * mg + 1e-9).all()


def test_qp_agrees_with_independent_long_run_admm(oracle_lib):
    """Second, algorithmically independent solver (numpy ADMM run to convergence)."""
    O = oracle_lib
    cfg = O.default_config()
    mg = cfg.mass * cfg.gravity
    lo, hi, mu = 0.1 * mg, 10 * mg, 0.45
    rng = np.random.default_rng(3)

    def proj(vv):
        V = vv.reshape(-1, 3)
        a, b, c = V[:, 0], V[:, 1], V[:, 2]
        aa, bb = np.abs(a), np.abs(b)
        zA = (c + mu * (aa + bb)) / (1 + 2 * mu * mu)
        zB = (c + mu * np.maximum(aa, bb)) / (1 + mu * mu)
        z = np.where(zA < np.minimum(aa, bb) / mu, zA, np.where(zB < np.maximum(aa, bb) / mu, zB, c))
        z = np.clip(z, lo, hi)
        return np.stack([np.clip(a, -mu * z, mu * z), np.clip(b, -mu * z, mu * z), z], 1).ravel()

    for contact in ([0, 1, 1, 0], [1, 1, 1, 1]):
        rpy, om, v, fp, cmd = _rand_state(rng)
        P, q, legs, Ad, Bd = O.mpc_build(cfg, rpy, om, v, fp.ravel(), contact, cmd)
        u, it, kkt = O.qp_solve(P, q, mu, lo, hi)
        n, rho = len(q), 1e-4
        M = np.linalg.inv(P + rho * np.eye(n))
        z, y = proj(np.zeros(n)), np.zeros(n)
        for _ in range(600):
            x = M @ (rho * (z - y) - q)
            xh = 1.8 * x - 0.8 * z
            z = proj(xh + y)
            y = y + xh - z
        np.testing.assert_allclose(z, u, rtol=0, atol=1e-7 * np.abs(u).max())


def _static_input(O, cfg, contact_all=True, yaw=0.0):
    inp = np.zeros(1, dtype=O.INPUT_DTYPE)
    inp["rpy"][0] = (0, 0, yaw)
    half = yaw / 2
    inp["quat"][0] = (0, 0, math.sin(half), math.cos(half))
    fp = HIP.copy()
    fp[:, 2] = -cfg.body_height
    inp["foot_pos"][0] = fp
    inp["jac"][0] = np.tile(np.eye(3), (4, 1, 1))
    inp["contact"][0] = 1
    return inp


def test_static_equilibrium_supports_weight(oracle_lib):
    """Level body at the desired height, zero velocity and command, symmetric feet:
    equal split, f_xy ~ 0, and sum f_z = m g (190 N, ghost/ctrl_constants.py:8) exactly when the
    force regulariser alpha -> 0.  With the upstream alpha = 1e-5 the regulariser outweighs the
    tracking cost of late-horizon forces, so the plan front-loads: first-step sum f_z ~ 1.06 m g."""
    O = oracle_lib
    stand = dict(duty_factor=(1.0,) * 4, init_phase=(0.0,) * 4, init_state=(1, 1, 1, 1))
    cfg0 = helpers.oracle_config(O, MPCConfig.for_robot("ghost", alpha=1e-11, **stand))
    g0 = O.OracleBatch(cfg0, 1).step(0.0, _static_input(O, cfg0))["grf"][0].reshape(4, 3)
    np.testing.assert_allclose(-g0[:, 2], 190.0 / 4, rtol=1e-4)  # forces are "foot on ground": negative z
    cfg = helpers.oracle_config(O, MPCConfig.for_robot("ghost", **stand))
    out = O.OracleBatch(cfg, 1).step(0.0, _static_input(O, cfg))
    grf = out["grf"][0].reshape(4, 3)
    assert 190.0 < -grf[:, 2].sum() < 1.1 * 190.0
    np.testing.assert_allclose(grf[:, 2], grf[0, 2], rtol=1e-9)
    assert np.abs(grf[:, :2]).max() < 1e-6
    # diagonal pair in stance: ~95 N each, the swing pair exactly zero
    trot = MPCConfig.for_robot("ghost")
    cfg2 = helpers.oracle_config(O, trot)
    ob2 = O.OracleBatch(cfg2, 1)
    out2 = ob2.step(0.25, _static_input(O, cfg2))  # t=0.25: legs 1,2 in stance, 0,3 swing
    assert list(out2["desired"][0]) == [0, 1, 1, 0]
    g2 = out2["grf"][0].reshape(4, 3)
    assert np.abs(g2[[0, 3]]).max() == 0.0
    assert 190.0 < -g2[[1, 2], 2].sum() < 1.1 * 190.0 and abs(g2[1, 2] - g2[2, 2]) < 1e-6


def test_yaw_invariance_and_mirror_symmetry(oracle_lib):
    O = oracle_lib
    cfg = helpers.oracle_config(O, MPCConfig.for_robot("ghost", duty_factor=(1.0,) * 4, init_phase=(0.0,) * 4, init_state=(1, 1, 1, 1)))
    rng = np.random.default_rng(5)
    inp = _static_input(O, cfg)
    inp["rpy"][0][:2] = (0.1, -0.05)
    inp["rpy_rate"][0] = rng.uniform(-1, 1, 3)
    inp["foot_pos"][0] += rng.uniform(-0.03, 0.03, (4, 3))
    inp["cmd"][0] = (0.3, 0.1, -0.2)
    a = O.OracleBatch(cfg, 1).step(0.0, inp)["grf"][0]
    inp2 = inp.copy()
    inp2["rpy"][0][2] = 2.1  # the controller zeroes yaw; v_world = 0 so the quaternion does not matter
    b = O.OracleBatch(cfg, 1).step(0.0, inp2)["grf"][0]
    np.testing.assert_allclose(a, b, rtol=0, atol=1e-9)
    # mirror left<->right: roll, yaw-rate, roll-rate, vy-command, wz flip sign; legs swap 0<->1, 2<->3
    m = inp.copy()
    m["rpy"][0] = inp["rpy"][0] * (-1, 1, -1)
    m["rpy_rate"][0] = inp["rpy_rate"][0] * (-1, 1, -1)
    m["cmd"][0] = inp["cmd"][0] * (1, -1, -1)
    m["foot_pos"][0] = inp["foot_pos"][0][[1, 0, 3, 2]] * (1, -1, 1)
    c = O.OracleBatch(cfg, 1).step(0.0, m)["grf"][0].reshape(4, 3)
    np.testing.assert_allclose(c[[1, 0, 3, 2]] * (1, -1, 1), a.reshape(4, 3), rtol=0, atol=1e-8)


def _gait_py(cfg, t, contact):
    """Independent pure-Python statement of SURVEY.md 8a-15."""
    des, st, ph = [], [], []
    for leg in range(4):
        T = cfg.stance_duration[leg] / cfg.duty_factor[leg]
        phi = math.fmod(t + cfg.init_phase[leg] * T, T) / T
        init = cfg.init_state[leg]
        r = cfg.duty_factor[leg] if init == 1 else 1 - cfg.duty_factor[leg]
        if phi < r:
            d, p = init, phi / r
        else:
            d, p = 1 - init, (phi - r) / (1 - r)
        s = d
        if p >= 0.1:
            if s == 0 and contact[leg]:
                s = 2
            if s == 1 and not contact[leg]:
                s = 3
        des.append(d); st.append(s); ph.append(p)
    return des, st, ph


def test_gait_table_bit_exact_over_a_cycle(oracle_lib):
    O = oracle_lib
    cfg = MPCConfig.for_robot("ghost")
    oc = helpers.oracle_config(O, cfg)
    i4, d4 = C.c_int * 4, C.c_double * 4
    rng = np.random.default_rng(9)
    # ghost constants: T = 0.5 s; at t = 0 legs 0,3 are past their initial SWING window
    d, s, p = i4(), i4(), d4()
    O.lib().orc_gait(C.byref(oc), 0.0, C.byref(i4(1, 1, 1, 1)), C.byref(d), C.byref(s), C.byref(p))
    assert list(d) == [1, 1, 1, 1] and abs(p[0] - (0.9 - 0.4) / 0.6) < 1e-15 and p[1] == 0.0
    for k in range(0, 1200):
        t = k * 0.001 * 0.5
        contact = [int(x) for x in rng.integers(0, 2, 4)]
        O.lib().orc_gait(C.byref(oc), t, C.byref(i4(*contact)), C.byref(d), C.byref(s), C.byref(p))
        de, se, pe = _gait_py(cfg, t, contact)
        assert list(d) == de and list(s) == se and list(p) == pe, (t, contact)


def test_velocity_filter_semantics(oracle_lib):
    O = oracle_lib
    cfg = helpers.oracle_config(O, MPCConfig.for_robot("ghost", duty_factor=(1.0,) * 4, init_phase=(0.0,) * 4, init_state=(1, 1, 1, 1)))
    ob = O.OracleBatch(cfg, 1)
    inp = _static_input(O, cfg)
    rng = np.random.default_rng(11)
    vs = rng.uniform(-1, 1, (45, 3))
    for k in range(45):
        inp["v_world"][0] = vs[k]
        out = ob.step(0.01 * k, inp)
        win = vs[max(0, k - 19):k + 1]
        # divides by the WINDOW SIZE even before the window is full (upstream behaviour)
        expect = np.array([math.fsum(win[:, a]) for a in range(3)]) / 20.0
        np.testing.assert_allclose(out["v_body"][0], expect, rtol=0, atol=1e-15)


def test_swing_trajectory_endpoints(oracle_lib):
    O = oracle_lib
    s, e, o = np.array([0.2, -0.1, -0.4]), np.array([0.3, -0.12, -0.41]), np.zeros(3)
    O.lib().orc_swing_trajectory(0.0, O._p(s), O._p(e), 0.1, O._p(o))
    np.testing.assert_allclose(o, s, atol=1e-15)
    O.lib().orc_swing_trajectory(1.0, O._p(s), O._p(e), 0.1, O._p(o))
    np.testing.assert_allclose(o, e, atol=1e-12)
    # s(p) = 0.8 sin(pi p) reaches the parabola's apex value 0.5 at p = asin(0.625)/pi
    O.lib().orc_swing_trajectory(math.asin(0.625) / math.pi, O._p(s), O._p(e), 0.1, O._p(o))
    assert abs(o[2] - (max(s[2], e[2]) + 0.1)) < 1e-12


@pytest.mark.parametrize("robot", ["ghost", "k3lso"])
def test_chain_kinematics_fk_jacobian_ik(oracle_lib, robot):
    O = oracle_lib
    cfg = MPCConfig.for_robot(robot, kin_mode=1)
    oc = helpers.oracle_config(O, cfg)
    ck = ChainKinematics(cfg)
    from robot_gym_amd.model.robots.robot_constants import ROBOTS
    q0 = np.array(ROBOTS[robot].init_motor_angles, dtype=np.float64)
    rng = np.random.default_rng(13)
    for leg in range(4):
        q = q0[3 * leg:3 * leg + 3] + rng.uniform(-0.3, 0.3, 3)
        p, J = np.zeros(3), np.zeros(9)
        O.lib().orc_leg_fk(C.byref(oc), leg, O._p(q), O._p(p), O._p(J))
        J = J.reshape(3, 3)
        p2, J2 = ck.foot_position_and_jacobian(leg, q)
        np.testing.assert_allclose(p, p2, atol=1e-14)
        np.testing.assert_allclose(J, J2, atol=1e-14)
        # Jacobian vs central finite differences
        for j in range(3):
            dq = np.zeros(3); dq[j] = 1e-6
            pp, pm = np.zeros(3), np.zeros(3)
            O.lib().orc_leg_fk(C.byref(oc), leg, O._p(q + dq), O._p(pp), None)
            O.lib().orc_leg_fk(C.byref(oc), leg, O._p(q - dq), O._p(pm), None)
            np.testing.assert_allclose((pp - pm) / 2e-6, J[:, j], atol=1e-8)
        # foot is below the hip, on the robot's side of the leg, at a plausible standing height
        assert p[2] < -0.2 and np.sign(p[1]) == np.sign(cfg.hip[3 * leg + 1]) and np.sign(p[0]) == np.sign(cfg.hip[3 * leg])
        # IK round trip from a perturbed start
        target = p.copy()
        qs = q + rng.uniform(-0.15, 0.15, 3)
        qo = np.zeros(3)
        O.lib().orc_leg_ik(C.byref(oc), leg, O._p(target), O._p(qs), O._p(qo))
        pr = np.zeros(3)
        O.lib().orc_leg_fk(C.byref(oc), leg, O._p(qo), O._p(pr), None)
        np.testing.assert_allclose(pr, target, atol=1e-9)
        ids, qn = ck.ComputeMotorAnglesFromFootLocalPosition(leg, target, np.concatenate([np.zeros(3 * leg), qs, np.zeros(9 - 3 * leg)]))
        assert ids == [3 * leg, 3 * leg + 1, 3 * leg + 2]
        np.testing.assert_allclose(qn, qo, atol=1e-12)


def test_first_update_after_reset_does_not_latch(oracle_lib):
    """Upstream list-aliasing quirk restated in the oracle (see orc_reset)."""
    O = oracle_lib
    # a gait whose leg 1 is already in SWING at the first tick after reset
    base = MPCConfig.for_robot("ghost", init_phase=(0.0, 0.7, 0.0, 0.0), init_state=(1, 1, 1, 1))
    cfg = helpers.oracle_config(O, base)
    ob = O.OracleBatch(cfg, 1)
    inp = _static_input(O, cfg)
    out = ob.step(0.0, inp)   # leg 1: phi = 0.7 >= 0.6 -> desired SWING at the very first update
    assert list(out["desired"][0]) == [1, 0, 1, 1]
    first = np.array(ob.states[0].latched[1][:])
    inp["foot_pos"][0][1] += (0.05, 0.0, 0.01)
    ob.step(0.01, inp)
    np.testing.assert_array_equal(np.array(ob.states[0].latched[1][:]), first)  # no transition seen -> unchanged


def test_first_update_latch_convention(oracle_lib):
    """conv_first_latch (a recall-sensitive convention): with the feet latched AT the reset
    (orc_reset given foot positions) and the robot in another pose at its first update, the default reading keeps the
    reset's latch through a STANCE->SWING edge seen by that first update; the other reading latches the current foot."""
    O = oracle_lib
    for conv in (0, 1):
        base = MPCConfig.for_robot("ghost", init_phase=(0.0, 0.7, 0.0, 0.0), init_state=(1, 1, 1, 1), conv_first_latch=conv)
        cfg = helpers.oracle_config(O, base)
        ob = O.OracleBatch(cfg, 1)
        inp = _static_input(O, cfg)
        at_reset = np.array(inp["foot_pos"][0], dtype=np.float64)
        ob.reset([0], 0.0, foot_pos=[at_reset.reshape(12)])
        inp["foot_pos"][0][1] += (0.05, 0.0, 0.01)   # leg 1 has moved by the first update, which sees it go STANCE -> SWING
        out = ob.step(0.0, inp)
        assert list(out["desired"][0]) == [1, 0, 1, 1]
        latched = np.array(ob.states[0].latched[1][:])
        np.testing.assert_array_equal(latched, inp["foot_pos"][0][1] if conv else at_reset[1])


def test_velocity_window_divide_convention(oracle_lib):
    """conv_window_divide = 1: the filling window divides by the samples it holds (a plain moving average)."""
    O = oracle_lib
    cfg = helpers.oracle_config(O, MPCConfig.for_robot("ghost", duty_factor=(1.0,) * 4, init_phase=(0.0,) * 4, init_state=(1, 1, 1, 1), conv_window_divide=1))
    ob = O.OracleBatch(cfg, 1)
    inp = _static_input(O, cfg)
    vs = np.random.default_rng(12).uniform(-1, 1, (30, 3))
    for k in range(30):
        inp["v_world"][0] = vs[k]
        out = ob.step(0.01 * k, inp)
        win = vs[max(0, k - 19):k + 1]
        np.testing.assert_allclose(out["v_body"][0], np.array([math.fsum(win[:, a]) for a in range(3)]) / len(win), rtol=0, atol=1e-15)


def test_contact_lookahead_extension(oracle_lib):
    """Opt-in extension: with a gait that never swings the look-ahead QP is the
    constant-contact QP; with a trot it differs, keeps swing legs force-free and satisfies the
    constraints (KKT residuals are checked inside the oracle solve)."""
    O = oracle_lib
    stand = dict(duty_factor=(1.0,) * 4, init_phase=(0.0,) * 4, init_state=(1, 1, 1, 1))
    inp = None
    outs = []
    for la in (0, 1):
        cfg = helpers.oracle_config(O, MPCConfig.for_robot("ghost", contact_lookahead=la, **stand))
        inp = _static_input(O, cfg)
        inp["rpy"][0][:2] = (0.05, -0.08)
        inp["cmd"][0] = (0.3, 0.0, 0.1)
        outs.append(O.OracleBatch(cfg, 1).step(0.0, inp)["grf"][0])
    np.testing.assert_allclose(outs[0], outs[1], rtol=0, atol=1e-9)
    res = []
    for la in (0, 1):
        cfg = helpers.oracle_config(O, MPCConfig.for_robot("ghost", contact_lookahead=la))
        inp = _static_input(O, cfg)
        inp["cmd"][0] = (0.3, 0.0, 0.0)
        out = O.OracleBatch(cfg, 1).step(0.22, inp)   # legs 1,2 in stance; 0,3 touch down at 0.25, 1,2 lift at 0.30
        assert list(out["desired"][0]) == [0, 1, 1, 0] and out["kkt"][0].max() < 1e-7
        g = out["grf"][0].reshape(4, 3)
        assert np.abs(g[[0, 3]]).max() == 0.0 and (-g[[1, 2], 2] >= 19.0 - 1e-9).all()
        res.append(g)
    assert np.abs(res[0] - res[1]).max() > 1.0   # the plan changes when the swap is anticipated


def test_caller_contact_schedule_and_per_robot_gaits(oracle_lib):
    """BASELINE config 5 inputs on the oracle: (1) a caller-supplied schedule equal to the open-loop one reproduces the
    gait-driven look-ahead exactly, bit 0 of the words is ignored, and dropping a planned contact changes the plan;
    (2) a batch with per-robot gait rows equals separately configured single-robot controllers."""
    O = oracle_lib
    base = MPCConfig.for_robot("ghost", contact_lookahead=1)
    B = 6
    state, cmd, t_off = synthetic.make_states(B, base, seed=41)
    gait = synthetic.random_gaits(B, base, seed=41)
    coff = helpers.cmd_with_offsets(base, cmd)
    contact = synthetic.gait_consistent_contacts(base, t_off, state["_flip"], gait)
    ocfg = helpers.oracle_config(O, base)

    def run(sched):
        ob = O.OracleBatch(ocfg, B, gait=gait)
        for b in range(B):
            ob.states[b].reset_time = -float(t_off[b])
        return ob.step(0.0, helpers.oracle_inputs(O, state, coff, contact, sched))

    ref = run(None)
    words = synthetic.contact_schedule(base, t_off, gait, dropout=0.0)
    same = run(words ^ 1)                                     # bit 0 flipped: must not matter
    np.testing.assert_array_equal(same["grf"], ref["grf"])
    np.testing.assert_array_equal(same["action"], ref["action"])
    dropped = run(synthetic.contact_schedule(base, t_off, gait, dropout=0.3, seed=3))
    assert np.abs(dropped["grf"] - ref["grf"]).max() > 0.5 and dropped["kkt"].max() < 1e-6
    # (2) per-robot rows == one config per robot
    for b in range(B):
        one = MPCConfig.for_robot("ghost", contact_lookahead=1, duty_factor=tuple(gait["duty_factor"][:, b]),
                                  stance_duration=tuple(gait["stance_duration"][:, b]), init_phase=tuple(gait["init_phase"][:, b]))
        ob1 = O.OracleBatch(helpers.oracle_config(O, one), 1)
        ob1.states[0].reset_time = -float(t_off[b])
        sub = {k: v[:, b:b + 1] for k, v in state.items() if k != "_flip"}
        o1 = ob1.step(0.0, helpers.oracle_inputs(O, sub, coff[:, b:b + 1], contact[:, b:b + 1]))
        np.testing.assert_array_equal(o1["action"][0], ref["action"][b])
        np.testing.assert_array_equal(o1["phase"][0], ref["phase"][b])


def test_admm_extrapolation_model_reaches_the_exact_optimum(oracle_lib):
    """The algorithm behind `admm_accel`, restated in numpy (tests/studies/admm_extrapolation_model.py: the kernels' two-stage
    ADMM, their stopping tests and the vote-time dominant-mode extrapolation with the kernels' constants), against the
    oracle's exact active-set solution on bench-workload trot QPs, three of which crawl (200-220 iterations from a cold
    start).  With and without the extrapolation the first-step forces land on the exact optimum, the extrapolation cuts
    the crawling robots by a third or more and costs nobody else more than two vote periods."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("admm_extrapolation_model", os.path.join(os.path.dirname(__file__), "studies", "admm_extrapolation_model.py"))
    model = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(model)
    O = oracle_lib
    cfg = MPCConfig.for_robot("ghost")
    ocfg = helpers.oracle_config(O, cfg)
    mg = cfg.mass * 9.8
    mu, lo, hi, tol = 0.45, 0.1 * mg, 10 * mg, 1e-6 * mg
    P, q = model.build_trot_qps(O, cfg, ocfg, 1500, 0)
    crawl = [26, 1273, 1423]                       # found by the study on this seeded batch
    idx = np.array(crawl + list(range(20)))
    P, q = P[idx], q[idx]
    z0 = np.zeros_like(q)
    z0[:, 2::3] = lo
    runs = {a: model.admm(P, q, z0, np.zeros_like(q), mu, lo, hi, tol, accel_from=a) for a in (0, 80)}
    for a, (z, y, it, done, jumps) in runs.items():
        assert done.all()
        for k in range(len(idx)):
            u, _, _ = O.qp_solve(P[k], q[k], mu, lo, hi)
            assert np.abs(z[k, :6] - u[:6]).max() <= 2e-5 * max(1.0, np.abs(u[:6]).max()), (a, k)
    it0, it1 = runs[0][2], runs[80][2]
    assert (it0[:3] >= 180).all() and (it1[:3] <= 0.67 * it0[:3]).all(), (it0[:3], it1[:3])
    assert (it1 <= it0 + 10).all() and runs[80][4][:3].min() >= 1 and runs[0][4].sum() == 0
