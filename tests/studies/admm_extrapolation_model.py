"""Study + reference model (CPU, numpy + the oracle's QP assembly): the kernels' warm-started two-stage ADMM with the
vote-time dominant-mode extrapolation (`admm_accel`, robot_gym_amd/csrc/rg_qp_common.inc: extrapolation_gain), batched
over trot robots of the bench workload.  Used to find out what the one or two robots per tick are that crawl for 200-300
iterations (they are slow from a cold start too: intrinsic to the QP, persistent for a given command) and to try remedies
offline: the extrapolation roughly halves them (230-290 -> 110-145 iterations) and leaves the population mean unchanged;
a cold restart at iteration 80 makes them worse.  tests/test_oracle_kat.py runs a small instance of this model against
the oracle's exact solver.
Usage: python tests/studies/admm_extrapolation_model.py [robots] [ticks]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def build_trot_qps(O, cfg, ocfg, B, tick, seed=0, amp=0.1):
    """(P [B,n,n], q [B,n]) of B trot robots of the bench workload at ring slab `tick` (diagonal pairs alternate)."""
    import bench
    from robot_gym_amd import synthetic
    from tests import helpers
    state0, cmd, _ = synthetic.make_states(B, cfg, seed=seed)
    state = bench.perturb_state(state0, tick, amp)
    coff = helpers.cmd_with_offsets(cfg, cmd)

    def quat_rot_inv(q, v):
        x, y, z, w = -q[0], -q[1], -q[2], q[3]
        t = 2 * np.cross([x, y, z], v)
        return v + w * t + np.cross([x, y, z], t)

    Ps, qs = [], []
    for b in range(B):
        vb = quat_rot_inv(state["quat"][:, b].astype(float), state["v_world"][:, b].astype(float))
        contact = [0, 1, 1, 0] if b % 2 else [1, 0, 0, 1]
        P, q, _, _, _ = O.mpc_build(ocfg, state["rpy"][:, b].astype(float), state["rpy_rate"][:, b].astype(float), vb,
                                    state["foot_pos"][:, b].astype(float), np.array(contact), coff[:, b].astype(float))
        Ps.append(P)
        qs.append(q)
    return np.array(Ps), np.array(qs)


def proj_pyramid(w, mu, lo, hi):
    """Exact Euclidean projection of every 3-vector of w [N, 3k] onto {|fx|, |fy| <= mu fz, lo <= fz <= hi} (closed form,
    as rg_mpc_dev.h: proj_pyramid)."""
    a, b, c = w[:, 0::3], w[:, 1::3], w[:, 2::3]
    aa, bb = np.abs(a), np.abs(b)
    mn, mx = np.minimum(aa, bb), np.maximum(aa, bb)
    zA = (c + mu * (aa + bb)) / (1 + 2 * mu * mu)
    zB = (c + mu * mx) / (1 + mu * mu)
    zz = np.clip(np.where(mu * zA < mn, zA, np.where(mu * zB < mx, zB, c)), lo, hi)
    lim = mu * zz
    out = np.empty_like(w)
    out[:, 0::3] = np.clip(a, -lim, lim)
    out[:, 1::3] = np.clip(b, -lim, lim)
    out[:, 2::3] = zz
    return out


def admm(P, q, z0, y0, mu, lo, hi, tol, rho1=1e-4, rho2=5e-4, switch=150, cap=450, chk=5, relax=1.8, extrap=5.0, accel_from=80,
         restart_at=0):
    """The kernels' ADMM, batched: two stages, votes every `chk` iterations (movement, primal residual, geometric distance
    estimate incl. the known rate of a jumped mode), extrapolation from iteration `accel_from` (0 = off).
    Returns (z, y in first-stage units, iterations, converged, jumps)."""
    N, n = q.shape
    I = np.eye(n)
    z, y = z0.copy(), y0.copy()
    rho = np.full(N, rho1)
    G = np.linalg.inv(P + rho1 * I)
    done = np.zeros(N, bool)
    iters = np.zeros(N, int)
    stage = np.zeros(N, int)
    jumps = np.zeros(N, int)
    zc, yc = z.copy(), y.copy()
    dz0 = np.full((N, n), np.inf)
    dy0 = np.zeros((N, n))
    hist = np.zeros(N, int)      # valid windows of displacement history before the current one
    gmax = np.zeros(N)           # largest extrapolation gain of the stage
    rfac = np.zeros(N)           # remaining_distance_factor of the previous vote
    for it in range(1, cap + 1):
        act = ~done
        x = np.einsum('nij,nj->ni', G, rho[:, None] * (z - y) - q)
        w = relax * x + (1 - relax) * z + y
        zn = proj_pyramid(w, mu, lo, hi)
        z = np.where(act[:, None], zn, z)
        y = np.where(act[:, None], w - zn, y)
        iters[act] = it
        if it % chk == 0:
            dz, dy = z - zc, y - yc
            m, mp = np.abs(dz), np.abs(dz0)
            with np.errstate(invalid="ignore", divide="ignore"):
                est = np.where((mp > m) & np.isfinite(mp), m * m / (mp - m), 0.0)
            jumped = gmax > 0           # extrapolated in this stage: known-rate / whole-iterate-rate estimates, two quiet windows
            est = np.maximum(est, m * np.where(jumped, np.maximum(gmax, rfac), 0.0)[:, None])
            t = (tol * np.where(stage == 1, rho1 / rho2, 1.0))[:, None]
            with np.errstate(invalid="ignore"):
                loud_before = jumped[:, None] & (mp > 4 * t)
            moving = ((m > t) | (est > extrap * t) | (np.abs(x - z) > 10 * t) | loud_before).any(1) | (jumped & (hist <= 0))
            done |= act & ~moving
            dz0f = np.where(np.isfinite(dz0) & (hist >= 1)[:, None], dz0, 0.0)
            dy0f = np.where((hist >= 1)[:, None], dy0, 0.0)
            d11 = (dz * dz).sum(1) + (dy * dy).sum(1)
            d00 = (dz0f * dz0f).sum(1) + (dy0f * dy0f).sum(1)
            d10 = (dz * dz0f).sum(1) + (dy * dy0f).sum(1)
            rate_now = (~done) & (hist >= 1) & (accel_from > 0) & (it >= accel_from)
            jump = rate_now & (d10 > 0) & (d10 * d10 > 0.9 * d11 * d00) & (d10 > 0.5 * d00) & (d10 < 0.98 * d00)
            with np.errstate(invalid="ignore", divide="ignore"):
                g = np.where(jump, d10 / (d00 - d10), 0.0)
                r = np.minimum(d10 / d00, 0.999)
                rf = np.where((d10 > 0) & (d10 * d10 > 0.5 * d11 * d00), r / (1 - r), 0.0)
            rfac = np.where(rate_now, np.where(jump, 0.0, rf), rfac)
            z = np.where(jump[:, None], z + g[:, None] * dz, z)
            y = np.where(jump[:, None], y + g[:, None] * dy, y)
            gmax = np.maximum(gmax, g)
            jumps += jump
            hist = np.where(jump, 0, hist + 1)
            dz0, dy0 = dz, dy
            zc, yc = z.copy(), y.copy()
        if restart_at and it == restart_at:      # remedy tried and rejected: an unconverged warm start restarts cold
            rs = ~done
            z[rs] = 0
            z[rs, 2::3] = lo
            y[rs] = 0
            zc[rs], yc[rs] = z[rs], 0
        if it == switch and rho2 > 0:
            sw = ~done
            if sw.any():
                y[sw] *= rho1 / rho2
                rho[sw] = rho2
                stage[sw] = 1
                G[sw] = np.linalg.inv(P[sw] + rho2 * I)
                zc[sw], yc[sw] = z[sw], y[sw]
                dz0[sw] = np.inf
                dy0[sw] = 0
                hist[sw] = 0
                gmax[sw] = 0
                rfac[sw] = 0
        if done.all():
            break
    return z, np.where((stage == 1)[:, None], y * rho2 / rho1, y), iters, done, jumps


if __name__ == "__main__":
    from oracle import oracle as O
    from robot_gym_amd.core.config import MPCConfig
    from tests import helpers
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    cfg = MPCConfig.for_robot("ghost")
    ocfg = helpers.oracle_config(O, cfg)
    mg = cfg.mass * 9.8
    mu, lo, hi, tol = 0.45, 0.1 * mg, 10 * mg, 1e-6 * mg
    n = 60
    cold_z = np.zeros((B, n))
    cold_z[:, 2::3] = lo
    warm = {a: (cold_z.copy(), np.zeros((B, n))) for a in (0, 80)}
    for j in range(ticks):
        P, q = build_trot_qps(O, cfg, ocfg, B, j)
        for a in (0, 80):
            z, y, it, done, jumps = admm(P, q, *warm[a], mu, lo, hi, tol, accel_from=a)
            warm[a] = (z.astype(np.float32).astype(float), y.astype(np.float32).astype(float))
            print(f"tick {j} accel {a:2d}: mean {it.mean():5.1f}  top {np.sort(it)[::-1][:6]}  unconverged {int((~done).sum())}  jumps/robot {jumps.mean():.3f}")
