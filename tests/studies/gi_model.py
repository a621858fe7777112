"""numpy model of the exact engine of rg_qp_exact_kernel (rg_qp_exact_kernel.inc): Goldfarb-Idnani dual active-set method in
range-space form on an explicit G = P^-1, the primal iterate recomputed from the multipliers once per iteration
(x = x0 + G C_A' lambda), optional warm start from the working set of the previous tick (its constraints added as
equalities, negative multipliers dropped: a valid S-pair).  Run as a script it compares cold and warm solves with the
oracle's solver on the bench workload and prints iteration / G-application counts.

    python tests/studies/gi_model.py [robots] [ticks]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def cons(pid, mu):
    blk, ty = divmod(pid, 6)
    i0 = 3 * blk + (0 if ty < 2 else (1 if ty < 4 else 2))
    i1 = 3 * blk + 2
    v0 = -1.0 if ty in (0, 2, 5) else 1.0
    v1 = mu if ty < 4 else 0.0
    return i0, i1, v0, v1


def rhs(pid, lo, hi):
    ty = pid % 6
    return lo if ty == 4 else (-hi if ty == 5 else 0.0)


def solve(G, qv, mu, lo, hi, warm=(), qmax=10 ** 9):
    """Returns (x or None on breakdown / overflow, constraint additions, applications of G, final working set)."""
    n = len(qv)
    nb = n // 3
    x0 = -G @ qv
    A, lam = [], []
    T = np.zeros((0, 0))
    napply = 1

    def cvec(pid):
        i0, i1, v0, v1 = cons(pid, mu)
        c = np.zeros(n)
        c[i0] += v0
        c[i1] += v1
        return c

    def slack(x, pid):
        i0, i1, v0, v1 = cons(pid, mu)
        return v0 * x[i0] + v1 * x[i1] - rhs(pid, lo, hi)

    def add(T, r, dz):
        q = T.shape[0]
        Tn = np.zeros((q + 1, q + 1))
        Tn[:q, :q] = T + np.outer(r, r) / dz
        Tn[:q, q] = -r / dz
        Tn[q, :q] = -r / dz
        Tn[q, q] = 1 / dz
        return Tn

    def drop(T, l):
        q = T.shape[0]
        t = T[:, l].copy()
        Tn = T - np.outer(t, t) / T[l, l]
        keep = [k for k in range(q) if k != l]
        return Tn[np.ix_(keep, keep)]

    # warm start: the stored constraints as equalities, one at a time (linearly dependent ones are skipped) ...
    for pid in warm:
        if len(A) >= qmax:
            break
        c = cvec(pid)
        d = G @ c
        napply += 1
        CA = np.array([cvec(a) for a in A]).reshape(len(A), n)
        sv = CA @ d
        r = T @ sv
        dz = c @ d - sv @ r
        if dz <= 1e-9 * (c @ d):
            continue
        T = add(T, r, dz)
        A.append(pid)
    # ... then the multipliers of the equality-constrained minimiser; negative ones are dropped until none is left
    while A:
        lam_ = -T @ np.array([slack(x0, a) for a in A])
        k = int(np.argmin(lam_))
        if lam_[k] >= 0:
            lam = list(lam_)
            break
        T = drop(T, k)
        A.pop(k)

    def xof(A, lam):
        w = np.zeros(n)
        for a, l in zip(A, lam):
            w += l * cvec(a)
        return x0 + G @ w

    x = x0.copy()
    if A:
        x = xof(A, lam)
        napply += 1
    vtol = 1e-9 * (1 + hi * 1e-3)
    its = 0
    for _ in range(8 * n + 80):
        best, pid = 0.0, -1
        act = set(A)
        for b in range(nb):
            for ty in range(6):
                p = 6 * b + ty
                if p in act:
                    continue
                s = slack(x, p)
                if s < best:
                    best, pid = s, p
        if pid < 0 or best >= -vtol:
            return x, its, napply, A
        if len(A) >= qmax:
            return None, its, napply, A
        c = cvec(pid)
        d = G @ c
        napply += 1
        sigma = c @ d
        s_p = best
        lam_p = 0.0
        while True:
            CA = np.array([cvec(a) for a in A]).reshape(len(A), n)
            sv = CA @ d
            r = T @ sv
            dz = sigma - sv @ r
            t1, l = np.inf, -1
            for k in range(len(A)):
                if r[k] > 0 and lam[k] / r[k] < t1:
                    t1, l = lam[k] / r[k], k
            have = dz > 1e-13 * (1 + abs(sigma))
            t2 = -s_p / dz if have else np.inf
            t = min(t1, t2)
            if not np.isfinite(t):
                return None, its, napply, A
            lam = [lk - t * rk for lk, rk in zip(lam, r)]
            lam_p += t
            if have:
                s_p += t * dz
            if have and t2 <= t1:
                T = add(T, r, dz)
                A.append(pid)
                lam.append(lam_p)
                its += 1
                break
            T = drop(T, l)
            A.pop(l)
            lam.pop(l)
        x = xof(A, lam)
        napply += 1
    return None, its, napply, A


def main():
    from oracle import oracle as O
    from tests import helpers
    from robot_gym_amd import synthetic
    from robot_gym_amd.core.config import MPCConfig
    import bench
    cfg = MPCConfig.for_robot("ghost")
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 96
    ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    state, cmd, t_off = synthetic.make_states(B, cfg, seed=1)
    ocfg = helpers.oracle_config(O, cfg)
    coff = helpers.cmd_with_offsets(cfg, cmd)
    ob = O.OracleBatch(ocfg, B, 0.0, 8)
    for b in range(B):
        ob.states[b].reset_time = -float(t_off[b])
    mg = cfg.mass * cfg.gravity
    mu, lo, hi = 0.45, 0.1 * mg, 10 * mg
    prev = [None] * B
    res = {}
    for j in range(ticks):
        st = bench.perturb_state(state, j, 0.1)
        contact = synthetic.gait_consistent_contacts(cfg, t_off + 0.01 * j, state["_flip"], None)
        inp = helpers.oracle_inputs(O, st, coff, contact, None)
        out = ob.step(0.01 * j, inp)
        for b in range(B):
            c4 = [int(v == 1) for v in out["desired"][b]]
            nc = sum(c4)
            if nc == 0:
                prev[b] = None
                continue
            P, q, _, _, _ = O.mpc_build(ocfg, np.array(inp["rpy"][b]), inp["rpy_rate"][b], out["v_body"][b], inp["foot_pos"][b].ravel(), c4, inp["cmd"][b])
            u, it, _ = O.qp_solve(P, q, mu, lo, hi)
            G = np.linalg.inv(P)
            xc, itc, nac, _ = solve(G, q, mu, lo, hi)
            warm = prev[b][1] if (prev[b] is not None and prev[b][0] == tuple(c4)) else ()
            xw, itw, naw, Aw = solve(G, q, mu, lo, hi, warm=warm)
            sc = max(np.abs(u).max(), 1)
            res.setdefault(nc, []).append((it, itc, nac, itw, naw, np.abs(xc - u).max() / sc, np.abs(xw - u).max() / sc, len(Aw), len(warm)))
            prev[b] = (tuple(c4), list(Aw))
    for nc, r in sorted(res.items()):
        r = np.array(r)
        print(f"nc={nc} n={len(r)} oracle its {r[:, 0].mean():.1f} | cold: its {r[:, 1].mean():.1f} G-applications {r[:, 2].mean():.1f} max {r[:, 2].max():.0f}"
              f" | warm: its {r[:, 3].mean():.2f} G-applications {r[:, 4].mean():.1f} p90 {np.percentile(r[:, 4], 90):.0f} max {r[:, 4].max():.0f}"
              f" | err cold {r[:, 5].max():.1e} warm {r[:, 6].max():.1e} | |A| mean {r[:, 7].mean():.1f} max {r[:, 7].max():.0f}")


if __name__ == "__main__":
    main()
