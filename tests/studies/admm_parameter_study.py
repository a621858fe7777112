"""Offline study (not a test, not collected by pytest): a numpy replica of the device ADMM loop on QPs built by
the CPU oracle from the bench's synthetic states (steady state: velocity filter full), used to map the
parameter space -- rho, relaxation, vote period, primal-residual test, per-step rho, accelerated ADMM.
Usage: python tests/studies/admm_parameter_study.py <robots> [v2|v3|v4|v5|v6|v7]   (results quoted in DESIGN.md section 4)"""
import sys, numpy as np
sys.path.insert(0, '/root/repo')
from oracle import oracle as O
from robot_gym_amd.core.config import MPCConfig
from robot_gym_amd import synthetic
from tests import helpers

cfg = MPCConfig.for_robot("ghost")
ocfg = helpers.oracle_config(O, cfg)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 200
state, cmd, t_off = synthetic.make_states(B, cfg, seed=0)
coff = helpers.cmd_with_offsets(cfg, cmd)
mg = cfg.mass * 9.8
mu, lo, hi = 0.45, 0.1 * mg, 10 * mg

def quat_rot_inv(q, v):
    x, y, z, w = -q[0], -q[1], -q[2], q[3]
    t = 2 * np.cross([x, y, z], v)
    return v + w * t + np.cross([x, y, z], t)

def proj(a, b, c):
    aa, bb = abs(a), abs(b)
    mn, mx = min(aa, bb), max(aa, bb)
    zA = (c + mu * (aa + bb)) / (1 + 2 * mu * mu)
    zB = (c + mu * mx) / (1 + mu * mu)
    zz = zA if mu * zA < mn else (zB if mu * zB < mx else c)
    zz = min(max(zz, lo), hi)
    lim = mu * zz
    return min(max(a, -lim), lim), min(max(b, -lim), lim), zz

def projv(w):
    out = np.empty_like(w)
    for i in range(0, len(w), 3):
        out[i:i+3] = proj(*w[i:i+3])
    return out

def admm(P, q, rho_vec, relax=1.8, chk=10, atol=1e-6 * mg, cap=400):
    n = len(q)
    G = np.linalg.inv(P + np.diag(rho_vec))
    z = np.zeros(n); z[2::3] = lo
    y = np.zeros(n)   # scaled dual: y = lambda / rho (per entry)
    zchk = z.copy()
    for it in range(1, cap + 1):
        x = G @ (rho_vec * (z - y) - q)
        w = relax * x + (1 - relax) * z + y
        zn = projv(w)
        y = w - zn
        z = zn
        if it % chk == 0:
            if np.max(np.abs(z - zchk)) <= atol:
                return z, it
            zchk = z.copy()
    return z, cap

problems = []
for b in range(B):
    vb = quat_rot_inv(state["quat"][:, b].astype(float), state["v_world"][:, b].astype(float))
    rpy = state["rpy"][:, b].astype(float)
    for contact in ([0, 1, 1, 0], [1, 1, 1, 1]) if b % 4 == 0 else ([0, 1, 1, 0] if b % 2 else [1, 0, 0, 1],):
        P, q, legs, _, _ = O.mpc_build(ocfg, rpy, state["rpy_rate"][:, b].astype(float), vb,
                                       state["foot_pos"][:, b].astype(float), np.array(contact), coff[:, b].astype(float))
        problems.append((P, q, len(legs)))

def run(name, rho_fn, **kw):
    its = {2: [], 4: []}
    errs = []
    for P, q, nc in problems:
        z, it = admm(P, q, rho_fn(P, nc), **kw)
        its[nc].append(it)
        if len(errs) < 40:
            u, _, _ = O.qp_solve(P, q, mu, lo, hi)
            errs.append(np.max(np.abs(z[:3 * nc] - u[:3 * nc])) / max(1.0, np.max(np.abs(u[:3 * nc]))))
    for nc in (2, 4):
        a = np.array(its[nc])
        if len(a): print(f"{name:34s} nc={nc} n={len(a):4d} mean {a.mean():6.1f} p90 {np.percentile(a,90):5.0f} max {a.max():4d}", end="  ")
    print(f"err(first step, rel) max {max(errs):.2e}")

H = 10
if __name__ == "__main__":
    run("baseline rho=1e-4", lambda P, nc: np.full(P.shape[0], 1e-4))
    for r in (5e-5, 2e-4):
        run(f"rho={r:g}", lambda P, nc, r=r: np.full(P.shape[0], r))
    # per-step rho proportional to the block's mean diagonal
    def per_step(P, nc, r0, gamma):
        d = np.diag(P).reshape(H, -1).mean(1)
        s = (d / d.mean()) ** gamma
        return np.repeat(r0 * s, 3 * nc)
    for r0 in (1e-4, 2e-4):
        for g in (0.5, 1.0):
            run(f"per-step r0={r0:g} gamma={g}", lambda P, nc, r0=r0, g=g: per_step(P, nc, r0, g))

if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[2] == "v2":
    print("---- v2")
    for r0 in (5e-5, 1e-4):
        for g in (-0.5, -1.0):
            run(f"per-step r0={r0:g} gamma={g}", lambda P, nc, r0=r0, g=g: per_step(P, nc, r0, g))
    for rl in (1.6, 1.9, 1.95):
        run(f"rho=1e-4 relax={rl}", lambda P, nc: np.full(P.shape[0], 1e-4), relax=rl)
    run(f"rho=5e-5 relax=1.9", lambda P, nc: np.full(P.shape[0], 5e-5), relax=1.9)
    # fz rows get a different rho than fx, fy
    for fz_scale in (0.25, 4.0):
        def axis(P, nc, s=fz_scale):
            r = np.full(P.shape[0], 1e-4); r[2::3] *= s; return r
        run(f"rho=1e-4, fz x{fz_scale}", axis)

def fast_admm(P, q, rho, relax=1.0, chk=10, atol=1e-6 * mg, cap=400, eta=0.999):
    n = len(q)
    G = np.linalg.inv(P + rho * np.eye(n))
    z = np.zeros(n); z[2::3] = lo
    y = np.zeros(n)
    zh, yh = z.copy(), y.copy()
    zprev, yprev = z.copy(), y.copy()
    t = 1.0; cprev = np.inf
    zchk = z.copy()
    for it in range(1, cap + 1):
        x = G @ (rho * (zh - yh) - q)
        w = relax * x + (1 - relax) * zh + yh
        zn = projv(w)
        yn = w - zn
        c = np.sum((yn - yh) ** 2) + np.sum((zn - zh) ** 2)
        if c < eta * cprev:
            tn = 0.5 * (1 + np.sqrt(1 + 4 * t * t))
            zh = zn + ((t - 1) / tn) * (zn - z)
            yh = yn + ((t - 1) / tn) * (yn - y)
            t = tn; cprev = c
            z, y = zn, yn
        else:
            t = 1.0; zh, yh = z.copy(), y.copy(); cprev = cprev / eta
        if it % chk == 0:
            if np.max(np.abs(z - zchk)) <= atol:
                return z, it
            zchk = z.copy()
    return z, cap

def run2(name, fn):
    its = {2: [], 4: []}; errs = []
    for P, q, nc in problems:
        z, it = fn(P, q)
        its[nc].append(it)
        if len(errs) < 40:
            u, _, _ = O.qp_solve(P, q, mu, lo, hi)
            errs.append(np.max(np.abs(z[:3 * nc] - u[:3 * nc])) / max(1.0, np.max(np.abs(u[:3 * nc]))))
    for nc in (2, 4):
        a = np.array(its[nc])
        print(f"{name:34s} nc={nc} mean {a.mean():6.1f} p90 {np.percentile(a,90):5.0f} max {a.max():4d}", end="  ")
    print(f"err max {max(errs):.2e}")

if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[2] == "v3":
    print("---- v3")
    for rho in (1e-4, 3e-4, 1e-3):
        for rl in (1.0, 1.5):
            run2(f"fast ADMM rho={rho:g} relax={rl}", lambda P, q, rho=rho, rl=rl: fast_admm(P, q, rho, relax=rl))

if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[2] == "v4":
    print("---- v4")
    base = lambda P, nc: np.full(P.shape[0], 1e-4)
    run("chk=10 atol=1e-6 (baseline)", base)
    run("chk=5 atol=1e-6", base, chk=5)
    run("chk=5 atol=5e-7", base, chk=5, atol=5e-7 * mg)
    run("chk=4 atol=4e-7", base, chk=4, atol=4e-7 * mg)
    run("chk=10 atol=2e-6", base, atol=2e-6 * mg)
    run("chk=8 atol=1e-6", base, chk=8)

def admm2(P, q, rho, relax=1.8, chk=5, atol=1e-6 * mg, ptol=None, cap=400):
    """baseline + primal residual test |x - z| <= ptol at the check iterations"""
    n = len(q)
    G = np.linalg.inv(P + rho * np.eye(n))
    z = np.zeros(n); z[2::3] = lo
    y = np.zeros(n)
    zchk = z.copy()
    for it in range(1, cap + 1):
        x = G @ (rho * (z - y) - q)
        w = relax * x + (1 - relax) * z + y
        zn = projv(w)
        y = w - zn
        z = zn
        if it % chk == 0:
            ok = np.max(np.abs(z - zchk)) <= atol
            if ptol is not None: ok = ok and np.max(np.abs(x - z)) <= ptol
            if ok: return z, it, np.max(np.abs(x - z))
            zchk = z.copy()
    return z, cap, np.max(np.abs(x - z))

if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[2] == "v5":
    print("---- v5")
    # unbalanced problems: bound (front pair / rear pair) and pace (left / right)
    unb = []
    for b in range(0, 60):
        vb = quat_rot_inv(state["quat"][:, b].astype(float), state["v_world"][:, b].astype(float))
        rpy = state["rpy"][:, b].astype(float)
        for contact in ([1, 1, 0, 0], [0, 0, 1, 1], [1, 0, 1, 0]):
            P, q, legs, _, _ = O.mpc_build(ocfg, rpy, state["rpy_rate"][:, b].astype(float), vb, state["foot_pos"][:, b].astype(float), np.array(contact), coff[:, b].astype(float))
            unb.append((P, q, 2))
    for name, probs in (("trot/4leg", problems), ("unbalanced", unb)):
        for label, kw in (("chk10", dict(chk=10)), ("chk5", dict(chk=5)), ("chk5+p1e-6", dict(chk=5, ptol=1e-6 * mg)), ("chk5+p1e-5", dict(chk=5, ptol=1e-5 * mg)), ("chk5+p1e-4", dict(chk=5, ptol=1e-4 * mg))):
            its, errs, pres = [], [], []
            for P, q, nc in probs[:150]:
                z, it, pr = admm2(P, q, 1e-4, **kw)
                its.append(it); pres.append(pr / mg)
                if len(errs) < 60:
                    u, _, _ = O.qp_solve(P, q, mu, lo, hi)
                    errs.append(np.max(np.abs(z[:6] - u[:6])) / max(1.0, np.max(np.abs(u[:6]))))
            its = np.array(its); errs = np.array(errs)
            print(f"{name:10s} {label:12s} mean {its.mean():6.1f} max {its.max():4d} capped {np.sum(its>=400):3d}  err max {errs.max():.2e}  worst err among 'converged' {errs[its[:len(errs)]<400].max() if np.any(its[:len(errs)]<400) else 0:.2e}  pres/mg max {max(pres):.1e}")

if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[2] == "v6":
    print("---- v6")
    for rho in (5e-5, 7e-5, 1e-4, 1.3e-4):
        for rl in (1.7, 1.8, 1.85):
            its = {2: [], 4: []}; errs = []
            for P, q, nc in problems:
                z, it, pr = admm2(P, q, rho, relax=rl, chk=5, ptol=1e-5 * mg)
                its[nc].append(it)
                if len(errs) < 40:
                    u, _, _ = O.qp_solve(P, q, mu, lo, hi)
                    errs.append(np.max(np.abs(z[:3 * nc] - u[:3 * nc])) / max(1.0, np.max(np.abs(u[:3 * nc]))))
            a2, a4 = np.array(its[2]), np.array(its[4])
            print(f"rho={rho:g} relax={rl}: nc2 mean {a2.mean():5.1f} p99 {np.percentile(a2,99):4.0f} max {a2.max():3d} | nc4 mean {a4.mean():5.1f} max {a4.max():3d} | err {max(errs):.1e}")

if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[2] == "v7":
    print("---- v7")
    for rho in (2e-5, 3e-5, 4e-5, 5e-5):
        for rl in (1.5, 1.6, 1.7):
            its = {2: [], 4: []}; errs = []
            for P, q, nc in problems:
                z, it, pr = admm2(P, q, rho, relax=rl, chk=5, ptol=1e-5 * mg)
                its[nc].append(it)
                if len(errs) < 40:
                    u, _, _ = O.qp_solve(P, q, mu, lo, hi)
                    errs.append(np.max(np.abs(z[:3 * nc] - u[:3 * nc])) / max(1.0, np.max(np.abs(u[:3 * nc]))))
            a2, a4 = np.array(its[2]), np.array(its[4])
            print(f"rho={rho:g} relax={rl}: nc2 mean {a2.mean():5.1f} p99 {np.percentile(a2,99):4.0f} max {a2.max():3d} | nc4 mean {a4.mean():5.1f} max {a4.max():3d} | err {max(errs):.1e}")
