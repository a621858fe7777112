"""CPU model of the symmetric 6 x 6-block sweep (robot_gym_amd/csrc/rg_qp_sym6.inc), lane by lane in numpy: the index algebra the
HIP code relies on, checked without a GPU.

* sym6_lane / sym6_lane_of: the fold of the lower block triangle onto lanes and its inverse;
* the turn-over sweep: lanes below the block diagonal start with the MIRROR IMAGE of their block, every publisher hands on a ROW
  of what it holds, and a lane turns its block over once, in front of the last pivot of the block-row above its own; the result
  is -(M^-1) with +2 on the diagonal, as tile_sweep's;
* sym6_to_tile8: the gather into 8 x 8 tiles -- staging offsets, the lower / mirrored choice per (lane, row), the reduce-scatter's
  row order, clamped padding -- reproduces -(M^-1) entry for entry.
The GPU parity tests exercise the real thing (every QP launch starts with it); this file is the restatement a reader can step
through.  What it accelerates: reference controllers/mpc/mpc_controller.py:102-106 (the QP solve inside get_action)."""
import numpy as np
import pytest


def sym6_lane(t, nb):
    pair, off = divmod(t, nb + 1)
    if t >= nb * (nb + 1) // 2:
        return 0, 0, False
    return (pair, off, True) if off <= pair else (nb - 1 - pair, off - pair - 1, True)


def sym6_lane_of(r, c, nb):
    return r * (nb + 1) + c if 2 * r <= nb - 1 else (nb - 1 - r) * (nb + 2) + 1 + c


def div6_u8(i):
    return (i * 43) >> 8


@pytest.mark.parametrize("nb", [5, 10, 20])
def test_lane_fold_and_its_inverse(nb):
    nt = 64 if nb <= 10 else 256
    seen = {}
    for t in range(nt):
        br, bc, on = sym6_lane(t, nb)
        if on:
            assert 0 <= bc <= br < nb and (br, bc) not in seen
            seen[(br, bc)] = t
            assert sym6_lane_of(br, bc, nb) == t
    assert len(seen) == nb * (nb + 1) // 2
    assert all(div6_u8(i) == i // 6 for i in range(131)) and div6_u8(131) != 131 // 6   # the callers stay below 128


def sweep_model(M, nb):
    """The lanes' blocks after the turn-over sweep: {lane: 6 x 6 array} (normal orientation, +2 on the diagonal)."""
    n = 6 * nb
    lanes = [t for t in range(nb * (nb + 1) // 2 + 3) if sym6_lane(t, nb)[2]]
    X, gr, gc = {}, {}, {}
    for t in lanes:
        br, bc, _ = sym6_lane(t, nb)
        gr[t], gc[t] = bc, br                                   # mirror image: block (bc, br)
        X[t] = M[6 * bc:6 * bc + 6, 6 * br:6 * br + 6].copy()

    def publish(ko, kb):
        p, d = np.full(n, np.nan), None
        for t in lanes:
            if gr[t] == kb:
                w = X[t][ko].copy()
                if gc[t] == kb:
                    d = w[ko]
                    w[ko] -= 1.0
                p[6 * gc[t]:6 * gc[t] + 6] = w
        return p, d
    p, d = publish(0, 0)
    for kb in range(nb):
        for ko in range(6):
            assert not np.isnan(p).any()                        # ten owners, six entries each: the whole pivot row
            if ko == 5:                                          # the lanes of block-row kb + 1 turn their block over
                for t in lanes:
                    br, bc, _ = sym6_lane(t, nb)
                    if br == kb + 1 and gr[t] != br:
                        X[t] = X[t].T.copy()
                        gr[t], gc[t] = br, bc
            pn = None
            kon, kbn = (ko + 1) % 6, kb + (ko == 5)
            for t in lanes:
                pr, pc = p[6 * gr[t]:6 * gr[t] + 6], p[6 * gc[t]:6 * gc[t] + 6]
                X[t] += np.outer(-pr / d, pc)
            if kbn < nb:
                p, d = publish(kon, kbn)
    for t in lanes:
        br, bc, _ = sym6_lane(t, nb)
        assert (gr[t], gc[t]) == (br, bc)                       # every lane ends in the normal orientation
    return X


@pytest.mark.parametrize("nb", [5, 10])
def test_turn_over_sweep_inverts_and_the_gather_rebuilds_the_tiles(nb):
    rng = np.random.default_rng(nb)
    n = 6 * nb
    A = rng.normal(size=(n, n))
    M = A @ A.T + n * np.eye(n)
    X = sweep_model(M, nb)
    want = -np.linalg.inv(M)
    for t, blk in X.items():
        br, bc, _ = sym6_lane(t, nb)
        ref = want[6 * br:6 * br + 6, 6 * bc:6 * bc + 6] + (2.0 * np.eye(6) if br == bc else 0.0)
        np.testing.assert_allclose(blk, ref, rtol=0, atol=1e-12)
    # ---- sym6_to_tile8: staging + gather (LG = 3: 8 x 8 lanes) ----
    stg = np.full(nb * (nb + 1) // 2 * 36, np.nan)
    for t, blk in X.items():
        br, bc, _ = sym6_lane(t, nb)
        b = blk - (2.0 * np.eye(6) if br == bc else 0.0)
        stg[(br * (br + 1) // 2 + bc) * 36:(br * (br + 1) // 2 + bc) * 36 + 36] = b.reshape(-1)
    for lane in range(64):
        lr, lc = lane >> 3, lane & 7
        perm = ((lc & 1) << 1) | (lc & 4)
        rsplit = 6 * (div6_u8(8 * lr) + 1)
        for ta in range(8):
            i = 8 * lr + (ta ^ perm)
            q = div6_u8(i); ia = i - 6 * q; ib = min(q, nb - 1)
            mir = lr < lc or (lr == lc and i < rsplit)
            rpart = ib * 36 + ia if mir else (ib * (ib + 1) // 2) * 36 + 6 * ia
            for tb in range(8):
                j = 8 * lc + tb
                q2 = div6_u8(j); ja = j - 6 * q2; jb = min(q2, nb - 1)
                v = stg[rpart + ((jb * (jb + 1) // 2) * 36 + 6 * ja if mir else jb * 36 + ja)]
                assert np.isfinite(v), (lane, ta, tb)          # padding reads real entries, never unwritten storage
                if i < n and j < n:
                    assert abs(v - want[i, j]) <= 1e-12, (lane, ta, tb, i, j)
