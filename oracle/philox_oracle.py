"""TEST INFRASTRUCTURE ONLY (tests/ import it; the product never does).

numpy restatement of the device random streams of the loader (superpoint_graph_amd/csrc/spg_loader.hip,
spg_loader_random): Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11 --
multipliers 0xD2511F53 / 0xCD9E8D57, Weyl keys 0x9E3779B9 / 0xBB67AE85; known-answer vectors of the Random123
distribution are checked in tests/test_host.py) keyed by (seed, superpoint id, step).  This stream has no counterpart in
the reference (which draws from numpy's MT19937, learning/spg.py:205-257): the oracle pins the kernel's arithmetic, the
distributional equivalence to the reference's draws is tested separately (tests/test_gpu_loader.py)."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """counters: uint32 arrays (broadcastable), key: python ints -> four uint32 arrays"""
    c0, c1, c2, c3 = [np.asarray(c, dtype=np.uint64) & MASK for c in np.broadcast_arrays(c0, c1, c2, c3)]
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        n0 = (p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0)
        n1 = p1 & MASK
        n2 = (p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1)
        n3 = p0 & MASK
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return [c.astype(np.uint32) for c in (c0, c1, c2, c3)]


def _words(kind, nblocks, sp_id, seed, step):
    idl = int(sp_id) & 0xFFFFFFFF
    idh = ((int(sp_id) >> 32) & 0xFFFFFFFF) ^ (int(step) & 0xFFFFFFFF)
    w = philox4x32_10(np.arange(nblocks, dtype=np.uint64), kind, idl, idh, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    return np.stack(w, axis=1).reshape(-1)                    # element 4*q4+e = word e of block q4


def loader_random(counts, ids, slot, npts, nfeat, n_valid, seed, step, augment, scale=0.0, rot=False, mirror_prob=0.0, jitter=False):
    S = len(counts)
    sidx = np.zeros((S, npts), dtype=np.int32)
    M = np.zeros((S, 3, 3)) if augment else None
    noise = np.zeros((n_valid, npts, nfeat), dtype=np.float32) if (augment and jitter) else None
    two32 = 4294967296.0
    for s in range(S):
        n = int(counts[s])
        w = _words(0, (npts + 3) // 4, ids[s], seed, step)[:npts].astype(np.uint64)
        draw = ((w * np.uint64(max(n, 1))) >> np.uint64(32)).astype(np.int32)
        q = np.arange(npts)
        sidx[s] = np.where((n <= npts) & (q < n), q, draw)
        if augment:
            u = _words(1, 1, ids[s], seed, step).astype(np.float64)
            sc = 1.0
            if np.float32(scale) > 1:
                f = float(np.float32(scale))
                sc = 1.0 / f + (u[0] + 0.5) / two32 * (f - 1.0 / f)
            c, sn = 1.0, 0.0
            if rot:
                a = (u[1] + 0.5) / two32 * 6.283185307179586
                c, sn = np.cos(a), np.sin(a)
            mx = my = 1.0
            if mirror_prob > 0:
                mp = float(np.float32(mirror_prob))
                mx = -1.0 if (u[2] + 0.5) / two32 < 0.5 * mp else 1.0
                my = -1.0 if (u[3] + 0.5) / two32 < 0.5 * mp else 1.0
            M[s] = [[mx * c * sc, -mx * sn * sc, 0], [my * sn * sc, my * c * sc, 0], [0, 0, sc]]
            if jitter and slot[s] >= 0:
                total = npts * nfeat
                w = _words(2, (total + 3) // 4, ids[s], seed, step).reshape(-1, 4)
                f32 = np.float32
                uu = ((w >> np.uint32(8)).astype(f32) + f32(0.5)) * f32(5.9604645e-8)
                r1 = np.sqrt(f32(-2) * np.log(uu[:, 0])); r2 = np.sqrt(f32(-2) * np.log(uu[:, 2]))
                t1 = f32(6.2831853) * uu[:, 1]; t2 = f32(6.2831853) * uu[:, 3]
                z = np.stack([r1 * np.cos(t1), r1 * np.sin(t1), r2 * np.cos(t2), r2 * np.sin(t2)], axis=1).astype(f32).reshape(-1)[:total]
                noise[slot[s]] = np.clip(f32(0.01) * z, f32(-0.05), f32(0.05)).reshape(npts, nfeat)
    return sidx, M, noise
