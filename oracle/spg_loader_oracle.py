"""ORACLE (test infrastructure, never the product path): numpy restatement of the reference's per-superpoint loader
`load_superpoint` (learning/spg.py:198-236) and `augment_cloud` (:239-258), split the way the device kernel consumes
it: the random streams (resampling indices, augmentation matrix, jitter) are drawn exactly as the reference draws
them, the arithmetic is a pure function of (raw rows, indices, matrix, noise).

Pinned by oracle/validate_against_reference.py::check_loader against the IMPORTED reference function (h5py replaced
by an in-memory stub; `transforms3d`, absent here, replaced by its published closed forms: zfdir2mat(s) = s*I,
zfdir2mat(-1, e_k) = I - 2 e_k e_k^T, axangle2mat(z, a) = Rz(a)).  Golden vectors: tests/golden/loader.npz."""
import math

import numpy as np

RAW_COLUMNS = {'xyz': (0, 1, 2), 'rgb': (3, 4, 5), 'e': (6,), 'lpsv': (7, 8, 9, 10), 'XYZ': (11, 12, 13)}


def column_map(pc_attribs):
    """Raw column of every output feature, in the reference's order (spg.py:224-232; substring tests!)."""
    cols = []
    for key in ('xyz', 'rgb', 'e', 'lpsv', 'XYZ'):
        if key in pc_attribs:
            cols += RAW_COLUMNS[key]
    if 'd' in pc_attribs:
        raise NotImplementedError("pc_attribs 'd': the reference appends a 1-D column and np.concatenate fails (spg.py:231)")
    return cols


def sample_indices(n, npts, rs):
    """Row indices after the resampling of spg.py:207-214 (`rs` = numpy RandomState or the np.random module)."""
    if n > npts:
        return rs.choice(n, npts).astype(np.int32)
    if n < npts:
        return np.concatenate([np.arange(n), rs.choice(n, npts - n)]).astype(np.int32)
    return np.arange(n, dtype=np.int32)


def test_rng(sp_id, test_seed_offset=0):
    """spg.py:205: the evaluation stream is a fresh RandomState(seed = id + test_seed_offset) per superpoint."""
    return np.random.RandomState(seed=sp_id + test_seed_offset)


def normalise_and_select(rows, pc_xyznormalize, pc_attribs):
    """rows: [npts, ncols] float32 (already resampled) -> (P [npts, F] float32, diameter float32[1]); spg.py:216-232."""
    P = rows.astype(np.float32).copy()
    if pc_xyznormalize:
        diameter = np.max(np.max(P[:, :3], axis=0) - np.min(P[:, :3], axis=0))
        P[:, :3] = (P[:, :3] - np.mean(P[:, :3], axis=0, keepdims=True)) / (diameter + 1e-10)
    else:
        diameter = 0.0
        P[:, :3] = (P[:, :3] - np.mean(P[:, :3], axis=0, keepdims=True))
    if pc_attribs != '':
        P = P[:, column_map(pc_attribs)]
    return P, np.array([diameter], dtype=np.float32)


def zoom(s):
    return np.eye(3) * s


def mirror(axis):
    m = np.eye(3)
    m[axis, axis] = -1.0
    return m


def rot_z(angle):
    c, s = math.cos(angle), math.sin(angle)
    return np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])


def augment_matrix(pc_augm_scale, pc_augm_rot, pc_augm_mirror_prob, pyrandom):
    """The 3x3 matrix of augment_cloud (spg.py:241-251), consuming `pyrandom` (the `random` module) in the same order."""
    M = zoom(1)
    if pc_augm_scale > 1:
        s = pyrandom.uniform(1 / pc_augm_scale, pc_augm_scale)
        M = np.dot(zoom(s), M)
    if pc_augm_rot == 1:
        angle = pyrandom.uniform(0, 2 * math.pi)
        M = np.dot(rot_z(angle), M)
    if pc_augm_mirror_prob > 0:
        if pyrandom.random() < pc_augm_mirror_prob / 2:
            M = np.dot(mirror(0), M)
        if pyrandom.random() < pc_augm_mirror_prob / 2:
            M = np.dot(mirror(1), M)
    return M


def jitter_noise(shape, nprandom, sigma=0.01, clip=0.05):
    """spg.py:255-257"""
    return np.clip(sigma * nprandom.randn(*shape), -1 * clip, clip).astype(np.float32)


def apply_augmentation(P, M=None, noise=None):
    P = P.copy()
    if M is not None:
        P[:, :3] = np.dot(P[:, :3], M.T)
    if noise is not None:
        P = P + noise
    return P


def load_batch(points, offsets, ids, minpts, npts, pc_xyznormalize, pc_attribs, train=False, test_seed_offset=0,
               augm=None, nprandom=np.random, pyrandom=None):
    """All superpoints of a ragged buffer, as `loader` does one by one (spg.py:150-167).
    -> dict(flag i64[S], clouds f32[Nv, F, npts], diam f32[Nv], slot i32[S], sample_idx i32[S, npts], M, noise)"""
    S = len(offsets) - 1
    flag = np.zeros(S, dtype=np.int64)
    slot = np.full(S, -1, dtype=np.int32)
    sidx = np.zeros((S, npts), dtype=np.int32)
    Ms = np.tile(np.eye(3), (S, 1, 1))
    clouds, diams, noises = [], [], []
    for s in range(S):
        raw = points[offsets[s]:offsets[s + 1]]
        n = raw.shape[0]
        if n < minpts:
            flag[s] = -1
            continue
        rs = nprandom if train else test_rng(int(ids[s]), test_seed_offset)
        sidx[s] = sample_indices(n, npts, rs)
        P, d = normalise_and_select(raw[sidx[s]], pc_xyznormalize, pc_attribs)
        if train and augm is not None:
            Ms[s] = augment_matrix(augm['scale'], augm['rot'], augm['mirror_prob'], pyrandom)
            nz = jitter_noise(P.shape, nprandom) if augm['jitter'] else None
            P = apply_augmentation(P, Ms[s], nz)
            if nz is not None:
                noises.append(nz)
        slot[s] = len(clouds)
        clouds.append(P.T)
        diams.append(d)
    F = len(column_map(pc_attribs))
    return dict(flag=flag, slot=slot, sample_idx=sidx, M=Ms,
                clouds=np.stack(clouds) if clouds else np.zeros((0, F, npts), np.float32),
                diam=np.concatenate(diams) if diams else np.zeros((0,), np.float32),
                noise=np.stack(noises) if noises else None)
