"""GPU: the device-side superpoint loader (spg_load_superpoints) against the clouds produced by the imported
reference's load_superpoint / augment_cloud (tests/golden/loader.npz) and against the numpy oracle on a larger
ragged buffer.  Evaluation mode (no augmentation) is bit-exact; with the fp64 augmentation matrix one float32 ulp."""
import os
import random as pyrandom
import types

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import spg_loader_oracle as L

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _args(attribs, norm, **kw):
    a = dict(ptn_minpts=40, ptn_npts=128, pc_xyznormalize=norm, pc_attribs=attribs, pc_augm_scale=0, pc_augm_rot=0,
             pc_augm_mirror_prob=0, pc_augm_jitter=0)
    a.update(kw)
    return types.SimpleNamespace(**a)


@pytest.mark.parametrize('tag,attribs,norm', [('s3dis', 'xyzrgbelpsvXYZ', 1), ('sema3d', 'xyzrgbelpsv', 1),
                                              ('nonorm', 'xyzelpsv', 0)])
def test_eval_mode_bit_exact_vs_reference(hip, tag, attribs, norm):
    from superpoint_graph_amd.learning import spg
    g = np.load(os.path.join(GOLDEN, 'loader.npz'))
    pts = torch.from_numpy(g['points']).to(DEV)
    flag, clouds, diam = spg.load_superpoints_device(_args(attribs, norm), pts, g['offsets'], g['ids'], False, 3)
    assert np.array_equal(flag.numpy(), g[f'{tag}/flag'])
    assert np.array_equal(clouds.cpu().numpy(), g[f'{tag}/clouds'])          # bit-exact, incl. the resampling stream
    assert np.array_equal(diam.cpu().numpy(), g[f'{tag}/diam'])


def test_train_mode_augmentation_vs_reference(hip):
    from superpoint_graph_amd.learning import spg
    g = np.load(os.path.join(GOLDEN, 'loader.npz'))
    pts = torch.from_numpy(g['points']).to(DEV)
    args = _args('xyzrgbelpsvXYZ', 1, pc_augm_scale=1.1, pc_augm_rot=1, pc_augm_mirror_prob=1.0, pc_augm_jitter=1)
    np.random.seed(5); pyrandom.seed(6)                                       # the seeds the golden run used
    flag, clouds, diam = spg.load_superpoints_device(args, pts, g['offsets'], g['ids'], True)
    ref = g['train/clouds']
    out = clouds.cpu().numpy()
    assert np.array_equal(diam.cpu().numpy(), g['train/diam'])
    assert np.array_equal(out[:, 3:], ref[:, 3:])                             # untouched by the rotation: bit-exact
    ulp = np.spacing(np.abs(ref[:, :3]).astype(np.float32))
    assert np.all(np.abs(out[:, :3] - ref[:, :3]) <= 2 * ulp + 1e-9)          # fp64 3x3 product, then + jitter
    assert np.mean(out[:, :3] == ref[:, :3]) > 0.99


def test_large_ragged_buffer_vs_oracle_and_pointnet_layout(hip):
    """5000 superpoints with log-normal sizes (SURVEY.md 8d scene statistics) against the numpy oracle."""
    from superpoint_graph_amd import ops
    rng = np.random.default_rng(3)
    S = 5000
    counts = np.clip(np.round(rng.lognormal(np.log(300), 1.0, S)), 1, 10000).astype(np.int64)
    offsets = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    points = rng.normal(size=(int(offsets[-1]), 14)).astype(np.float32)
    points[:, :3] += rng.normal(size=(1, 3)).astype(np.float32) * 20
    ids = np.arange(S)
    m = L.load_batch(points, offsets, ids, 40, 128, 1, 'xyzrgbelpsv', train=False)
    dev = torch.device(DEV)
    clouds, diam = ops.load_superpoints(torch.from_numpy(points).to(dev), torch.from_numpy(offsets).to(dev),
                                        torch.from_numpy(m['slot']).to(dev), torch.from_numpy(m['sample_idx']).to(dev),
                                        L.column_map('xyzrgbelpsv'), True, int((m['flag'] == 0).sum()))
    assert np.array_equal(clouds.cpu().numpy(), m['clouds']) and np.array_equal(diam.cpu().numpy(), m['diam'])
    assert clouds.shape[1:] == (11, 128) and clouds.is_contiguous()          # [Nv, F, P]: what PointNet.forward takes
    with pytest.raises(RuntimeError):
        ops.load_superpoints(torch.from_numpy(points).to(dev), torch.from_numpy(offsets).to(dev),
                             torch.from_numpy(m['slot']).to(dev), torch.from_numpy(m['sample_idx']).to(dev), [0, 1, 99], True, 1)


# ---------------------------------------------------------------------------------------------------------------------
# device random streams (`--loader_rng device`)
# ---------------------------------------------------------------------------------------------------------------------
def test_device_random_streams_vs_philox_oracle(hip):
    from oracle import philox_oracle as P
    from superpoint_graph_amd import ops
    rng = np.random.default_rng(11)
    S, npts, F = 300, 128, 11
    counts = np.clip(np.round(rng.lognormal(np.log(150), 1.0, S)), 1, 5000).astype(np.int64)
    counts[:3] = [128, 1, 127]
    ids = rng.integers(0, 2 ** 40, S).astype(np.int64)
    flag = counts >= 40
    slot = np.full(S, -1, dtype=np.int32); slot[flag] = np.arange(flag.sum(), dtype=np.int32)
    nv = int(flag.sum())
    seed, step = 0x1234567890ABCDEF, 7
    t = lambda a: torch.from_numpy(a).to(DEV)
    sidx, M, noise = ops.loader_random(t(counts), t(ids), t(slot), npts, F, nv, seed, step, True, 1.1, True, 0.6, True)
    o_sidx, o_M, o_noise = P.loader_random(counts, ids, slot, npts, F, nv, seed, step, True, 1.1, True, 0.6, True)
    assert np.array_equal(sidx.cpu().numpy(), o_sidx)                        # integer stream: bit-exact
    assert np.all((o_sidx >= 0) & (o_sidx < np.maximum(counts, 1)[:, None]))
    np.testing.assert_allclose(M.cpu().numpy(), o_M, rtol=0, atol=1e-14)     # fp64 cos / sin of the device vs numpy
    np.testing.assert_allclose(noise.cpu().numpy(), o_noise, rtol=0, atol=2e-7)   # fp32 logf / cosf: ~1e-5 relative of 0.01
    # a different step / seed gives a different stream; the same key the same stream
    s2, _, _ = ops.loader_random(t(counts), t(ids), t(slot), npts, F, nv, seed, step + 1, True, 1.1, True, 0.6, True)
    s3, _, _ = ops.loader_random(t(counts), t(ids), t(slot), npts, F, nv, seed, step, False)
    assert not torch.equal(s2, sidx) and torch.equal(s3, sidx)


def test_device_random_streams_distributions(hip):
    """Same distributions as the reference's draws (learning/spg.py:207-214, 241-257): uniform resampling with
    replacement, U(1/s, s) scale, U(0, 2pi) rotation, two Bernoulli(p/2) mirrors, N(0, 0.01^2) jitter clipped at 0.05."""
    from superpoint_graph_amd import ops
    S, npts, F = 4000, 128, 8
    counts = np.full(S, 1000, dtype=np.int64); counts[::2] = 50
    ids = np.arange(S, dtype=np.int64)
    slot = np.arange(S, dtype=np.int32)
    t = lambda a: torch.from_numpy(a).to(DEV)
    sidx, M, noise = ops.loader_random(t(counts), t(ids), t(slot), npts, F, S, 3, 1, True, 1.1, True, 0.8, True)
    sidx, M, noise = sidx.cpu().numpy(), M.cpu().numpy(), noise.cpu().numpy()
    big = sidx[1::2]                                                         # n = 1000 > npts: all drawn
    hist = np.bincount(big.reshape(-1), minlength=1000)
    exp = big.size / 1000
    chi2 = ((hist - exp) ** 2 / exp).sum()
    assert 800 < chi2 < 1200, chi2                                           # 999 dof: mean 999, sd 45
    small = sidx[::2]                                                        # n = 50 < npts: identity then draws
    assert np.array_equal(small[:, :50], np.tile(np.arange(50), (S // 2, 1)))
    assert small[:, 50:].min() == 0 and small[:, 50:].max() == 49
    sc = M[:, 2, 2]
    assert sc.min() >= 1 / 1.1 - 1e-6 and sc.max() <= 1.1 + 1e-6 and abs(sc.mean() - (1.1 + 1 / 1.1) / 2) < 3e-3
    det = np.linalg.det(M / sc[:, None, None])
    np.testing.assert_allclose(np.abs(det), 1.0, atol=1e-12)
    mx = np.sign(M[:, 0, 0] * M[:, 1, 1] - M[:, 0, 1] * M[:, 1, 0])           # -1 iff exactly one mirror
    p = 0.4
    assert abs((mx < 0).mean() - 2 * p * (1 - p)) < 0.03
    ang = np.arctan2(-M[:, 0, 1] * np.sign(M[:, 0, 0] ** 2 + 1), M[:, 0, 0])  # angle up to the mirror sign: uniform either way
    h, _ = np.histogram(ang, bins=8, range=(-np.pi, np.pi))
    assert h.min() > 0.8 * S / 8 and h.max() < 1.2 * S / 8
    assert abs(noise.mean()) < 2e-5 and abs(noise.std() - 0.01) < 1e-4 and np.abs(noise).max() <= 0.05 + 1e-9
    z = noise.reshape(-1) / 0.01
    assert abs((np.abs(z) < 1).mean() - 0.6827) < 2e-3 and abs(np.mean(z ** 4) - 3.0) < 0.05


def test_device_rng_loader_end_to_end(hip):
    """rng='device' through load_superpoints_device: evaluation = the oracle's clouds for the oracle's indices, and
    reproducible; training = augmented clouds whose per-superpoint statistics match the host-stream mode."""
    from oracle import philox_oracle as P
    from superpoint_graph_amd.learning import spg
    g = np.load(os.path.join(GOLDEN, 'loader.npz'))
    pts = torch.from_numpy(g['points']).to(DEV)
    args = _args('xyzrgbelpsvXYZ', 1, loader_rng='device', seed=5)
    flag, clouds, diam = spg.load_superpoints_device(args, pts, g['offsets'], g['ids'], False, 3)
    flag2, clouds2, _ = spg.load_superpoints_device(args, pts, g['offsets'], g['ids'], False, 3)
    assert torch.equal(clouds, clouds2) and np.array_equal(flag.numpy(), g['s3dis/flag'])
    counts = np.diff(g['offsets'])
    slot = np.full(len(counts), -1, dtype=np.int32); slot[flag.numpy() == 0] = np.arange(int((flag == 0).sum()))
    sidx, _, _ = P.loader_random(counts, g['ids'], slot, 128, 14, int((flag == 0).sum()), 5 + 3, 0, False)
    both = [L.normalise_and_select(g['points'][g['offsets'][s]:g['offsets'][s + 1]][sidx[s]], 1, 'xyzrgbelpsvXYZ')
            for s in range(len(counts)) if slot[s] >= 0]
    assert np.array_equal(clouds.cpu().numpy(), np.stack([b[0].T for b in both]))
    assert np.array_equal(diam.cpu().numpy(), np.concatenate([b[1] for b in both]))       # the diameter of the RESAMPLED cloud (spg.py:216-219)
    # training: streams advance between calls, the clouds stay valid augmentations of the same superpoints
    args = _args('xyzrgbelpsvXYZ', 1, loader_rng='device', seed=5, pc_augm_scale=1.1, pc_augm_rot=1, pc_augm_mirror_prob=1.0, pc_augm_jitter=1)
    _, c1, d1 = spg.load_superpoints_device(args, pts, g['offsets'], g['ids'], True)
    _, c2, d2 = spg.load_superpoints_device(args, pts, g['offsets'], g['ids'], True)
    assert not torch.equal(c1, c2)
    assert torch.isfinite(c1).all()
    r1 = c1[:, :2].pow(2).sum(1).sqrt().amax(1)                              # horizontal extent: within the scale band of the raw cloud
    r0 = clouds[:, :2].pow(2).sum(1).sqrt().amax(1)
    assert ((r1 / r0.clamp_min(1e-6)) < 1.1 * 1.5).all()
