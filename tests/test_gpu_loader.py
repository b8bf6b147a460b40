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
