"""Fused backward of the convolutions with 64 / 128 input channels (round 4; superpoint_graph_amd/csrc/spg_gemm.hip: spg_bwdpair_kernel):
the data gradient and the weight gradient of a layer come from ONE pass over dz instead of two launches that each re-read
g and y.  Per element the arithmetic is that of the separate kernels, the summation order of dW and of the BatchNorm-backward
sums differs -- so a whole training step is compared with the separate launches (spg_tune key 14 = 1) at fp32 round-off, not
bit for bit; the forward (loss, logits, embeddings) must be IDENTICAL, the fused path must be deterministic from run to run, and
the instrumented launch table must show that the fused kernel actually ran."""
import ctypes

import pytest
import torch

from conftest import build_model, load_golden, maxrel, noise_grad
from test_gpu_grouped import _step

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _fused_launches(hip):
    n = 0
    for mode in (3, 4):
        ms, cnt, fl = ctypes.c_double(), ctypes.c_long(), ctypes.c_double()
        hip.spg_prof_read_tag(hip.spg_prof_tag(4, 64, 64, mode, 1, 1), ctypes.byref(ms), ctypes.byref(cnt), ctypes.byref(fl))
        n += cnt.value
    return n


def _run(hip, spec, batch, state0, cw, separate, flat=True):
    from superpoint_graph_amd.flat import FlatParameters
    old = hip.spg_tune(14, 1 if separate else 0)
    try:
        model = build_model(spec, state0).to(DEV).train()
        arena = FlatParameters(model, lazy_zero=True) if flat else None
        hip.spg_prof_read(None, None, None, 1)
        hip.spg_prof_enable(1)
        out = _step(model, batch, cw, arena)
        hip.spg_prof_enable(0)
        n = _fused_launches(hip)
        hip.spg_prof_read(None, None, None, 1)
        return out, n
    finally:
        hip.spg_prof_enable(0)
        hip.spg_tune(14, old)


def _compare(a, b, tol):
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])      # the forward is untouched
    worst = ('', 0.0)
    for k in a[3]:
        if noise_grad(k, a[3]):
            continue
        e = maxrel(b[3][k], a[3][k])
        if e > worst[1]:
            worst = (k, e)
    assert worst[1] < tol, worst
    for k in a[4]:
        assert torch.equal(a[4][k], b[4][k]), k
    return worst


@pytest.mark.parametrize('tag', ['s3dis_gru10_matrix', 'vector_gru4_small'])
def test_fused_narrow_backward_matches_separate_launches(hip, tag):
    spec, batch, state0, g = load_golden(tag)
    cw = torch.from_numpy(g['class_weights']).to(DEV) if 'class_weights' in g.files else None
    sep, n_sep = _run(hip, spec, batch, state0, cw, True)
    fus, n_fus = _run(hip, spec, batch, state0, cw, False)
    assert n_sep == 0
    # conv2 / conv3 / conv4 of the main network, conv2 / conv3 of the STN (where they have 64 / 128 input channels), and since round 6 the
    # pooled layer conv5 (128 -> 256) as TWO launches over the halves of its output channels
    assert n_fus == (7 if tag == 's3dis_gru10_matrix' else 2)
    worst = _compare(sep, fus, 2e-5)
    print(f'{tag}: {n_fus} fused launches, worst gradient difference {worst[1]:.2e} ({worst[0]})')
    again, _ = _run(hip, spec, batch, state0, cw, False)
    for k in fus[3]:
        assert torch.equal(fus[3][k], again[3][k]), k      # fixed tile -> workgroup map, fixed-point statistics: deterministic


@pytest.mark.parametrize('n_sp,n_edges', [(1000, 5000), (300, 1500), (7, 20)])
def test_fused_narrow_backward_on_scenes(hip, n_sp, n_edges):
    """BASELINE-size scene (four tiles per workgroup on 256 CUs), a scene with about one tile per workgroup, a tiny one."""
    from oracle import spg_oracle as O
    from superpoint_graph_amd import synth
    spec = O.ModelSpec()
    col = synth.collate_numpy([synth.scene(5, n_sp=n_sp, n_edges=n_edges)])
    idxn, degs, ef, _ = O.set_batch(col['edge_lists'], col['vcounts'], col['edge_feats'])
    batch = dict(clouds_flag=torch.from_numpy(col['clouds_flag']), clouds=torch.from_numpy(col['clouds']),
                 clouds_global=torch.from_numpy(col['clouds_global']), idxn=torch.from_numpy(idxn), degs=torch.from_numpy(degs),
                 edgefeats=torch.from_numpy(ef), label_mode=torch.from_numpy(col['targets'][:, 0].copy()))
    torch.manual_seed(2)
    ref = build_model(spec)
    with torch.no_grad():
        ref.ptn.stn.proj.weight.normal_(0, 0.02)
    state0 = {k: v.clone() for k, v in ref.state_dict().items()}
    sep, n_sep = _run(hip, spec, batch, state0, None, True)
    fus, n_fus = _run(hip, spec, batch, state0, None, False)
    assert n_sep == 0 and n_fus == 7      # conv2 / conv3 of the STN, conv2 / conv3 / conv4 of the main network, conv5 in two launches (round 6)
    worst = _compare(sep, fus, 2e-5)
    print(f'{n_sp} superpoints: worst gradient difference {worst[1]:.2e} ({worst[0]})')


def test_pooled_layer_two_pass_pair_matches_its_separate_launches(hip):
    """Round 6: the pooled layer conv5 (128 -> 256; its weight matrix does not fit LDS) runs the fused pair as TWO launches over the
    halves of its output channels -- pass 1 stores its partial data gradient, pass 2 adds it and runs the epilogue.  Against the
    same step with ONLY that layer on the separate weight- / data-gradient launches (spg_tune key 22 = 1; every other layer
    fused on both sides): forward identical, every gradient within 2e-5 of its tensor's maximum, deterministic."""
    from oracle import spg_oracle as O
    from superpoint_graph_amd import synth
    spec = O.ModelSpec()
    col = synth.collate_numpy([synth.scene(9, n_sp=1000, n_edges=5000)])
    idxn, degs, ef, _ = O.set_batch(col['edge_lists'], col['vcounts'], col['edge_feats'])
    batch = dict(clouds_flag=torch.from_numpy(col['clouds_flag']), clouds=torch.from_numpy(col['clouds']),
                 clouds_global=torch.from_numpy(col['clouds_global']), idxn=torch.from_numpy(idxn), degs=torch.from_numpy(degs),
                 edgefeats=torch.from_numpy(ef), label_mode=torch.from_numpy(col['targets'][:, 0].copy()))
    torch.manual_seed(4)
    ref = build_model(spec)
    with torch.no_grad():
        ref.ptn.stn.proj.weight.normal_(0, 0.02)
    state0 = {k: v.clone() for k, v in ref.state_dict().items()}
    old = hip.spg_tune(22, 1)
    try:
        one, n_one = _run(hip, spec, batch, state0, None, False)
    finally:
        hip.spg_tune(22, old)
    two, n_two = _run(hip, spec, batch, state0, None, False)
    assert n_one == 5 and n_two == 7
    worst = _compare(one, two, 2e-5)
    print(f'two-pass pooled layer: worst gradient difference {worst[1]:.2e} ({worst[0]})')
    again, _ = _run(hip, spec, batch, state0, None, False)
    for k in two[3]:
        assert torch.equal(two[3][k], again[3][k]), k
