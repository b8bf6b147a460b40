"""BASELINE-size parity (BASELINE.json configs[1]/[2]: 1000 superpoints x 128 points x 14 features, 5000 superedges, the S3DIS
production model `gru_10_0,f_13`, fp32):

  * against the IMPORTED REFERENCE run on this very scene (tests/golden/baseline_size.npz, written by
    oracle/validate_against_reference.py::check_baseline_size): eval / train embeddings and logits element-wise at 1e-4, loss,
    the gradients behind the max-pool and the running statistics;
  * ALL gradients with the DECISIONS HELD EQUAL: ReLU (which side of zero; PointNet and the filter network) and max-pool
    (which point wins) are the non-smooth operations of the path; on a near-tie two correct fp32 implementations may decide
    differently and their gradients then differ by far more than round-off.  The test (1) reads the HIP path's own decisions
    out of its workspace, (2) asserts they differ from the fp64 oracle's only on near-ties (stated distance), (3) runs the
    oracle backward in fp64 WITH the HIP decisions and requires EVERY gradient tensor within 1e-4 -- a backward bug of
    relative size 5e-3 in a convolution layer fails this test (round 2 allowed 1e-2 there)."""
import ctypes
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN, assert_elementwise, build_model, maxrel, noise_grad
from oracle import spg_oracle as O
from oracle import validate_against_reference as V
from test_gpu_model import _run

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.fixture(scope='module')
def setup():
    g = np.load(os.path.join(GOLDEN, 'baseline_size.npz'))
    spec = O.ModelSpec()
    state0 = {k[7:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('state0/')}      # the reference run's initial state
    assert V.state_digest(state0) == str(g['state0_sha256'])
    return g, spec, V.baseline_size_batch(0), state0


def test_baseline_size_vs_reference_golden(hip, setup):
    g, spec, batch, state0 = setup
    model = build_model(spec, state0).to(DEV).eval()
    with torch.no_grad():
        emb, logits, _ = _run(model, batch)
    assert_elementwise(emb, g['eval/emb'], what='eval embeddings vs reference')
    assert_elementwise(logits, g['eval/logits'], what='eval logits vs reference')
    assert torch.equal(logits.argmax(1).cpu(), torch.from_numpy(g['eval/logits']).argmax(1))
    model.load_state_dict(state0)
    model.train()
    cw = torch.from_numpy(g['class_weights']).to(DEV)
    emb, logits, embedder = _run(model, batch)
    loss = F.cross_entropy(logits, batch['label_mode'].to(DEV), weight=cw)
    model.zero_grad()
    loss.backward()
    embedder.bw_hook()
    assert_elementwise(emb, g['train/emb'], what='train embeddings vs reference')
    assert_elementwise(logits, g['train/logits'], what='train logits vs reference')
    assert abs(float(loss) - float(g['train/loss'])) <= 1e-5 * abs(float(g['train/loss']))
    grads = {k: p.grad for k, p in model.named_parameters()}
    err = {}
    gref = {k: torch.from_numpy(g['grad/' + k]) for k in grads}
    for k in grads:
        if not noise_grad(k, gref):
            err[k] = maxrel(grads[k], gref[k])
    # Tensors with NO ReLU / max-pool decision of their own network upstream of them in the backward pass: the recurrent
    # cell, the classifier, the filter network from its BatchNorm on, PointNet's FC head.  For the others (PointNet / STN
    # convolutions; the first two filter-network layers) the REFERENCE's fp32 run itself sits on the other side of a few
    # near-ties than float64 does (16 of 1.2e8 ReLU decisions; oracle/validate_against_reference.py shows the reference at
    # 3.5e-3 from float64 on ecc.0._fnet.2.weight): they are compared with the decisions held equal in the next test, here
    # only loosely.
    smooth = {k: e for k, e in err.items() if k.startswith('ecc.0._cell') or k.startswith('ecc.1') or k.startswith('ptn.fcs.') or
              k.startswith('ecc.0._fnet.4') or k.startswith('ecc.0._fnet.5') or k.startswith('ecc.0._fnet.7')}
    rest = {k: e for k, e in err.items() if k not in smooth}
    print('gradients vs the reference: decision-free tensors worst %.2e; decision-dependent ones worst %.2e (see the conditioned test)'
          % (max(smooth.values()), max(rest.values())))
    assert max(smooth.values()) < 1e-4, {k: e for k, e in smooth.items() if e >= 1e-4}
    # (loose by construction: WHICH near-ties fall on the reference's side depends on the last bits of the forward.  Round 5's one-pass
    #  first layers take the first BatchNorm's statistics from the exact Gram matrix instead of from rounded fp32 outputs: one
    #  max-pool winner of the 128 -> 256 layer now differs from the reference's fp32 run -- 3.4e-2 of that layer's weight gradient,
    #  1e-2 before.  The decision-conditioned tests below hold every tensor to 1e-4.)
    # FROZEN (round 6): this bound is not to move again.  A kernel change that flips a further near-tie shows up
    # as a COUNT in the conditioned test's log (`decisions: ReLU a / b differ, max-pool c / d differ`, run with -s;
    # profiles/r06_parity_decisions.txt holds round 6's) and has to be argued there, not here.
    assert max(rest.values()) < 5e-2, {k: e for k, e in rest.items() if e >= 5e-2}
    sd = model.state_dict()
    for k in g.files:
        if k.startswith('state1/'):
            assert maxrel(sd[k[7:]].double(), torch.from_numpy(g[k]).double()) < 1e-5, k


def _hip_decisions(ptn, st):
    """ReLU masks and max-pool winners of the HIP forward, read from its workspace (spg_pointnet_debug_offset): the kernels
    decide `fmaf(y, s, t) > 0` on the raw layer output y with the BatchNorm scale / shift (spg_gemm.hip backward epilogue,
    spg_common.h AFFINE prologue); the sign of the exactly rounded fma equals the sign of the exact y*s + t, evaluated here in
    float64 (the product of two floats is exact in float64)."""
    from superpoint_graph_amd import _lib
    L = _lib.lib()
    cfg, B, ws = st.cfg, st.B, st.ws
    Pn = cfg.npts

    def buf(layer, what, n, dtype=torch.float32):
        off = L.spg_pointnet_debug_offset(ctypes.byref(cfg), B, 1, layer, what)
        assert off >= 0, (layer, what)
        return ws[off:off + 4 * n].view(dtype)

    dec, li = {}, 0

    def segment(pfx, convs, fcs_bn, pool_layer):
        nonlocal li
        for i, c in enumerate(convs):
            y = buf(li, 0, B * Pn * c).view(B * Pn, c).double()
            v = y * buf(li, 1, c).double() + buf(li, 2, c).double()
            dec[f'{pfx}.convs.{3 * i + 1}'] = (v > 0).cpu()
            li += 1
        ld = int(L.spg_pointnet_debug_offset(ctypes.byref(cfg), B, 1, pool_layer, 2))
        dec[f'{pfx}.pool'] = buf(pool_layer, 1, B * ld, torch.int32).view(B, ld)[:, :convs[-1]].long().cpu()
        for i, c in enumerate(fcs_bn):
            y = buf(li, 0, B * c).view(B, c).double()
            v = y * buf(li, 1, c).double() + buf(li, 2, c).double()
            dec[f'{pfx}.fcs.{3 * i + 1}'] = (v > 0).cpu()
            li += 1
        li += 1          # the segment's last, plain layer (STN projection / embedding)

    segment('ptn.stn', list(ptn.stn._nf_conv), list(ptn.stn._nf_fc), -1)
    segment('ptn', list(ptn._nf_conv), list(ptn._nf_fc[:-1]), -2)
    return dec


def _hip_fnet_decisions(st, spec):
    """ReLU decisions of the filter-generating network (5000 edges x 32 / 128 / 64 channels): `y > 0` for the plain layers,
    `fmaf(y, s, t) > 0` behind the BatchNorm layer -- from the RNN-ECC forward workspace (spg_eccrnn_debug_offset)."""
    from superpoint_graph_amd import _lib
    L = _lib.lib()
    cfg, ws, N, E = st.cfg, st.ws, st.graph.N, st.graph.E
    dec = {}
    for k, c in enumerate(spec.fnet_widths):
        def buf(what, n):
            off = L.spg_eccrnn_debug_offset(ctypes.byref(cfg), N, E, 1, k, what)
            assert off >= 0, (k, what)
            return ws[off:off + 4 * n].view(torch.float32)
        v = buf(0, E * c).view(E, c).double()
        if spec.fnet_bnidx == k:
            v = v * buf(1, c).double() + buf(2, c).double()
        dec[f'ecc.0._fnet.relu{k}'] = (v > 0).cpu()
    return dec


def test_decision_conditioned_gradients_at_baseline_size(hip, setup, monkeypatch):
    g, spec, batch, state0 = setup
    _decision_conditioned(spec, batch, state0, torch.from_numpy(g['class_weights']), monkeypatch, free_run=True)


def _fused_train_step(spec, batch, state0, cw):
    """One training step through superpoint_graph_amd.fused.FusedStep with its DEFAULTS -- what bench.py times and what the CLI
    runs with --fused_step 1: spg_train_step, classifier + cross entropy inside the one-launch recurrence, fused convolution
    backward, grouped launches.  -> (model, step, loss, logits, embeddings [N, nf])"""
    from superpoint_graph_amd.flat import FlatParameters
    from superpoint_graph_amd.fused import FusedStep
    from superpoint_graph_amd.learning import ecc
    model = build_model(spec, state0).to(DEV).train()
    arena = FlatParameters(model, lazy_zero=True, host_counters=True)
    gi = ecc.GraphConvInfo.from_buffers(batch['idxn'].clone(), batch['degs'].clone(), batch['edgefeats'].clone())
    model.ecc.set_info([gi], 1)
    step = FusedStep(model, arena, class_weights=None if cw is None else cw.to(DEV))
    arena.zero_grad()
    loss, logits = step(batch['clouds_flag'], batch['clouds'], batch['clouds_global'], gi, batch['label_mode'].to(DEV))
    torch.cuda.synchronize()
    return model, step, loss, logits, step.embeddings


def _decision_conditioned(spec, batch, state0, cw, monkeypatch, free_run, fused=False, tol=1e-4, tie_tol=1e-4, loss_tol=1e-5):
    """free_run: also run the fp64 oracle with its OWN decisions (near-tie statistics against the unconditioned values, the
    unconditioned gradient error for the log) -- one more fp64 pass; the large configurations check the near-ties on the
    values of the conditioned pass instead.  fused: the step under test is FusedStep (spg_train_step) instead of the modules.
    tol: bound on every gradient tensor (max-norm relative); tie_tol: how far from a tie (relative to the layer's largest value) a
    decision may differ from the fp64 one; loss_tol -- the fp32 path's are the defaults, the opt-in precision modes state their own
    (tests/test_gpu_precision.py).  -> (worst gradient error, its tensor)"""
    from superpoint_graph_amd import ops
    captured = {}
    if fused:
        model, step, loss, logits, _ = _fused_train_step(spec, batch, state0, cw)
        captured['state'], captured['ecc'] = step.debug_states()
    else:
        model = build_model(spec, state0).to(DEV).train()
        real_forward = ops.pointnet_forward

        def spy(*a, **kw):
            out = real_forward(*a, **kw)
            captured['state'] = out[1]
            return out
        monkeypatch.setattr(ops, 'pointnet_forward', spy)
        real_ecc = ops.eccrnn_forward

        def spy_ecc(*a, **kw):
            out = real_ecc(*a, **kw)
            captured['ecc'] = out[1]
            return out
        monkeypatch.setattr(ops, 'eccrnn_forward', spy_ecc)
        emb, logits, embedder = _run(model, batch)
        loss = F.cross_entropy(logits, batch['label_mode'].to(DEV), weight=None if cw is None else cw.to(DEV))
        model.zero_grad()
        loss.backward()
        embedder.bw_hook()
    torch.cuda.synchronize()
    grads = {k: p.grad.detach().cpu() for k, p in model.named_parameters()}
    dec = _hip_decisions(model.ptn, captured['state'])
    dec.update(_hip_fnet_decisions(captured['ecc'], spec))

    # (2) the fp64 oracle's own view of every decision
    torch.set_num_threads(min(32, torch.get_num_threads()))
    rec = {}
    st = {k: v.clone() for k, v in state0.items()}
    if free_run:
        l64, _, _, g64_free = O.train_step(batch, spec, st, cw, dtype=torch.float64, update_running_stats=False, rec=rec)
    else:       # ONE fp64 pass: the conditioned one, recording the values its ReLUs / max-pools saw
        lc, _, _, g64 = O.train_step(batch, spec, st, cw, dtype=torch.float64, update_running_stats=False, dec=dec, rec=rec)
        g64_free = g64
    n_relu = n_relu_diff = n_pool = n_pool_diff = 0
    for key, d in dec.items():
        if key.endswith('.pool'):
            h = rec[key]                                           # [B, Pn, C] post-ReLU values the max-pool saw
            best = h.max(1)[0]
            at_hip = h.gather(1, d.unsqueeze(1)).squeeze(1)
            gap = (best - at_hip)                                  # >= 0; 0 where the HIP winner attains the maximum
            scale = float(h.abs().max())
            n_pool += d.numel(); n_pool_diff += int((gap > 0).sum())
            print(f'  {key}: {int((gap > 0).sum())} of {d.numel()} winners are not the fp64 maximum; worst gap {float(gap.max()):.2e} (scale {scale:.2e})')
            assert float(gap.max()) <= tie_tol * scale, key           # a different winner only on a near-tie
        else:
            v = rec[key].reshape(d.shape)                          # value the ReLU saw (fp64 oracle)
            differ = (v > 0) != d
            scale = float(v.abs().max())
            worst = float(v[differ].abs().max()) if bool(differ.any()) else 0.0
            n_relu += d.numel(); n_relu_diff += int(differ.sum())
            print(f'  {key}: {int(differ.sum())} of {d.numel()} ReLU decisions differ; worst |v| there {worst:.2e} (scale {scale:.2e})')
            assert worst <= tie_tol * scale, key                      # a different side of zero only within round-off of zero
            assert int(differ.sum()) <= 10 * tie_tol * d.numel(), key
    print(f'decisions: ReLU {n_relu_diff} / {n_relu} differ, max-pool {n_pool_diff} / {n_pool} differ')
    del rec

    # (3) fp64 oracle backward with the HIP path's decisions: every gradient tensor within 1e-4
    if free_run:
        st = {k: v.clone() for k, v in state0.items()}
        lc, _, _, g64 = O.train_step(batch, spec, st, cw, dtype=torch.float64, update_running_stats=False, dec=dec)
    assert abs(float(loss) - float(lc)) <= loss_tol * abs(float(lc))
    err, err_free = {}, {}
    for k, ref in g64.items():
        if not noise_grad(k, g64):
            err[k] = maxrel(grads[k], ref)
            err_free[k] = maxrel(grads[k], g64_free[k])
    worst = max(err, key=err.get)
    print('conditioned: worst %s %.2e; unconditioned (fp64 oracle with its own decisions): worst %.2e' %
          (worst, err[worst], max(err_free.values())))
    for k in sorted(err, key=err.get, reverse=True)[:8]:
        print(f'  {k}: conditioned {err[k]:.2e}  unconditioned {err_free[k]:.2e}')
    assert err[worst] <= tol, {k: e for k, e in err.items() if e > tol}
    return err[worst], worst


def test_two_scene_batch_vs_reference_golden(hip, setup):
    """The reference's default batch (learning/main.py:49 --batch_size 2; BASELINE.json configs[2]): the IMPORTED reference on
    two BASELINE-shaped scenes (tests/golden/two_scenes.npz, oracle/validate_against_reference.py::check_two_scenes; initial
    state = the BASELINE-size golden's).  2000 nodes: the RNN-ECC recurrence of a multi-scene batch, BatchNorm statistics
    over both scenes.  Embeddings / logits element-wise at 1e-4, loss, decision-free gradients 1e-4, the rest loosely (near-ties
    of the reference's own fp32 run; the conditioned test below is the sharp one), running statistics."""
    g0, spec, _, state0 = setup
    g = np.load(os.path.join(GOLDEN, 'two_scenes.npz'))
    assert str(g['state0_sha256']) == str(g0['state0_sha256'])
    batch = V.two_scene_batch()
    model = build_model(spec, state0).to(DEV).train()
    cw = torch.from_numpy(g['class_weights']).to(DEV)
    emb, logits, embedder = _run(model, batch)
    loss = F.cross_entropy(logits, batch['label_mode'].to(DEV), weight=cw)
    model.zero_grad()
    loss.backward()
    embedder.bw_hook()
    assert_elementwise(emb, g['train/emb'], what='2 scenes: train embeddings vs reference')
    assert_elementwise(logits, g['train/logits'], what='2 scenes: train logits vs reference')
    assert abs(float(loss) - float(g['train/loss'])) <= 1e-5 * abs(float(g['train/loss']))
    grads = {k: p.grad for k, p in model.named_parameters()}
    gref = {k: torch.from_numpy(g['grad/' + k]) for k in grads}
    err = {k: maxrel(grads[k], gref[k]) for k in grads if not noise_grad(k, gref)}
    # decision-free: nothing non-smooth of PointNet upstream of them in the backward.  (PointNet's FC head counted as such at
    # one scene; with 2000 rows behind its two BatchNorm + ReLU layers the reference's fp32 run and the kernels fall on different
    # sides of a near-tie somewhere -- one flipped ReLU moves a row of dW by ~1e-2 of the tensor maximum, measured 4.8e-2 on
    # ptn.fcs.0.weight; the decision-conditioned tests are the sharp check of those tensors, incl. at 8 scenes.)
    smooth = {k: e for k, e in err.items() if k.startswith('ecc.0._cell') or k.startswith('ecc.1') or
              k.startswith('ecc.0._fnet.4') or k.startswith('ecc.0._fnet.5') or k.startswith('ecc.0._fnet.7')}
    rest = {k: e for k, e in err.items() if k not in smooth}
    print('2 scenes, gradients vs the reference: decision-free tensors worst %.2e; decision-dependent ones worst %.2e' % (max(smooth.values()), max(rest.values())))
    assert max(smooth.values()) < 1e-4, {k: e for k, e in smooth.items() if e >= 1e-4}
    assert max(rest.values()) < 1e-1, {k: e for k, e in rest.items() if e >= 1e-1}
    sd = model.state_dict()
    for k in g.files:
        if k.startswith('state1/'):
            assert maxrel(sd[k[7:]].double(), torch.from_numpy(g[k]).double()) < 1e-5, k


_SMOOTH_1 = ('ecc.0._cell', 'ecc.1', 'ptn.fcs.', 'ecc.0._fnet.4', 'ecc.0._fnet.5', 'ecc.0._fnet.7')
_SMOOTH_2 = ('ecc.0._cell', 'ecc.1', 'ecc.0._fnet.4', 'ecc.0._fnet.5', 'ecc.0._fnet.7')


@pytest.mark.parametrize('which', ['baseline_size', 'two_scenes'])
def test_fused_step_vs_reference_golden(hip, setup, which):
    """the path bench.py TIMES -- FusedStep with its defaults (spg_train_step: classifier + cross entropy inside
    the recurrence, fused convolution backward, grouped launches, riders) -- against the IMPORTED REFERENCE's outputs directly, at
    BASELINE size (baseline_size.npz) and on the reference's default 2-scene batch (two_scenes.npz; 2000 nodes = one round of the
    two-workgroups-per-CU recurrence): embeddings / logits element-wise 1e-4, loss 1e-5, decision-free gradients 1e-4, running
    statistics 1e-5.  Until now this path was tied to the reference only through HIP-vs-HIP comparisons with the module path."""
    g0, spec, batch, state0 = setup
    g, smooth_pfx, loose = g0, _SMOOTH_1, 5e-2
    if which == 'two_scenes':
        g = np.load(os.path.join(GOLDEN, 'two_scenes.npz'))
        assert str(g['state0_sha256']) == str(g0['state0_sha256'])
        batch, smooth_pfx, loose = V.two_scene_batch(), _SMOOTH_2, 1e-1
    cw = torch.from_numpy(g['class_weights'])
    model, step, loss, logits, emb = _fused_train_step(spec, batch, state0, cw)
    assert_elementwise(emb, g['train/emb'], what=f'{which}: FusedStep embeddings vs reference')
    assert_elementwise(logits, g['train/logits'], what=f'{which}: FusedStep logits vs reference')
    assert abs(float(loss) - float(g['train/loss'])) <= 1e-5 * abs(float(g['train/loss']))
    grads = {k: p.grad for k, p in model.named_parameters()}
    gref = {k: torch.from_numpy(g['grad/' + k]) for k in grads}
    err = {k: maxrel(grads[k], gref[k]) for k in grads if not noise_grad(k, gref)}
    smooth = {k: e for k, e in err.items() if k.startswith(smooth_pfx)}
    rest = {k: e for k, e in err.items() if k not in smooth}
    print('%s, FusedStep gradients vs the reference: decision-free tensors worst %.2e; decision-dependent ones worst %.2e'
          % (which, max(smooth.values()), max(rest.values())))
    assert max(smooth.values()) < 1e-4, {k: e for k, e in smooth.items() if e >= 1e-4}
    assert max(rest.values()) < loose, {k: e for k, e in rest.items() if e >= loose}
    sd = model.state_dict()
    for k in g.files:
        if k.startswith('state1/'):
            assert maxrel(sd[k[7:]].double(), torch.from_numpy(g[k]).double()) < 1e-5, k
    from superpoint_graph_amd import _lib
    assert _lib.lib().spg_ecc_persistent_errors() == 0


def test_fused_step_decision_conditioned_gradients_at_baseline_size(hip, setup, monkeypatch):
    """... and the sharp check on the same path: the decisions FusedStep's kernels took (read from ITS workspaces), the fp64 oracle
    backward with those decisions, every one of the 69 gradient tensors within 1e-4."""
    g, spec, batch, state0 = setup
    _decision_conditioned(spec, batch, state0, torch.from_numpy(g['class_weights']), monkeypatch, free_run=False, fused=True)


_LARGE = {
    # BASELINE.json configs[3]: 8 S3DIS-shaped scenes in one step (S3DIS.md:26)
    's3dis_8_scenes': dict(spec=dict(), seeds=list(range(8)), n_sp=1000, n_edges=5000, n_feat=14, n_classes=13, need_gb=40),
    # BASELINE.json configs[4] shape: Semantic3D scale, 11 point features, vector filters, 8 classes (Semantic3D.md:20-22)
    'semantic3d_scale': dict(spec=dict(model_config='gru_10,f_8', node_feats=11, ptn_nfeat_stn=11), seeds=[0], n_sp=10000, n_edges=50000,
                             n_feat=11, n_classes=8, need_gb=56),
}


@pytest.mark.parametrize('name', sorted(_LARGE))
def test_decision_conditioned_gradients_at_large_configs(hip, name, monkeypatch):
    """the SHARP gradient check (HIP decisions -> fp64 oracle backward, every one of the gradient tensors within
    1e-4) at the sizes of BASELINE.json configs[3] (8 scenes per step: 8000 superpoints, 1.0 M points) and configs[4] (Semantic3D
    scale: 10 000 superpoints, 50 000 superedges, vector filters).  The unconditioned comparison at these sizes
    (tests/test_gpu_model.py::test_large_configs_train_step_vs_oracle) can only bound the PointNet gradients loosely -- a few
    hundred of 1e8 ReLU / max-pool decisions sit on fp32 near-ties; a 1 % backward bug that shows only with multi-scene batches
    or vector filters at 10 k nodes passed it.  One fp64 oracle pass on the host (~30-60 s, tens of GB of autograd state)."""
    import psutil
    from superpoint_graph_amd import synth
    cfg = _LARGE[name]
    if psutil.virtual_memory().available < cfg['need_gb'] * 2 ** 30:
        pytest.skip(f'needs ~{cfg["need_gb"]} GB of host memory for the float64 oracle pass')
    spec = O.ModelSpec(**cfg['spec'])
    torch.manual_seed(1)
    ref = build_model(spec)
    with torch.no_grad():
        for m in ref.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.weight.normal_(1, 0.2); m.bias.normal_(0, 0.1)
        ref.ptn.stn.proj.weight.normal_(0, 0.02)
    state0 = {k: v.detach().clone() for k, v in ref.state_dict().items()}
    scenes = [synth.scene(s, n_sp=cfg['n_sp'], n_edges=cfg['n_edges'], n_feat=cfg['n_feat'], n_classes=cfg['n_classes']) for s in cfg['seeds']]
    col = synth.collate_numpy(scenes)
    idxn, degs, ef, _ = O.set_batch(col['edge_lists'], col['vcounts'], col['edge_feats'])
    batch = dict(clouds_flag=torch.from_numpy(col['clouds_flag']), clouds=torch.from_numpy(col['clouds']),
                 clouds_global=torch.from_numpy(col['clouds_global']), idxn=torch.from_numpy(idxn), degs=torch.from_numpy(degs),
                 edgefeats=torch.from_numpy(ef), label_mode=torch.from_numpy(col['targets'][:, 0].copy()))
    _decision_conditioned(spec, batch, state0, None, monkeypatch, free_run=False)
