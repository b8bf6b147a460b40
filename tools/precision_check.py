#!/usr/bin/env python3
"""Opt-in bf16 / split-bf16 MFMA modes of the wide row-GEMMs (spg_tune key 7) against the fp32-MFMA default on the bench
scene: embeddings, logits, loss and every gradient (max-norm relative difference), and the step time of each mode."""
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from superpoint_graph_amd import _lib, ops  # noqa: E402
from superpoint_graph_amd.learning import pointnet  # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    L = _lib.lib()
    model = bench.build_model('gru_10_0,f_13', dev, 14)
    model.train()
    state0 = {k: v.clone() for k, v in model.state_dict().items()}
    targets, GIs, flag, clouds, diam, scenes = bench.make_batch([0], 1000, 5000, 14, 13)
    clouds_d, diam_d, label = clouds.to(dev), diam.to(dev), targets[:, 0].to(dev)
    model.ecc.set_info(GIs, 1)
    embedder = pointnet.CloudEmbedder(types.SimpleNamespace(cuda=1, ptn_mem_monger=1))

    def step():
        model.zero_grad()
        emb = embedder.run(model, None, flag, clouds_d, diam_d)
        out = model.ecc(emb)
        loss = ops.cross_entropy(out, label)
        loss.backward()
        embedder.bw_hook()
        return emb.detach().clone(), out.detach().clone(), float(loss), {k: p.grad.detach().clone() for k, p in model.named_parameters()}

    res = {}
    for name, prec in (('fp32', 0), ('bf16x3', 3), ('bf16', 1)):
        L.spg_tune(7, prec)
        model.load_state_dict(state0)
        res[name] = step()
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            step()
        torch.cuda.synchronize()
        print(f'{name}: {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms/step (fwd+bwd, no optimizer)', flush=True)
    L.spg_tune(7, 0)
    # sensitivity of the fp32 step itself: the same scene with the point features perturbed by 1e-6 / 1e-5 relative noise
    # (what a different summation order or the split-bf16 products do to the first activations)
    base = clouds_d.clone()
    for eps in (1e-6, 1e-5):
        torch.manual_seed(5)
        clouds_d.copy_(base * (1 + eps * torch.randn_like(base)))
        model.load_state_dict(state0)
        res[f'fp32 + {eps:g} input noise'] = step()
    clouds_d.copy_(base)
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
    for name in [k for k in res if k != 'fp32']:
        e, o, l, g = res[name]
        e0, o0, l0, g0 = res['fp32']
        worst = max(((rel(g[k], g0[k]), k) for k in g0 if float(g0[k].abs().max()) > 1e-6))
        top = sorted(((rel(g[k], g0[k]), k) for k in g0 if float(g0[k].abs().max()) > 1e-6), reverse=True)[:6]
        print('   ', ', '.join(f'{k} {v:.1e}' for v, k in top))
        print(f'{name} vs fp32: emb {rel(e, e0):.2e} logits {rel(o, o0):.2e} loss {abs(l - l0) / abs(l0):.2e} worst grad {worst[0]:.2e} ({worst[1]})')


if __name__ == '__main__':
    main()
