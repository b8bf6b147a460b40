"""Bit-for-bit fingerprint of the training step (tools only): python tools/bitcheck.py [--scenes S] [--n-sp N] [--n-feat F] [--model-config C]
runs three optimiser steps of the bench workload from fixed seeds and prints SHA-256 digests of the loss, the gradients of the third
step, the parameters, Adam's moments and the BatchNorm running statistics.  Run it under two library builds (SPG_HIP_LIB=...): equal
digests = the change between them is bit-identical on this workload (the gate of round 6's packed-VALU rewrite)."""
import argparse
import hashlib
import os
import sys
import types

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench  # noqa: E402


def digest(t):
    return hashlib.sha256(t.detach().contiguous().cpu().numpy().tobytes()).hexdigest()[:16]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--scenes', type=int, default=1)
    ap.add_argument('--n-sp', type=int, default=1000)
    ap.add_argument('--n-edges', type=int, default=5000)
    ap.add_argument('--n-feat', type=int, default=14)
    ap.add_argument('--model-config', default='gru_10_0,f_13')
    ap.add_argument('--tune', default='')
    a = ap.parse_args()
    from superpoint_graph_amd import _lib, fused
    from superpoint_graph_amd.flat import FlatParameters
    dev = torch.device('cuda:0')
    for kv in [t for t in a.tune.split(',') if t]:
        k, v = kv.split(':')
        _lib.lib().spg_tune(int(k), int(v))
    model = bench.build_model(a.model_config, dev, a.n_feat).train()
    n_classes = int(a.model_config.split('f_')[-1].split(',')[0])
    targets, GIs, flag, clouds, diam, _ = bench.make_batch(list(range(a.scenes)), a.n_sp, a.n_edges, a.n_feat, n_classes)
    clouds_d, diam_d, label = clouds.to(dev), diam.to(dev), targets[:, 0].to(dev)
    model.ecc.set_info(GIs, 1)
    arena = FlatParameters(model, lazy_zero=True, host_counters=True)
    step = fused.FusedStep(model, arena, reduction='mean', ptn_mem_monger=True)
    out = {}
    for it in range(3):
        arena.zero_grad()
        loss, _ = step(flag, clouds_d, diam_d, GIs[0], label)
        if it == 2:
            torch.cuda.synchronize()
            out['loss'] = digest(loss)
            out['grads'] = digest(torch.cat([p.grad.reshape(-1) for p in model.parameters()]))
        arena.adam_step(lr=1e-2, weight_decay=0.0, grad_clip=1.0)
    torch.cuda.synchronize()
    out['params'] = digest(torch.cat([p.reshape(-1) for p in model.parameters()]))
    out['adam_m'], out['adam_v'] = digest(arena._m), digest(arena._v)
    out['running'] = digest(torch.cat([b.reshape(-1).float() for n, b in model.named_buffers() if 'running' in n]))
    model.eval()
    with torch.no_grad():
        from superpoint_graph_amd.learning import pointnet
        emb = pointnet.CloudEmbedder(types.SimpleNamespace(cuda=1, ptn_mem_monger=1)).run(model, None, flag, clouds_d, diam_d)
        out['eval_logits'] = digest(model.ecc(emb))
    print(os.environ.get('SPG_HIP_LIB', 'in-tree'), vars(a), ' '.join(f'{k}={v}' for k, v in out.items()))


if __name__ == '__main__':
    main()
