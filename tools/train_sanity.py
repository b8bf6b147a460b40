#!/usr/bin/env python3
"""Sanity: a few hundred optimiser steps on one synthetic scene (random labels) must drive the loss down -- the whole loop
(HIP forward / backward, flat arena, fused clamp + Adam) learns."""
import os
import sys
import types

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    from superpoint_graph_amd import _lib
    from superpoint_graph_amd.flat import FlatParameters
    prec = sys.argv[1] if len(sys.argv) > 1 else 'f32'          # f32 | bf16x3 | bf16 (opt-in arithmetic of the wide GEMMs)
    assert _lib.lib().spg_tune(7, {'f32': 0, 'bf16': 1, 'bf16x3': 3}[prec]) >= 0
    print(f'precision mode: {prec}')
    from superpoint_graph_amd.learning import pointnet
    dev = torch.device('cuda')
    model = bench.build_model('gru_10_0,f_13', dev).train()
    targets, GIs, flag, clouds, diam, _ = bench.make_batch([0], 1000, 5000)
    clouds_d, diam_d, label = clouds.to(dev), diam.to(dev), targets[:, 0].to(dev)
    model.ecc.set_info(GIs, 1)
    emb_er = pointnet.CloudEmbedder(types.SimpleNamespace(cuda=1, ptn_mem_monger=1))
    fused = len(sys.argv) > 2 and sys.argv[2] == 'fused'       # the step as ONE library call (spg_train_step), as bench.py / the CLI run it
    arena = FlatParameters(model, lazy_zero=fused, host_counters=fused)
    fstep = None
    if fused:
        from superpoint_graph_amd.fused import FusedStep
        fstep = FusedStep(model, arena)
        print('step: spg_train_step')
    losses = []
    for it in range(301):
        arena.zero_grad()
        if fused:
            loss, out = fstep(flag, clouds_d, diam_d, GIs[0], label)
        else:
            out = model.ecc(emb_er.run(model, None, flag, clouds_d, diam_d))
            loss = F.cross_entropy(out, label)
            loss.backward()
            emb_er.bw_hook()
        arena.adam_step(lr=1e-3, weight_decay=0.0, grad_clip=1.0)
        if it % 50 == 0:
            acc = float((out.argmax(1) == label)[label >= 0].float().mean())
            losses.append(float(loss))
            print(f'step {it:4d}  loss {float(loss):.4f}  train accuracy {acc:.3f}', flush=True)
    assert losses[-1] < 0.7 * losses[0], losses
    assert _lib.lib().spg_ecc_persistent_errors() == 0
    print('ok: loss decreased')


if __name__ == '__main__':
    main()
