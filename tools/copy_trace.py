#!/usr/bin/env python3
"""Which host-side ops of the train step launch memcpy / fill kernels (torch.profiler with python stacks)."""
import os
import sys
import types

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    from superpoint_graph_amd import dist as spd
    from superpoint_graph_amd.flat import FlatParameters
    from superpoint_graph_amd.learning import pointnet
    dev = torch.device('cuda')
    model = bench.build_model('gru_10_0,f_13', dev).train()
    targets, GIs, flag, clouds, diam, scenes = bench.make_batch([0], 1000, 5000)
    clouds_d, diam_d, label = clouds.to(dev), diam.to(dev), targets[:, 0].to(dev)
    model.ecc.set_info(GIs, 1)
    emb_er = pointnet.CloudEmbedder(types.SimpleNamespace(cuda=1, ptn_mem_monger=1))
    arena = FlatParameters(model)

    def step():
        if os.environ.get('TRACE_FWD'):
            model.eval()
            with torch.no_grad():
                model.ecc(emb_er.run(model, None, flag, clouds_d, diam_d))
            return
        arena.zero_grad()
        emb = emb_er.run(model, None, flag, clouds_d, diam_d)
        out = model.ecc(emb)
        F.cross_entropy(out, label).backward()
        emb_er.bw_hook()
        arena.adam_step(lr=1e-2, weight_decay=0.0, grad_clip=1.0)

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        for _ in range(3):
            step()
        torch.cuda.synchronize()
    rows = []
    for e in prof.events():
        n = e.name
        if ('Memcpy' in n or 'Memset' in n or 'copyBuffer' in n or 'fillBuffer' in n or n in ('aten::copy_', 'aten::fill_', 'aten::zero_')):
            st = [s for s in (e.stack or []) if 'superpoint_graph_amd' in s or 'copy_trace' in s or 'bench' in s][:2]
            rows.append((n[:50], str(e.device_type)[-4:], ' <- '.join(st)))
    from collections import Counter
    for (k, c) in Counter(rows).most_common(40):
        print(c, k)


if __name__ == '__main__':
    main()
