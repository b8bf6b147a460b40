#!/usr/bin/env python3
"""Which host-side ops of the train step issue device copies (hipMemcpyWithStream / hipMemcpyAsync -> __amd_rocclr_copyBuffer
blit kernels): torch.profiler event tree, every copy API call printed with its chain of enclosing aten / autograd ops."""
import os
import sys
import types
from collections import Counter

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    from superpoint_graph_amd import ops
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
        arena.zero_grad()
        emb = emb_er.run(model, None, flag, clouds_d, diam_d)
        out = model.ecc(emb)
        ops.cross_entropy(out, label).backward()
        emb_er.bw_hook()
        arena.adam_step(lr=1e-2, weight_decay=0.0, grad_clip=1.0)

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    nsteps = 3
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for _ in range(nsteps):
            step()
        torch.cuda.synchronize()
    rows = Counter()
    for e in prof.events():
        if 'Memcpy' in e.name or 'Memset' in e.name:
            chain, p = [], e.cpu_parent
            while p is not None and len(chain) < 6:
                chain.append(p.name)
                p = p.cpu_parent
            rows[(e.name[:40], ' <- '.join(chain))] += 1
    for (k, c) in rows.most_common(60):
        print(f'{c / nsteps:6.1f}/step  {k[0]:<40s} {k[1]}')


if __name__ == '__main__':
    main()
