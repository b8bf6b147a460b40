import os, sys, types
from collections import Counter
import torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import bench
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
for _ in range(5): step()
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3): step()
    torch.cuda.synchronize()
rows = Counter()
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA or 'spg_' in e.name: continue
    # CPU-side op that launched a kernel directly
    if e.name.startswith('hipLaunchKernel') or e.name.startswith('hipMemset') or e.name.startswith('hipMemcpy') or e.name.startswith('hipExtModuleLaunch'):
        chain, p = [], e.cpu_parent
        while p is not None and len(chain) < 5:
            chain.append(p.name); p = p.cpu_parent
        rows[(e.name[:28], ' <- '.join(chain))] += 1
for (k, c) in rows.most_common(40):
    print(f'{c/3:5.1f}/step {k[0]:<28s} {k[1][:150]}')
