"""Diagnostic: BatchNorm fold (fixed-point slots) vs the finalize-kernel path on the LocalCloudEmbedder chunk case and on the
BASELINE scene: embeddings, running statistics and gradients of the two paths side by side (GPU only)."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import numpy as np, torch
from oracle import validate_against_reference as V
from superpoint_graph_amd import _lib
from superpoint_graph_amd.learning import pointnet
L = _lib.lib()
g = np.load(os.path.join(os.path.dirname(__file__), '..', '..', 'tests', 'golden', 'local_embedder.npz'))
ARGS = types.SimpleNamespace(ptn_nfeat_stn=2, stn_as_global=1)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65535
res = {}
for mode in (1, 0):
    L.spg_tune(10, mode)
    model = V.make_local_model(pointnet)
    model.load_state_dict({k[7:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('state0/')})
    clouds, cg, w = V.local_inputs(n)
    model.cuda().train()
    emb = pointnet.LocalCloudEmbedder(ARGS).run_batch(model, clouds.cuda(), cg.cuda())
    (emb * w.cuda()).sum().backward()
    torch.cuda.synchronize()
    res[mode] = (emb.detach().cpu(), {k: v.detach().cpu().clone() for k, v in model.state_dict().items() if 'running' in k},
                 {k: p.grad.detach().cpu() for k, p in model.named_parameters()})
L.spg_tune(10, 0)
a, b = res[1], res[0]
print('emb max diff', float((a[0] - b[0]).abs().max()))
if n == int(g['chunk/n']):
    for k in a[1]:
        ref = torch.from_numpy(g['chunk/state1/' + k])
        print(f'  vs reference {k}: finalize path {float((a[1][k] - ref).abs().max() / ref.abs().max()):.2e}   fold path {float((b[1][k] - ref).abs().max() / ref.abs().max()):.2e}')
    for k in [kk[13:] for kk in g.files if kk.startswith('chunk/grad64/')]:
        r64 = torch.from_numpy(g['chunk/grad64/' + k])
        print(f'  grad {k} vs float64: finalize path {float((a[2][k].double() - r64).abs().max() / r64.abs().max()):.2e}   fold path {float((b[2][k].double() - r64).abs().max() / r64.abs().max()):.2e}')
for k in a[1]:
    print(f'  {k}: rel diff {float((a[1][k] - b[1][k]).abs().max() / a[1][k].abs().max()):.2e}')
for k in a[2]:
    d = float(a[2][k].abs().max())
    if d > 1e-3:
        print(f'  grad {k}: rel diff {float((a[2][k] - b[2][k]).abs().max() / d):.2e}')
