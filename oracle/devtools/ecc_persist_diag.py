"""Diagnostic: persistent vs per-iteration RNN-ECC on one graph; prints where the two differ (GPU only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import numpy as np, torch
from test_gpu_ecc_persistent import _graph, _run
from superpoint_graph_amd import _lib
from superpoint_graph_amd.learning import ecc, graphnet
L = _lib.lib()
cfg, n, e = sys.argv[1] if len(sys.argv) > 1 else 'gru_10_0,f_13', int(sys.argv[2]) if len(sys.argv) > 2 else 1000, int(sys.argv[3]) if len(sys.argv) > 3 else 5000
hubs = (sys.argv[4] != '0') if len(sys.argv) > 4 else True
idxn, degs = _graph(n, e, seed=n + e, hubs=hubs)
edgefeats = torch.randn(e, 13, generator=torch.Generator().manual_seed(1))
x = torch.randn(n, 32, generator=torch.Generator().manual_seed(2)).cuda()
torch.manual_seed(7)
net = graphnet.GraphNetwork(cfg, 32, [13, 32, 128, 64], 1, 0, 2, 30000, use_pyg=0, cuda=1).cuda().train()
gi = ecc.GraphConvInfo.from_buffers(idxn.clone(), degs.clone(), edgefeats.clone(), None, None)
with torch.no_grad():
    net.set_info([gi], 1); width = net(x).shape[1]
go = torch.randn(n, width, generator=torch.Generator().manual_seed(3)).cuda()
st0 = {k: v.clone() for k, v in net.state_dict().items()}
ref = _run(net, gi, x, go, True)
net.load_state_dict(st0)
got = _run(net, gi, x, go, False)
print('errors', L.spg_ecc_persistent_errors())
print('out equal', torch.equal(ref[0], got[0]))
d = (ref[1] - got[1]).abs()
rows = (d.max(1)[0] > 0).nonzero().reshape(-1)
print('x.grad: rows differing', rows.numel(), 'of', n, 'max abs', float(d.max()), 'rel', float(d.max() / ref[1].abs().max()))
print('first rows', rows[:20].tolist())
outdeg = np.bincount(idxn.numpy(), minlength=n)
print('out-degrees of differing rows', outdeg[rows[:20].cpu().numpy()].tolist(), 'in-degrees', degs[rows[:20].cpu()].tolist())
for k in ref[2]:
    dd = (ref[2][k] - got[2][k]).abs().max()
    print(f'  {k}: max abs diff {float(dd):.3e} (max {float(ref[2][k].abs().max()):.3e})')
