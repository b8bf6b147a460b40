"""Where does the time of the persistent RNN-ECC forward go?  Times the module forward (eval mode, no_grad: filter network +
recurrence) for R = 1, 2, 5, 10 iterations on graphs that isolate the pieces:
  none  : no edges at all            -> GRU arithmetic per iteration, no exchange
  self  : one self-loop per node     -> + one granule round trip per iteration (own data, no waiting on others)
  ring  : i -> i+1                   -> + waiting on ONE other wave
  scene : the BASELINE graph (5000 edges, in-degree 1..15)
GPU only (devtool; uses nothing from oracle/)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from superpoint_graph_amd import synth
from superpoint_graph_amd.learning import ecc, graphnet

n = 1000
sc = synth.scene(0, n_sp=n, n_edges=5000)
e = np.asarray(sc['edges']).reshape(-1, 2)
order = np.argsort(e[:, 1], kind='stable')
graphs = {
    'none': (np.zeros(0, np.int64), np.zeros(n, np.int64)),
    'self': (np.arange(n), np.ones(n, np.int64)),
    'ring': ((np.arange(n) - 1) % n, np.ones(n, np.int64)),
    'scene': (e[order, 0], np.bincount(e[:, 1], minlength=n)),
}
# where does a hop go?  node i lives in workgroup i // 4, workgroup b on XCD b % 8
graphs['ring4'] = ((np.arange(n) - 4) % n, np.ones(n, np.int64))        # source in the NEXT workgroup: another XCD, always
graphs['ring32'] = ((np.arange(n) - 32) % n, np.ones(n, np.int64))      # source 8 workgroups away: the SAME XCD, another CU
rs = np.random.default_rng(0)
anyw = rs.integers(0, n, (n, 5))
xcd = (np.arange(n) // 4) % 8
same = np.stack([rs.choice(np.flatnonzero(xcd == xcd[i]), 5) for i in range(n)])
graphs['rand5'] = (anyw.reshape(-1), np.full(n, 5, np.int64))           # 5 in-neighbours anywhere
graphs['rand5x'] = (same.reshape(-1), np.full(n, 5, np.int64))          # 5 in-neighbours on the node's own XCD
x = torch.randn(n, 32).cuda()
for name, (idxn, degs) in graphs.items():
    E = len(idxn)
    ef = torch.randn(max(E, 0), 13)
    line = f'{name:6s} E={E:5d}: '
    for R in (1, 2, 5, 10):
        torch.manual_seed(1)
        net = graphnet.GraphNetwork(f'gru_{R}_0', 32, [13, 32, 128, 64], 1, 0, 2, 30000, use_pyg=0, cuda=1).cuda().eval()
        gi = ecc.GraphConvInfo.from_buffers(torch.from_numpy(idxn.astype(np.int64)), torch.from_numpy(degs.astype(np.int64)), ef.clone(), None, None)
        with torch.no_grad():
            net.set_info([gi], 1)
            for _ in range(5):
                net(x)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ts = []
            for _ in range(20):
                a.record(); net(x); b.record(); torch.cuda.synchronize()
                ts.append(a.elapsed_time(b) * 1e3)
        line += f' R={R}: {sorted(ts)[len(ts) // 2]:6.1f} us'
    print(line)
