import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from superpoint_graph_amd import ops
rng = np.random.default_rng(0)
for n, e in ((1000, 5000), (2000, 10000), (4000, 20000)):
    edges = torch.from_numpy(rng.integers(0, n, size=(e, 2)).astype(np.int64))
    feats = torch.from_numpy(rng.standard_normal((e, 13)).astype(np.float32))
    for _ in range(5): ops.batch_graph_build(edges, feats, n)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(20):
        a.record(); ops.batch_graph_build(edges, feats, n); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
    t0 = time.perf_counter()
    for _ in range(50): ops.batch_graph_build(edges, feats, n)
    th = (time.perf_counter() - t0) / 50
    torch.cuda.synchronize()
    ed = edges.cuda(); fd = feats.cuda()
    ts2 = []
    for _ in range(20):
        a.record(); i2, d2, p2, _e = ops.set_batch(ed, n); f2 = ops.gather_rows(fd, p2); g2 = ops.DeviceGraph(i2, d2); b.record(); torch.cuda.synchronize(); ts2.append(a.elapsed_time(b) * 1e3)
    print(f'N={n} E={e}: single launch (incl. 2 uploads) {sorted(ts)[10]:.1f} us GPU, host {th*1e6:.0f} us per call; multi-launch (resident inputs) {sorted(ts2)[10]:.1f} us')
