"""Experiment: can the latency-bound small-grid chains of the step hide under the wide PointNet kernels
when they run on a second stream?  Measures, on the unit scene,
   forward : PointNet forward (wide persistent GEMMs)  vs  the RNN-ECC module forward (filter net: 5 few-row GEMMs + finalize,
             then the recurrence)                      -- alone, back to back on one stream, and concurrently on two streams
   backward: PointNet backward                         vs  the RNN-ECC backward (recurrence, edge gradient, filter-net chain,
             GRU weight gradients, reduction)          -- the same three ways.
The concurrent numbers are an UPPER bound of what a real split could gain (in the real step the recurrence itself depends on
PointNet's output; only the filter network and the parameter-gradient tail are independent).  GPU only; prints a small table."""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from superpoint_graph_amd import ops
from superpoint_graph_amd.learning import pointnet

dev = torch.device('cuda', 0)
model = bench.build_model('gru_10_0,f_13', dev).train()
targets, GIs, flag, clouds, diam, scenes = bench.make_batch([0], 1000, 5000)
clouds_d, diam_d, lab = clouds.to(dev), diam.to(dev), targets[:, 0].to(dev)
model.ecc.set_info(GIs, 1)
emb_er = pointnet.CloudEmbedder(types.SimpleNamespace(cuda=1, ptn_mem_monger=1))
side = torch.cuda.Stream()
main = torch.cuda.current_stream()
emb_fixed = emb_er.run(model, None, flag, clouds_d, diam_d).detach()
g_emb = torch.randn(int((flag == 0).sum()), 32, device=dev)


def ptn_fwd():
    return model.ptn(clouds_d[:], diam_d)


def ecc_fwd():
    return model.ecc(emb_fixed.clone().requires_grad_(True))


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def both_serial_fwd():
    a = ptn_fwd(); b = ecc_fwd(); return a, b


def both_parallel_fwd():
    side.wait_stream(main)
    with torch.cuda.stream(side):
        b = ecc_fwd()
    a = ptn_fwd()
    main.wait_stream(side)
    return a, b


rows = []
rows.append(('forward: PointNet alone', timeit(ptn_fwd)))
rows.append(('forward: RNN-ECC module alone', timeit(ecc_fwd)))
rows.append(('forward: both, one stream', timeit(both_serial_fwd)))
rows.append(('forward: both, two streams', timeit(both_parallel_fwd)))


def ptn_fwd_bwd():
    out = ptn_fwd(); out.backward(g_emb)


def ecc_fwd_bwd():
    out = ecc_fwd(); ops.cross_entropy(out, lab).backward()


def serial_fb():
    ptn_fwd_bwd(); ecc_fwd_bwd()


def parallel_fb():
    side.wait_stream(main)
    with torch.cuda.stream(side):
        ecc_fwd_bwd()
    ptn_fwd_bwd()
    main.wait_stream(side)


rows.append(('fwd+bwd: PointNet alone', timeit(ptn_fwd_bwd)))
rows.append(('fwd+bwd: RNN-ECC module alone', timeit(ecc_fwd_bwd)))
rows.append(('fwd+bwd: both, one stream', timeit(serial_fb)))
rows.append(('fwd+bwd: both, two streams', timeit(parallel_fb)))
for k, v in rows:
    print(f'{k:36s} {v:8.3f} ms')
print('(persistent RNN-ECC launches on the side stream fall back to the per-iteration kernels when the main stream owns the exchange buffer)')
