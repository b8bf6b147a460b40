"""Worker of tests/test_gpu_dist.py::test_native_rccl_single_rank (own process: the communicator is process-global)."""
import os
import sys
import types

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    from superpoint_graph_amd import _lib, dist as spd
    from superpoint_graph_amd.flat import FlatParameters
    from superpoint_graph_amd.learning import pointnet
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    assert spd.native_rccl_world_size() == 0
    assert spd.init_native_rccl() == 1 and spd.native_rccl_world_size() == 1
    L = _lib.lib()
    x = torch.randn(279410, device=dev)
    y = x.clone()
    _lib.check(L.spg_rccl_allreduce_sum_f32(y.data_ptr(), y.numel(), torch.cuda.current_stream().cuda_stream), 'allreduce')
    torch.cuda.synchronize()
    assert torch.equal(x, y)                                  # sum over one rank
    model = bench.build_model('gru_10_0,f_13', dev).train()
    state0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    targets, GIs, flag, clouds, diam, _ = bench.make_batch([0], 300, 1400)
    clouds_d, diam_d, label = clouds.to(dev), diam.to(dev), targets[:, 0].to(dev)
    model.ecc.set_info(GIs, 1)

    def run():
        model.load_state_dict(state0)
        model.zero_grad()
        emb_er = pointnet.CloudEmbedder(types.SimpleNamespace(cuda=1, ptn_mem_monger=1))
        out = model.ecc(emb_er.run(model, None, flag, clouds_d, diam_d))
        F.cross_entropy(out, label).backward()
        emb_er.bw_hook()
        return out.detach().clone(), {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    # (local BatchNorm with the separate launches the synchronised mode uses: spg_tune key 14, tests/test_gpu_bwdpair.py; keys 17 / 18,
    #  tests/test_gpu_narrow.py)
    old = [L.spg_tune(k, 1) for k in (14, 17, 18)]
    out0, g0 = run()
    for k, v in zip((14, 17, 18), old):
        L.spg_tune(k, v)
    st = spd.enable_sync_bn(dev, mode='finalize')
    assert st.get('native') is True
    try:
        out1, g1 = run()
    finally:
        spd.disable_sync_bn()
    assert torch.equal(out0, out1)
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k
    # slot-synchronised BatchNorm (round 5): the library all-reduces the fixed-point statistics slots in place (ncclInt64) and
    # the row counts come from the device buffer -- at one rank bit-identical to the plain per-rank step, every fast path kept
    out2, g2 = run()
    st = spd.enable_sync_bn(dev, mode='slots')
    assert st.get('native') is True and spd.sync_bn_mode() == 'slots'
    try:
        out3, g3 = run()
    finally:
        spd.disable_sync_bn()
    assert torch.equal(out2, out3)
    for k in g2:
        assert torch.equal(g2[k], g3[k]), k
    _lib.check(L.spg_rccl_destroy(), 'destroy')
    assert spd.native_rccl_world_size() == 0
    print('native rccl ok')


if __name__ == '__main__':
    main()
