"""LocalCloudEmbedder (reference learning/pointnet.py:182-218 as configured by
supervized_partition/supervized_partition.py:411-421): stand-alone STN + PointNet without inner STN + STN output as
global feature + L2 normalisation, forward and all gradients against a plain-torch restatement of run_batch built from
the same parameters."""
import types

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _torch_reference(model, clouds, clouds_global, nfeat_stn):
    """run_batch of the reference, op for op (learning/pointnet.py:188-205), on stock torch modules holding the SAME
    parameter tensors (so autograd accumulates into the same .grad fields)."""
    def seq(convs, x):
        for m in convs:
            if isinstance(m, nn.Conv1d):
                x = F.conv1d(x, m.weight, m.bias)
            elif isinstance(m, nn.Linear):
                x = F.linear(x, m.weight, m.bias)
            elif isinstance(m, nn.BatchNorm1d):
                x = F.batch_norm(x, None, None, m.weight, m.bias, True, 0.1, m.eps)
            elif isinstance(m, nn.ReLU):
                x = F.relu(x)
        return x
    stn = model.stn
    h = seq(stn.convs, clouds[:, :nfeat_stn, :])
    h = F.max_pool1d(h, h.size(2)).squeeze(2)
    h = seq(stn.fcs, h)
    T = F.linear(h, stn.proj.weight, stn.proj.bias).view(-1, 2, 2) + torch.eye(2, device=clouds.device).unsqueeze(0)
    xy = torch.bmm(clouds[:, :2, :].transpose(1, 2), T).transpose(1, 2)
    x = torch.cat([xy, clouds[:, 2:, :]], 1)
    g = torch.cat([clouds_global, T.reshape(-1, 4)], 1)
    x = seq(model.ptn.convs, x)
    x = F.max_pool1d(x, x.size(2)).squeeze(2)
    x = torch.cat([x, g], 1)
    return F.normalize(seq(model.ptn.fcs, x))


@pytest.mark.parametrize('n,k', [(700, 20), (129, 20), (300, 32)])
def test_local_cloud_embedder_forward_backward(n, k):
    from superpoint_graph_amd.learning import pointnet
    torch.manual_seed(3)
    model = nn.Module()
    model.stn = pointnet.STNkD(2, [16, 64], [32, 16])
    model.ptn = pointnet.PointNet([32, 128], [34, 32, 32, 4], [], [], 6, 0, prelast_do=0, nfeat_global=11, is_res=False, last_bn=True)
    nn.init.normal_(model.stn.proj.weight, std=0.05)          # the zero-initialised projection would hide the transform path
    nn.init.normal_(model.stn.proj.bias, std=0.05)
    model.cuda().train()
    g = torch.Generator().manual_seed(1)
    clouds = torch.randn(n, 6, k, generator=g).cuda()
    clouds_global = torch.randn(n, 7, generator=g).cuda()
    w = torch.randn(n, 4, generator=g).cuda()
    emb = pointnet.LocalCloudEmbedder(types.SimpleNamespace(ptn_nfeat_stn=2, stn_as_global=1)).run_batch(model, clouds, clouds_global)
    (emb * w).sum().backward()
    ours = {kk: p.grad.clone() for kk, p in model.named_parameters()}
    for p in model.parameters():
        p.grad = None
    ref = _torch_reference(model, clouds, clouds_global, 2)
    (ref * w).sum().backward()
    assert emb.shape == (n, 4)
    assert float((emb - ref).abs().max()) <= 2e-5
    worst = 0.0
    gmax = max(float(p.grad.abs().max()) for p in model.parameters())
    for kk, p in model.named_parameters():
        den = float(p.grad.abs().max())
        if den < 1e-5 * gmax:   # biases in front of a train-mode BatchNorm: exactly zero here, round-off in torch
            assert float(ours[kk].abs().max()) <= 1e-5 * gmax
            continue
        err = float((ours[kk] - p.grad).abs().max()) / den
        print(f'  {kk}: {err:.3e} (max|ref| {den:.3e})')
        worst = max(worst, err)
    print('worst gradient error (max|d| / max|ref|):', worst)
    assert worst < 5e-4
