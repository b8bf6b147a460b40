"""Worker of tests/test_gpu_dist.py: one of WORLD_SIZE data-parallel ranks (all on cuda:0, gloo rendezvous).
Each rank takes one scene of the golden 2-scene batch, runs the train step with synchronised BatchNorm and the
weighted gradient all-reduce, and compares against the REFERENCE's single-process results for the whole batch
(tests/golden/s3dis_gru10_matrix.npz: logits, loss, every gradient, running statistics).
argv: sync (0 / 1), mode ('slots': the ranks all-reduce the exact fixed-point statistics slots, round 5; 'finalize': the fp64 sums of a
finalize launch per layer, rounds 1-4), fused (1: the step as ONE library call, spg_train_step -- slots mode only)."""
import os
import sys
import types

import torch
import torch.distributed as dist
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from conftest import build_model, load_golden, maxrel  # noqa: E402


def main():
    from superpoint_graph_amd import dist as spd
    from superpoint_graph_amd.flat import FlatParameters
    from superpoint_graph_amd.learning import ecc, pointnet
    sync = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    mode = sys.argv[2] if len(sys.argv) > 2 else 'slots'
    fused = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    native = os.environ.get('SPG_NATIVE_RCCL', '0') == '1'         # multi-GPU node: one GPU per rank, the library's own RCCL communicator
    if native:
        torch.cuda.set_device(rank)
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', rank))
        spd.init_native_rccl()
        assert spd.native_rccl_world_size() == world
    else:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    dev = torch.device('cuda', rank if native else 0)
    spec, batch, state0, g = load_golden('s3dis_gru10_matrix')
    # ---- this rank's scene out of the collated batch (nodes / edges of a scene are contiguous) ----
    n_nodes = [int(g[f'graph/{i}/n']) for i in range(2)]
    assert world == 2 and sum(n_nodes) == batch['degs'].numel()
    n0 = sum(n_nodes[:rank]); n1 = n0 + n_nodes[rank]
    e0 = int(batch['degs'][:n0].sum()); e1 = e0 + int(batch['degs'][n0:n1].sum())
    flag = batch['clouds_flag'][n0:n1]
    v0 = int((batch['clouds_flag'][:n0] == 0).sum()); v1 = v0 + int((flag == 0).sum())
    gi = ecc.GraphConvInfo.from_buffers(batch['idxn'][e0:e1] - n0, batch['degs'][n0:n1].clone(),
                                        batch['edgefeats'][e0:e1].clone(), None, None)
    labels = batch['label_mode'][n0:n1].to(dev)
    cw = torch.from_numpy(g['class_weights']).to(dev)

    model = build_model(spec, state0).to(dev).train()
    arena = FlatParameters(model, lazy_zero=bool(fused), host_counters=bool(fused))
    state = spd.enable_sync_bn(dev, mode=mode) if sync else None
    assert not sync or spd.sync_bn_mode() == mode
    model.ecc.set_info([gi], 1)
    arena.zero_grad()
    clouds_r, global_r = batch['clouds'][v0:v1].to(dev), batch['clouds_global'][v0:v1].to(dev)
    w = spd.loss_weight(labels, cw)
    if fused:
        from superpoint_graph_amd.fused import FusedStep
        step = FusedStep(model, arena, class_weights=cw, reduction='sum' if sync else 'mean')
        loss, logits = step(flag, clouds_r, global_r, gi, labels)
    else:
        embedder = pointnet.CloudEmbedder(types.SimpleNamespace(cuda=1, ptn_mem_monger=1))
        emb = embedder.run(model, None, flag, clouds_r, global_r)
        logits = model.ecc(emb)
        if sync:      # the backward couples the ranks: scale the loss first
            loss = F.cross_entropy(logits, labels, weight=cw, reduction='sum')
        else:
            loss = F.cross_entropy(logits, labels, weight=cw)
        loss.backward()
        embedder.bw_hook()
    arena.allreduce(w, prescaled=bool(sync))
    torch.cuda.synchronize()
    if state is not None:
        assert state['error'] is None, state['error']
        assert native or state['calls'] >= 26, state['calls']      # 13 BatchNorm layers, forward + backward

    ref_logits = torch.from_numpy(g['train/logits'])[n0:n1]
    err_logits = maxrel(logits, ref_logits)
    wsum = torch.tensor([float(loss) if sync else float(loss) * w, w], dtype=torch.float64, device=dev if native else 'cpu')
    dist.all_reduce(wsum)
    wsum = wsum.cpu()
    err_loss = abs(float(wsum[0] / wsum[1]) - float(g['train/loss'])) / abs(float(g['train/loss']))
    worst, worst_k = 0.0, ''
    for k, p in model.named_parameters():
        ref = torch.from_numpy(g['grad/' + k])
        if float(ref.abs().max()) < 1e-6:
            continue
        e = maxrel(p.grad, ref)
        if e > worst:
            worst, worst_k = e, k
    sd = model.state_dict()
    err_run = max(maxrel(sd[k[7:]].double(), torch.from_numpy(g[k]).double()) for k in g.files
                  if k.startswith('state1/') and 'running' in k)
    print(f'rank {rank} sync={sync} mode={mode} fused={fused}: logits {err_logits:.2e} loss {err_loss:.2e} worst grad {worst:.2e} ({worst_k}) '
          f'running stats {err_run:.2e}', flush=True)
    if sync:
        assert err_logits < 1e-4 and err_loss < 1e-4 and worst < 5e-4 and err_run < 1e-5
    else:
        # per-rank statistics are a different (documented) model: it must NOT reproduce the single-process batch
        assert err_logits > 1e-3
    spd.disable_sync_bn()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
