"""Host time of the fresh-batch loop (bench.trainer_window with the one-call step), by function: cProfile of 60 steps after warm-up.
GPU only.  python tools/tw_host_profile.py [top]"""
import cProfile, os, pstats, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from superpoint_graph_amd import ops, synth, fused as spg_fused
from superpoint_graph_amd.flat import FlatParameters
from superpoint_graph_amd.learning import ecc, pointnet, spg
from superpoint_graph_amd.learning.prefetch import SideStreamBatches

dev = torch.device('cuda', 0)
model = bench.build_model('gru_10_0,f_13', dev).train()
arena = FlatParameters(model, lazy_zero=True, host_counters=True)
fstep = spg_fused.FusedStep(model, arena, reduction='mean', ptn_mem_monger=True)
batches = []
for b in range(4):
    sc = synth.scene(3000 + b, n_sp=1000, n_edges=5000, n_feat=14, n_classes=13)
    smp = spg.sample_from_scene(sc, f'a{b}')
    targets, _, (meta, flag, clouds, diam) = spg.eccpc_collate([smp])
    batches.append((targets, [smp[1]], flag, clouds.pin_memory(), diam))


def fresh(n):
    for it in range(n):
        targets, graphs, flag, clouds, diam = batches[it % 4]
        flag = flag.clone()
        iv, slot = pointnet.flag_index_vectors(flag)
        gi = ecc.GraphConvInfo()
        gi.set_batch_device(graphs, spg.cloud_edge_feats, extras=[iv, slot, targets[:, 0].contiguous(), diam])
        iv_d, slot_d, lab_d, diam_d = gi.extras_dev
        pointnet.attach_staged_flags(flag, iv_d, slot_d)
        yield gi, flag, ops.upload(clouds, dev), diam_d, lab_d


def run(n):
    for gi, flag, c, d, lab in SideStreamBatches(fresh(n)):
        model.ecc.set_info([gi], 1)
        arena.zero_grad()
        fstep(flag, c, d, gi, lab)
        arena.adam_step(lr=1e-2, weight_decay=0.0, grad_clip=1.0)


run(12)
torch.cuda.synchronize()
t0 = time.perf_counter(); run(60); th = time.perf_counter() - t0; torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f'fresh-batch loop: host {th / 60 * 1e3:.3f} ms/step, GPU done {dt / 60 * 1e3:.3f} ms/step')
pr = cProfile.Profile()
pr.enable(); run(60); pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats('cumulative')
st.print_stats(int(sys.argv[1]) if len(sys.argv) > 1 else 45)
