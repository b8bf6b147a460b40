"""Runs only the fresh-batch loop (bench.trainer_window) -- for a rocprofv3 kernel trace of that window:
    rocprofv3 --kernel-trace --output-format rocpd -d /tmp/p_tw -- python tools/tw_trace.py
    python tools/prof_summary.py $(find /tmp/p_tw -name '*.db') 90"""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from superpoint_graph_amd.flat import FlatParameters
from superpoint_graph_amd.learning import pointnet

dev = torch.device('cuda', 0)
args = types.SimpleNamespace(scenes=1, n_sp=1000, n_edges=5000, n_feat=14)
model = bench.build_model('gru_10_0,f_13', dev).train()
embedder = pointnet.CloudEmbedder(types.SimpleNamespace(cuda=1, ptn_mem_monger=1))
arena = FlatParameters(model, lazy_zero=True, host_counters=True)
out = bench.trainer_window(args, dev, model, embedder, arena, [0], 13, lambda s: print(s, flush=True), iters=30)
print(out['ms_per_step'])
