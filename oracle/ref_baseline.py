"""ORACLE / test infrastructure: the CPU baseline of bench.py timed over the IMPORTED reference modules (BASELINE.md
section 3) -- only where /root/reference exists (the build container); on the GPU box bench.py falls back to the oracle
port (oracle/spg_oracle.py).  The reference's matrix-filter ECC backward raises on torch >= 1.5
(GraphConvModule.py:146), so `GraphConvFunction` is the restated oracle.EccFunction (pinned in
oracle/validate_against_reference.py); everything else -- PointNet, STN, CloudEmbedder with memory mongering, filter
network, GRUCellEx, the per-node aggregation loop of GraphConvModule.py:82-88 in the forward -- is the reference's code."""
import os
import sys
import time
import types

import numpy as np
import torch

REF = os.environ.get('SPG_REFERENCE', '/root/reference')


def available():
    return os.path.isdir(os.path.join(REF, 'learning'))


def time_reference_step(model_config, batch, n_feat, state, max_seconds=25.0):
    """-> (median seconds per fwd+bwd step, number of timed steps) with the reference's own modules on this host."""
    from oracle import spg_oracle as O
    if 'igraph' not in sys.modules:
        sys.modules['igraph'] = types.ModuleType('igraph')
    sys.path.insert(0, REF)
    from learning import ecc, graphnet, pointnet
    import learning.modules as refmodules
    ecc.GraphConvFunction = O.EccFunction
    refmodules.ecc.GraphConvFunction = O.EccFunction
    torch.manual_seed(1)
    model = torch.nn.Module()
    model.ecc = graphnet.GraphNetwork(model_config, 32, [13, 32, 128, 64], 1, 0, 2, 30000, use_pyg=0, cuda=0)
    model.ptn = pointnet.PointNet([64, 64, 128, 128, 256], [256, 64, 32], [64, 64, 128], [128, 64], n_feat, n_feat, prelast_do=0)
    model.load_state_dict(state)
    model.train()
    gi = ecc.GraphConvInfo()
    gi._idxn, gi._idxe, gi._degrees, gi._degrees_gpu, gi._edgefeats = batch['idxn'], None, batch['degs'], None, batch['edgefeats']
    gi._edge_indexes = None                         # only read by the --use_pyg 1 path (modules.py:156)
    model.ecc.set_info([gi], 0)
    embedder = pointnet.CloudEmbedder(types.SimpleNamespace(cuda=0, ptn_mem_monger=1))

    def step():
        for p in model.parameters():
            p.grad = None
        emb = embedder.run(model, None, batch['clouds_flag'], batch['clouds'], batch['clouds_global'])
        loss = torch.nn.functional.cross_entropy(model.ecc(emb), batch['label_mode'])
        loss.backward()
        embedder.bw_hook()
    step()
    times, t_begin = [], time.perf_counter()
    while len(times) < 5 and (time.perf_counter() - t_begin) < max_seconds:
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
    return float(np.median(times)), len(times)
