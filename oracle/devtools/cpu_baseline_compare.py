#!/usr/bin/env python3
"""Reference vs. port on ONE host: times the IMPORTED reference modules (oracle/ref_baseline.py, needs /root/reference)
and the oracle port (oracle/spg_oracle.py) on the bench scene, fwd+bwd, same threads.  bench.py's cpu_baseline on the GPU
box can only run the port (the reference checkout does not travel); this ratio, measured in the build container, is what
relates that number to the reference's own code.  CPU only (no GPU needed)."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from oracle import ref_baseline, spg_oracle as O  # noqa: E402
from superpoint_graph_amd import synth  # noqa: E402


def main():
    cfg, n_feat = 'gru_10_0,f_13', 14
    model = bench.build_model(cfg, torch.device('cpu'), n_feat)
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    scenes = [synth.scene(0, n_sp=1000, n_edges=5000, n_feat=n_feat, n_classes=13)]
    ref = bench.cpu_baseline(cfg, scenes, state, max_seconds=120.0, n_feat=n_feat)
    assert ref['kind'] == 'reference', 'no reference checkout on this machine'
    # the port, same scene, same thread count
    spec = O.ModelSpec(model_config=cfg, node_feats=n_feat, ptn_nfeat_stn=n_feat)
    col = synth.collate_numpy(scenes)
    idxn, degs, ef, _ = O.set_batch(col['edge_lists'], col['vcounts'], col['edge_feats'])
    batch = dict(clouds_flag=torch.from_numpy(col['clouds_flag']), clouds=torch.from_numpy(col['clouds']),
                 clouds_global=torch.from_numpy(col['clouds_global']), idxn=torch.from_numpy(idxn), degs=torch.from_numpy(degs),
                 edgefeats=torch.from_numpy(ef), label_mode=torch.from_numpy(col['targets'][:, 0].copy()))
    st = {k: v.clone() for k, v in state.items()}
    O.train_step(batch, spec, st, None)
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        O.train_step(batch, spec, st, None)
        times.append(time.perf_counter() - t0)
    port = 1000 / float(np.median(times))
    print(json.dumps({'host': bench.cpu_model(), 'threads': torch.get_num_threads(), 'scene': '1000 superpoints x 128 pts, 5000 superedges, gru_10_0,f_13, fwd+bwd',
                      'reference_superpoints_per_s': ref['value'], 'reference_sample': ref['sample'],
                      'port_superpoints_per_s': port, 'port_over_reference': port / ref['value']}, indent=1))


if __name__ == '__main__':
    main()
