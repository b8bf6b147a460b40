#!/usr/bin/env python3
"""Forward-only timing of the hot path on the BASELINE unit scene (config 2 of BASELINE.json: PointNet + RNN-ECC
forward): eval mode (BatchNorm folded into the consumers, no activations stored for a backward) and train mode."""
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from superpoint_graph_amd.learning import pointnet  # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    only = os.environ.get('FWD_ONLY', '')            # e.g. FWD_ONLY='gru_1_0,f_13:eval' (profiling)
    for cfgname in ('gru_10_0,f_13', 'gru_1_0,f_13', 'gru_10,f_13'):
        if only and only.split(':')[0] != cfgname:
            continue
        model = bench.build_model(cfgname, dev)
        targets, GIs, flag, clouds, diam, scenes = bench.make_batch([0], 1000, 5000)
        clouds_d, diam_d = clouds.to(dev), diam.to(dev)
        model.ecc.set_info(GIs, 1)
        emb = pointnet.CloudEmbedder(types.SimpleNamespace(cuda=1, ptn_mem_monger=1))
        for mode in ('eval', 'train'):
            if only and only.split(':')[1] != mode:
                continue
            model.train(mode == 'train')
            def fwd():
                with torch.no_grad():
                    e = emb.run(model, None, flag, clouds_d, diam_d)
                    return model.ecc(e)
            for _ in range(5):
                fwd()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 30
            for _ in range(n):
                fwd()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
            print(f'{cfgname:16s} {mode:5s} forward: {dt * 1e3:7.3f} ms  = {flag.numel() / dt:10.0f} superpoints/s', flush=True)


if __name__ == '__main__':
    main()
