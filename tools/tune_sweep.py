#!/usr/bin/env python3
"""Sweep of the row-GEMM tuning knobs (spg_tune) on the real training step of the unit scene: for every setting the
wall time per step (un-instrumented) and the hipEvent time of every (kernel instantiation, layer shape) of the
instrumented pass.  Attribution switches (key 3) produce WRONG results by design -- timing only.
Keys 3 and 5 (timing attribution) exist only in a library built with `make -C superpoint_graph_amd/csrc clean all ATTRIBUTION=1`;
the production build rejects them.
Round-2 findings (gpurun_out/r2_tune*.txt, profiles/r02_gemm_attribution.txt): see DESIGN.md section 4.1.

    python tools/tune_sweep.py [--configs name=k:v,k:v ...] > gpurun_out/tune.txt
"""
import argparse
import ctypes
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import bench  # noqa: E402
from superpoint_graph_amd import _lib, ops  # noqa: E402
from superpoint_graph_amd.flat import FlatParameters  # noqa: E402
from superpoint_graph_amd.learning import pointnet  # noqa: E402

KNOBS = 8


def decode_tag(tag):
    kind = tag // 1000000
    it = (tag // 100000) % 10 * 32
    jt = (tag // 10000) % 10 * 32
    x = (tag // 100) % 100 - 1
    y = (tag // 10) % 10 - 1
    full = tag % 10
    return f'{"gemm" if kind == 1 else "wgrad"}<{it},{jt},{x},{y},{full}>'


def read_shapes(L, nprof):
    keys = (ctypes.c_int * (4 * 256))()
    vals = (ctypes.c_double * (2 * 256))()
    n = L.spg_prof_read_shapes(keys, vals, 256)
    rows = []
    for j in range(n):
        tag, N, K, cnt = keys[4 * j], keys[4 * j + 1], keys[4 * j + 2], keys[4 * j + 3]
        ms, fl = vals[2 * j], vals[2 * j + 1]
        rows.append((ms / nprof * 1e3, decode_tag(tag), N, K, cnt / nprof, ms / cnt * 1e3, fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0))
    rows.sort(reverse=True)
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--configs', nargs='*', default=None)
    ap.add_argument('--steps', type=int, default=60)
    ap.add_argument('--top', type=int, default=14)
    ap.add_argument('--scenes', type=int, default=1)
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    L = _lib.lib()
    model = bench.build_model('gru_10_0,f_13', dev, 14)
    model.train()
    targets, GIs, flag, clouds, diam, scenes = bench.make_batch(list(range(args.scenes)), 1000, 5000, 14, 13)
    clouds_d, diam_d = clouds.to(dev), diam.to(dev)
    label_mode = targets[:, 0].to(dev)
    model.ecc.set_info(GIs, 1)
    embedder = pointnet.CloudEmbedder(types.SimpleNamespace(cuda=1, ptn_mem_monger=1))
    arena = FlatParameters(model)

    def step():
        arena.zero_grad()
        emb = embedder.run(model, None, flag, clouds_d, diam_d)
        out = model.ecc(emb)
        loss = ops.cross_entropy(out, label_mode)
        loss.backward()
        embedder.bw_hook()
        arena.adam_step(lr=1e-3, weight_decay=0.0, grad_clip=1.0)

    configs = args.configs or ['default=', 'one_wg_per_tile=0:1', 'slice_above_256=4:256', 'no_stat_accum=5:1', 'bf16x3=7:3', 'bf16=7:1', 'default2=']
    for cfg in configs:
        name, _, kv = cfg.partition('=')
        for k in range(KNOBS):
            L.spg_tune(k, 0)
        for item in filter(None, kv.split(',')):
            k, v = item.split(':')
            L.spg_tune(int(k), int(v))
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        L.spg_prof_enable(1)
        nprof = 3
        for _ in range(nprof):
            step()
        torch.cuda.synchronize()
        rows = read_shapes(L, nprof)
        tot, launches, flops = ctypes.c_double(), ctypes.c_long(), ctypes.c_double()
        L.spg_prof_read(ctypes.byref(tot), ctypes.byref(launches), ctypes.byref(flops), 1)
        L.spg_prof_enable(0)
        print(f'=== {name} [{kv}]: {ms:.3f} ms/step wall; instrumented GEMM time {tot.value / nprof:.3f} ms/step in '
              f'{launches.value / nprof:.0f} launches = {flops.value / (tot.value * 1e-3) / 1e12:.1f} TF', flush=True)
        for us_step, kname, N, K, cnt, avg, tf in rows[:args.top]:
            print(f'   {us_step:8.1f} us/step  {kname:<28s} N={N:<4d} K={K:<4d} x{cnt:<4.1f} avg {avg:7.1f} us  {tf:6.1f} TF', flush=True)
    for k in range(KNOBS):
        L.spg_tune(k, 0)


if __name__ == '__main__':
    main()
