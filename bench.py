#!/usr/bin/env python3
"""bench.py -- superpoints/sec of the superpoint-graph learning hot path on MI355X.

One "step" = one pass of the hot path over one batch of synthetic S3DIS-shaped scenes that are already resident
in HBM: zero_grad -> CloudEmbedder.run (PointNet) -> model.ecc (filter MLP + 10 x {ECC, GRU} + classifier) ->
weighted cross entropy -> backward -> bw_hook (PointNet backward) -> [N>1: ONE flat-bucket RCCL all-reduce]
-> element-wise gradient clamp -> Adam step      (the reference's trainer window, learning/main.py:199-213).

    python bench.py --gpus N --steps K --warmup W
For N > 1 one rank per GPU: either the caller launches it with torch.distributed.run (RANK/LOCAL_RANK/WORLD_SIZE in the
environment) or -- a plain `python bench.py --gpus N` without WORLD_SIZE -- the script re-launches ITSELF under
torch.distributed.run with N ranks on 127.0.0.1 (it refuses when the node has fewer than N GPUs).  Every rank holds its
own scene(s) (weak scaling), no data-path collective besides the gradient all-reduce.  Rank 0 prints ONE JSON line."""
import argparse
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.nn.functional as F  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense fp32
PEAK_BF16_MFMA_TFLOPS = 2500.0     # dense bf16 MFMA (same guide; the opt-in precision modes only)


def build_model(model_config, device, n_feat=14):
    """create_model of learning/main.py:414-431 with the S3DIS production flags (S3DIS.md:26-28); n_feat = 11 with
    --ptn_nfeat_stn 11 is the Semantic3D configuration (Semantic3D.md:20-22)."""
    from superpoint_graph_amd.learning import graphnet, pointnet
    torch.manual_seed(1)
    model = torch.nn.Module()
    model.ecc = graphnet.GraphNetwork(model_config, 32, [13, 32, 128, 64], 1, 0, 2, 30000, use_pyg=0, cuda=1)
    model.ptn = pointnet.PointNet([64, 64, 128, 128, 256], [256, 64, 32], [64, 64, 128], [128, 64], n_feat, n_feat, prelast_do=0)
    return model.to(device)


def make_batch(seeds, n_sp, n_edges, n_feat=14, n_classes=13):
    from superpoint_graph_amd import synth
    from superpoint_graph_amd.learning import spg
    scenes = [synth.scene(s, n_sp=n_sp, n_edges=n_edges, n_feat=n_feat, n_classes=n_classes) for s in seeds]
    targets, GIs, (meta, flag, clouds, diam) = spg.eccpc_collate([spg.sample_from_scene(s, f'scene{i}') for i, s in enumerate(scenes)])
    return targets, GIs, flag, clouds, diam, scenes


def cpu_model():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(model_config, scenes, state, max_seconds=25.0, n_feat=14):
    """The CPU path timed on this host's cores on the same scene, fwd+bwd like `value`.  With a reference checkout
    (/root/reference: the build container) the IMPORTED reference modules are timed (kind "reference",
    oracle/ref_baseline.py); on the GPU box, where the reference does not exist, the oracle port (oracle/spg_oracle.py,
    kind "port").  This is the only place bench.py touches oracle/ -- as the reported baseline, never as the measured path."""
    from oracle import ref_baseline
    from oracle import spg_oracle as O
    from superpoint_graph_amd import synth
    spec = O.ModelSpec(model_config=model_config, node_feats=n_feat, ptn_nfeat_stn=n_feat)
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    # torch's CPU kernels stop scaling (and oversubscribe badly) beyond a few dozen threads on these small layer shapes:
    # use up to 64, report both the threads used (`cores`) and what the host has (`host_cores`)
    torch.set_num_threads(max(1, min(ncpu, 64)))
    col = synth.collate_numpy(scenes[:1])
    idxn, degs, ef, _ = O.set_batch(col['edge_lists'], col['vcounts'], col['edge_feats'])
    batch = dict(clouds_flag=torch.from_numpy(col['clouds_flag']), clouds=torch.from_numpy(col['clouds']),
                 clouds_global=torch.from_numpy(col['clouds_global']), idxn=torch.from_numpy(idxn), degs=torch.from_numpy(degs),
                 edgefeats=torch.from_numpy(ef), label_mode=torch.from_numpy(col['targets'][:, 0].copy()))
    st = {k: v.detach().cpu().clone() for k, v in state.items()}
    n = int(batch['clouds_flag'].numel())
    sweep = ''
    if ref_baseline.available():
        med, cnt = ref_baseline.time_reference_step(model_config, batch, n_feat, st, max_seconds)
        kind, what = 'reference', 'imported reference modules (learning/pointnet.py, graphnet.py, modules.py, ecc/*; restated matrix-filter backward)'
    else:
        # torch's CPU kernels do not scale on these small layer shapes (64 threads are SLOWER than 8 on a 64-core EPYC): give
        # the CPU its best thread count -- one step each at 8 / 16 / 32 / 64 threads, then the median of up to 4 more steps
        # at the fastest setting
        t_begin = time.perf_counter()
        O.train_step(batch, spec, st, None)           # warm-up
        trial = {}
        for t in [c for c in (8, 16, 32, 64) if c <= max(ncpu, 8)]:
            torch.set_num_threads(min(t, ncpu))
            t0 = time.perf_counter()
            O.train_step(batch, spec, st, None)
            trial[torch.get_num_threads()] = time.perf_counter() - t0
            if time.perf_counter() - t_begin > 0.6 * max_seconds:
                break
        best = min(trial, key=trial.get)
        torch.set_num_threads(best)
        times = [trial[best]]
        while len(times) < 5 and (time.perf_counter() - t_begin) < max_seconds:
            t0 = time.perf_counter()
            O.train_step(batch, spec, st, None)
            times.append(time.perf_counter() - t0)
        med, cnt = float(np.median(times)), len(times)
        sweep = '; one step at ' + ', '.join(f'{k} threads {v * 1e3:.0f} ms' for k, v in trial.items())
        kind, what = 'port', ('oracle/spg_oracle.py train_step on torch-CPU: the reference\'s own op sequence (Conv1d / BatchNorm1d / Linear '
                              'through torch, restated ECC) -- no reference checkout on this machine; on the build container the port runs at '
                              '0.9-1.25x the speed of the imported reference modules, profiles/r03_cpu_reference_vs_port.json, oracle/devtools/cpu_baseline_compare.py)')
    return {'value': n / med, 'unit': 'superpoints/s', 'cores': torch.get_num_threads(), 'host_cores': ncpu, 'cpu_model': cpu_model(), 'kind': kind,
            'sample': f'{cnt} fwd+bwd steps of one {n}-superpoint scene (median {med * 1e3:.0f} ms/step), {what}{sweep}'}


def gemm_traffic_live(args, log):
    """HBM bytes per row-GEMM launch MEASURED in this run: two rocprofv3 passes of this very script (--pmc FETCH_SIZE and --pmc
    WRITE_SIZE in separate runs with --kernel-trace only, as MI355X_MICROARCH.md prescribes; 6 training steps each), summarised by
    tools/pmc_traffic.py with the gfx950 unit corrections (FETCH_SIZE x2, WRITE_SIZE x1: calibrated on known byte counts,
    profiles/r03_pmc_calib_*.txt).  -> (bytes per launch, all-kernel bytes per step, description) or None when rocprofv3 is not
    there / a pass fails (the committed file is used then)."""
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which('rocprofv3')
    if exe is None or os.environ.get('SPG_BENCH_NO_LIVE_PMC'):
        return None
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    try:
        import pmc_traffic
        dbs = []
        for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
            out = tempfile.mkdtemp(prefix=f'spg_pmc_{counter}_', dir='/tmp')
            cmd = [exe, '--kernel-trace', '--pmc', counter, '--output-format', 'rocpd', '-d', out, '--', sys.executable, os.path.abspath(__file__),
                   '--steps', '4', '--warmup', '2', '--no-cpu-baseline', '--no-forward-only', '--no-trainer-window', '--no-roofline', '--no-extras',
                   '--fused-step', str(args.fused_step),
                   '--precision', args.precision, '--scenes', str(args.scenes), '--n-sp', str(args.n_sp), '--n-edges', str(args.n_edges),
                   '--n-feat', str(args.n_feat), '--model-config', args.model_config]
            env = dict(os.environ, TMPDIR='/tmp', SPG_BENCH_NO_LIVE_PMC='1')
            subprocess.run(cmd, cwd='/tmp', env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=120, check=True)
            found = glob.glob(os.path.join(out, '**', '*.db'), recursive=True)
            if not found:
                return None
            dbs.append(found[0])
        t = pmc_traffic.summarise(dbs[0], dbs[1])
        for d in dbs:
            shutil.rmtree(os.path.dirname(os.path.dirname(d)), ignore_errors=True)
        log(f'live PMC passes: {t["hbm_mb_per_launch"]:.1f} MB per GEMM launch, {t["all_kernels_hbm_mb_per_step"]:.0f} MB per step')
        return t['hbm_mb_per_launch'] * 1e6, t['all_kernels_hbm_mb_per_step'] * 1e6, t.get('ecc'), 'measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes of 6 steps of this script), FETCH_SIZE x2.000 / WRITE_SIZE x1.000 (gfx950 units, calibrated: profiles/r03_pmc_calib_*.txt)'
    except Exception as e:      # a profiler problem must never take the bench line down
        log(f'live PMC passes failed ({type(e).__name__}: {e}); using the committed traffic file')
        return None


def gemm_traffic(args):
    """HBM bytes per row-GEMM launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate
    runs, gfx950 correction applied; profiles/*_gemm_traffic.json) -- only for the workload they were collected on.
    Returns (bytes per launch, source file) or (None, None): the value is STATIC (read from the file), not measured in
    this run -- PMC collection needs rocprofv3 around the process."""
    import glob
    default = (args.scenes == 1 and args.n_sp == 1000 and args.n_edges == 5000 and args.model_config == 'gru_10_0,f_13'
               and args.n_feat == 14)
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_gemm_traffic.json')))
    if not default or not files:
        return None, None
    with open(files[-1]) as f:
        t = json.load(f)
    return t['hbm_mb_per_launch'] * 1e6, os.path.relpath(files[-1], ROOT)


def trainer_window(args, dev, model, embedder, arena, seeds, n_classes, log, iters=40, fstep=None):
    """The reference's own "trainer time" window (learning/main.py:192-215): every step gets a FRESH batch -- the clouds
    come from pinned host memory (H2D on a side stream, overlapped with the previous step) and the batched graph is built
    anew on the GPU (GraphConvInfo.set_batch_device, as the CLI's collate does: ordering by target, edge-feature reordering,
    CSR / reverse CSR; the host concatenates and counts degrees) -- then zero_grad ... optimizer step as in the headline
    measurement.  Reported next to `value` (which keeps its inputs resident, as the metric definition says)."""
    from superpoint_graph_amd import ops, synth
    from superpoint_graph_amd.learning import ecc, pointnet, spg
    nb = 4
    batches = []
    for b in range(nb):
        scenes = [synth.scene(1000 + b * args.scenes + i, n_sp=args.n_sp, n_edges=args.n_edges, n_feat=args.n_feat, n_classes=n_classes)
                  for i in range(args.scenes)]
        samples = [spg.sample_from_scene(s, f'w{b}_{i}') for i, s in enumerate(scenes)]
        targets, _, (meta, flag, clouds, diam) = spg.eccpc_collate(samples)
        batches.append((targets, [s[1] for s in samples], flag, clouds.pin_memory(), diam))
    from superpoint_graph_amd.learning.prefetch import SideStreamBatches

    def fresh_batches(n):
        """what the CLI's collate does for every batch (eccpc_collate(device_batch=True) + the trainer's uploads): issued on the
        side stream by SideStreamBatches, so it overlaps the step in flight"""
        for it in range(n):
            targets, graphs, flag, clouds, diam = batches[it % nb]
            flag = flag.clone()                                    # a fresh batch object, as a collate produces it
            if os.environ.get('SPG_TW_LEGACY'):                    # (A/B only: the round-5 sequence, one staging copy per vector)
                gi = ecc.GraphConvInfo()
                gi.set_batch_device(graphs, spg.cloud_edge_feats)
                pointnet.stage_flags(flag)
                yield gi, flag, ops.upload(clouds, dev), ops.upload(diam, dev), ops.upload(targets[:, 0].contiguous(), dev)
                continue
            # as eccpc_collate(device_batch=True): the edge list, the edge features, CloudEmbedder's two index vectors, the labels
            # and the diameters travel in ONE staging copy (round 6; seven copies before), then ordering + CSR on the device
            iv, slot = pointnet.flag_index_vectors(flag)
            gi = ecc.GraphConvInfo()
            gi.set_batch_device(graphs, spg.cloud_edge_feats, extras=[iv, slot, targets[:, 0].contiguous(), diam])
            iv_d, slot_d, lab_d, diam_d = gi.extras_dev
            pointnet.attach_staged_flags(flag, iv_d, slot_d)
            yield gi, flag, ops.upload(clouds, dev), diam_d, lab_d

    waited = [0.0]

    def run(n):
        ssb = SideStreamBatches(fresh_batches(n))
        try:
            _run(ssb)
        finally:
            waited[0] += ssb.host_wait_seconds

    def _run(ssb):
        for gi, flag, c, d, lab in ssb:
            model.ecc.set_info([gi], 1)
            arena.zero_grad()
            if fstep is not None:
                fstep(flag, c, d, gi, lab)
                arena.adam_step(lr=1e-2, weight_decay=0.0, grad_clip=1.0)
                continue
            emb = embedder.run(model, None, flag, c, d)
            loss = ops.cross_entropy(model.ecc(emb), lab)
            loss.backward(arena.one)
            embedder.bw_hook()
            arena.adam_step(lr=1e-2, weight_decay=0.0, grad_clip=1.0)
    run(12)               # the side stream's allocator pool and the staging ring fill during the first batches
    torch.cuda.synchronize()
    runs = []
    for _ in range(3):    # (the window is bimodal on some boxes -- roughly one run in four lands 5-30 % above the others, with the
        torch.cuda.synchronize()      # round-5 sequence as well --, so: three runs, the median is reported, all three are listed)
        t0 = time.perf_counter()
        run(iters)
        th = (time.perf_counter() - t0) / iters          # the host over the WHOLE loop: once it runs ahead of the GPU by more than the
        torch.cuda.synchronize()                         # hardware queue holds, its launches block -- this figure then tends to dt
        runs.append(((time.perf_counter() - t0) / iters, th))
    dt, t_host_long = sorted(runs)[1]
    # the host's own cost of a fresh-batch step: short bursts from an idle GPU (12 steps = ~450 launches: they fit the queue, nothing
    # blocks), median of 5
    bursts = []
    for _ in range(5):
        torch.cuda.synchronize()
        waited[0] = 0.0
        t0 = time.perf_counter()
        run(12)
        t1 = time.perf_counter()
        bursts.append(((t1 - t0 - waited[0]) / 12, (t1 - t0) / 12))
    torch.cuda.synchronize()
    t_host, t_host_with_waits = sorted(bursts)[2]
    n = int(batches[0][2].numel())
    log(f'trainer window: {dt * 1e3:.3f} ms/step (host enqueue {t_host * 1e3:.3f} ms/step in 12-step bursts, {t_host_long * 1e3:.3f} over the {iters}-step loop)')
    return {'ms_per_step': dt * 1e3, 'ms_per_step_runs': [r[0] * 1e3 for r in runs], 'superpoints_per_s': n / dt, 'host_enqueue_ms_per_step': t_host * 1e3,
            'host_enqueue_ms_per_step_long_loop': t_host_long * 1e3, 'host_enqueue_plus_waits_ms_per_step': t_host_with_waits * 1e3,
            'host_enqueue_how': 'host WORK per fresh-batch step: median of five 12-step bursts started on an idle GPU (the launches fit the hardware '
                                'queue), minus the time SideStreamBatches spent waiting on the host for a batch under construction (it waits there '
                                'instead of putting a cross-stream dependency on the training stream; ..._plus_waits includes it); the long-loop figure '
                                'also includes the time the host waits for queue space once it is ahead of the GPU',

            'what': 'fresh batch every step: pinned H2D of the clouds + ONE packed staging copy of everything else (edge list, edge features, index vectors, labels, diameters) + GraphConvInfo.set_batch_device (ordering by '
                    'target + CSR / reverse CSR as kernels), both on a side stream (SideStreamBatches, as the CLI does) + zero_grad..Adam '
                    '(learning/main.py:192-215)'}


def other_workloads(args, log):
    """Short runs (subprocesses of this script: fresh process, own spg_tune state) of the other BASELINE.json configurations
    next to the headline: configs[2] = the reference's default --batch_size 2, configs[3] = 8 scenes per step, configs[4] =
    Semantic3D scale (10 000 superpoints, 50 000 superedges, 11 point features, vector filters `gru_10,f_8`) in fp32 and with
    the opt-in split-bf16 MFMA operands.  -> {name: {superpoints_per_s, ms_per_step, roofline fractions, ...}}"""
    import subprocess
    common = ['--steps', '12', '--warmup', '4', '--no-cpu-baseline', '--no-forward-only', '--no-trainer-window', '--no-live-pmc',
              '--no-extras', '--fused-step', str(args.fused_step)]
    sema = ['--n-sp', '10000', '--n-edges', '50000', '--n-feat', '11', '--model-config', 'gru_10,f_8']
    runs = {'s3dis_2_scenes_per_step_f32': ['--scenes', '2'], 's3dis_8_scenes_per_step_f32': ['--scenes', '8'],
            'semantic3d_scale_f32': sema + ['--infer-line'], 'semantic3d_scale_bf16x3': sema + ['--precision', 'bf16x3']}
    out = {}
    for name, extra in runs.items():
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__)] + common + extra, capture_output=True, text=True, timeout=150)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1]
            d = json.loads(line)
            rf = d.get('roofline', {})
            out[name] = {'superpoints_per_s': d['value'], 'ms_per_step': d['ms_per_step'], 'dtype': d['dtype'].split(' ')[0],
                         'workload': d['config']['workload'], 'roofline_bound': rf.get('bound'), 'roofline_frac': rf.get('frac'), 'roofline_note': rf.get('bound_note'),
                         'roofline_unit': rf.get('unit'), 'roofline_achieved': rf.get('achieved'), 'dominant_frac_of_fp32_mfma_peak': rf.get('dominant_frac'),
                         'launches_per_step_gemm': rf.get('launches_per_step')}
            if 'inference' in d:      # the evaluation forward at this scale (eval_final: 10 passes per scene)
                out[name + '_inference'] = d['inference']
            log(f'{name}: {d["value"]:.0f} superpoints/s, {d["ms_per_step"]:.3f} ms/step, roofline {rf.get("bound")} {rf.get("frac")}')
        except Exception as e:      # a side measurement must never take the headline down
            out[name] = {'error': f'{type(e).__name__}: {e}'}
            log(f'{name}: failed ({e})')
    return out


def inference_line(args, dev, model, flag, clouds_d, diam_d, GIs, iters=30):
    """the evaluation forward of the FULL model of this run (learning/main.py:246-262; eval_final runs it 10
    times per scene, :267-311) as ONE library call (spg_infer_step), with its roofline: algorithmic forward FLOP of the two
    PointNet segments and the filter network / time against the fp32-MFMA peak (the fused inference stacks of DESIGN 4.7 are
    MFMA-bound: activations never leave the CU)."""
    from superpoint_graph_amd.flat import FlatParameters
    from superpoint_graph_amd.fused import FusedStep
    was_training = model.training
    model.eval()
    try:
        step = FusedStep(model, FlatParameters(model, lazy_zero=True))
        model.ecc.set_info(GIs, 1)
        for _ in range(5):
            step.infer(flag, clouds_d, diam_d, GIs[0])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            step.infer(flag, clouds_d, diam_d, GIs[0])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
    finally:
        model.train(was_training)
    B, P = int(clouds_d.shape[0]), int(clouds_d.shape[2])
    ptn = model.ptn
    conv_mac = lambda nin, widths: sum(a * b for a, b in zip([nin] + list(widths[:-1]), widths))
    stn = ptn.stn if getattr(ptn, 'nfeat_stn', 0) > 0 else None
    mac_pt = conv_mac(int(clouds_d.shape[1]), list(ptn._nf_conv)) + (conv_mac(stn._nfeat, list(stn._nf_conv)) if stn is not None else 0)
    mac_sp = conv_mac(ptn._nf_conv[-1] + ptn._nfeat_global, list(ptn._nf_fc)) + (conv_mac(stn._nf_conv[-1], list(stn._nf_fc) + [stn._K * stn._K]) if stn is not None else 0)
    E = int(GIs[0].get_buffers()[0].numel())
    fnet = model.ecc.gconvs[0]._fnet
    mac_edge = sum(m.weight.shape[0] * m.weight.shape[1] for m in fnet if hasattr(m, 'weight') and m.weight.dim() == 2)
    gflop = 2.0 * (B * (P * mac_pt + mac_sp) + E * mac_edge) / 1e9
    return {'workload': f'evaluation forward (running BatchNorm statistics) of {args.model_config} as ONE call (spg_infer_step): '
                        f'{int(flag.numel())} superpoints x {P} pts x {int(clouds_d.shape[1])} f, {E} edges',
            'ms': dt * 1e3, 'superpoints_per_s': int(flag.numel()) / dt, 'algorithmic_gflop': gflop,
            'roofline': {'bound': 'mfma', 'achieved': gflop / dt / 1e3, 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': gflop / dt / 1e3 / PEAK_FP32_MFMA_TFLOPS,
                         'what': 'PointNet segments + filter network forward FLOP (the RNN-ECC iterations are latency / L2-bound and not counted) / wall time of the whole call'}}


def forward_only(dev, flag, clouds_d, diam_d, GIs, n_feat, iters=40):
    """BASELINE.json configs[1]: PointNet + 1 x ECC, FORWARD only, on the same scene (model `gru_1_0,f_13`), eval-mode
    BatchNorm (inference) and train-mode BatchNorm (batch statistics), no autograd."""
    import types
    from superpoint_graph_amd.learning import pointnet
    model = build_model('gru_1_0,f_13', dev, n_feat)
    model.ecc.set_info(GIs, 1)
    embedder = pointnet.CloudEmbedder(types.SimpleNamespace(cuda=1, ptn_mem_monger=1))
    out = {'workload': 'same scene, PointNet + gru_1_0,f_13 (one ECC iteration), forward only, no_grad; eval_bn: the evaluation '
                       'forward as ONE library call (spg_infer_step; eval_bn_modules: the same through the Python modules -- host-bound), '
                       'train_bn: batch statistics, through the modules'}
    from superpoint_graph_amd.flat import FlatParameters
    from superpoint_graph_amd.fused import FusedStep
    step = FusedStep(model, FlatParameters(model, lazy_zero=True))

    def modules():
        with torch.no_grad():
            return model.ecc(embedder.run(model, None, flag, clouds_d, diam_d))
    for name, train, fwd, n in (('eval_bn', False, lambda: step.infer(flag, clouds_d, diam_d, GIs[0]), 5 * iters),
                                ('eval_bn_modules', False, modules, iters), ('train_bn', True, modules, iters)):
        model.train(train)
        for _ in range(5):
            fwd()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fwd()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        out[name] = {'ms': dt * 1e3, 'superpoints_per_s': int(flag.numel()) / dt}
    return out


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) without a torchrun environment: run N ranks of this very command line under
    torch.distributed.run on this node and hand their exit code back.  The reference has no counterpart (single process,
    learning/main.py:180); this is the launcher the multi-GPU rows of BASELINE.json need."""
    import socket
    import subprocess
    if args.device_index < 0:           # one rank per GPU: every rank needs its own device (RCCL refuses two ranks on one)
        have = torch.cuda.device_count()
        if have < args.gpus:
            raise SystemExit(f'bench.py: --gpus {args.gpus} but this node has {have} GPU(s) visible; refusing to report fewer ranks '
                             f'than asked for (use --device-index D --backend gloo to put all ranks on one device for a control-flow test)')
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'), SPG_BENCH_SELF_LAUNCHED='1')
    print(f'[bench] --gpus {args.gpus} without WORLD_SIZE: launching {args.gpus} ranks under torch.distributed.run (port {port})', file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--scenes', type=int, default=1, help='scenes per GPU per step')
    ap.add_argument('--n-sp', type=int, default=1000)
    ap.add_argument('--n-edges', type=int, default=5000)
    ap.add_argument('--model-config', default='gru_10_0,f_13')
    ap.add_argument('--n-feat', type=int, default=14, help='point features (14: S3DIS xyzrgbelpsvXYZ, 11: Semantic3D)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--backend', default=None, help='torch.distributed backend (default: nccl = RCCL); gloo is for the control-flow test on a 1-GPU box')
    ap.add_argument('--device-index', type=int, default=-1, help='GPU of this rank (default: LOCAL_RANK)')
    ap.add_argument('--no-trainer-window', action='store_true', help='skip the fresh-batch-per-step side measurement (learning/main.py:200-215 window)')
    ap.add_argument('--no-forward-only', action='store_true', help='skip the forward-only side measurement (BASELINE configs[1])')
    ap.add_argument('--sync-bn', type=int, default=0, help='1: BatchNorm statistics all-reduced over the ranks (exact single-process batch semantics); 0: per-rank statistics')
    ap.add_argument('--native-rccl', type=int, default=0, help='1: the C library issues the RCCL collectives itself (own communicator, spg_rccl_*: no Python between a BatchNorm finalize kernel and its all-reduce); 0 (default): torch.distributed calls -- the library\'s communicator has only ever run at world size 1 on the 1-GPU test box, so it stays opt-in until it has been exercised on a multi-GPU node (tests/test_gpu_dist.py runs it whenever device_count() > 1)')
    ap.add_argument('--precision', default='f32', choices=['f32', 'bf16x3', 'bf16'],
                    help="arithmetic of the wide row-GEMMs: f32 = fp32 MFMA (default, the headline); bf16x3 = split-bf16 products (three "
                         "bf16 MFMAs per operand pair, fp32 accumulate, ~2^-16 per product); bf16 = bf16 operands.  A SEPARATE line: never "
                         "replaces the f32 headline (tolerances: tests/test_gpu_precision.py)")
    ap.add_argument('--group-trace', type=int, default=0, help='tools: with an ATTRIBUTION build of the library (SPG_HIP_LIB), print the per-job time spans of the grouped launches of ONE step after the warm-up, and exit')
    ap.add_argument('--tune', default='', help='A/B switches of the library for experiments: comma-separated key:value pairs of spg_tune (include/spg_hip.h), e.g. 8:1 = per-iteration RNN-ECC launches, 9:1 = no side stream')
    ap.add_argument('--no-live-pmc', action='store_true', help='do not run the two rocprofv3 PMC passes for roofline.traffic (use the committed file)')
    ap.add_argument('--hipgraph', type=int, default=0, help='capture the step in a hipGraph (torch.cuda.CUDAGraph) and replay it')
    ap.add_argument('--fused-step', type=int, default=1, help='1 (default): forward + backward as ONE library call (superpoint_graph_amd/fused.py: spg_train_step; same kernels and results as the module path, tests/test_gpu_fused.py); 0: CloudEmbedder.run -> model.ecc -> cross_entropy -> backward -> bw_hook through the modules')
    ap.add_argument('--infer-line', action='store_true', help='also time the evaluation forward of this run\'s model as one call (spg_infer_step) and report it with its roofline (`inference`)')
    ap.add_argument('--no-extras', action='store_true', help='skip the short runs of the other BASELINE.json configurations (2 / 8 scenes per step, Semantic3D scale in f32 and split-bf16) and the sustained repeat')
    args = ap.parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(self_launch(args))

    import faulthandler
    faulthandler.enable()
    faulthandler.dump_traceback_later(240, exit=False, file=sys.stderr)     # if anything hangs, say where

    def log(msg):
        print(f'[bench +{time.perf_counter() - T0:.1f}s] {msg}', file=sys.stderr, flush=True)

    T0 = time.perf_counter()
    from superpoint_graph_amd import _lib, dist as spd, ops
    from superpoint_graph_amd.learning import pointnet
    PREC = {'f32': 0, 'bf16': 1, 'bf16x3': 3}[args.precision]
    if args.device_index >= 0:
        torch.cuda.set_device(args.device_index)
    rank, local, world = spd.init_from_env(args.backend)
    if args.device_index >= 0:
        local = args.device_index
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    assert torch.cuda.is_available(), 'bench.py needs a GPU (no CPU fallback for the product path)'
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)

    model = build_model(args.model_config, dev, args.n_feat)
    n_classes = int(args.model_config.split('f_')[-1].split(',')[0])
    model.train()
    seeds = [rank * args.scenes + i for i in range(args.scenes)]          # every rank its own scenes (weak scaling)
    targets, GIs, flag, clouds, diam, scenes = make_batch(seeds, args.n_sp, args.n_edges, args.n_feat, n_classes)
    # inputs resident in HBM before the timed region
    clouds_d, diam_d = clouds.to(dev), diam.to(dev)
    label_mode = targets[:, 0].to(dev)
    model.ecc.set_info(GIs, 1)                                             # H2D of the index buffers + device CSR build
    embedder = pointnet.CloudEmbedder(types.SimpleNamespace(cuda=1, ptn_mem_monger=1))
    from superpoint_graph_amd.flat import FlatParameters
    # parameters / gradients as views of one flat buffer each; as in the training CLI (learning/main.py of this package):
    # zero_grad() launches nothing (the backward kernels overwrite every gradient), BatchNorm batch counters on the host
    arena = FlatParameters(model, lazy_zero=True, host_counters=True)
    if _lib.lib().spg_tune(7, PREC) < 0:             # precision mode of the wide row-GEMMs (0 = fp32 MFMA)
        raise RuntimeError('libspg_hip.so has no precision switch (spg_tune key 7)')
    shared_gpu = world > 1 and args.device_index >= 0
    if shared_gpu:
        # several ranks on ONE device (the control-flow test of a 1-GPU box): the one-launch RNN-ECC recurrence needs all of ITS
        # workgroups resident at once, which two processes sharing the CUs cannot promise -- its bounded spins would time out (the
        # self-check below catches exactly that).  The per-iteration kernels have no such assumption.
        _lib.lib().spg_tune(8, 1)
    for kv in [t for t in args.tune.split(',') if t]:
        k, v = kv.split(':')
        if _lib.lib().spg_tune(int(k), int(v)) < 0:
            raise RuntimeError(f'spg_tune rejected {kv}')
    state0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    n_sp_step = int(flag.numel())

    native = False
    if world > 1 and args.native_rccl and dist.get_backend() == 'nccl':
        spd.init_native_rccl()                       # gradient / BatchNorm collectives enqueued by libspg_hip itself
        native = True
    if args.sync_bn:       # BatchNorm statistics over the scenes of ALL ranks (exact single-process-batch semantics)
        spd.enable_sync_bn(dev)

    dp = world > 1 or bool(args.sync_bn)
    exchange = {}
    from superpoint_graph_amd import fused as spg_fused
    fstep = None
    if args.fused_step and spg_fused.supports(model) and (not args.sync_bn or spd.sync_bn_mode() == 'slots') and not args.hipgraph:
        fstep = spg_fused.FusedStep(model, arena, reduction='sum' if dp else 'mean', ptn_mem_monger=True)

    def fwd_bwd():
        arena.zero_grad()
        if fstep is not None:      # the same step as ONE call into the library (include/spg_hip.h: spg_train_step)
            exchange['loss'], _ = fstep(flag, clouds_d, diam_d, GIs[0], label_mode)
            exchange['w'] = fstep.normaliser
            return
        emb = embedder.run(model, None, flag, clouds_d, diam_d)
        out = model.ecc(emb)
        if dp:
            # data parallel: back-propagate the SUM-reduced loss (the gradients carry this rank's loss weight, superpoint_graph_amd/
            # dist.py); the weight stays on the device and rides along with the gradient all-reduce
            loss, exchange['w'] = ops.cross_entropy(out, label_mode, reduction='sum', return_normaliser=True)
        else:
            loss = ops.cross_entropy(out, label_mode)    # learning/main.py:205, one launch each way
        exchange['loss'] = loss.detach()
        loss.backward(arena.one)                         # (a cached 1: autograd's implicit ones_like(loss) is a fill launch)
        embedder.bw_hook()

    def update():
        # clamp (main.py:210-212) + Adam (main.py:213), one launch; data parallel: the division by the summed loss weights is in it
        arena.adam_step(lr=1e-2, weight_decay=0.0, grad_clip=1.0, grad_div=arena.normaliser if dp else None)

    def eager_step():
        fwd_bwd()
        if dp:
            arena.allreduce_sums(exchange['w'])          # ONE collective: [gradients | loss weight], no host synchronisation
        update()

    step = eager_step
    if args.hipgraph:
        # HIP graph capture of the launch-bound step (~230 dependent kernel launches): the forward/backward and the
        # clamp+Adam update are captured separately so that the RCCL all-reduce between them stays an eager call.
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                eager_step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g_fb, g_up = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_fb):
            fwd_bwd()
        with torch.cuda.graph(g_up):
            update()

        def graph_step():
            g_fb.replay()
            if dp:
                arena.allreduce_sums(exchange['w'])
            g_up.replay()
        step = graph_step
        log('hipGraph captured')

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    log('model and batch ready')
    for _ in range(args.warmup):
        step()
    barrier()
    log('warm-up done')
    if args.group_trace:
        import ctypes
        L = _lib.lib()
        MAXL, MAXJ = 64, 16
        buf = torch.zeros(MAXL * MAXJ * 2, dtype=torch.int64, device=dev)
        buf[0::2] = -1
        if L.spg_group_trace(buf.data_ptr(), MAXL) != 0:
            raise RuntimeError('--group-trace needs an attribution build of the library: make -C superpoint_graph_amd/csrc ATTRIBUTION=1, SPG_HIP_LIB=<that .so>')
        step()
        torch.cuda.synchronize()
        n = L.spg_group_trace_read(None, 0)
        ints = (ctypes.c_int * max(n, 1))()
        L.spg_group_trace_read(ints, n)
        L.spg_group_trace(None, 0)
        t = buf.cpu().view(MAXL, MAXJ, 2)
        kinds = {1: 'gemm', 2: 'wgrad', 3: 'colsum', 4: 'edge_wgrad', 5: 'pad', 6: 'zero', 7: 'reduce'}
        pos, li = 0, 0
        first = None
        while pos < n:
            nj, heavy = ints[pos], ints[pos + 1]
            pos += 2
            starts = [int(t[li, j, 0]) for j in range(nj)]
            ends = [int(t[li, j, 1]) for j in range(nj)]
            l0, l1 = min(starts), max(ends)
            first = l0 if first is None else first
            print(f'launch {li}: {"heavy" if heavy else "light"} {nj} jobs, starts at {(l0 - first) / 100:.1f} us, span {(l1 - l0) / 100:.1f} us')
            for j in range(nj):
                kind, var, gx, gy, gz, w = (ints[pos + k] for k in range(6))
                pos += 6
                print(f'    job {j}: {kinds.get(kind, kind)} variant {var} grid ({gx},{gy},{gz}) = {gx * gy * gz} wgs, weight {w}: '
                      f'{(starts[j] - l0) / 100:6.1f} .. {(ends[j] - l0) / 100:6.1f} us')
            li += 1
        return
    # the timed region: EXACTLY K steps between barrier + synchronize on both sides, nothing else in the stream (round 5: the
    # per-step event pairs moved to a pass of their own below -- an event record is a marker packet that drains the queue's
    # pipeline, ~6 us per step that no training loop pays)
    # per-step device times (min / median / max next to the mean): the same K steps with one event pair per step.  This pass runs
    # FIRST (round 6): with the driver's W = 5 / K = 20 the timed region used to start 6 ms after a second of host-side set-up, on a
    # device whose clocks were still coming up -- the mean of the region sat 0.6-2.5 % above the median of this pass and above the
    # sustained figure of the same run.  The order changes nothing in what is timed below.
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    ev[0].record()
    for i in range(args.steps):
        step()
        ev[i + 1].record()
    barrier()
    step_ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps))
    t0 = time.perf_counter()
    for i in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = n_sp_step * world * args.steps / dt
    log(f'timed region done: {ms_per_step:.3f} ms/step')

    # who took part: every rank contributes 1 (all-reduce on the bench's process group) -- the line says how many ranks the
    # collectives actually saw, not what --gpus asked for
    ranks_seen = 1
    if world > 1:
        t = torch.ones(1, dtype=torch.float64, device=dev if dist.get_backend() == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        ranks_seen = int(t.item())
    # the gradient exchange alone, event-bracketed on the compute stream in a short pass OUTSIDE the timed region
    ar_us = None
    if dp:
        e0 = [torch.cuda.Event(enable_timing=True) for _ in range(8)]
        e1 = [torch.cuda.Event(enable_timing=True) for _ in range(8)]
        for i in range(8):
            fwd_bwd()
            e0[i].record()
            arena.allreduce_sums(exchange['w'])
            e1[i].record()
            update()
        barrier()
        ar = sorted(a.elapsed_time(b) for a, b in zip(e0, e1))
        ar_us = ar[len(ar) // 2] * 1e3
        if world > 1:
            t = torch.tensor([ar_us], dtype=torch.float64, device=dev if dist.get_backend() == 'nccl' else 'cpu')
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ar_us = float(t.item())

    # the run checks itself: the loss of the last step is finite and no dataflow-synchronised RNN-ECC launch timed out (a spin
    # time-out carries on with stale states -- the number would still print; learning/main.py of this package checks the same)
    def self_check(where):
        torch.cuda.synchronize()
        nerr, withheld = ops.persistent_ecc_status()       # (time-outs, optimiser updates the device withheld because of them)
        loss_v = float(exchange['loss'].item())
        if dp:                                   # SUM-reduced loss of this rank / its own normaliser = its mean loss
            w = float(exchange['w'].item())
            loss_v = loss_v / w if w > 0 else float('nan')
        gfin = bool(torch.isfinite(arena.flat.grad).all().item())
        if nerr != 0 or withheld != 0 or not gfin or not np.isfinite(loss_v):
            raise SystemExit(f'bench.py self-check FAILED after the {where}: persistent RNN-ECC time-outs {nerr}, optimiser updates withheld {withheld}, loss {loss_v}, gradients finite {gfin}')
        return {'loss': loss_v, 'persistent_errors': nerr, 'updates_withheld': withheld, 'grads_finite': gfin, 'checked_after': where}
    check = self_check('timed region')

    wl = f'{args.model_config} fwd+bwd+Adam; {args.scenes} scene/GPU x {args.n_sp} sp x 128 pts x {args.n_feat} f, {args.n_edges} edges x 13 f'
    result = {
        'metric': 'superpoints/sec (embed+ECC fwd+bwd), S3DIS-shaped SPG', 'value': value, 'unit': 'superpoints/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step,
        'ms_per_step_min': step_ms[0], 'ms_per_step_median': step_ms[len(step_ms) // 2], 'ms_per_step_max': step_ms[-1],
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': {'f32': 'f32', 'bf16x3': 'bf16x3 (split-bf16 MFMA operands of the wide row-GEMMs, f32 accumulate and f32 everywhere else; NOT the headline arithmetic)', 'bf16': 'bf16 (MFMA operands of the wide row-GEMMs, f32 accumulate and f32 everywhere else; NOT the headline arithmetic)'}[args.precision], 'data': 'synthetic',
        'config': {'workload': wl,
                   'workload_detail': f'synthetic S3DIS-shaped SPG (SURVEY.md 8d scene(seed)): PointNet + {args.model_config}' + (' = S3DIS production model, matrix filters, 10 GRU iterations' if args.model_config == 'gru_10_0,f_13' else '') + '; one step = zero_grad, forward, weighted CE, backward, bw_hook, clamp + Adam on resident inputs',
                   'superpoints_per_step': n_sp_step * world, 'hipgraph': bool(args.hipgraph), 'step_call': 'spg_train_step (one library call: forward + backward)' if fstep is not None else 'module API (CloudEmbedder.run, model.ecc, cross_entropy, backward, bw_hook)', 'parallelism': (f'dp{world} (one scene shard per GPU, one flat-bucket RCCL all-reduce, ' + ('issued by libspg_hip' if native else 'torch.distributed') + ')') if world > 1 else 'single GPU',
                   'batchnorm': ('synchronised over ranks (' + str(spd.sync_bn_mode()) + ')') if args.sync_bn else 'per-rank statistics', 'precision': args.precision},
        'ranks_seen': ranks_seen, 'allreduce_us_per_step': ar_us, 'batchnorm': 'sync' if args.sync_bn else 'per-rank',
        'ranks_share_one_gpu': shared_gpu,
        'self_check': check,
    }
    if ranks_seen != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but the collectives saw {ranks_seen} rank(s)')

    if not args.no_roofline:
        # dominant kernels = the MFMA row-GEMMs (PointNet convs/FCs + filter net, fwd/dgrad/wgrad): each launch bracketed by
        # hipEvents on its stream in a separate instrumented pass of the SAME step (events inside the timed region would
        # perturb `value`).  EVERY rank runs the pass (the step contains the gradient all-reduce); rank 0 reports.
        L = _lib.lib()
        import ctypes
        torch.cuda.synchronize()
        L.spg_prof_enable(1)
        nprof = 3
        for _ in range(nprof):
            eager_step()                     # instrumented launches cannot be replayed from a graph
        torch.cuda.synchronize()
        ms, launches, flops = ctypes.c_double(), ctypes.c_long(), ctypes.c_double()
        # the single instantiation with the largest share of the step: forward row-GEMM 128x128 tile, BN+ReLU prologue,
        # full-tile fast path (conv3/conv4/conv5 + the STN's last conv) -- its average duration is what the rocprofv3
        # kernel-trace summary in profiles/ lists for spg_rowgemm_kernel<128, 128, 2, 2, false, 1, true>
        dms, dl, dfl = ctypes.c_double(), ctypes.c_long(), ctypes.c_double()
        L.spg_prof_read_tag(L.spg_prof_tag(1, 128, 128, 0, 1, 1), ctypes.byref(dms), ctypes.byref(dl), ctypes.byref(dfl))
        # ... and the single heaviest LAUNCH SHAPE of the step (one (instantiation, N, K) row of the per-shape table): the
        # instantiation average above mixes the MFMA-bound 128->256 layer with HBM-bound 64->128 ones
        keys, vals = (ctypes.c_int * (4 * 256))(), (ctypes.c_double * (2 * 256))()
        nshape = L.spg_prof_read_shapes(keys, vals, 256)
        # (tag kind 3 = grouped launches of several few-row GEMMs / weight gradients: not a shape)
        shapes = [j for j in range(nshape) if keys[4 * j] // 1000000 != 3]
        top = max(shapes, key=lambda j: vals[2 * j]) if shapes else -1
        L.spg_prof_read(ctypes.byref(ms), ctypes.byref(launches), ctypes.byref(flops), 1)
        L.spg_prof_enable(0)
        log('instrumented pass done')
        ach = flops.value / (ms.value * 1e-3) / 1e12 if ms.value > 0 else 0.0
        live = gemm_traffic_live(args, log) if (world == 1 and not args.no_live_pmc) else None
        if live is not None:
            traffic, traffic_all, traffic_src, traffic_note = live[0], live[1], 'live', live[3]
            live = (live[0], live[1], live[3], live[2])
        else:
            traffic, traffic_src = gemm_traffic(args)
            traffic_all = None
            traffic_src = ('static:' + traffic_src) if traffic_src else None
            traffic_note = 'HBM bytes per launch (rocprofv3 PMC passes of the same command, tools/collect_profiles.sh; read from the file, not measured in this run)'
        gflop_step = flops.value / nprof / 1e9
        # flat keys only (nested objects are dropped by the driver's parser): aggregate over every MFMA GEMM launch of a
        # step, the dominant instantiation, and the whole step (algorithmic GEMM FLOP / wall time per step)
        result['roofline'] = {'bound': 'mfma', 'achieved': ach, 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                              'frac': ach / PEAK_FP32_MFMA_TFLOPS, 'traffic': traffic,
                              'traffic_source': traffic_src, 'traffic_unit': 'HBM bytes per GEMM launch; ' + traffic_note,
                              'traffic_all_kernels_per_step': traffic_all,
                              'kernel': 'spg_rowgemm_kernel + spg_bwdpair_kernel + spg_wgrad_kernel (fp32 MFMA 32x32x2)',
                              'peak_note': 'gfx950 executes v_mfma_f32_32x32x2_f32 on the SIMD\'s fp32 lanes: a partner wave\'s VALU work ADDS to the matrix time instead of hiding behind it (profiles/r06_coissue_probe.txt: MFMA stream 3.53 ms + VALU stream 2.25 ms = 5.64 ms together; bf16 MFMA 6.98 -> 7.04 ms) -- the fused prologues / epilogues of these kernels are priced in MFMA time; under load the chip clocks ~2.0-2.1 GHz (shader cycles / kernel time, profiles/r06_bwdpair_roles.txt), the peak is quoted at 2.4 GHz',
                              'launches_per_step': launches.value / nprof, 'gemm_ms_per_step': ms.value / nprof,
                              'algorithmic_gflop_per_step': gflop_step,
                              'step_achieved': gflop_step / ms_per_step, 'step_frac': gflop_step / ms_per_step / PEAK_FP32_MFMA_TFLOPS}
        if traffic_all:
            # the second roof of the step: HBM bytes of ALL kernels of a step / step time against 8 TB/s (6.29 TB/s achievable)
            result['roofline'].update({'hbm_bytes_per_step': traffic_all, 'hbm_achieved_gbs': traffic_all / (ms_per_step * 1e-3) / 1e9,
                                       'hbm_frac': traffic_all / (ms_per_step * 1e-3) / 8e12, 'hbm_peak_gbs': 8000.0})
        if live is not None and len(live) > 3 and live[3]:
            result['roofline']['ecc'] = live[3]      # the RNN-ECC kernels: duration, HBM bytes, what bounds them
        if PREC != 0 and not traffic:
            # no PMC pass for this workload: the algorithmic operand bytes of the instrumented GEMM launches (every operand and the
            # output once, fp32 in memory) -- a LOWER bound of the traffic, hence of the achieved rate
            alg = 0.0
            for j in range(nshape):
                n_, k_, cnt = keys[4 * j + 1], keys[4 * j + 2], keys[4 * j + 3]
                if n_ > 0 and k_ > 0 and cnt > 0:
                    m_ = vals[2 * j + 1] / (2.0 * n_ * k_ * cnt)
                    alg += cnt * 4.0 * (m_ * k_ + n_ * k_ + m_ * n_)
            if alg > 0:
                traffic = alg / max(launches.value, 1)
                result['roofline'].update({'traffic': traffic, 'traffic_source': 'algorithmic',
                                           'traffic_unit': 'ALGORITHMIC bytes per GEMM launch (operands and output once, fp32 in memory): a lower bound of the HBM traffic'})
        if PREC != 0 and traffic:
            # the bf16 modes move the wide GEMMs under the HBM roof (activations stay fp32 in memory: same bytes, less matrix
            # time): report the GEMM launches against HBM -- static traffic of the same launches ÷ their measured time
            gbs = traffic * launches.value / (ms.value * 1e-3) / 1e9
            # (at ~0.25 of 8 TB/s that is the NEARER roof, not a bound -- the mode's GEMMs are limited by converting
            # and splitting fp32 operands on the fly (VALU issue), so both fractions are reported and the note says so)
            nprod = 3 if PREC == 3 else 1
            result['roofline'].update({'bound': 'hbm', 'achieved': gbs, 'peak': 8000.0, 'unit': 'GB/s', 'frac': gbs / 8000.0,
                                       'mfma_bf16_frac': ach * nprod / PEAK_BF16_MFMA_TFLOPS,
                                       'bound_note': 'nearer roof of two, neither is reached: fp32 operands are split into bf16 planes while '
                                                     'staging (VALU issue bound); mfma_bf16_frac = %d bf16 products per fp32 product against the '
                                                     'dense bf16 MFMA peak' % nprod,
                                       'fp32_equivalent_tflops': ach,
                                       'kernel': 'spg_rowgemm_kernel (bf16 MFMA 32x32x16, %s) + spg_wgrad_kernel' % args.precision})
        if dl.value > 0:
            dach = dfl.value / (dms.value * 1e-3) / 1e12
            result['roofline'].update({
                'dominant_kernel': 'spg_rowgemm_kernel<128, 128, 2, 2, false, 1, true, true, %d>' % PREC, 'dominant_launches_per_step': dl.value / nprof,
                'dominant_avg_us': dms.value / dl.value * 1e3, 'dominant_gflop_per_launch': dfl.value / dl.value / 1e9,
                'dominant_achieved': dach, 'dominant_frac': dach / PEAK_FP32_MFMA_TFLOPS})
        if top >= 0 and vals[2 * top] > 0:
            tms, tfl, tcnt = vals[2 * top], vals[2 * top + 1], keys[4 * top + 3]
            result['roofline'].update({
                'heaviest_shape': f'tag {keys[4 * top]} (tile/mode code of spg_prof_tag), N={keys[4 * top + 1]}, K={keys[4 * top + 2]}',
                'heaviest_shape_launches_per_step': tcnt / nprof, 'heaviest_shape_avg_us': tms / tcnt * 1e3,
                'heaviest_shape_gflop_per_launch': tfl / tcnt / 1e9,
                'heaviest_shape_frac': tfl / (tms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS})
    if world == 1 and not args.no_extras:
        # a sustained repeat of the very same step: >= 2 s of back-to-back steps (the timed region above is K steps = tens of ms)
        torch.cuda.synchronize()
        n_sus, t0 = 0, time.perf_counter()
        while True:
            for _ in range(100):
                step()
            n_sus += 100
            torch.cuda.synchronize()
            if time.perf_counter() - t0 >= 2.0:
                break
        sus = (time.perf_counter() - t0) / n_sus
        result['sustained'] = {'seconds': time.perf_counter() - t0, 'steps': n_sus, 'ms_per_step': sus * 1e3, 'superpoints_per_s': n_sp_step / sus}
        log(f'sustained: {n_sus} steps, {sus * 1e3:.3f} ms/step')
        result['self_check'] = self_check('timed region and the sustained repeat')
        if 'roofline' in result:
            result['roofline']['sustained_ms_per_step'] = sus * 1e3
            result['roofline']['sustained_superpoints_per_s'] = n_sp_step / sus
    if world == 1 and not args.no_extras:
        # how long the HOST needs to enqueue one step (GPU idle at the start: no back-pressure from a full queue)
        hs = []
        for _ in range(12):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            step()
            hs.append(time.perf_counter() - t1)
        torch.cuda.synchronize()
        result['host_step_enqueue_ms'] = sorted(hs)[len(hs) // 2] * 1e3
        if 'roofline' in result:
            result['roofline']['host_step_enqueue_ms'] = result['host_step_enqueue_ms']
        log(f'host enqueue of one step: {result["host_step_enqueue_ms"]:.3f} ms')
    if world == 1 and not args.no_trainer_window:
        result['trainer_window'] = trainer_window(args, dev, model, embedder, arena, seeds, n_classes, log, fstep=fstep)
    if world == 1 and args.infer_line:
        result['inference'] = inference_line(args, dev, model, flag, clouds_d, diam_d, GIs)
        log(f'inference: {result["inference"]["ms"]:.3f} ms per forward, {result["inference"]["superpoints_per_s"]:.0f} superpoints/s, {result["inference"]["roofline"]["frac"]:.3f} of the fp32-MFMA peak')
    if world > 1:
        dist.barrier()
    if rank == 0:
        if world == 1 and not args.no_forward_only:
            model.ecc.set_info(GIs, 1)
            result['forward_only'] = forward_only(dev, flag, clouds_d, diam_d, GIs, args.n_feat)
        if world == 1 and not args.no_cpu_baseline:
            result['cpu_baseline'] = cpu_baseline(args.model_config, scenes, state0, n_feat=args.n_feat)
        elif world > 1:      # (a reported baseline of rank 0 at N = 1 only; say so instead of leaving the key out)
            result['cpu_baseline'] = {'skipped': 'n_gpus>1: the CPU baseline is timed by the N=1 run only'}
            if 'roofline' in result and result['roofline'].get('traffic_source') != 'live':
                result['roofline']['traffic_note'] = 'n_gpus>1: no live PMC passes; HBM bytes come from the N=1 run / the committed file'
        else:
            result['cpu_baseline'] = {'skipped': '--no-cpu-baseline'}
        if world == 1 and not args.no_extras:
            ow = other_workloads(args, log)
            result['other_workloads'] = ow
            if 'roofline' in result:      # (the driver's parser keeps `roofline` / `config` / `cpu_baseline` and only the NAMES of other keys)
                result['roofline']['other_workloads'] = ow
        print(json.dumps(result), flush=True)
    faulthandler.cancel_dump_traceback_later()
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
