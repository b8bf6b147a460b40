"""Training / evaluation CLI of the superpoint-graph network on the HIP kernels: the command line, checkpoint format,
random streams and printed scores of the reference's `learning/main.py` (file:line cited per function), so that a user of

    python learning/main.py --dataset s3dis --S3DIS_PATH ... --cvfold 5 --epochs 350 ...      (reference)

runs the same experiment with

    python -m superpoint_graph_amd.learning.main --dataset s3dis --S3DIS_PATH ... --cvfold 5 --epochs 350 ...

What is different underneath: the model is built from this package's modules (libspg_hip.so), the superpoint clouds are
built on the GPU from scenes resident in HBM (`--loader_device`), the evaluation accounting runs on the device, and with
`--fused_optim 1` the clamp + Adam update is one launch over a flat parameter arena.  There is no CPU execution path:
`--cuda 0` raises."""
import argparse
import ast
import functools
import json
import logging
import math
import os
import random
import sys
import time
from collections import defaultdict

import numpy as np
import torch
import torch.nn as nn
import torch.optim as optim
from torch.optim.lr_scheduler import MultiStepLR

from .. import ops
from . import datasets, graphnet, meters, metrics, pointnet, spg


def build_parser():
    """Flags of learning/main.py:41-120, same names / defaults / help, plus the three HIP-specific ones at the end."""
    p = argparse.ArgumentParser(description='Large-scale Point Cloud Semantic Segmentation with Superpoint Graphs (MI355X / HIP)')
    a = p.add_argument
    # optimisation
    a('--wd', default=0, type=float, help='Weight decay')
    a('--lr', default=1e-2, type=float, help='Initial learning rate')
    a('--lr_decay', default=0.7, type=float, help='Multiplicative factor used on learning rate at `lr_steps`')
    a('--lr_steps', default='[]', help='List of epochs where the learning rate is decreased by `lr_decay`')
    a('--momentum', default=0.9, type=float, help='Momentum')
    a('--epochs', default=10, type=int, help='Number of epochs to train. If <=0, only testing will be done.')
    a('--batch_size', default=2, type=int, help='Batch size')
    a('--optim', default='adam', help='Optimizer: sgd|adam')
    a('--grad_clip', default=1, type=float, help='Element-wise clipping of gradient. If 0, does not clip')
    a('--loss_weights', default='none', help='[none, proportional, sqrt] how to weight the loss function')
    # learning process
    a('--cuda', default=1, type=int, help='Bool, use cuda')
    a('--nworkers', default=0, type=int, help='Num subprocesses to use for data loading. 0 means that the data will be loaded in the main process')
    a('--test_nth_epoch', default=1, type=int, help='Test each n-th epoch during training')
    a('--save_nth_epoch', default=1, type=int, help='Save model each n-th epoch during training')
    a('--test_multisamp_n', default=10, type=int, help='Average logits obtained over runs with different seeds')
    # dataset
    a('--dataset', default='sema3d', help='Dataset name: sema3d|s3dis|vkitti|custom_dataset|<registered name>')
    a('--cvfold', default=0, type=int, help='Fold left-out for testing in leave-one-out setting (S3DIS)')
    a('--odir', default='results', help='Directory to store results')
    a('--resume', default='', help='Loads a previously saved model.')
    a('--db_train_name', default='train')
    a('--db_test_name', default='test')
    a('--use_val_set', type=int, default=0)
    a('--SEMA3D_PATH', default='datasets/semantic3d')
    a('--S3DIS_PATH', default='datasets/s3dis')
    a('--VKITTI_PATH', default='datasets/vkitti')
    a('--CUSTOM_SET_PATH', default='datasets/custom_set')
    a('--use_pyg', default=0, type=int, help='Wether to use Pytorch Geometric for graph convolutions (not available on the HIP path)')
    # model
    a('--model_config', default='gru_10,f_8', help='Sequence of layers, see graphnet.py: rectype_repeats_mv_layernorm_ingate_concat ...')
    a('--seed', default=1, type=int, help='Seed for random initialisation')
    a('--edge_attribs', default='delta_avg,delta_std,nlength/ld,surface/ld,volume/ld,size/ld,xyz/d', help='Edge attribute definition, see spg_edge_features() in spg.py for definitions.')
    # point clouds
    a('--pc_attribs', default='xyzrgbelpsvXYZ', help='Point attributes fed to PointNets, if empty then all possible.')
    a('--pc_augm_scale', default=0, type=float, help='Training augmentation: Uniformly random scaling in [1/scale, scale]')
    a('--pc_augm_rot', default=1, type=int, help='Training augmentation: Bool, random rotation around z-axis')
    a('--pc_augm_mirror_prob', default=0, type=float, help='Training augmentation: Probability of mirroring about x or y axes')
    a('--pc_augm_jitter', default=1, type=int, help='Training augmentation: Bool, Gaussian jittering of all attributes')
    a('--pc_xyznormalize', default=1, type=int, help='Bool, normalize xyz into unit ball, i.e. in [-0.5,0.5]')
    # filter generating network
    a('--fnet_widths', default='[32,128,64]', help='List of width of hidden filter gen net layers')
    a('--fnet_llbias', default=0, type=int, help='Bool, use bias in the last layer in filter gen net')
    a('--fnet_orthoinit', default=1, type=int, help='Bool, use orthogonal weight initialization for filter gen net.')
    a('--fnet_bnidx', default=2, type=int, help='Layer index to insert batchnorm to. -1=do not insert.')
    a('--edge_mem_limit', default=30000, type=int, help='Accepted for compatibility (the HIP kernels do not shard edges)')
    # superpoint graph
    a('--spg_attribs01', default=1, type=int, help='Bool, normalize edge features to 0 mean 1 deviation')
    a('--spg_augm_nneigh', default=100, type=int, help='Number of neighborhoods to sample in SPG')
    a('--spg_augm_order', default=3, type=int, help='Order of neighborhoods to sample in SPG')
    a('--spg_augm_hardcutoff', default=512, type=int, help='Maximum number of superpoints larger than args.ptn_minpts to sample in SPG')
    a('--spg_superedge_cutoff', default=-1, type=float, help='Artificially constrained maximum length of superedge, -1=do not constrain')
    # PointNet
    a('--ptn_minpts', default=40, type=int, help='Minimum number of points in a superpoint for computing its embedding.')
    a('--ptn_npts', default=128, type=int, help='Number of input points for PointNet.')
    a('--ptn_widths', default='[[64,64,128,128,256], [256,64,32]]', help='PointNet widths')
    a('--ptn_widths_stn', default='[[64,64,128], [128,64]]', help='PointNet\'s Transformer widths')
    a('--ptn_nfeat_stn', default=11, type=int, help='PointNet\'s Transformer number of input features')
    a('--ptn_prelast_do', default=0, type=float)
    a('--ptn_mem_monger', default=1, type=int, help='Kept for compatibility: activations stay in HBM, nothing is recomputed')
    a('--sp_decoder_config', default='[]', type=str, help='Size of the decoder : sp_embedding -> sp_class.')
    # HIP path
    a('--loader_device', default=1, type=int, help='Bool, build the superpoint clouds on the GPU from scenes resident in HBM')
    a('--batch_device', default=1, type=int,
      help='Bool, build the batched graph (edge ordering by target, edge-feature reordering, CSR) on the GPU in eccpc_collate; '
           'needs collation in the main process, i.e. it is only active together with --loader_device 1')
    a('--batch_stream', default=1, type=int,
      help='Bool, run the device half of the loader (cloud kernels, graph construction, uploads) on a second HIP stream so that it '
           'overlaps the training step in flight (superpoint_graph_amd/learning/prefetch.py); active with --batch_device 1')
    a('--gemm_precision', default='f32', choices=['f32', 'bf16x3', 'bf16'],
      help="Arithmetic of the wide PointNet GEMMs: f32 = fp32 MFMA (the reference's arithmetic, default); bf16x3 = split-bf16 "
           "products (~2^-16 per product, fp32 accumulate); bf16 = bf16 operands.  Tolerances: tests/test_gpu_precision.py")
    a('--loader_rng', default='host', choices=['host', 'device'],
      help="Random streams of the cloud loader: 'host' = numpy / python streams in the reference's order (seeded runs reproduce "
           "the reference's clouds), 'device' = counter-based generator on the GPU (no per-superpoint host loop)")
    a('--fused_optim', default=1, type=int, help='Bool, clamp + Adam as one launch over a flat parameter arena (adam only)')
    a('--fused_step', default=1, type=int,
      help='Bool, forward + backward of a training step as ONE call into the library (superpoint_graph_amd/fused.py: same kernels and '
           'results as the module-level path; the filter network and the RNN-ECC parameter gradients travel next to PointNet\'s '
           'launches).  Active with --fused_optim 1 for the standard model (gru_R / lstm_R followed by f_K); otherwise the modules run')
    a('--max_train_iters', default=0, type=int, help='Stop every training epoch after this many batches (0 = whole epoch)')
    a('--ecc_check_every', default=25, type=int,
      help='Fail-safe of the one-launch RNN-ECC recurrences: their bounded waits raise a sticky device word when a neighbour state did '
           'not arrive in time (GPU shared with another job, profiler); the clamp + Adam launch reads it and WITHHOLDS its update, so '
           'wrong gradients never reach the parameters.  Every this many steps the host reads the word back (16 bytes, synchronises '
           'the device), switches to the per-iteration kernels if it is set, corrects Adam\'s step count and repeats the batch at '
           'hand.  0 = only at the end of an epoch / before a checkpoint (ops.check_persistent_ecc)')
    # data parallel (one process per GPU: `python -m torch.distributed.run --nproc-per-node N -m superpoint_graph_amd.learning.main ...`)
    a('--sync_bn', default=0, type=int, help='Data parallel: BatchNorm statistics over the scenes of ALL ranks (= the single-process '
      'batch of the reference); 0 = per-rank statistics')
    a('--dist_backend', default='', help="torch.distributed backend (default: nccl = RCCL; 'gloo' for tests on one GPU)")
    a('--dist_device', default=-1, type=int, help='GPU of this rank (default: LOCAL_RANK)')
    a('--dp_replicate_loader', default=0, type=int,
      help='Data parallel: 1 = every rank runs the loader for the WHOLE batch and keeps its shard, so the random streams of the '
           'sub-sampling / augmentation are consumed exactly as in a single process (bit-reproducible against it, N-fold loader '
           'work); 0 = every rank loads only its own scenes')
    return p


def parse_args(argv=None):
    args = build_parser().parse_args(argv)
    args.start_epoch = 0
    for k in ('lr_steps', 'fnet_widths', 'ptn_widths', 'sp_decoder_config', 'ptn_widths_stn'):
        setattr(args, k, ast.literal_eval(getattr(args, k)))
    return args


def set_seed(seed, cuda=True):
    """Sets the seeds of all frameworks (learning/main.py:439-445)."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if cuda:
        torch.cuda.manual_seed(seed)
    from . import spg
    spg._device_rng_step[0] = 0            # `--loader_rng device`: the call counter is part of the stream key


def filter_valid(output, target, other=None):
    """Removes predictions for nodes without ground truth (learning/main.py:447-452)."""
    idx = target != -100
    if other is not None:
        return output[idx, :], target[idx], other[idx, ...]
    return output[idx, :], target[idx]


def _labels_on_device(targets):
    """(majority label, per-class point counts) of a batch on the device (learning/main.py:202-205): uploaded by the device collate in
    the batch's one staging copy (learning/spg.py: eccpc_collate(device_batch=True)), or here."""
    staged = getattr(targets, '_spg_labels_dev', None)
    if staged is not None:
        return staged
    return tuple(ops.upload_packed([targets[:, 0].contiguous(), targets[:, 2:].contiguous()]))


def _log_bounded(log, entry, keep=2048):
    """Per-batch records for tests and tools: 4-byte device scalars (clones -- a view would keep the loss kernel's whole
    [N+2] buffer alive), the oldest half dropped beyond `keep` entries so that a 350-epoch run does not grow without bound."""
    log.append(entry)
    if len(log) > keep:
        del log[:keep // 2]


def meter_value(meter):
    return meter.value()[0] if meter.n > 0 else 0


def create_model(args, dbinfo):
    """ecc first, then ptn -- the order fixes the parameter initialisation under a seed (learning/main.py:414-431)."""
    if 'use_pyg' not in args:
        args.use_pyg = 0
    model = nn.Module()
    nfeat = args.ptn_widths[1][-1]
    model.ecc = graphnet.GraphNetwork(args.model_config, nfeat, [dbinfo['edge_feats']] + args.fnet_widths, args.fnet_orthoinit,
                                      args.fnet_llbias, args.fnet_bnidx, args.edge_mem_limit, use_pyg=args.use_pyg, cuda=args.cuda)
    model.ptn = pointnet.PointNet(args.ptn_widths[0], args.ptn_widths[1], args.ptn_widths_stn[0], args.ptn_widths_stn[1],
                                  dbinfo['node_feats'], args.ptn_nfeat_stn, prelast_do=args.ptn_prelast_do)
    print('Total number of parameters: {}'.format(sum([p.numel() for p in model.parameters()])))
    print(model)
    if args.cuda:
        model.cuda()
    return model


def create_optimizer(args, model):
    if args.optim == 'sgd':
        return optim.SGD(model.parameters(), lr=args.lr, momentum=args.momentum, weight_decay=args.wd)
    if args.optim == 'adam':
        return optim.Adam(model.parameters(), lr=args.lr, weight_decay=args.wd)
    raise NotImplementedError(args.optim)


_STALE_KEYS = ('ecc.0._cell.inh.running_mean', 'ecc.0._cell.inh.running_var', 'ecc.0._cell.ini.running_mean', 'ecc.0._cell.ini.running_var')


def resume(args, dbinfo):
    """Loads model and optimizer state from a checkpoint written by this CLI or by the reference (learning/main.py:390-412)."""
    print("=> loading checkpoint '{}'".format(args.resume))
    checkpoint = torch.load(args.resume, weights_only=False)
    checkpoint['args'].model_config = args.model_config
    checkpoint['args'].cuda = args.cuda
    model = create_model(checkpoint['args'], dbinfo)
    optimizer = create_optimizer(args, model)
    model.load_state_dict({k: v for k, v in checkpoint['state_dict'].items() if k not in _STALE_KEYS})
    if 'optimizer' in checkpoint:
        optimizer.load_state_dict(checkpoint['optimizer'])
    for group in optimizer.param_groups:
        group['initial_lr'] = args.lr
    args.start_epoch = checkpoint['epoch']
    try:
        with open(os.path.join(os.path.dirname(args.resume), 'trainlog.json')) as f:
            stats = json.loads(f.read())
    except Exception:
        stats = []
    return model, optimizer, stats


class _ShardedBatches(torch.utils.data.Sampler):
    """Batch sampler of one data-parallel rank: the epoch's permutation is drawn from a generator seeded by (seed, epoch) --
    the same on every rank --, cut into global batches of `batch_size` scenes (incomplete last one dropped, as the
    reference's drop_last=True), and the rank keeps its contiguous shard of every batch (dist.shard_scenes)."""

    def __init__(self, n, batch_size, rank, world, seed, shuffle, costs=None):
        self.n, self.bs, self.rank, self.world, self.seed, self.shuffle, self.epoch = n, batch_size, rank, world, seed, shuffle, 0
        # superpoints per scene (if the dataset knows them): the scenes of a global batch are then dealt to the ranks by size
        # (dist.balanced_shards) instead of in contiguous blocks -- a step lasts as long as its slowest rank
        self.costs = None if costs is None or len(costs) != n else [float(c) for c in costs]

    def set_epoch(self, epoch):
        self.epoch = epoch

    def _batches(self):
        from ..dist import shard_scenes
        if self.shuffle:
            order = torch.randperm(self.n, generator=torch.Generator().manual_seed(1000003 * self.seed + self.epoch)).tolist()
        else:
            order = list(range(self.n))
        full = self.n // self.bs if self.shuffle else (self.n + self.bs - 1) // self.bs
        for b in range(full):
            chunk = order[b * self.bs:(b + 1) * self.bs]
            if self.costs is not None:
                from ..dist import balanced_shards
                yield [chunk[i] for i in balanced_shards([self.costs[j] for j in chunk], self.world)[self.rank]]
            else:
                yield [chunk[i] for i in shard_scenes(len(chunk), self.rank, self.world)]

    def __iter__(self):
        return self._batches()

    def __len__(self):
        return self.n // self.bs if self.shuffle else (self.n + self.bs - 1) // self.bs


class Session:
    """One run of the CLI: model, optimiser, datasets and the three loops of learning/main.py:176-311.  Data parallel
    (WORLD_SIZE > 1, one process per GPU): every global batch of `--batch_size` scenes is sharded over the ranks, the flat
    gradient arena is summed by ONE all-reduce per step with the loss weights riding along (FlatParameters.allreduce_sums; the
    normalisation happens inside the clamp + Adam launch), evaluation scenes are dealt round-robin and the confusion matrices /
    loss sums are all-reduced, rank 0 writes the files."""

    def __init__(self, args, dbinfo, create_dataset, model, optimizer, stats):
        self.args, self.dbinfo, self.create_dataset = args, dbinfo, create_dataset
        self.model, self.optimizer, self.stats = model, optimizer, stats
        self.rank, self.world = getattr(args, 'rank', 0), getattr(args, 'world', 1)
        self.dp = self.world > 1
        self.train_dataset, self.test_dataset, self.valid_dataset, self.scaler = create_dataset(args)
        self.log('Train dataset: %i elements - Test dataset: %i elements - Validation dataset: %i elements' %
                 (len(self.train_dataset), len(self.test_dataset), len(self.valid_dataset)))
        if self.dp:
            if not (args.fused_optim and args.optim == 'adam'):
                raise NotImplementedError('data parallel training uses the flat gradient arena: --fused_optim 1 --optim adam')
            if args.batch_size % self.world != 0:
                raise ValueError(f'--batch_size {args.batch_size} must be a multiple of the number of ranks ({self.world}): scenes are the shards')
        self._epoch = args.start_epoch
        self.embedder = pointnet.CloudEmbedder(args)
        self.scheduler = MultiStepLR(optimizer, milestones=args.lr_steps, gamma=args.lr_decay, last_epoch=args.start_epoch - 1)
        self.arena = None
        if args.fused_optim and args.optim == 'adam':
            from ..flat import FlatParameters
            # parameters / gradients become views of one flat buffer each; zero_grad() launches nothing (the backward kernels
            # overwrite every gradient) and the BatchNorm batch counters live on the host
            self.arena = FlatParameters(model, lazy_zero=True, host_counters=True)
            self.arena.attach_optimizer(optimizer)    # the Adam moments live in `optimizer.state` (checkpoint format kept)
        self.fused = None
        # (with synchronised BatchNorm only in its slot form -- the all-reduces of the fixed-point statistics are issued by the library
        #  itself between producer and consumer launches; the finalize-based form of rounds 1-4 needs the module path)
        from .. import dist as _spd
        if self.arena is not None and getattr(args, 'fused_step', 1) and args.cuda and (not getattr(args, 'sync_bn', 0) or _spd.sync_bn_mode() in (None, 'slots')):
            from .. import fused
            if fused.supports(model):
                self.fused = fused.FusedStep(model, self.arena, class_weights=dbinfo['class_weights'],
                                             reduction='sum' if self.dp else 'mean', ptn_mem_monger=bool(args.ptn_mem_monger))
        self.iter_log = []                            # (loss, trainer ms) per training batch, for tests and tools
        self.eval_log = []                            # loss per evaluation batch

    def log(self, *a):
        if self.rank == 0:
            print(*a)

    def _nworkers(self):
        """--nworkers as given, except with the device loader: `spg.loader` then launches kernels inside `__getitem__`, which
        cannot run in forked DataLoader workers (the CUDA/HIP context of the parent does not survive a fork) and needs no
        workers anyway (one launch builds all clouds of a graph).  The reference's documented commands pass --nworkers 2."""
        a = self.args
        if a.nworkers > 0 and a.cuda and a.loader_device:
            if not getattr(self, '_warned_workers', False):
                logging.warning('--nworkers %d ignored: --loader_device 1 builds the superpoint clouds on the GPU in the main process '
                                '(use --loader_device 0 for the host loader with worker processes)', a.nworkers)
                self._warned_workers = True
            return 0
        return a.nworkers

    def _collate(self):
        a = self.args
        if a.cuda and a.loader_device and getattr(a, 'batch_device', 1) and self._nworkers() == 0:
            return functools.partial(spg.eccpc_collate, device_batch=True)      # GraphConvInfo.set_batch_device
        return spg.eccpc_collate

    def _loader(self, dataset, train):
        loader = self._host_loader(dataset, train)
        a = self.args
        if a.cuda and getattr(a, 'batch_stream', 1) and self._collate() is not spg.eccpc_collate:
            from .prefetch import SideStreamBatches
            return SideStreamBatches(loader)        # the collate's kernels / copies overlap the step in flight
        return loader

    def _host_loader(self, dataset, train):
        a = self.args
        collate, nw = self._collate(), self._nworkers()
        if self.dp and a.dp_replicate_loader:
            # the single-process loaders, unchanged (same sampler, same consumption of the random streams); the collate keeps
            # this rank's shard of the batch.  Evaluation batches hold ONE scene: rank b % world owns batch b, the others skip it.
            from ..dist import shard_scenes
            counter = [0]

            def shard(batch):
                if train:
                    return collate([batch[i] for i in shard_scenes(len(batch), self.rank, self.world)])
                b = counter[0]
                counter[0] += 1
                return collate(batch) if b % self.world == self.rank else None
            if train:
                return torch.utils.data.DataLoader(dataset, batch_size=a.batch_size, collate_fn=shard, num_workers=nw, shuffle=True, drop_last=True)
            return torch.utils.data.DataLoader(dataset, batch_size=1, collate_fn=shard, num_workers=nw)
        if self.dp:
            if train:
                sampler = _ShardedBatches(len(dataset), a.batch_size, self.rank, self.world, a.seed, True, costs=self._scene_costs(dataset))
                sampler.set_epoch(self._epoch)
                return torch.utils.data.DataLoader(dataset, batch_sampler=sampler, collate_fn=collate, num_workers=nw)
            own = list(range(self.rank, len(dataset), self.world))              # evaluation: scenes dealt round-robin
            return torch.utils.data.DataLoader(torch.utils.data.Subset(dataset, own), batch_size=1, collate_fn=collate, num_workers=nw)
        if train:
            return torch.utils.data.DataLoader(dataset, batch_size=a.batch_size, collate_fn=collate, num_workers=nw,
                                               shuffle=True, drop_last=True)
        return torch.utils.data.DataLoader(dataset, batch_size=1, collate_fn=collate, num_workers=nw)

    def _scene_costs(self, dataset):
        """Superpoints per training scene as the loader will deliver it (the graphs are read up front, learning/spg.py:67-106;
        `--spg_augm_hardcutoff` caps the sampled sub-graph), or None when the dataset's elements are not graphs."""
        try:
            cut = getattr(self.args, 'spg_augm_hardcutoff', 0)
            sizes = [int(g.vcount()) for g in dataset.list]
        except (AttributeError, TypeError):      # a dataset whose elements are not graphs (or that keeps no list): contiguous shards
            self.log('data parallel: the training set exposes no per-scene sizes -- contiguous scene shards (no balancing by size)')
            return None
        self.log(f'data parallel: scenes of a global batch are dealt to the ranks by size ({len(sizes)} training scenes, '
                 f'{min(sizes)}-{max(sizes)} superpoints each); --dp_replicate_loader 1 keeps the single-process order instead')
        return [min(v, cut) if cut and cut > 0 else v for v in sizes]

    def _forward(self, targets, GIs, clouds_data):
        self.model.ecc.set_info(GIs, self.args.cuda)
        # through the staging ring (ops.upload): a pageable `.cuda()` would stall the host until the previous step has drained
        label_mode, label_vec = _labels_on_device(targets) if self.args.cuda else (targets[:, 0].contiguous(), targets[:, 2:].contiguous())
        embeddings = self.embedder.run(self.model, *clouds_data)
        outputs = self.model.ecc(embeddings)
        return outputs, label_mode, label_vec

    # ---- learning/main.py:176-226 ----
    def train(self):
        """Trains for one epoch -> (accuracy, mean loss, overall accuracy, mean IoU) over the training batches."""
        a = self.args
        self.model.train()
        loss_meter, acc_meter = meters.AverageValueMeter(), meters.ClassErrorMeter(accuracy=True)
        cm = metrics.ConfusionMatrix(self.dbinfo['classes'])

        def step_batch(targets, GIs, clouds_data):
            """zero_grad -> forward -> loss -> backward -> bw_hook -> [all-reduce] -> clamp + Adam (learning/main.py:199-213) of ONE batch
            -> (loss of the whole batch, outputs, label_vec, label_mode), everything still on the device"""
            if self.arena is not None:
                self.arena.zero_grad()
            else:
                self.optimizer.zero_grad()
            if self.fused is not None and len(clouds_data[2]) > 1:
                # forward + backward + bw_hook as one library call (learning/main.py:199-208); gradients land in the arena
                self.model.ecc.set_info(GIs, a.cuda)
                label_mode, label_vec = _labels_on_device(targets)
                loss, outputs = self.fused(clouds_data[1], clouds_data[2], clouds_data[3], GIs[0], label_mode)
                if self.dp:
                    self.arena.allreduce_sums(self.fused.normaliser, loss)
                    loss = (self.arena.loss_sum / self.arena.normaliser).reshape(())
                    self.arena.optimizer_step(grad_clip=a.grad_clip, grad_div=self.arena.normaliser)
                else:
                    self.arena.optimizer_step(grad_clip=a.grad_clip)
                return loss, outputs, label_vec, label_mode
            outputs, label_mode, label_vec = self._forward(targets, GIs, clouds_data)
            if self.dp:
                # data parallel: back-propagate the SUM-reduced loss (gradients carry this rank's loss weight w_r), ONE all-reduce of
                # [gradients | w_r | loss_r], division by sum_r w_r inside the clamp + Adam launch -- no host synchronisation
                loss_sum, w = ops.cross_entropy(outputs, label_mode, weight=self.dbinfo['class_weights'], reduction='sum',
                                                return_normaliser=True)
                loss_sum.backward(self.arena.one)
                self.embedder.bw_hook()
                self.arena.allreduce_sums(w, loss_sum)
                loss = (self.arena.loss_sum / self.arena.normaliser).reshape(())       # the loss of the WHOLE batch (main.py:205)
                self.arena.optimizer_step(grad_clip=a.grad_clip, grad_div=self.arena.normaliser)
                return loss, outputs.detach(), label_vec, label_mode
            loss = ops.cross_entropy(outputs, label_mode, weight=self.dbinfo['class_weights'])      # main.py:205
            loss.backward(self.arena.one if self.arena is not None else None)      # cached seed: no fill launch for ones_like(loss)
            self.embedder.bw_hook()
            if self.arena is not None:
                self.arena.optimizer_step(grad_clip=a.grad_clip)          # clamp (main.py:210-212) + Adam in one launch
            else:
                if a.grad_clip > 0:
                    for p in self.model.parameters():
                        p.grad.data.clamp_(-a.grad_clip, a.grad_clip)
                self.optimizer.step()
            return loss.detach(), outputs.detach(), label_vec, label_mode

        t0 = time.time()
        guard_every = int(getattr(a, 'ecc_check_every', 0) or 0) if (a.cuda and self.arena is not None) else 0
        for bidx, (targets, GIs, clouds_data) in enumerate(self._loader(self.train_dataset, True)):
            t_loader = 1000 * (time.time() - t0)
            t0 = time.time()
            loss, outputs, label_vec, label_mode = step_batch(targets, GIs, clouds_data)
            if guard_every and (bidx + 1) % guard_every == 0:
                # per-step fail-safe, host half (the device half: the clamp + Adam launch withholds its update while the one-launch
                # RNN-ECC recurrences' time-out word is set).  One 16-byte read-back every `guard_every` steps.
                withheld = ops.recover_persistent_ecc(self.arena)
                if withheld:
                    logging.warning('persistent RNN-ECC time-out: %d optimiser update(s) were withheld on the device (parameters untouched); '
                                    'switched to the per-iteration kernels; repeating the current batch, %d earlier batch(es) of this window '
                                    'are dropped', withheld, max(withheld - 1, 0))
                    loss, outputs, label_vec, label_mode = step_batch(targets, GIs, clouds_data)
            t_trainer = 1000 * (time.time() - t0)
            loss_meter.add(loss)
            cm.count_predicted_batch_device(label_vec, outputs, label_mode)   # filter_valid + argmax + counts, on the GPU
            _log_bounded(self.iter_log, (loss.clone(), t_trainer))
            logging.debug('Batch loader time %f ms, trainer time %f ms.', t_loader, t_trainer)
            t0 = time.time()
            if a.max_train_iters and bidx + 1 >= a.max_train_iters:
                break
        cm.allreduce()
        acc_meter.add_counts(*cm.accuracy_counts())
        if a.cuda:
            ops.check_persistent_ecc('this training epoch')      # (the meters above already synchronised the device)
        return acc_meter.value()[0], loss_meter.value()[0], cm.get_overall_accuracy(), cm.get_average_intersection_union()

    # ---- learning/main.py:229-264 ----
    def eval(self, is_valid=False):
        """Evaluates the model on the test (or validation) set."""
        self.model.eval()
        loss_meter, acc_meter = meters.AverageValueMeter(), meters.ClassErrorMeter(accuracy=True)
        cm = metrics.ConfusionMatrix(self.dbinfo['classes'])
        lsum = torch.zeros(2, dtype=torch.float64, device='cuda')       # data parallel: (sum of the batch losses, batches)
        for item in self._loader(self.valid_dataset if is_valid else self.test_dataset, False):
            if item is None:                          # --dp_replicate_loader: a scene of another rank
                continue
            targets, GIs, clouds_data = item
            with torch.no_grad():
                outputs, label_mode, label_vec = self._forward(targets, GIs, clouds_data)
                loss = ops.cross_entropy(outputs, label_mode, weight=self.dbinfo['class_weights'])
            loss_meter.add(loss)
            lsum += torch.stack([loss.double(), torch.ones((), dtype=torch.float64, device=loss.device)])
            _log_bounded(self.eval_log, loss.clone())
            cm.count_predicted_batch_device(label_vec, outputs, label_mode)
        cm.allreduce()
        acc_meter.add_counts(*cm.accuracy_counts())
        mean_loss = loss_meter.value()[0] if loss_meter.n > 0 else 0.0
        if self.dp:                                   # mean of the per-scene losses over ALL ranks' scenes
            import torch.distributed as tdist
            host = lsum.cpu() if tdist.get_backend() == 'gloo' else lsum
            tdist.all_reduce(host)
            mean_loss = float(host[0] / host[1].clamp(min=1))
        return (meter_value(acc_meter), mean_loss, cm.get_overall_accuracy(), cm.get_average_intersection_union(),
                cm.get_mean_class_accuracy())

    # ---- learning/main.py:267-311 ----
    def eval_final(self):
        """Multi-sample evaluation: the logits of `test_multisamp_n` differently seeded samplings of every scene are
        averaged (on the device, in sample order) before the arg-max; returns the predictions per scene as well."""
        a = self.args
        self.model.eval()
        acc_meter = meters.ClassErrorMeter(accuracy=True)
        cm = metrics.ConfusionMatrix(self.dbinfo['classes'])
        collected, labels, predictions = defaultdict(list), {}, {}
        for ss in range(a.test_multisamp_n):
            test_dataset_ss = self.create_dataset(a, ss)[1]
            for item in self._loader(test_dataset_ss, False):
                if item is None:
                    continue
                targets, GIs, clouds_data = item
                with torch.no_grad():
                    outputs, label_mode, label_vec = self._forward(targets, GIs, clouds_data)
                fname = clouds_data[0][0][:clouds_data[0][0].rfind('.')]
                collected[fname].append(outputs)
                labels.setdefault(fname, (label_mode, label_vec))
        for fname, outs in collected.items():
            label_mode, label_vec = labels[fname]
            logits = torch.stack(outs, 0) if a.test_multisamp_n > 1 else outs[0]
            predictions[fname] = cm.count_predicted_batch_device(label_vec, logits.contiguous(), label_mode).cpu().numpy()
        cm.allreduce()
        if self.dp:                                   # every rank predicted its own scenes: rank 0 writes them all
            import torch.distributed as tdist
            parts = [None] * self.world
            tdist.all_gather_object(parts, predictions)
            predictions = {k: v for part in parts for k, v in part.items()}
        acc_meter.add_counts(*cm.accuracy_counts())
        per_class_iou = {name: cm.get_intersection_union_per_class()[c] for c, name in self.dbinfo['inv_class_map'].items()}
        return (meter_value(acc_meter), cm.get_overall_accuracy(), cm.get_average_intersection_union(), per_class_iou, predictions,
                cm.get_mean_class_accuracy(), cm.confusion_matrix)

    def checkpoint(self, epoch, with_scaler=True):
        if self.args.cuda:
            ops.check_persistent_ecc('the steps behind this checkpoint')       # never persist parameters trained on corrupted ECC outputs
        if self.rank != 0:                  # replicas are identical after every step: rank 0 writes
            return
        state = {'epoch': epoch + 1, 'args': self.args, 'state_dict': self.model.state_dict(), 'optimizer': self.optimizer.state_dict()}
        if with_scaler:
            state['scaler'] = self.scaler
        cache = self.args.__dict__.pop('_device_point_cache', None)        # device buffers do not belong into the checkpoint
        try:
            torch.save(state, os.path.join(self.args.odir, 'model.pth.tar'))
        finally:
            if cache is not None:
                self.args._device_point_cache = cache

    # ---- learning/main.py:313-388 ----
    def run(self):
        a = self.args
        try:
            best_iou = self.stats[-1]['best_iou']
        except Exception:
            best_iou = 0
        epoch = a.start_epoch
        for epoch in range(a.start_epoch, a.epochs):
            self.log('Epoch {}/{} ({}):'.format(epoch, a.epochs, a.odir))
            self._epoch = epoch
            self.scheduler.step()
            acc, loss, oacc, avg_iou = self.train()
            self.log('-> Train Loss: %1.4f   Train accuracy: %3.2f%%' % (loss, acc))
            new_best = False
            if a.use_val_set:
                acc_val, loss_val, oacc_val, avg_iou_val, avg_acc_val = self.eval(True)
                self.log('-> Val Loss: %1.4f  Val accuracy: %3.2f%%  Val oAcc: %3.2f%%  Val IoU: %3.2f%%  best ioU: %3.2f%%' %
                      (loss_val, acc_val, 100 * oacc_val, 100 * avg_iou_val, 100 * max(best_iou, avg_iou_val)))
                if avg_iou_val > best_iou:
                    self.log('-> New best model achieved!')
                    best_iou, new_best = avg_iou_val, True
                    self.checkpoint(epoch)
            elif epoch % a.save_nth_epoch == 0 or epoch == a.epochs - 1:
                self.checkpoint(epoch)
            if (not a.use_val_set and (epoch + 1) % a.test_nth_epoch == 0) or (a.use_val_set and new_best and epoch > 5):
                acc_test, loss_test, oacc_test, avg_iou_test, avg_acc_test = self.eval(False)
                self.log('-> Test Loss: %1.4f  Test accuracy: %3.2f%%  Test oAcc: %3.2f%%  Test avgIoU: %3.2f%%' %
                      (loss_test, acc_test, 100 * oacc_test, 100 * avg_iou_test))
            else:
                acc_test, loss_test, oacc_test, avg_iou_test, avg_acc_test = 0, 0, 0, 0, 0
            self.stats.append({'epoch': epoch, 'acc': acc, 'loss': loss, 'oacc': oacc, 'avg_iou': avg_iou, 'acc_test': acc_test,
                               'oacc_test': oacc_test, 'avg_iou_test': avg_iou_test, 'avg_acc_test': avg_acc_test, 'best_iou': best_iou})
            if math.isnan(loss):
                break
            if self.rank == 0:
                with open(os.path.join(a.odir, 'trainlog.json'), 'w') as outfile:
                    json.dump(self.stats, outfile, indent=4)
        if a.use_val_set:
            if self.dp:
                import torch.distributed as tdist
                tdist.barrier()                       # rank 0 has written the best model
            a.resume = a.odir + '/model.pth.tar'
            self.model, self.optimizer, self.stats = resume(a, self.dbinfo)
            self.arena = None
            self.checkpoint(epoch, with_scaler=False)
        if a.test_multisamp_n > 0 and 'test' in a.db_test_name:
            acc_test, oacc_test, avg_iou_test, per_class_iou_test, predictions_test, avg_acc_test, confusion = self.eval_final()
            self.log('-> Multisample {}: Test accuracy: {}, \tTest oAcc: {}, \tTest avgIoU: {}, \tTest mAcc: {}'.format(
                a.test_multisamp_n, acc_test, oacc_test, avg_iou_test, avg_acc_test))
            if self.rank == 0:
                write_predictions(os.path.join(a.odir, 'predictions_' + a.db_test_name), predictions_test)
                with open(os.path.join(a.odir, 'scores_' + a.db_test_name + '.json'), 'w') as outfile:
                    json.dump([{'epoch': a.start_epoch, 'acc_test': acc_test, 'oacc_test': oacc_test, 'avg_iou_test': avg_iou_test,
                                'per_class_iou_test': per_class_iou_test, 'avg_acc_test': avg_acc_test}], outfile)
                np.save(os.path.join(a.odir, 'pointwise_cm.npy'), confusion)


def write_predictions(stem, predictions):
    """predictions_<db_test_name>.h5: one dataset of 0-based class ids per scene (learning/main.py:383-385; read by
    partition/visualize.py:74-77).  Without h5py the same arrays go to <stem>.npz."""
    try:
        import h5py
    except ImportError:
        np.savez(stem + '.npz', **predictions)
        return stem + '.npz'
    with h5py.File(stem + '.h5', 'w') as hf:
        for fname, o_cpu in predictions.items():
            hf.create_dataset(name=fname, data=o_cpu)
    return stem + '.h5'


def main(argv=None):
    args = parse_args(argv)
    if not args.cuda:
        raise RuntimeError('superpoint_graph_amd has no CPU execution path: run with --cuda 1 on a ROCm device')
    # data parallel: one process per GPU (torch.distributed.run sets RANK / LOCAL_RANK / WORLD_SIZE); a plain launch is rank 0 of 1
    from .. import dist as spd
    if args.dist_device >= 0:
        torch.cuda.set_device(args.dist_device)
    args.rank, local, args.world = spd.init_from_env(args.dist_backend or None)
    if args.world > 1:
        torch.cuda.set_device(args.dist_device if args.dist_device >= 0 else local)
        if args.sync_bn:
            spd.enable_sync_bn(torch.device('cuda', torch.cuda.current_device()))
    if args.rank == 0:
        print('Will save to ' + args.odir)
        os.makedirs(args.odir, exist_ok=True)
        with open(os.path.join(args.odir, 'cmdline.txt'), 'w') as f:
            f.write(' '.join(["'" + a + "'" if (len(a) == 0 or a[0] != '-') else a for a in (argv if argv is not None else sys.argv)]))
    set_seed(args.seed, args.cuda)
    from .. import _lib
    if _lib.lib().spg_tune(7, {'f32': 0, 'bf16': 1, 'bf16x3': 3}[args.gemm_precision]) < 0:
        raise RuntimeError('libspg_hip.so has no precision switch (spg_tune key 7)')
    logging.getLogger().setLevel(logging.INFO)
    get_info, create_dataset = datasets.provider(args.dataset)
    dbinfo = get_info(args)
    if args.resume != '':
        if args.resume == 'RESUME':
            args.resume = args.odir + '/model.pth.tar'
        model, optimizer, stats = resume(args, dbinfo)
    else:
        model = create_model(args, dbinfo)
        optimizer = create_optimizer(args, model)
        stats = []
    session = Session(args, dbinfo, create_dataset, model, optimizer, stats)
    try:
        session.run()
    finally:
        if args.world > 1 and args.sync_bn:
            spd.disable_sync_bn()
    return session


if __name__ == '__main__':
    main()
