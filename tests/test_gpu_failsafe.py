"""Per-step fail-safe of the one-launch RNN-ECC recurrences (round 6; include/spg_hip.h: spg_ecc_persistent_status).

The recurrences synchronise by dataflow with BOUNDED waits: a wave whose neighbour state does not arrive in time raises a sticky
device word and goes on with stale data -- the gradients of that step are wrong.  The fused clamp + Adam launch reads the word on
the device and WITHHOLDS its update while it is set (no host synchronisation), the host half (ops.recover_persistent_ecc) clears it,
switches to the per-iteration kernels, corrects Adam's step count, and the caller repeats the batch.  The reference's only guard on
this path is the NaN-loss check of learning/main.py:367.

A time-out is FORCED here with spg_tune key 20 (a spin bound of one sweep group)."""
import numpy as np
import pytest
import torch

from conftest import build_model

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _scene():
    from oracle import spg_oracle as O
    from superpoint_graph_amd import synth
    spec = O.ModelSpec()
    col = synth.collate_numpy([synth.scene(5, n_sp=1000, n_edges=5000)])
    idxn, degs, ef, _ = O.set_batch(col['edge_lists'], col['vcounts'], col['edge_feats'])
    batch = dict(clouds_flag=torch.from_numpy(col['clouds_flag']), clouds=torch.from_numpy(col['clouds']),
                 clouds_global=torch.from_numpy(col['clouds_global']), idxn=torch.from_numpy(idxn), degs=torch.from_numpy(degs),
                 edgefeats=torch.from_numpy(ef), label_mode=torch.from_numpy(col['targets'][:, 0].copy()))
    torch.manual_seed(1)
    ref = build_model(spec)
    with torch.no_grad():
        ref.ptn.stn.proj.weight.normal_(0, 0.02)
    return spec, batch, {k: v.clone() for k, v in ref.state_dict().items()}


def _setup(spec, state0):
    from superpoint_graph_amd.flat import FlatParameters
    from superpoint_graph_amd.fused import FusedStep
    model = build_model(spec, state0).to(DEV).train()
    arena = FlatParameters(model, lazy_zero=True, host_counters=True)
    return model, arena, FusedStep(model, arena, ptn_mem_monger=True)


def _step(model, arena, step, batch, lr=1e-3):
    from superpoint_graph_amd.learning import ecc
    gi = ecc.GraphConvInfo.from_buffers(batch['idxn'].clone(), batch['degs'].clone(), batch['edgefeats'].clone())
    model.ecc.set_info([gi], 1)
    arena.zero_grad()
    loss, _ = step(batch['clouds_flag'].to(DEV), batch['clouds'].to(DEV), batch['clouds_global'].to(DEV), gi, batch['label_mode'].to(DEV))
    arena.adam_step(lr=lr, grad_clip=1.0)
    return loss


def _params(model):
    # (BatchNorm running statistics are updated by the forward kernels whatever happens later: only what the optimiser owns)
    return {k: p.detach().clone() for k, p in model.named_parameters()}


def test_timed_out_step_leaves_parameters_and_moments_untouched_and_is_replayed(hip):
    from superpoint_graph_amd import ops
    spec, batch, state0 = _scene()
    old8, old20 = hip.spg_tune(8, 0), hip.spg_tune(20, 0)
    try:
        ops.persistent_ecc_status(clear=True)
        # reference run: one good step, then a second one on the per-iteration kernels (what the recovery switches to)
        model_r, arena_r, step_r = _setup(spec, state0)
        _step(model_r, arena_r, step_r, batch)
        after_one = _params(model_r)
        hip.spg_tune(8, 1)
        _step(model_r, arena_r, step_r, batch)
        after_two = _params(model_r)
        hip.spg_tune(8, 0)
        assert ops.persistent_ecc_status() == (0, 0)

        model, arena, step = _setup(spec, state0)
        _step(model, arena, step, batch)
        torch.cuda.synchronize()
        for k, v in _params(model).items():
            assert torch.equal(v, after_one[k]), k
        m0, v0, t0 = arena._m.clone(), arena._v.clone(), arena._t

        hip.spg_tune(20, 1)                      # every wait gives up after its first sweeps: the recurrence times out
        for _ in range(3):                       # the word is sticky: the two steps behind the failed one are withheld as well
            _step(model, arena, step, batch)
        hip.spg_tune(20, 0)
        torch.cuda.synchronize()
        errors, withheld = ops.persistent_ecc_status()
        assert errors > 0 and withheld == 3, (errors, withheld)
        for k, v in _params(model).items():
            assert torch.equal(v, after_one[k]), ('parameter touched by a withheld update', k)
        assert torch.equal(arena._m, m0) and torch.equal(arena._v, v0)
        assert arena._t == t0 + 3                # (the host counted the launches it issued ...)

        n = ops.recover_persistent_ecc(arena)
        assert n == 3 and arena._t == t0         # (... and takes the withheld ones back)
        assert ops.persistent_ecc_status() == (0, 0)
        assert hip.spg_tune(8, 1) == 1           # the process is on the per-iteration kernels now
        _step(model, arena, step, batch)         # the repeated batch
        torch.cuda.synchronize()
        assert ops.persistent_ecc_status() == (0, 0)
        for k, v in _params(model).items():
            assert torch.equal(v, after_two[k]), ('replayed step differs from the undisturbed run', k)
    finally:
        hip.spg_tune(8, old8); hip.spg_tune(20, old20)
        ops.persistent_ecc_status(clear=True)


def test_guard_switch_and_epoch_end_check(hip):
    """spg_tune key 21 switches the device half off (A/B only): the update then goes through; ops.check_persistent_ecc still raises."""
    from superpoint_graph_amd import ops
    spec, batch, state0 = _scene()
    old8, old20, old21 = hip.spg_tune(8, 0), hip.spg_tune(20, 0), hip.spg_tune(21, 0)
    try:
        ops.persistent_ecc_status(clear=True)
        model, arena, step = _setup(spec, state0)
        before = _params(model)
        hip.spg_tune(20, 1); hip.spg_tune(21, 1)
        _step(model, arena, step, batch)
        hip.spg_tune(20, 0); hip.spg_tune(21, 0)
        torch.cuda.synchronize()
        errors, withheld = ops.persistent_ecc_status()
        assert errors > 0 and withheld == 0
        assert any(not torch.equal(v, before[k]) for k, v in _params(model).items())
        with pytest.raises(RuntimeError, match='time-out'):
            ops.check_persistent_ecc('the test', group=False)
        assert ops.persistent_ecc_status() == (0, 0)
    finally:
        hip.spg_tune(8, old8); hip.spg_tune(20, old20); hip.spg_tune(21, old21)
        ops.persistent_ecc_status(clear=True)


def test_another_ranks_time_out_withholds_the_update_here(hip):
    """Data parallel: the time-out flag travels as a slot of the gradient all-reduce (FlatParameters.allreduce_sums), the clamp + Adam
    launch of every rank reads the SUM -- a rank whose own recurrence was fine withholds as well when another rank's was not (the
    summed gradients contain that rank's wrong ones), so the replicas stay identical.  Simulated at world size 1 by writing the
    summed flag a peer would have contributed."""
    from superpoint_graph_amd import ops
    from superpoint_graph_amd.learning import ecc
    spec, batch, state0 = _scene()
    ops.persistent_ecc_status(clear=True)
    model, arena, step = _setup(spec, state0)
    gi = ecc.GraphConvInfo.from_buffers(batch['idxn'].clone(), batch['degs'].clone(), batch['edgefeats'].clone())
    model.ecc.set_info([gi], 1)

    def one(flag_from_peer):
        arena.zero_grad()
        loss, _ = step(batch['clouds_flag'], batch['clouds'].to(DEV), batch['clouds_global'].to(DEV), gi, batch['label_mode'].to(DEV))
        arena.allreduce_sums(step.normaliser, loss)
        if flag_from_peer:
            arena._gbuf[arena.numel + 2] += 1.0            # what the all-reduce would have added
        arena.adam_step(lr=1e-3, grad_clip=1.0, grad_div=arena.normaliser)
        torch.cuda.synchronize()

    one(False)
    p1, m1 = _params(model), arena._m.clone()
    assert ops.persistent_ecc_status() == (0, 0)
    one(True)
    assert ops.persistent_ecc_status() == (0, 1)           # no local time-out, one update withheld
    for k, v in _params(model).items():
        assert torch.equal(v, p1[k]), k
    assert torch.equal(arena._m, m1)
    ops.persistent_ecc_status(clear=True)
    one(False)                                             # the flag is per exchange: the next step goes through
    assert any(not torch.equal(v, p1[k]) for k, v in _params(model).items())
    assert ops.persistent_ecc_status() == (0, 0)
