"""ctypes loader of libspg_hip.so (the C-ABI boundary declared in include/spg_hip.h).

The product path has NO fallback: if the library is missing or a symbol is absent, importing the
kernels raises.  torch must be imported before the library so that the HIP runtime bundled with
torch (libamdhip64) is the one the library binds to.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import torch  # noqa: F401  (must precede CDLL: see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, 'csrc')
# SPG_HIP_LIB: another build of the same library (tools/ab_builds.sh: A/B timing of two builds on ONE box); never set in production
LIB_PATH = os.environ.get('SPG_HIP_LIB') or os.path.join(CSRC, 'libspg_hip.so')

SPG_MAX_LAYERS = 8
SPG_MAX_PARTS = 64
c_void_pp = ctypes.POINTER(ctypes.c_void_p)


class PointNetCfg(ctypes.Structure):
    _fields_ = [('nfeat', ctypes.c_int), ('nfeat_stn', ctypes.c_int), ('nfeat_global', ctypes.c_int), ('npts', ctypes.c_int),
                ('n_stn_conv', ctypes.c_int), ('n_stn_fc', ctypes.c_int), ('n_conv', ctypes.c_int), ('n_fc', ctypes.c_int),
                ('stn_conv', ctypes.c_int * SPG_MAX_LAYERS), ('stn_fc', ctypes.c_int * SPG_MAX_LAYERS),
                ('conv', ctypes.c_int * SPG_MAX_LAYERS), ('fc', ctypes.c_int * SPG_MAX_LAYERS),
                ('last_ac', ctypes.c_int), ('bn_eps', ctypes.c_float), ('bn_momentum', ctypes.c_float)]


class EccRnnCfg(ctypes.Structure):
    _fields_ = [('nc', ctypes.c_int), ('nrepeats', ctypes.c_int), ('matrix', ctypes.c_int), ('layernorm', ctypes.c_int),
                ('ingate', ctypes.c_int), ('cat_all', ctypes.c_int), ('n_fnet', ctypes.c_int),
                ('fnet_widths', ctypes.c_int * (SPG_MAX_LAYERS + 1)), ('bnidx', ctypes.c_int), ('llbias', ctypes.c_int),
                ('bn_eps', ctypes.c_float), ('bn_momentum', ctypes.c_float), ('cell', ctypes.c_int),
                ('n_parts', ctypes.c_int), ('part_ptr', ctypes.c_int * (SPG_MAX_PARTS + 1))]


_i, _l, _p, _sz = ctypes.c_int, ctypes.c_long, ctypes.c_void_p, ctypes.c_size_t

SPG_EF_MAX_COLS = 32


class EdgeFeatureSpec(ctypes.Structure):
    _fields_ = [('data', ctypes.c_void_p), ('ld', ctypes.c_long), ('column', ctypes.c_int), ('kind', ctypes.c_int),
                ('is_f64', ctypes.c_int), ('pad_', ctypes.c_int)]


class StepArgs(ctypes.Structure):
    """spg_step_args of include/spg_hip.h (spg_train_step)."""
    _fields_ = [('ptn_cfg', ctypes.POINTER(PointNetCfg)), ('B', ctypes.c_int), ('bn_update_times', ctypes.c_int),
                ('clouds', ctypes.c_void_p), ('clouds_global', ctypes.c_void_p), ('ptn_params', c_void_pp), ('ptn_grads', c_void_pp),
                ('ptn_ws', ctypes.c_void_p), ('ptn_bwd_ws', ctypes.c_void_p), ('emb', ctypes.c_void_p), ('grad_emb', ctypes.c_void_p),
                ('N', ctypes.c_int), ('nf', ctypes.c_int), ('slot_of_row', ctypes.c_void_p), ('idx_valid', ctypes.c_void_p),
                ('desc', ctypes.c_void_p), ('grad_desc', ctypes.c_void_p),
                ('ecc_cfg', ctypes.POINTER(EccRnnCfg)), ('E', ctypes.c_int), ('graph_ws', ctypes.c_void_p), ('edgefeats', ctypes.c_void_p),
                ('ecc_params', c_void_pp), ('ecc_grads', c_void_pp), ('ecc_ws', ctypes.c_void_p), ('ecc_bwd_ws', ctypes.c_void_p),
                ('ecc_out', ctypes.c_void_p), ('grad_ecc_out', ctypes.c_void_p),
                ('nout', ctypes.c_int), ('n_classes', ctypes.c_int), ('cls_W', ctypes.c_void_p), ('cls_b', ctypes.c_void_p),
                ('cls_dW', ctypes.c_void_p), ('cls_db', ctypes.c_void_p), ('cls_work', ctypes.c_void_p), ('logits', ctypes.c_void_p),
                ('grad_logits', ctypes.c_void_p),
                ('target', ctypes.c_void_p), ('class_weight', ctypes.c_void_p), ('ignore_index', ctypes.c_int64),
                ('reduction_mean', ctypes.c_int), ('loss_buf', ctypes.c_void_p), ('ptn_slots_clean', ctypes.c_int)]


class EdgeFeatureSpecs(ctypes.Structure):
    _fields_ = [('ncols', ctypes.c_int), ('pad_', ctypes.c_int), ('col', EdgeFeatureSpec * SPG_EF_MAX_COLS)]


# name -> (restype, argtypes): every symbol include/spg_hip.h declares
SIGNATURES = {
    'spg_last_error': (ctypes.c_char_p, []),
    'spg_version': (_i, []),
    'spg_graph_workspace_bytes': (_sz, [_i, _i, _i]),
    'spg_graph_build': (_i, [_p, _p, _i, _i, _i, _p, _p]),
    'spg_graph_export': (_i, [_p, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p]),
    'spg_ecc_aggregate_fwd': (_i, [_i, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p]),
    'spg_ecc_aggregate_bwd': (_i, [_i, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p]),
    'spg_gru_scratch_floats': (_sz, [_i]),
    'spg_gru_cell_fwd': (_i, [_p, _p, _i, c_void_pp, _i, _i, _p, _p, _p]),
    'spg_gru_cell_bwd': (_i, [_p, _p, _p, _i, c_void_pp, _i, _i, _p, _p, c_void_pp, _p, _p]),
    'spg_lstm_scratch_floats': (_sz, [_i]),
    'spg_lstm_cell_fwd': (_i, [_p, _p, _p, _i, c_void_pp, _i, _i, _p, _p, _p, _p]),
    'spg_lstm_cell_bwd': (_i, [_p, _p, _p, _p, _p, _i, c_void_pp, _i, _i, _p, _p, _p, c_void_pp, _p, _p]),
    'spg_linear_fwd': (_i, [_p, _l, _i, _i, _p, _p, _i, _p, _p, _i, _p, _l, _p]),
    'spg_linear_dgrad': (_i, [_p, _l, _i, _i, _p, _i, _p, _l, _p]),
    'spg_colsum': (_i, [_p, _l, _l, _i, _p, _p, _p]),
    'spg_linear_wgrad_work_floats': (_sz, [_i, _i, _i]),
    'spg_linear_wgrad': (_i, [_p, _l, _p, _l, _i, _i, _i, _p, _p, _i, _p, _p, _p]),
    'spg_linear_wgrad_bias_work_floats': (ctypes.c_size_t, [_i, _i, _i]),
    'spg_linear_wgrad_bias': (_i, [_p, _l, _p, _l, _i, _i, _i, _p, _p, _i, _p, _p, _p, _p]),
    'spg_linear_backward': (_i, [_p, _l, _p, _l, _p, _i, _i, _i, _p, _l, _p, _p, _p, _p]),
    'spg_pointnet_num_layers': (_i, [ctypes.POINTER(PointNetCfg)]),
    'spg_pointnet_workspace_bytes': (_sz, [ctypes.POINTER(PointNetCfg), _i, _i]),
    'spg_pointnet_forward': (_i, [ctypes.POINTER(PointNetCfg), _i, _p, _p, c_void_pp, _p, _p, _i, _i, _p]),
    'spg_pointnet_forward_ext': (_i, [ctypes.POINTER(PointNetCfg), _i, _p, _p, _p, c_void_pp, _p, _p, _i, _i, _p]),
    'spg_pointnet_backward_ext': (_i, [ctypes.POINTER(PointNetCfg), _i, _p, _p, _p, c_void_pp, _p, c_void_pp, _p, _p, _p, _p, _p]),
    'spg_pointnet_debug_offset': (_l, [ctypes.POINTER(PointNetCfg), _i, _i, _i, _i]),
    'spg_pointnet_bwd_workspace_bytes': (_sz, [ctypes.POINTER(PointNetCfg), _i]),
    'spg_pointnet_backward': (_i, [ctypes.POINTER(PointNetCfg), _i, _p, _p, c_void_pp, _p, c_void_pp, _p, _p, _p]),
    'spg_eccrnn_workspace_bytes': (_sz, [ctypes.POINTER(EccRnnCfg), _i, _i, _i]),
    'spg_eccrnn_forward': (_i, [ctypes.POINTER(EccRnnCfg), _i, _i, _p, _p, _p, c_void_pp, _p, _p, _i, _i, _p]),
    'spg_eccrnn_bwd_workspace_bytes': (_sz, [ctypes.POINTER(EccRnnCfg), _i, _i]),
    'spg_eccrnn_debug_offset': (_l, [ctypes.POINTER(EccRnnCfg), _i, _i, _i, _i, _i]),
    'spg_ecc_persistent_errors': (_i, []),
    'spg_ecc_persistent_errors_clear': (_i, []),
    'spg_ecc_persistent_status': (_i, [ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), _i]),
    'spg_eccrnn_backward': (_i, [ctypes.POINTER(EccRnnCfg), _i, _i, _p, _p, c_void_pp, _p, _p, c_void_pp, _p, _p, _p]),
    'spg_adam_clamp_step': (_i, [_p, _p, _p, _p, _l, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                 ctypes.c_float, ctypes.c_float, _i, _p]),
    'spg_adam_clamp_step_scaled': (_i, [_p, _p, _p, _p, _l, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                        ctypes.c_float, ctypes.c_float, _i, _p, _p]),
    'spg_adam_clamp_step_guarded': (_i, [_p, _p, _p, _p, _l, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                        ctypes.c_float, ctypes.c_float, _i, _p, _p, _p]),
    'spg_ecc_persistent_flag': (_i, [_p, _p]),
    'spg_load_superpoints': (_i, [_p, _i, _p, _i, _p, _p, _i, _i, _p, _i, _p, _p, _p, _p, _p]),
    'spg_set_batch_workspace_bytes': (_sz, [_i, _i]),
    'spg_set_batch': (_i, [_p, _i, _i, _p, _p, _p, _p, _p, _p]),
    'spg_gather_rows': (_i, [_p, _l, _p, _l, _i, _p, _l, _p]),
    'spg_upload': (_i, [_p, _sz, _p, _p]),
    'spg_batch_graph_scratch_bytes': (_sz, [_i, _i, _i]),
    'spg_batch_graph_build': (_i, [_p, _p, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p]),
    'spg_batch_graph_build_dev': (_i, [_p, _p, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p]),
    'spg_upload_packed': (_i, [_p, _p, _p, _i, _p, _sz, _p]),
    'spg_spg_workspace_bytes': (_sz, [_i, _l]),
    'spg_spg_tet_edges': (_i, [_p, _l, _p, _p, _l, _p, _p]),
    'spg_spg_unique_edges': (_i, [_p, _l, _p, _p, _l, ctypes.c_float, _p, _p, _p, _p, _sz, _p]),
    'spg_spg_group_edges': (_i, [_p, _p, _l, _p, _p, _p, _p, _p, _p, _sz, _p]),
    'spg_spg_superpoints': (_i, [_p, _l, _p, _i, _p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    'spg_spg_superedges': (_i, [_p, _p, _p, _l, _l] + [_p] * 16 + [_p]),
    'spg_compute_geof': (_i, [_p, _p, _l, _i, _p, _p]),
    'spg_prune_workspace_bytes': (_sz, [_l]),
    'spg_prune_voxels': (_i, [_p, _l, ctypes.c_float, _p, _p, _p, _sz, _p]),
    'spg_prune_reduce': (_i, [_p, _p, _p, _p, _l, _l, _i, _i, _p, _p, _p, _p, _p, _p, _sz, _p]),
    'spg_edge_features': (_i, [ctypes.POINTER(EdgeFeatureSpecs), _p, _l, _p, _p, _p, _p]),
    'spg_loader_random': (_i, [_p, _p, _p, _i, _i, _i, ctypes.c_uint64, ctypes.c_uint32, _i, ctypes.c_float, _i, ctypes.c_float, _i, _p, _p, _p, _p]),
    'spg_cross_entropy_fwd': (_i, [_p, _p, _p, _i, _i, ctypes.c_int64, _i, _p, _p, _p, _p]),
    'spg_cross_entropy_bwd': (_i, [_p, _p, _p, _p, _p, _p, _i, _i, ctypes.c_int64, _i, _p, _p]),
    'spg_cross_entropy_fwd_bwd': (_i, [_p, _p, _p, _i, _i, ctypes.c_int64, _i, _p, _p, _p, _p, _p]),
    'spg_train_step': (_i, [ctypes.POINTER(StepArgs), _p]),
    'spg_infer_step': (_i, [ctypes.POINTER(StepArgs), _p]),
    'spg_eval_accumulate': (_i, [_p, _i, _l, _i, _i, _p, _p, _p, _p, _p, _p]),
    'spg_set_bn_allreduce': (_i, [_p, _p, _p, _l]),
    'spg_rccl_unique_id': (_i, [_p]),
    'spg_rccl_init': (_i, [_p, _i, _i]),
    'spg_rccl_world_size': (_i, []),
    'spg_rccl_allreduce_sum_f32': (_i, [_p, _l, _p]),
    'spg_rccl_sync_bn': (_i, [_p, _l]),
    'spg_rccl_sync_slots': (_i, [_i]),
    'spg_group_trace': (_i, [_p, _i]),
    'spg_group_trace_read': (_i, [_p, _i]),
    'spg_rccl_allreduce_sum_f64': (_i, [_p, _l, _p]),
    'spg_set_slot_allreduce': (_i, [_p, _p, _i]),
    'spg_rccl_destroy': (_i, []),
    'spg_prof_enable': (None, [_i]),
    'spg_prof_tag': (_i, [_i, _i, _i, _i, _i, _i]),
    'spg_prof_read_tag': (_i, [_i, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_long), ctypes.POINTER(ctypes.c_double)]),
    'spg_prof_read_shapes': (_i, [ctypes.POINTER(_i), ctypes.POINTER(ctypes.c_double), _i]),
    'spg_tune': (_i, [_i, _i]),
    'spg_prof_read': (_i, [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_long), ctypes.POINTER(ctypes.c_double), _i]),
}

_lib = None


def build(verbose: bool = False) -> str:
    """Compile the HIP sources for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    out = subprocess.run(['make', '-C', CSRC, '-j8'], capture_output=True, text=True)
    if out.returncode != 0:
        raise RuntimeError('building libspg_hip.so failed:\n' + out.stdout[-4000:] + out.stderr[-4000:])
    if verbose:
        print(out.stdout[-2000:])
    return LIB_PATH


def lib():
    """The loaded library; raises (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f'{LIB_PATH} not found: the HIP extension is not built '
                               f'(run `python -c "import __graft_entry__ as g; g.build()"` or `make -C {CSRC}`). '
                               'There is no CPU fallback for the product path.')
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc: int, what: str = ''):
    if rc != 0:
        msg = lib().spg_last_error()
        raise RuntimeError(f'libspg_hip {what} failed (rc={rc}): {msg.decode() if msg else ""}')
