"""Cycles per role and phase of spg_bwdpair_kernel (attribution build: tools/build_variant.sh attr "-DSPG_ATTRIBUTION").
   SPG_HIP_LIB=<variant .so> python tools/bwdpair_timing.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from superpoint_graph_amd import _lib

def main():
    import types
    from superpoint_graph_amd.learning import pointnet
    from superpoint_graph_amd.flat import FlatParameters
    from superpoint_graph_amd import fused as spg_fused
    dev = torch.device('cuda', 0)
    model = B.build_model('gru_10_0,f_13', dev, 14)
    model.train()
    targets, GIs, flag, clouds, diam, scenes = B.make_batch([0], 1000, 5000, 14, 13)
    clouds_d, diam_d, label = clouds.to(dev), diam.to(dev), targets[:, 0].to(dev)
    model.ecc.set_info(GIs, 1)
    arena = FlatParameters(model, lazy_zero=True, host_counters=True)
    fstep = spg_fused.FusedStep(model, arena, reduction='mean', ptn_mem_monger=True)

    def run():
        arena.zero_grad()
        fstep(flag, clouds_d, diam_d, GIs[0], label)
        arena.adam_step(lr=1e-2, weight_decay=0.0, grad_clip=1.0, grad_div=None)

    h = ctypes.CDLL(_lib.LIB_PATH)
    h.spg_pair_role_times.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    for _ in range(5): run()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 36)()
    h.spg_pair_role_times(buf, 1)
    n = 10
    for _ in range(n): run()
    torch.cuda.synchronize()
    h.spg_pair_role_times(buf, 0)
    h.spg_pair_entry_times.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    eb = (ctypes.c_ulonglong * 12)()
    h.spg_pair_entry_times(eb, 0)
    names = ['64->64 (x2 launches)', '64->128 (CO 128, CI 64; x2)', '128->128']
    launches = [2, 2, 1]
    phases = {0: ['mfma', 'epilogue', 'barrier wait', '-'], 1: ['mfma', '-', 'barrier wait', '-'], 2: ['finish+write tile (incl. data wait)', 'issue loads', 'barrier wait', '-']}
    for sh in range(3):
        print(names[sh])
        print('   entry of the data-gradient wave 0 (cycles per launch and workgroup; includes the warm-up launches): up to the fold %.0f, fold %.0f, constants + W -> LDS + barrier %.0f, first tile ready %.0f' % tuple(eb[sh * 4 + k] / ((n + 5) * launches[sh] * 256) for k in range(4)))
        for role, rn in enumerate(['data gradient', 'weight gradient', 'loader']):
            # summed over 256 workgroups x (1 wave for the matrix roles | 2 waves (tid % 256 == 0) for the loaders)
            div = n * launches[sh] * 256 * (2 if role == 2 else 1)
            v = [buf[(sh * 3 + role) * 4 + k] / div for k in range(4)]
            tot = sum(v[:3])
            print('   %-16s total %8.0f cycles per launch and wave: ' % (rn, tot) + ', '.join('%s %.0f (%.0f%%)' % (phases[role][k], v[k], 100 * v[k] / max(tot, 1)) for k in range(3) if phases[role][k] != '-') + ', entry -> loop %.0f' % v[3])

if __name__ == '__main__':
    main()
