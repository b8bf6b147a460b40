"""Cycles per phase of the one-launch forward RNN-ECC recurrence (attribution build: tools/build_variant.sh attr "-DSPG_ATTRIBUTION" spg_gemm.hip spg_ecc.hip).
   SPG_HIP_LIB=<variant .so> python tools/ecc_phase_timing.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from superpoint_graph_amd import _lib


def main():
    from superpoint_graph_amd.flat import FlatParameters
    from superpoint_graph_amd import fused as spg_fused
    dev = torch.device('cuda', 0)
    model = B.build_model('gru_10_0,f_13', dev, 14)
    model.train()
    targets, GIs, flag, clouds, diam, _ = B.make_batch([0], 1000, 5000, 14, 13)
    clouds_d, diam_d, label = clouds.to(dev), diam.to(dev), targets[:, 0].to(dev)
    model.ecc.set_info(GIs, 1)
    arena = FlatParameters(model, lazy_zero=True, host_counters=True)
    fstep = spg_fused.FusedStep(model, arena, reduction='mean', ptn_mem_monger=True)

    def run():
        arena.zero_grad()
        fstep(flag, clouds_d, diam_d, GIs[0], label)
        arena.adam_step(lr=1e-2, weight_decay=0.0, grad_clip=1.0, grad_div=None)

    h = ctypes.CDLL(_lib.LIB_PATH)
    h.spg_ecc_phase_times.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    for _ in range(5): run()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 12)()
    h.spg_ecc_phase_times(buf, 1)
    n = 10
    for _ in range(n): run()
    torch.cuda.synchronize()
    h.spg_ecc_phase_times(buf, 0)
    waves = max(buf[6], 1)
    names = ['entry -> first iteration', 'own-state part (W_hh h, classifier share)', 'waiting for + gathering the neighbours', 'filters + mean', 'input part + gates + publish', 'head + exit']
    print('   inside the entry: node / graph / filter-load issue %.0f, cell weights -> LDS %.0f, classifier weights -> LDS + loss-weight partials %.0f, barrier + gate rows -> registers %.0f; the rest up to the first iteration is in line 1' % tuple(buf[k] / waves for k in (7, 8, 9, 10)))
    tot = sum(buf[k] for k in range(6)) / waves
    print(f'forward recurrence, 10 iterations: {waves / n:.0f} node waves per launch, {tot:.0f} cycles per wave')
    for k in range(6):
        v = buf[k] / waves
        print(f'   {names[k]:45s} {v:8.0f} cycles ({100 * v / tot:4.1f} %)' + (f' = {v / 10:.0f} per iteration' if 1 <= k <= 4 else ''))


if __name__ == '__main__':
    main()
