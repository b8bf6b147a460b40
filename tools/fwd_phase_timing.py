"""Cycles per phase of the persistent forward row-GEMM streams (attribution build: tools/build_variant.sh attr "-DSPG_ATTRIBUTION").
   SPG_HIP_LIB=<variant .so> python tools/fwd_phase_timing.py [scenes]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from superpoint_graph_amd import _lib


def main():
    from superpoint_graph_amd.flat import FlatParameters
    from superpoint_graph_amd import fused as spg_fused
    scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    dev = torch.device('cuda', 0)
    model = B.build_model('gru_10_0,f_13', dev, 14)
    model.train()
    targets, GIs, flag, clouds, diam, _ = B.make_batch(list(range(scenes)), 1000, 5000, 14, 13)
    clouds_d, diam_d, label = clouds.to(dev), diam.to(dev), targets[:, 0].to(dev)
    model.ecc.set_info(GIs, 1)
    arena = FlatParameters(model, lazy_zero=True, host_counters=True)
    fstep = spg_fused.FusedStep(model, arena, reduction='mean', ptn_mem_monger=True)

    def run():
        arena.zero_grad()
        fstep(flag, clouds_d, diam_d, GIs[0], label)
        arena.adam_step(lr=1e-2, weight_decay=0.0, grad_clip=1.0, grad_div=None)

    h = ctypes.CDLL(_lib.LIB_PATH)
    h.spg_fwd_phase_times.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    for _ in range(5): run()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 56)()
    h.spg_fwd_phase_times(buf, 1)
    n = 10
    for _ in range(n): run()
    torch.cuda.synchronize()
    h.spg_fwd_phase_times(buf, 0)
    names = ['K 64 -> 128 (main conv3)', 'K 128 -> 128 (conv4)', 'K 128 -> 256 pooled (conv5)', 'K 64 -> 128 pooled (STN conv3)']
    mfma = [128, 256, 256, 128]       # MFMAs per wave and tile
    print(f'{scenes} scene(s) per step; cycles of wave 0, averaged over the workgroups of a launch')
    for c in range(4):
        v = [buf[c * 14 + k] for k in range(14)]
        wgs, tiles = max(v[4], 1), max(v[5], 1)
        tot = sum(v[:4]) / wgs
        print('%-32s entry: first loads issued %.0f, BatchNorm fold %.0f, constants + first chunk into LDS + barrier %.0f, bias / sign / stagger / accumulators %.0f' % (names[c], v[10] / max(v[4], 1), v[11] / max(v[4], 1), v[12] / max(v[4], 1), v[13] / max(v[4], 1)))
        print('%-32s workgroups/launch %5.0f tiles/workgroup %.2f | total %7.0f cycles: entry->first chunk %6.0f, chunk loops %6.0f (%.0f per tile; MFMA issue alone %d), '
              'epilogues+between tiles %6.0f (%.0f per tile: store %.0f, statistics %.0f, pool %.0f + %.0f behind its barrier), end %5.0f' % (names[c], wgs / n, tiles / wgs, tot, v[0] / wgs, v[1] / wgs, v[1] / tiles, mfma[c] * 64,
                                                                              v[2] / wgs, v[2] / tiles, v[6] / tiles, v[7] / tiles, v[8] / tiles, v[9] / tiles, v[3] / wgs))


if __name__ == '__main__':
    main()
