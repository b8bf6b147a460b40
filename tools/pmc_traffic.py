#!/usr/bin/env python3
"""HBM traffic of the GEMM kernels from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass) of
`bench.py`, corrected as MI355X_MICROARCH.md prescribes (FETCH_SIZE x2 on gfx950) and, when calibration passes of
tools/pmc_calib.py are given, by the ratios measured there on known byte counts (WRITE_SIZE is uncalibrated otherwise).

    python tools/pmc_traffic.py fetch.db write.db [calib_fetch.db calib_write.db calib_MiB] > profiles/rNN_gemm_traffic.json"""
import json
import sqlite3
import sys
from collections import defaultdict


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    out = defaultdict(lambda: [0, 0.0])
    for name, val in c.execute("select kernel_name, sum(value) from counters_collection where counter_name = ? "
                               "group by dispatch_id, kernel_name", (counter,)):
        short = name.replace('void ', '').split('(')[0]
        out[short][0] += 1
        out[short][1] += val
    return out


def summarise(fetch_db, write_db, calib_fetch_db=None, calib_write_db=None, calib_mib=None):
    """-> dict (the content of profiles/rNN_gemm_traffic.json)"""
    fetch, write = per_kernel(fetch_db, 'FETCH_SIZE'), per_kernel(write_db, 'WRITE_SIZE')
    kb = 1024.0
    fcorr, wcorr, calib = 2.0, 1.0, None
    if calib_fetch_db is not None:
        mib = float(calib_mib)
        cf, cw = per_kernel(calib_fetch_db, 'FETCH_SIZE'), per_kernel(calib_write_db, 'WRITE_SIZE')
        add = [k for k in cf if 'CUDAFunctorOnSelf_add' in k][0]
        fill = [k for k in cw if 'FillFunctor' in k][0]
        fcorr = mib * 1024 * 1024 / (cf[add][1] / cf[add][0] * kb)
        wadd = cw[add][1] / cw[add][0] * kb
        wfill = cw[fill][1] / cw[fill][0] * kb
        wcorr = mib * 1024 * 1024 / wfill
        calib = {'MiB': mib, 'fetch_true_over_reported': fcorr, 'write_true_over_reported_fill': wcorr,
                 'write_true_over_reported_add': mib * 1024 * 1024 / wadd}
    steps = sum(v[0] for k, v in fetch.items() if k.startswith('spg_adam_clamp_kernel'))
    # (spg_multi_*: grouped launches of few-row GEMMs / weight gradients, round 4 -- counted as one launch each, like bench.py's events)
    gemm = lambda d: {k: v for k, v in d.items() if k.startswith('spg_rowgemm_kernel') or k.startswith('spg_wgrad_kernel') or k.startswith('spg_multi')}
    gf, gw = gemm(fetch), gemm(write)
    launches = sum(v[0] for v in gf.values())
    f_b = sum(v[1] for v in gf.values()) * kb * fcorr
    w_b = sum(v[1] for v in gw.values()) * kb * wcorr
    rows = []
    for k in sorted(gf, key=lambda k: -(gf[k][1])):
        n = gf[k][0]
        rows.append({'kernel': k, 'launches_per_step': n / steps, 'fetch_mb_per_launch': gf[k][1] / n * kb * fcorr / 1e6,
                     'write_mb_per_launch': gw.get(k, [1, 0.0])[1] / max(gw.get(k, [1, 0.0])[0], 1) * kb * wcorr / 1e6})
    allk = lambda d: sum(v[1] for v in d.values())
    # the RNN-ECC kernels (latency-bound dataflow launches: no roof applies; reported with their HBM bytes and duration)
    ecc = {}
    try:
        c = sqlite3.connect(fetch_db)
        for name, n, avg_ns in c.execute("select name, count(*), avg(end - start) from kernels where name like '%spg_ecc_%' group by name"):
            short = name.replace('void ', '').split('(')[0]
            f, w = fetch.get(short, [0, 0.0]), write.get(short, [0, 0.0])
            fb = f[1] / max(f[0], 1) * kb * fcorr
            wb = w[1] / max(w[0], 1) * kb * wcorr
            ecc[short] = {'launches_per_step': n / steps, 'avg_us': avg_ns / 1e3, 'hbm_fetch_mb': fb / 1e6, 'hbm_write_mb': wb / 1e6,
                          'achieved_gbs': (fb + wb) / (avg_ns * 1e-9) / 1e9 if avg_ns else None,
                          'bound': 'latency (dataflow-synchronised recurrence: one wavefront per node waits for its neighbours\' states each '
                                   'iteration; HBM traffic = the filters once)' if 'persist' in short else 'hbm/l2'}
    except Exception as e:
        ecc = {'error': str(e)}
    return {
        'ecc': ecc,
        'source': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) of bench.py; FETCH_SIZE x%.3f, '
                  'WRITE_SIZE x%.3f (%s)' % (fcorr, wcorr, 'calibrated on tools/pmc_calib.py in the same session' if calib else
                                             'gfx950 rule of MI355X_MICROARCH.md; WRITE_SIZE uncalibrated'),
        'calibration': calib, 'steps': steps, 'kernels': 'spg_rowgemm_kernel + spg_wgrad_kernel',
        'launches_per_step': launches / steps, 'fetch_mb_per_step': f_b / steps / 1e6, 'write_mb_per_step': w_b / steps / 1e6,
        'hbm_mb_per_step': (f_b + w_b) / steps / 1e6, 'hbm_mb_per_launch': (f_b + w_b) / launches / 1e6,
        'all_kernels_hbm_mb_per_step': (allk(fetch) * kb * fcorr + allk(write) * kb * wcorr) / steps / 1e6,
        'per_kernel': rows[:24]}


if __name__ == '__main__':
    print(json.dumps(summarise(*sys.argv[1:6]), indent=1))
