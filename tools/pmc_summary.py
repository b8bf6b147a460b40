#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counters (rocpd sqlite) per kernel name and grid: mean counter values per dispatch.
usage: pmc_summary.py run_results.db [min_total_us]"""
import sqlite3
import sys
from collections import defaultdict

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select kernel_name, grid_size_x, grid_size_y, grid_size_z, workgroup_size_x, counter_name, value, (end-start)/1e3, dispatch_id "
                 "from counters_collection")
agg = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(set)
dur = defaultdict(float)
for name, gx, gy, gz, wx, cname, val, d, disp in rows:
    key = (name.replace('void ', '').split('(')[0], gx // max(wx, 1), gy, gz)
    agg[key][cname] += val
    if disp not in cnt[key]:
        cnt[key].add(disp)
        dur[key] += d
names = sorted({k for v in agg.values() for k in v})
print('# mean per dispatch;', ' '.join(names))
for key in sorted(agg, key=lambda k: -dur[k]):
    n = len(cnt[key])
    if dur[key] < (float(sys.argv[2]) if len(sys.argv) > 2 else 0):
        continue
    vals = ' '.join(f'{cn}={agg[key][cn] / n:.4g}' for cn in names if cn in agg[key])
    print(f'{dur[key] / n:9.1f}us n={n:4d} {key[0][:52]:52s} grid=({key[1]},{key[2]},{key[3]})  {vals}')
