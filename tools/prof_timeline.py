#!/usr/bin/env python3
"""Print the kernel sequence of the LAST step of a rocprofv3 (rocpd sqlite) kernel trace.
usage: prof_timeline.py run_results.db nsteps"""
import sqlite3
import sys
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x from kernels order by start"))
per = len(rows) // int(sys.argv[2])
last = rows[-per:]
t0 = last[0][1]
prev = None
for nm, s, e, gx, gy, gz, wx in last:
    gap = (s - prev) / 1e3 if prev else 0.0
    prev = e
    print(f'{(s - t0) / 1e3:9.1f} dur={(e - s) / 1e3:7.1f} gap={gap:6.1f} grid=({gx // max(wx, 1)},{gy},{gz}) {nm.replace("void ", "")[:90]}')
