#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel count / total / average, like --stats."""
import sqlite3
import sys

db, nsteps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1
c = sqlite3.connect(db)
rows = list(c.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
                      "from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
s, e = c.execute("select min(start), max(end) from kernels").fetchone()
print(f'# {db}: {sum(r[1] for r in rows)} kernel launches, {tot / 1e3:.2f} ms of kernel time over {nsteps} steps '
      f'= {tot / 1e3 / nsteps:.3f} ms/step; first-to-last span {(e - s) / 1e6:.1f} ms')
print(f'{"total_us":>10s} {"pct":>6s} {"calls":>6s} {"calls/step":>10s} {"avg_us":>9s} {"min_us":>9s} {"max_us":>9s}  name')
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 45]:
    print(f'{r[2]:10.0f} {100 * r[2] / tot:6.2f} {r[1]:6d} {r[1] / nsteps:10.1f} {r[3]:9.1f} {r[4]:9.1f} {r[5]:9.1f}  {r[0][:120]}')
