#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel count / total / average, like --stats, restricted to the
STEADY-STATE training steps: the window from the end of the first spg_adam_clamp_kernel (= end of step 1) to the end of
the last one, so that set-up work (model.to(device): one small copy kernel per parameter, graph build, warm-up
allocations) is not attributed to the steps.

    python tools/prof_summary.py <results.db> [max_rows]"""
import sqlite3
import sys

db = sys.argv[1]
c = sqlite3.connect(db)
marks = [r[0] for r in c.execute("select end from kernels where name like 'spg_adam_clamp_kernel%' order by end")]
if len(marks) >= 2:
    t0, t1, nsteps = marks[0], marks[-1], len(marks) - 1
    where = f'where start >= {t0} and end <= {t1}'
else:
    t0, t1 = c.execute('select min(start), max(end) from kernels').fetchone()
    nsteps, where = 1, ''
rows = list(c.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
                      f"from kernels {where} group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
print(f'# {db}: steady-state window of {nsteps} training steps: {sum(r[1] for r in rows)} kernel launches '
      f'({sum(r[1] for r in rows) / nsteps:.1f} per step), {tot / 1e3:.2f} ms of kernel time = {tot / 1e3 / nsteps:.3f} ms/step; '
      f'window span {(t1 - t0) / 1e6:.2f} ms = {(t1 - t0) / 1e6 / nsteps:.3f} ms/step wall')
print(f'{"total_us":>10s} {"pct":>6s} {"calls":>6s} {"calls/step":>10s} {"avg_us":>9s} {"min_us":>9s} {"max_us":>9s}  name')
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 60]:
    print(f'{r[2]:10.0f} {100 * r[2] / tot:6.2f} {r[1]:6d} {r[1] / nsteps:10.1f} {r[3]:9.1f} {r[4]:9.1f} {r[5]:9.1f}  {r[0][:120]}')
