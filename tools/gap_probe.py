#!/usr/bin/env python3
"""What sits in the gap between the optimiser launch and the next step's first kernel: lists, for a rocpd database of
`rocprofv3 --hip-trace --kernel-trace --memory-copy-trace`, every HIP API call and memory copy whose start lies between the end of an
spg_adam_clamp_kernel dispatch and the start of the next kernel dispatch (steady state: the last 5 such gaps)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
tables = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]


def cols(t):
    return [r[1] for r in db.execute(f'pragma table_info({t})')]


kt = 'kernels' if 'kernels' in tables else [t for t in tables if 'kernel' in t][0]
kc = cols(kt)
print(kt, kc)
rows = list(db.execute(f'select name, start, end from {kt} order by start'))
gaps = []
for i, (n, s, e) in enumerate(rows[:-1]):
    if 'adam' in n:
        gaps.append((e, rows[i + 1][1], rows[i + 1][0]))
for (a, b, nxt) in gaps[-6:-1]:
    print(f'gap {(b - a) / 1e3:.1f} us before {nxt[:60]}')
    for t in ('memory_copies', 'memory_allocations'):
        if t in tables:
            c = cols(t)
            if 'start' in c and 'name' in c:
                for r in db.execute(f'select name, start, end from {t} where end >= ? and start <= ? order by start', (a - 20000, b)):
                    print(f'   {t}: {r[0][:70]} start {(r[1] - a) / 1e3:+.1f} us end {(r[2] - a) / 1e3:+.1f} us')
