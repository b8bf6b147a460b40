#!/usr/bin/env python3
"""Which HIP API calls of a profiled run are memory copies / fills (rocprofv3 --hip-trace --kernel-trace, rocpd sqlite):
prints the memcpy / memset API calls per name and size, to find the origin of small __amd_rocclr_copyBuffer dispatches."""
import sqlite3
import sys
from collections import Counter

db = sqlite3.connect(sys.argv[1])
tables = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
print('tables:', [t for t in tables if 'region' in t or 'memory' in t or 'kernel' in t][:20])
for t in ('regions', 'memory_copies', 'memory_allocations'):
    if t in tables:
        cols = [r[1] for r in db.execute(f'pragma table_info({t})')]
        print(t, cols)
if 'regions' in tables:
    c = Counter(r[0] for r in db.execute('select name from regions'))
    for k, v in c.most_common(40):
        print(f'{v:7d} {k}')
if 'memory_copies' in tables:
    cols = [r[1] for r in db.execute('pragma table_info(memory_copies)')]
    rows = list(db.execute('select * from memory_copies limit 2000'))
    print('memory_copies rows', len(rows))
    c = Counter()
    for r in rows:
        d = dict(zip(cols, r))
        c[(d.get('name'), d.get('size'), d.get('src_agent_type', d.get('src_agent_abs_index')), d.get('dst_agent_type', d.get('dst_agent_abs_index')))] += 1
    for k, v in c.most_common(40):
        print(v, k)
