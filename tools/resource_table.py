#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS table of one HIP source, from the compiler's own remarks:

    python tools/resource_table.py superpoint_graph_amd/csrc/spg_gemm.hip [more.hip ...] > profiles/rNN_kernel_resources.txt

Runs `hipcc --offload-arch=gfx950 -O3 -Rpass-analysis=kernel-resource-usage -c` (no GPU needed: hipcc cross-compiles) and prints one
row per __global__ function: SGPRs, VGPRs, AGPRs, scratch bytes per lane, occupancy (waves per SIMD), spilled SGPRs / VGPRs, static LDS.
Spills are what to look at first: a spilled SGPR costs a v_writelane / v_readlane pair (and scratch when the lanes run out), a spilled
VGPR a scratch store + load per use."""
import os
import re
import subprocess
import sys
import tempfile

HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FIELDS = [('sgpr', r'TotalSGPRs'), ('vgpr', r'VGPRs'), ('agpr', r'AGPRs'), ('scratch', r'ScratchSize \[bytes/lane\]'),
          ('occ', r'Occupancy \[waves/SIMD\]'), ('sspill', r'SGPRs Spill'), ('vspill', r'VGPRs Spill'), ('lds', r'LDS Size \[bytes/block\]')]


def table(src, extra=()):
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.abspath(src)
        r = subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Rpass-analysis=kernel-resource-usage', *extra,
                            '-c', src, '-o', os.path.join(tmp, 'x.o')], capture_output=True, text=True, cwd=os.path.dirname(src))
    if r.returncode != 0:
        sys.exit(r.stderr[-3000:])
    blocks = re.split(r'remark: Function Name: ', r.stderr)[1:]
    rows = []
    for b in blocks:
        name = b.split()[0]
        vals = {}
        for key, pat in FIELDS:
            m = re.search(r'remark:\s+' + pat + r': (\d+)', b)
            vals[key] = int(m.group(1)) if m else -1
        rows.append((name, vals))
    names = subprocess.run(['c++filt'], input='\n'.join(n for n, _ in rows), capture_output=True, text=True).stdout.splitlines()
    return [(re.sub(r'^void ', '', n), v) for n, (_, v) in zip(names, rows)]


def main():
    only = None
    args = [a for a in sys.argv[1:]]
    if '--spills-only' in args:
        args.remove('--spills-only')
        only = 'spills'
    for src in args:
        rows = table(src)
        print(f'# {src}: {len(rows)} kernels (hipcc -O3 --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage)')
        print(f'# {"SGPR":>4} {"VGPR":>4} {"AGPR":>4} {"scratch":>7} {"occ":>3} {"SGPRspill":>9} {"VGPRspill":>9} {"LDS":>6}  kernel')
        for name, v in sorted(rows, key=lambda r: r[0]):
            if only == 'spills' and v['sspill'] <= 0 and v['vspill'] <= 0:
                continue
            print(f'  {v["sgpr"]:4d} {v["vgpr"]:4d} {v["agpr"]:4d} {v["scratch"]:7d} {v["occ"]:3d} {v["sspill"]:9d} {v["vspill"]:9d} {v["lds"]:6d}  {name[:150]}')
        ns = sum(1 for _, v in rows if v['sspill'] > 0)
        nv = sum(1 for _, v in rows if v['vspill'] > 0)
        print(f'# kernels with spilled SGPRs: {ns}, with spilled VGPRs: {nv}')


if __name__ == '__main__':
    main()
