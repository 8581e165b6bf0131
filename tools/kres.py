#!/usr/bin/env python3
"""Register / scratch / LDS usage of every kernel of one translation unit (runs hipcc here: no GPU needed).

    python tools/kres.py world_modelz_amd/csrc/attn_bwd_row16.hip [-DWMZ_X=1 ...] [--asm out.s] [--filter substr]

One line per kernel from hipcc's -Rpass-analysis=kernel-resource-usage; exits 1 when any kernel spills (scratch > 0).
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from world_modelz_amd.build import COMMON, PER_FILE, _hipcc  # noqa: E402


def main():
    args = sys.argv[1:]
    src = args[0]
    asm, flt, extra = None, None, []
    i = 1
    while i < len(args):
        if args[i] == '--asm':
            asm = args[i + 1]
            i += 2
        elif args[i] == '--filter':
            flt = args[i + 1]
            i += 2
        else:
            extra.append(args[i])
            i += 1
    out = asm or '/dev/null'
    cmd = [_hipcc()] + [f for f in COMMON if f != '-fPIC'] + PER_FILE.get(os.path.basename(src), []) + extra + \
        ['-S', '--cuda-device-only', '-Rpass-analysis=kernel-resource-usage', src, '-o', out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stderr)
        sys.exit(r.returncode)
    rows, cur = [], None
    for line in r.stderr.splitlines():
        m = re.search(r'remark:\s+(.*?)\s*\[-Rpass', line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith('Function Name:') or t.startswith('Name:'):
            cur = {'name': t.split(':', 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ':' in t:
            k, v = t.split(':', 1)
            cur[k.strip()] = v.strip()
    bad = False
    for row in rows:
        name = subprocess.run(['c++filt', row['name']], capture_output=True, text=True).stdout.strip() or row['name']
        name = re.sub(r'\(anonymous namespace\)::', '', name)
        name = re.sub(r'\(.*$', '', name)
        if flt and flt not in name:
            continue
        scratch = int(row.get('ScratchSize [bytes/lane]', '0'))
        bad |= scratch > 0
        print(f"{name:70s} vgpr {row.get('VGPRs', '?'):>4s} agpr {row.get('AGPRs', '?'):>3s} sgpr {row.get('TotalSGPRs', '?'):>3s} "
              f"scratch {scratch:>4d} occ {row.get('Occupancy [waves/SIMD]', '?')} lds {row.get('LDS Size [bytes/block]', '?')}")
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
