"""Build-time check for csrc/conv_direct.hip (convr_kernel): the weight fragments are fetched by inline-asm loads hipcc does not
track; load k is retired by the first inline-asm `s_waitcnt vmcnt(7)` behind load k + 7 (or any vmcnt(0)).  Nothing may read or
write a load's destination registers between the load and that wait.  Compiles the file to ISA and scans every instantiation.

    python tools/check_untracked_conv.py [-D... of a variant build]"""
import os
import re
import subprocess
import sys
src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'world_modelz_amd', 'csrc', 'conv_direct.hip')
asm = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only'] + [a for a in sys.argv[1:] if a.startswith('-D')] + [src, '-o', '-'],
                     capture_output=True, text=True).stdout
bad_total = 0
for name, body in re.findall(r'^(_ZN\S*convr_kernel\S*):\s*;.*?\n(.*?)\.end_amdhsa_kernel', asm, flags=re.S | re.M):
    lines = body.split('\n')
    in_asm = False
    ev = []                       # (line, kind, regs)
    for i, l in enumerate(lines):
        if '#ASMSTART' in l:
            in_asm = True
            continue
        if '#ASMEND' in l:
            in_asm = False
            continue
        m = re.match(r'\s*global_load_dwordx4 v\[(\d+):(\d+)\], v\[\d+:\d+\], off\s*$', l)
        if m and in_asm:
            ev.append((i, 'load', (int(m.group(1)), int(m.group(2)))))
        m = re.match(r'\s*s_waitcnt vmcnt\((\d+)\)', l)
        if m:
            ev.append((i, 'wait', int(m.group(1))))
    loads = [e for e in ev if e[1] == 'load']
    bad = []
    for k, (li, _, (a, b)) in enumerate(loads):
        # retired at: first wait vmcnt(n) located behind load k + n (n younger loads allowed), searching forward
        end = None
        for (wi, kind, n) in ev:
            if kind != 'wait' or wi < li:
                continue
            younger = sum(1 for (lj, _, _) in loads if li < lj < wi)
            if younger <= n:
                end = wi
                break
        if end is None:
            bad.append((li, 'never retired'))
            continue
        for i in range(li + 1, end):
            l = lines[i]
            if not re.match(r'\s+[a-z]', l):
                continue
            for m in re.finditer(r'v\[(\d+):(\d+)\]|\bv(\d+)\b', l):
                rr = [int(m.group(3))] if m.group(3) else range(int(m.group(1)), int(m.group(2)) + 1)
                if any(a <= r <= b for r in rr):
                    bad.append((i, l.strip()))
    # the same for the inline-asm LDS reads (the A-fragment window): read k is retired by the first lgkmcnt(n) wait with at most n
    # asm reads between them
    in_asm = False
    lev = []
    for i, l in enumerate(lines):
        if '#ASMSTART' in l:
            in_asm = True
            continue
        if '#ASMEND' in l:
            in_asm = False
            continue
        m = re.match(r'\s*ds_read_b128 v\[(\d+):(\d+)\], v\d+', l)
        if m and in_asm:
            lev.append((i, 'read', (int(m.group(1)), int(m.group(2)))))
        m = re.search(r's_waitcnt .*lgkmcnt\((\d+)\)', l)
        if m:
            lev.append((i, 'wait', int(m.group(1))))
    reads = [e for e in lev if e[1] == 'read']
    nbad0 = len(bad)
    for k, (li, _, (a, b)) in enumerate(reads):
        end = None
        for (wi, kind, n) in lev:
            if kind != 'wait' or wi < li:
                continue
            if sum(1 for (lj, _, _) in reads if li < lj < wi) <= n:
                end = wi
                break
        if end is None:
            bad.append((li, 'LDS read never retired'))
            continue
        for i in range(li + 1, end):
            l = lines[i]
            if not re.match(r'\s+[a-z]', l):
                continue
            for m in re.finditer(r'v\[(\d+):(\d+)\]|\bv(\d+)\b', l):
                rr = [int(m.group(3))] if m.group(3) else range(int(m.group(1)), int(m.group(2)) + 1)
                if any(a <= r <= b for r in rr):
                    bad.append((i, l.strip()))
    print(f'{name[:64]}: {len(loads)} untracked loads + {len(reads)} asm LDS reads, touches before their waits: {len(bad)}')
    for x in bad[:5]:
        print('   ', x)
    bad_total += len(bad)
sys.exit(1 if bad_total else 0)
