"""Build-time check for csrc/layer_fused_bwd.hip.  Its kernels fetch their operands with inline-asm loads hipcc does not
track and retire them with counted `s_waitcnt vmcnt(n)` (fused_common.h vm_wait_since): safe only if
  * the kernels are straight-line code (the counts are compile-time constants; predicated row stores are the only branches
    and are taken as issued -- the entry points require whole 32-token tiles),
  * nothing spills (scratch traffic would count in vmcnt),
  * no instruction reads or writes a destination register of such a load before a wait that retires it.
Compiles the file to ISA and simulates the vmcnt queue.        python tools/check_untracked_bwd.py"""
import os, re, subprocess, sys
src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'world_modelz_amd', 'csrc', 'layer_fused_bwd.hip')
asm = subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only', src, '-o', '-'],
                     capture_output=True, text=True).stdout
VMEM = re.compile(r'\s*(global_load|global_store|global_atomic|buffer_|scratch_|flat_)')
bad_total, kernels = 0, 0
for name, body in re.findall(r'^(_ZN\S*(?:qkv_bwd_kernel|ff_bwd_kernel)\S*):\s*;.*?\n(.*?)s_endpgm', asm, flags=re.S | re.M):
    kernels += 1
    lines = body.split('\n')
    issued = 0
    pending = []            # (seq, lo, hi, line) of untracked loads not yet retired
    nloads = nwaits = 0
    bad = []
    for i, l in enumerate(lines):
        if not re.match(r'\s+[a-z]', l):
            continue
        code = l.split(';')[0]
        if re.match(r'\s*s_cbranch_(?!execz)', code) or re.match(r'\s*s_(branch|setpc)', code):
            bad.append((i, 'control flow: ' + code.strip()))
        if 'scratch_' in code:
            bad.append((i, 'scratch traffic: ' + code.strip()))
        m = re.match(r'\s*s_waitcnt\b(.*)', code)
        if m:
            v = re.search(r'vmcnt\((\d+)\)', m.group(1))
            if v:
                n = int(v.group(1))
                pending = [p for p in pending if p[0] > issued - n]
                nwaits += 1
            continue
        # register uses against the loads still in flight
        for r in re.finditer(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b', code):
            rr = (int(r.group(3)),) * 2 if r.group(3) else (int(r.group(1)), int(r.group(2)))
            for (sq, lo, hi, li) in pending:
                if rr[0] <= hi and rr[1] >= lo:
                    bad.append((i, f'{code.strip()}   <- in flight since line {li}'))
        if VMEM.match(code):
            issued += 1
            a = re.match(r'\s*global_load_dword(x\d)? v(?:\[(\d+):(\d+)\]|(\d+)), v\[\d+:\d+\], off', code)
            if a and '#ASMSTART' in lines[i - 1]:
                lo, hi = (int(a.group(4)),) * 2 if a.group(4) else (int(a.group(2)), int(a.group(3)))
                pending.append((issued, lo, hi, i))
                nloads += 1
    if pending:
        bad.append((len(lines), f'{len(pending)} untracked loads never retired'))
    print(f'{name[:60]}: {nloads} untracked loads, {nwaits} vmcnt waits, {issued} VMEM ops, violations: {len(bad)}')
    for b in bad[:10]:
        print('   line', b[0], b[1])
    bad_total += len(bad)
if kernels < 8:
    print('expected 8 kernel instantiations, found', kernels)
    sys.exit(1)
sys.exit(1 if bad_total else 0)
