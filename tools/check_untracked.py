"""Build-time check for csrc/layer_fused.hip: the residual rows are fetched by inline-asm loads hipcc does not track, so
nothing may read or overwrite their destination registers between the load and the counted s_waitcnt that retires them.
Compiles the kernel to ISA and scans the span.   python tools/check_untracked.py"""
import re, subprocess, sys, os
src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'world_modelz_amd', 'csrc', 'layer_fused.hip')
asm = subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only', src, '-o', '-'],
                     capture_output=True, text=True).stdout
bad_total = 0
for name, body in re.findall(r'^(_ZN\S*layer_fused_kernel\S*):\s*;.*?\n(.*?)s_endpgm', asm, flags=re.S | re.M):
    lines = body.split('\n')
    loads = []
    for i, l in enumerate(lines):
        m = re.match(r'\s*global_load_dwordx4 v\[(\d+):(\d+)\], v\[\d+:\d+\], off\s*$', l)
        if m and i and '#ASMSTART' in lines[i - 1]:
            loads.append((i, int(m.group(1)), int(m.group(2))))
    if not loads:
        continue
    waits = [i for i, l in enumerate(lines) if re.search(r's_waitcnt vmcnt\((6|4)\)', l) and '#ASMSTART' in lines[i - 1] and i > loads[-1][0]]
    end = waits[0]
    bad = []
    for i in range(loads[0][0], end):
        l = lines[i]
        if not re.match(r'\s+[a-z]', l) or ('global_load_dwordx4' in l and '#ASMSTART' in lines[i - 1]):
            continue
        for m in re.finditer(r'v\[(\d+):(\d+)\]|v(\d+)', l):
            rr = [int(m.group(3))] if m.group(3) else range(int(m.group(1)), int(m.group(2)) + 1)
            for r in rr:
                for li, a, b in loads:
                    if a <= r <= b and i > li:
                        bad.append((i, l.strip()))
    print(f'{name[:70]}: {len(loads)} untracked loads, wait at line {end}, touches in between: {len(bad)}')
    bad_total += len(bad)
sys.exit(1 if bad_total else 0)
