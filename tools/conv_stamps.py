"""Diagnostic: in-kernel s_memtime stamps of convr_kernel (library built with -DWMZ_CONV_STAMPS, loaded through WMZ_LIB_PATH): per
workgroup the cycles from start to the patch requests issued / landed, in the MFMA loop, waiting for the epilogue's barrier, in the
epilogue; and, per CU (HW_ID / XCC_ID), how the two resident workgroups' phases lie against each other.

    python tools/build_variant.py stamps conv_direct.hip -DWMZ_CONV_STAMPS
    WMZ_LIB_PATH=tools/variants/libwmz_stamps.so python tools/conv_stamps.py [Cin Cout]"""
import ctypes, sys, numpy as np, torch
sys.path.insert(0, '.')
from world_modelz_amd import ops, _lib as L
Ci, Co = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 128)
B, H = 256, 64
x = torch.randn(B, H, H, Ci, device='cuda').bfloat16()
w = (torch.randn(Co, 9 * Ci, device='cuda') * 0.05).bfloat16()
for _ in range(5):
    ops.conv2d_nhwc(x, w, 3, 3, 1, 1, stats=True)
torch.cuda.synchronize()
n = 4096
buf = (ctypes.c_ulonglong * (n * 8))()
fn = L.lib().wmz_debug_conv_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert fn(buf, n * 8) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(n, 8).astype(np.int64)
a = a[a[:, 0] > 0]
t0 = a[:, 0].min()
med = lambda v: int(np.median(v))
print(f'{len(a)} workgroups, span {a[:, 5].max() - t0} memtime ticks (100 MHz: {1e-2 * (a[:, 5].max() - t0):.1f} us)')
ph = [('patch requests issued', 0, 1), ('patch landed (+ barrier)', 1, 2), ('MFMA loop', 2, 3), ('barrier before the epilogue', 3, 4), ('epilogue', 4, 5), ('whole workgroup', 0, 5)]
for name, i, j in ph:
    d = a[:, j] - a[:, i]
    print(f'  {name:28s} median {med(d):6d}  p10 {int(np.percentile(d, 10)):6d}  p90 {int(np.percentile(d, 90)):6d}')
# per CU: hw id fields (gfx9: wave 0-3, simd 4-5, pipe 6-7, cu 8-11, sh 12, se 13-15 ...) + xcc
hw = a[:, 6] & 0xFFFFFFFF
xcc = (a[:, 6] >> 32) & 0xF
cu = ((hw >> 8) & 0xF) | (((hw >> 12) & 0x1) << 4) | (((hw >> 13) & 0x7) << 5) | (xcc << 8)
ids = np.unique(cu)
print(f'  {len(ids)} distinct (XCC, SE, SH, CU) ids; workgroups per id: min {min((cu == i).sum() for i in ids)} max {max((cu == i).sum() for i in ids)}')
# for one CU: its workgroups in start order with phase boundaries relative to t0
one = ids[len(ids) // 2]
rows = a[cu == one]
rows = rows[np.argsort(rows[:, 0])]
print(f'  CU id {one:#x}: start, landed, loop end, end (ticks from the launch start), overlap of MFMA loops of consecutive workgroups')
prev = None
for r in rows[:12]:
    ov = ''
    if prev is not None:
        o = min(prev[3], r[3]) - max(prev[2], r[2])
        ov = f'   loop overlap with the previous {max(o, 0)}'
    print(f'     {r[0] - t0:7d} {r[2] - t0:7d} {r[3] - t0:7d} {r[5] - t0:7d}{ov}')
    prev = r
