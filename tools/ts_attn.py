"""Phase timeline of the row16 attention forward kernel (workgroup 0), from the s_memtime probe."""
import sys, torch
sys.path.insert(0, '.')
from world_modelz_amd import ops, _lib as L
torch.manual_seed(0)
import os
B, S = 8, 32
DBG = int(os.environ.get('WMZ_DBG', '0'))
qkv = torch.randn(B, S, 16, 16, 384, device='cuda').bfloat16()
q, k, v = qkv[..., :128], qkv[..., 128:256], qkv[..., 256:]
ts = torch.zeros(16 * 64, dtype=torch.int64, device='cuda')
for _ in range(3): ops.local3d_attention_fwd(q, k, v, (3, 3, 3), 1)
L.call('wmz_debug_attn_knobs', DBG, 0)
for _ in range(300): ops.local3d_attention_fwd(q, k, v, (3, 3, 3), 1)
L.call('wmz_debug_attn_timestamps', ts.data_ptr())
ops.local3d_attention_fwd(q, k, v, (3, 3, 3), 1)
torch.cuda.synchronize()
L.call('wmz_debug_attn_timestamps', None)
t = ts.cpu().view(16, 64)
for w in (0, 8):
    dc, dr = int(t[w, 62] - t[w, 0]), int(t[w, 61] - t[w, 60])
    print(f'wave {w}: {dc} shader cycles in {dr} ticks of 100 MHz -> in-kernel clock {dc / max(dr, 1) * 0.1:.2f} GHz')
t[:, 60:62] = 0
names = {0: 'start', 1: 'q loaded/setup', 2: 'prime issued', 63: 'loop end', 62: 'stored'}
for j in range(15):
    names[3 + 4 * j] = f'slab {j} vmcnt'; names[4 + 4 * j] = f'slab {j} barrier'; names[5 + 4 * j] = f'slab {j} issue'; names[6 + 4 * j] = f'slab {j} compute'
import os
NWV = 8 if os.environ.get('WMZ_ATTN_QT') == '2' else 16
for w in (0, NWV // 2 - 1, NWV - 1):
    base = int(t[w, 0]); prev = base
    print(f'--- wave {w} total {int(t[w].max()) - base}')
    for kk in sorted((k for k in names if int(t[w, k]) != 0), key=lambda k: int(t[w, k])):
        vv = int(t[w, kk]); print(f'  {names[kk]:18s} +{vv - prev:6d}  @{vv - base:7d}'); prev = vv
print('per-wave slab-3 phases: vmcnt, barrier, issue, compute (cycles)')
for w in range(NWV):
    a = [int(t[w, 3 + 4 * 3 + i]) for i in range(4)]; p = int(t[w, 6 + 4 * 2])
    print(f'  wave {w:2d}: {a[0]-p:6d} {a[1]-a[0]:6d} {a[2]-a[1]:6d} {a[3]-a[2]:6d}   compute end @{a[3]-int(t[w,0])}')
