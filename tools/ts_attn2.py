"""Per-wave, per-slab table of the row16 attention forward (workgroup 0): when each wave leaves the barrier, how long it
computes, when it arrives at the next barrier.  WMZ_DBG=2: compute only (no staging)."""
import os, sys, torch
sys.path.insert(0, '.')
from world_modelz_amd import ops, _lib as L
torch.manual_seed(0)
DBG = int(os.environ.get('WMZ_DBG', '0'))
qkv = torch.randn(8, 32, 16, 16, 384, device='cuda').bfloat16()
q, k, v = qkv[..., :128], qkv[..., 128:256], qkv[..., 256:]
ts = torch.zeros(16 * 64, dtype=torch.int64, device='cuda')
L.call('wmz_debug_attn_knobs', DBG, 0)
for _ in range(300): ops.local3d_attention_fwd(q, k, v, (3, 3, 3), 1)
L.call('wmz_debug_attn_timestamps', ts.data_ptr())
ops.local3d_attention_fwd(q, k, v, (3, 3, 3), 1)
torch.cuda.synchronize()
L.call('wmz_debug_attn_timestamps', None)
L.call('wmz_debug_attn_knobs', 0, 0)
t = ts.cpu().view(16, 64)
t0 = int(t[:, 0].min())
print('slab: barrier release (min..max over waves) | per wave: leave-barrier offset, issue, compute | arrive (vmcnt) offset from first arrival')
for j in range(1, 7):
    rel = [int(t[w, 4 + 4 * j]) - t0 for w in range(16)]
    iss = [int(t[w, 5 + 4 * j] - t[w, 4 + 4 * j]) for w in range(16)]
    cmp_ = [int(t[w, 6 + 4 * j] - t[w, 5 + 4 * j]) for w in range(16)]
    arr = [int(t[w, 3 + 4 * (j + 1)]) - t0 for w in range(16)]
    nxt = min(int(t[w, 4 + 4 * (j + 1)]) - t0 for w in range(16))
    print(f'slab {j}: release {min(rel)}..{max(rel)}  next release {nxt}  period {nxt - min(rel)}')
    print('   leave  ' + ' '.join(f'{r - min(rel):5d}' for r in rel))
    print('   issue  ' + ' '.join(f'{x:5d}' for x in iss))
    print('   compute' + ' '.join(f'{x:5d}' for x in cmp_))
    print('   arrive ' + ' '.join(f'{a - min(rel):5d}' for a in arr))
