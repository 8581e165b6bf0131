"""In-kernel timeline of attn_fwd_row32 (variant build with -DWMZ_ATTN32_TS, loaded through WMZ_LIB_PATH): per wave of workgroup 0,
cycles spent per slab in: wait for own DMA, barrier, DMA issue, and per tile H2(carried) / H1 / H2."""
import ctypes, os, sys, torch
sys.path.insert(0, '.')
from world_modelz_amd import ops, _lib as L
torch.manual_seed(0)
r = torch.randn(3, 8, 32, 16, 16, 128, device='cuda').bfloat16()
var = int(sys.argv[1]) if len(sys.argv) > 1 else 0
dbg = int(sys.argv[2]) if len(sys.argv) > 2 else 0
buf = torch.zeros(8 * 256, dtype=torch.int64, device='cuda')
lib = L.lib()
lib.wmz_debug_attn32_ts.argtypes = [ctypes.c_void_p]
assert lib.wmz_debug_attn32_ts(buf.data_ptr()) == 0
L.call('wmz_debug_attn_knobs', dbg, var)
for _ in range(20): ops.local3d_attention_fwd(r[0], r[1], r[2], (3, 3, 3), 1)
torch.cuda.synchronize()
buf.zero_()
ops.local3d_attention_fwd(r[0], r[1], r[2], (3, 3, 3), 1)
torch.cuda.synchronize()
t = buf.cpu().view(8, 256).tolist()
L.call('wmz_debug_attn_knobs', 0, 0)
for w in range(8):
    T = t[w]
    t0 = T[240]
    print(f'wave {w}: total {T[250] - t0} cycles (loop start -> end)')
    tot = dict(wait=0, barrier=0, dma=0, h2c=0, h1=0, h2=0, other=0)
    for j in range(14):
        b = 16 * j
        if T[b] == 0: continue
        line = f'  slab {j:2d} @{T[b] - t0:6d}: wait {T[b+1]-T[b]:5d} barrier {T[b+2]-T[b+1]:5d} dma {T[b+3]-T[b+2]:4d} |'
        tot['wait'] += T[b+1]-T[b]; tot['barrier'] += T[b+2]-T[b+1]; tot['dma'] += T[b+3]-T[b+2]
        for i in range(3):
            s = b + 4 + 4 * i
            if T[s] == 0: break
            line += f' tile{i}: h2c {T[s+1]-T[s]:5d} h1 {T[s+2]-T[s+1]:5d} h2 {T[s+3]-T[s+2]:5d} |'
            tot['h2c'] += T[s+1]-T[s]; tot['h1'] += T[s+2]-T[s+1]; tot['h2'] += T[s+3]-T[s+2]
        if w in (0, 2, 6) : print(line)
    print('  totals:', tot)
