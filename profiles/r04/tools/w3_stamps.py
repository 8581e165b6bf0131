"""Diagnostic: in-kernel s_memtime stamps of wgrad3_kernel (library built with -DWMZ_W3_STAMPS, loaded through WMZ_LIB_PATH):
per workgroup, cycles to the first slab, in the slab loop, parked at its DMA waits + barriers, in the tile stores."""
import ctypes, sys, numpy as np, torch
sys.path.insert(0, '.')
from world_modelz_amd import ops, _lib as L
N, bf = 65536, torch.bfloat16
def t(*s): return torch.randn(*s, device='cuda', dtype=bf)
def w(*s): return torch.zeros(*s, device='cuda', dtype=torch.float32)
for name, D, M, I in (('dim 384 / mlp 512', 384, 512, 128), ('dim 256 / mlp 256', 256, 256, 128)):
    probs = [(t(N, D), t(N, M), w(D, M), w(D), False), (t(N, M), t(N, D), w(M, D), w(M), True),
             (t(N, 2 * I), t(N, D), w(2 * I, D), w(2 * I), True), (t(N, D), t(N, I), w(D, I), w(D), False),
             (t(N, I), t(N, D), w(I, D), None, False)]
    for _ in range(30): ops.linear_wgrad_batch(probs)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (256 * 8))()
    assert L.lib().wmz_debug_w3_stamps(buf, 256 * 8) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(256, 8).astype(np.int64)
    a = a[a[:, 0] > 0]
    t0 = a[:, 0].min()
    med = lambda v: int(np.median(v))
    print(f'{name}: {len(a)} workgroups; 100 MHz-domain ticks? no: s_memtime = shader clock cycles')
    print(f'  start skew (max t0 - min t0) {a[:, 0].max() - t0}, whole span {a[:, 4].max() - t0}')
    print(f'  median: to first slab {med(a[:, 2] - a[:, 0])}, slab loop {med(a[:, 3] - a[:, 2])} (parked {med(a[:, 5])}), stores {med(a[:, 4] - a[:, 3])}')
    print(f'  loop min / max {(a[:, 3] - a[:, 2]).min()} / {(a[:, 3] - a[:, 2]).max()}; parked min / max {a[:, 5].min()} / {a[:, 5].max()}')
    print(f'  end skew: first done {a[:, 4].min() - t0}, last done {a[:, 4].max() - t0}', flush=True)
