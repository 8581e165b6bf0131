"""Over-read detector for single library calls: operands are placed so that their last byte is the last byte of a dedicated hipMalloc
region (a whole number of 2 MB granules: what lies behind is normally unmapped), so a kernel that reads past the end of an operand
faults instead of silently reading a neighbour.  Each probe announces itself before it runs (the last announced name is the
culprit).  Found with it: see the commit that added this file."""
import ctypes, sys, torch
sys.path.insert(0, '.')
from world_modelz_amd import ops
hip = ctypes.CDLL('libamdhip64.so')
GRAN = 2 << 20
_keep = []
def at_end(t):
    """a copy of the CUDA tensor t whose storage ends exactly at the end of its own hipMalloc region"""
    t = t.contiguous()
    nbytes = t.numel() * t.element_size()
    size = (nbytes + GRAN - 1) // GRAN * GRAN
    p = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(size)) == 0
    class Raw:
        pass
    r = Raw()
    r.__cuda_array_interface__ = {'shape': tuple(t.shape), 'typestr': {torch.bfloat16: '<u2', torch.float32: '<f4', torch.int64: '<i8'}[t.dtype],
                                  'data': (p.value + size - nbytes, False), 'version': 2}
    g = torch.as_tensor(r, device='cuda')
    if t.dtype == torch.bfloat16:
        g = g.view(torch.bfloat16)
    g.copy_(t)
    _keep.append((p, r))
    return g
def probe(name, fn):
    print('PROBE', name, flush=True)
    out = fn()
    torch.cuda.synchronize()
    print('   ok', flush=True)
    return out
torch.manual_seed(0)
B, n, heads, dh = 2, 16, 4, 128
I = heads * dh
for H, W in ((1, 16), (2, 16), (3, 16), (2, 8), (4, 8), (1, 8), (5, 16)):
    for layout in ('thirds', 'separate'):
        if layout == 'thirds':
            qkv = at_end(torch.randn(B, 1, H, W, 3 * I, device='cuda').bfloat16())
            q, k, v = qkv[..., :I], qkv[..., I:2 * I], qkv[..., 2 * I:]
        else:
            q, k, v = (at_end(torch.randn(B, 1, H, W, I, device='cuda').bfloat16()) for _ in range(3))
        ext = (0, H, W)
        o, lse, _ = probe(f'attn fwd {(H, W)} {layout}', lambda: ops.local3d_attention_fwd(q, k, v, ext, heads, need_lse=True))
        o, lse = at_end(o), at_end(lse)
        do = at_end(torch.randn_like(o))
        probe(f'attn bwd {(H, W)} {layout}', lambda: ops.local3d_attention_bwd(q, k, v, o, lse, do, ext, heads))
print('ALL PROBES PASSED')
