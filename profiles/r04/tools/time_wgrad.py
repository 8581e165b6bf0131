"""The layer's weight-gradient launch pair alone (wmz_linear_wgrad_batch), at the published dim-384 widths and the default widths,
65 536 tokens: microseconds per call.  With WMZ_LIB_PATH = a -DWMZ_W3_ABL variant: the ablations of wgrad3_kernel."""
import sys, torch
sys.path.insert(0, '.')
from world_modelz_amd import ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
bf = torch.bfloat16
def t(*s): return torch.randn(*s, device='cuda', dtype=bf)
def w(*s): return torch.zeros(*s, device='cuda', dtype=torch.float32)
for name, D, M, I in (('dim 384 / mlp 512', 384, 512, 128), ('dim 256 / mlp 256', 256, 256, 128)):
    probs = [(t(N, D), t(N, M), w(D, M), w(D), False), (t(N, M), t(N, D), w(M, D), w(M), True),
             (t(N, 2 * I), t(N, D), w(2 * I, D), w(2 * I), True), (t(N, D), t(N, I), w(D, I), w(D), False),
             (t(N, I), t(N, D), w(I, D), None, False)]
    for sel, label in ((probs, 'all five'), (probs[:3], 'wide three'), (probs[3:], 'narrow two')):
        for _ in range(3): ops.linear_wgrad_batch(sel)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.linear_wgrad_batch(sel)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        flops = sum(2.0 * N * p[2].shape[0] * p[2].shape[1] for p in sel)
        byts = sum(2.0 * N * (p[2].shape[0] + p[2].shape[1]) for p in sel)
        print(f'{name} {label}: {us:.1f} us  ({flops / us / 1e6:.0f} TFLOP/s, operands once {byts / us / 1e6:.2f} TB/s)', flush=True)
