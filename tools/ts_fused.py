"""Stage timeline of the fused layer kernel (workgroup 0, per wave), from the s_memtime probe."""
import sys, torch
sys.path.insert(0, '.')
from world_modelz_amd import config, fused, _lib as L
from world_modelz_amd.main import VqVideoDiffusionModel
torch.manual_seed(0)
m = VqVideoDiffusionModel(data_shape=(32, 16, 16), dim=256, num_classes=1024, extents=(3, 3, 3), depth=4, dim_head=128, mlp_dim=256, heads=1).cuda()
config.set_compute_dtype(torch.bfloat16)
planes = int(sys.argv[1]) if len(sys.argv) > 1 else 256
x = torch.randn(planes, 16, 16, 256, device='cuda').bfloat16()
o = torch.randn(planes, 16, 16, 128, device='cuda').bfloat16()
Ls = list(m.transformer.layers)
ts = torch.zeros(8 * 64, dtype=torch.int64, device='cuda')
with torch.no_grad():
    for _ in range(3): fused.layer_fused(o, x, Ls[0], Ls[1], xflags=3)
    L.call('wmz_debug_fused_timestamps', ts.data_ptr())
    fused.layer_fused(o, x, Ls[0], Ls[1], xflags=3)
    torch.cuda.synchronize()
    L.call('wmz_debug_fused_timestamps', None)
t = ts.cpu().view(8, 64)
names = {44: 'o+vec landed', 45: 'to_out gemm', 0: 'start', 1: 'all issued', 2: 'init bout', 3: '+x', 4: 'LN2', 30: 'b2+LN1+pack', 31: 'x stores', 32: 'q gemm', 33: 'q pack/store', 34: 'k gemm', 35: 'k pack/store', 36: 'v gemm', 37: 'end'}
names.update({48: ' W1[4] vmcnt done', 49: ' W1[4] barrier done', 50: ' W1[4] dma issued', 51: ' W1[4] group0 done', 52: ' W2[3] vmcnt done', 53: ' W2[3] barrier done', 54: ' W2[3] dma issued', 55: ' W2[3] group0 done'})
names[5] = 'W1[0] gemm'; names[6] = 'gelu[0]'; names[29] = 'W2[7] gemm'
for c in range(1, 8):
    names[5 + 3 * c] = f'W1[{c}] gemm'; names[7 + 3 * c] = f'W2[{c-1}]+gelu[{c}]'
for w in (0, 3, 7):
    base = int(t[w, 0]); prev = base
    print(f'--- wave {w} (100 MHz ticks? see total) total {int(t[w,37]) - base}')
    for k in sorted(names, key=lambda k: int(t[w, k])):
        v = int(t[w, k])
        if v == 0: continue
        print(f'  {names[k]:16s} +{v - prev:7d}   @{v - base:8d}')
        prev = v
