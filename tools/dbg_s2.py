import sys, torch
sys.path.insert(0, '.')
from world_modelz_amd import ops
torch.manual_seed(31)
B,H,W,Ci,Co = 1,16,32,64,128
x = torch.randn(B, H, W, Ci, device='cuda').bfloat16()
for tap in range(9):
    w4 = torch.zeros(Co, 3, 3, Ci, device='cuda')
    w4[:, tap // 3, tap % 3, :] = torch.randn(Co, Ci, device='cuda') * 0.05
    w = w4.reshape(Co, 9 * Ci).bfloat16()
    ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w.float().view(Co, 3, 3, Ci).permute(0, 3, 1, 2), stride=2, padding=1).permute(0, 2, 3, 1)
    y = ops.conv2d_nhwc(x, w, 3, 3, 2, 1)
    err = (y.float() - ref).abs()
    print('tap', tap, 'rel', float((y.float() - ref).norm() / ref.norm()), 'bad rows', (err.amax(-1) > 0.05).nonzero()[:6].tolist())
