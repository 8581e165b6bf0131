import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from world_modelz_amd import config, ops
from world_modelz_amd.autoencoder import Residual
from oracle import autoencoder as oae
def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))
for stride in (1, 2):
  for dtype in (torch.float32, torch.bfloat16):
    torch.manual_seed(4)
    with config.compute_dtype(dtype):
        blk = Residual(64, 128, stride).cuda()
        x0 = torch.randn(4, 64, 32, 32)
        dyo = torch.randn(4, 64, 32 // stride, 32 // stride)
        x = x0.cuda().requires_grad_(True)
        y = blk(x)
        (y * dyo.cuda()).sum().backward()
    leaves = {'b.' + k: (v.detach().cpu().clone().requires_grad_(True) if v.is_floating_point() and 'running' not in k else v.detach().cpu().clone()) for k, v in blk.state_dict().items()}
    for k in leaves:
        if k.endswith('running_mean'): leaves[k] = torch.zeros_like(leaves[k])
        elif k.endswith('running_var'): leaves[k] = torch.ones_like(leaves[k])
    xo = x0.clone().requires_grad_(True)
    yo = oae.residual_block(leaves, 'b.', xo, stride, True)
    (yo * dyo).sum().backward()
    print(stride, dtype, 'y', rel(y, yo), 'dx', rel(x.grad, xo.grad))
    for n, prm in blk.named_parameters():
        print('   ', n, rel(prm.grad, leaves['b.' + n].grad))
    # same with bf16-rounded input to see input rounding effect
