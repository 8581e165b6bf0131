"""GPU parity of the config-5 path (minecraft/sparse_diffusion.py + transformer.py drop-ins) against the reference
capture and the oracle's autograd."""
import pytest
import torch

from conftest import load_golden, sub

pytestmark = pytest.mark.gpu

from oracle import denoiser as oden          # noqa: E402


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(scope='module')
def wmz():
    assert torch.cuda.is_available()
    from world_modelz_amd import config, sparse_diffusion
    return dict(config=config, sd=sparse_diffusion)


def test_sparse_model_vs_golden_and_grads(wmz):
    g = load_golden('sparse_tiny')
    sd = sub(g, 'sd/')
    shape = tuple(int(e) for e in g['shape'])
    heads = int(g['heads'])
    m = wmz['sd'].VqSparseDiffusionModel(shape=shape, dim=32, num_classes=40, depth=2, dim_head=16, mlp_dim=48, heads=heads)
    m.load_state_dict(sd, strict=True)
    m = m.cuda()
    x, idx = g['x'].cuda(), g['indices'].cuda()
    with wmz['config'].compute_dtype(torch.float32):
        with torch.no_grad():
            y = m(x, idx)
        assert y.shape == g['logits'].shape and rel(y, g['logits']) < 1e-5
        # gradients against torch.autograd over the oracle
        leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        ref = oden.sparse_denoiser_forward(leaves, g['x'], g['indices'], shape, heads)
        w = torch.randn_like(ref)
        (ref * w).sum().backward()
        out = m(x, idx)
        (out * w.cuda()).sum().backward()
    for n, p in m.named_parameters():
        assert rel(p.grad, leaves[n].grad) < 5e-5, n


@pytest.mark.parametrize('n', [512, 100])
def test_sparse_model_default_width_bf16(wmz, n):
    """dim 512, 4 heads x 128, mlp 1024 (sparse_diffusion.py:233-257), n = 512 tokens (16-wide fast path) and a
    ragged n = 100 (general kernel), bf16 vs the fp32 oracle."""
    torch.manual_seed(3)
    shape = (8, 16, 16)
    m = wmz['sd'].VqSparseDiffusionModel(shape=shape, dim=512, num_classes=256, depth=2, dim_head=128, mlp_dim=1024, heads=4)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    x = torch.randint(0, 257, (2, n))
    idx = torch.stack([torch.randperm(8 * 256)[:n] for _ in range(2)])
    ref = oden.sparse_denoiser_forward(sd, x, idx, shape, 4)
    m = m.cuda()
    with torch.no_grad():
        with wmz['config'].compute_dtype(torch.float32):
            y32 = m(x.cuda(), idx.cuda())
        with wmz['config'].compute_dtype(torch.bfloat16):
            y16 = m(x.cuda(), idx.cuda())
    assert rel(y32, ref) < 1e-5
    assert rel(y16, ref) < 2e-2


def test_position_samplers(wmz):
    sdm = wmz['sd']
    p = sdm.sample_flat_positions(3, 100, 6, 4, 4, 'cuda')
    assert p.shape == (3, 100) and int(p.min()) >= 0 and int(p.max()) < 96
    t = torch.tensor([0.0, 0.5, 1.0])
    q = sdm.sample_time_dependent(3, 32, 8, 4, 4, t, 'cuda', o=torch.tensor([0.0, 0.5, 0.99]))
    assert q.shape == (3, 32)
    for b in range(3):
        assert len(set(q[b].tolist())) == 32                      # without replacement
    # t = 0: window = min_sample_window = 2 frames -> positions within 2*16 of the offset
    assert int(q[0].max()) - int(q[0].min()) < 2 * 16
    assert int(q.max()) < 8 * 16
