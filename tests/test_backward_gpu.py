"""GPU parity of the backward kernels: gradients of the HIP path (fp32 mode) against torch.autograd over the
CPU oracle and against the gradients captured from the reference (golden vectors)."""
import math

import pytest
import torch

from conftest import load_golden, sub

pytestmark = pytest.mark.gpu

from oracle import attention as oat          # noqa: E402
from oracle import train_step as ots         # noqa: E402


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(scope='module')
def wmz():
    assert torch.cuda.is_available()
    from world_modelz_amd import config, local_3d_attention, main, ops
    return dict(config=config, l3a=local_3d_attention, main=main, ops=ops)


@pytest.mark.parametrize('shape,heads,dh,ext,dtype', [
    ((2, 5, 6, 7), 1, 32, (3, 3, 3), torch.float32),
    ((1, 3, 4, 4), 3, 16, (2, 2, 2), torch.float32),
    ((1, 4, 16, 16), 2, 64, (1, 2, 3), torch.float32),
    ((1, 3, 6, 40), 1, 8, (1, 1, 2), torch.float32),
    ((1, 8, 8, 8), 1, 128, (3, 3, 3), torch.float32),
    ((1, 8, 16, 16), 1, 128, (3, 3, 3), torch.bfloat16),
])
def test_attention_core_backward_vs_oracle(wmz, shape, heads, dh, ext, dtype):
    torch.manual_seed(11)
    ops = wmz['ops']
    B, S, H, W = shape
    I = heads * dh
    q, k, v, do = (torch.randn(B, S, H, W, I).to(dtype).float() for _ in range(4))
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, k, v))
    out_ref = oat.local_attention(kr, vr, qr, ext, heads)
    out_ref.backward(do)
    qd, kd, vd, dod = (t.cuda().to(dtype) for t in (q, k, v, do))
    out, lse, _ = ops.local3d_attention_fwd(qd, kd, vd, ext, heads, need_lse=True)
    dq, dkv = ops.local3d_attention_bwd(qd, kd, vd, out, lse, dod, ext, heads)
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    assert rel(dq, qr.grad) < tol
    assert rel(dkv[..., :I], kr.grad) < tol
    assert rel(dkv[..., I:], vr.grad) < tol


@pytest.mark.parametrize('tag', ['a', 'b'])
def test_attention_module_grads_vs_golden(wmz, tag):
    """Local3dAttention.forward(x, q) with x != q: out, dx, dq and every parameter gradient vs the reference."""
    g = load_golden(f'attn_module_{tag}')
    sd = sub(g, 'sd/')
    heads = int(g['heads'])
    I, D = sd['to_q.weight'].shape
    m = wmz['l3a'].Local3dAttention(tuple(int(e) for e in g['extents']), D, heads=heads, dim_head=I // heads)
    m.load_state_dict(sd, strict=True)
    m = m.cuda()
    x = g['x'].cuda().requires_grad_(True)
    q = g['q'].cuda().requires_grad_(True)
    with wmz['config'].compute_dtype(torch.float32):
        out = m(x, q=q)
        out.square().sum().backward()
    assert rel(out, g['out']) < 1e-5
    assert rel(x.grad, g['dx']) < 2e-5
    assert rel(q.grad, g['dq']) < 2e-5
    for n, p in m.named_parameters():
        assert rel(p.grad, g['grad/' + n]) < 2e-5, n


def test_training_step_grads_vs_golden(wmz):
    """One step of main.py:train at the tiny shape: logits, loss, every gradient, grad-norm (a15)."""
    g = load_golden('step_tiny')
    sd0 = sub(g, 'sd0/')
    ext = tuple(int(e) for e in g['extents'])
    heads = int(g['heads'])
    C = g['logits'].shape[-1]
    m = wmz['main'].VqVideoDiffusionModel(data_shape=(3, 4, 4), dim=16, num_classes=C, extents=ext, depth=2, dim_head=8,
                                          mlp_dim=24, heads=heads)
    m.load_state_dict(sd0, strict=True)
    m = m.cuda()
    with wmz['config'].compute_dtype(torch.float32):
        y = m(g['corrupted'].cuda())
        loss = torch.nn.functional.cross_entropy(y.reshape(-1, C), g['target'].cuda().reshape(-1), reduction='none')
        per_sample = loss.view(2, -1).mean(dim=1)
        loss.mean().backward()
    assert rel(y, g['logits']) < 1e-5
    assert torch.allclose(per_sample.cpu(), g['per_sample_loss'], rtol=1e-5)
    grads = {n: p.grad for n, p in m.named_parameters()}
    for n, p in m.named_parameters():
        assert p.grad is not None and p.grad.dtype == torch.float32, n
        assert rel(p.grad, g['grad/' + n]) < 5e-5, n
    gn = math.sqrt(sum(float((gr.double() ** 2).sum()) for gr in grads.values()))
    assert math.isclose(gn, float(g['grad_norm']), rel_tol=1e-4)


def test_default_model_backward_bf16_vs_oracle(wmz):
    """Default denoiser on a 2x6x16x16 grid: bf16 training-step gradients against the fp32 oracle's autograd
    (reported, bounded at 5e-2 relative per tensor family; the parity gate is the fp32 test above)."""
    torch.manual_seed(42)
    C = 64
    m = wmz['main'].VqVideoDiffusionModel(data_shape=(6, 16, 16), dim=256, num_classes=C, extents=(3, 3, 3), depth=2,
                                          dim_head=128, mlp_dim=256, heads=1)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    z = torch.randint(0, C + 1, (2, 6, 16, 16))
    target = torch.randint(0, C, (2, 16, 16))
    _, _, loss_ref, grads_ref = ots.step_grads(sd, z, target, (3, 3, 3), 1)
    m = m.cuda()
    for dtype, tol in [(torch.float32, 2e-4), (torch.bfloat16, 6e-2)]:
        m.zero_grad()
        with wmz['config'].compute_dtype(dtype):
            y = m(z.cuda())
            loss = torch.nn.functional.cross_entropy(y.reshape(-1, C), target.cuda().reshape(-1))
            loss.backward()
        assert abs(float(loss) - float(loss_ref)) < (1e-4 if dtype == torch.float32 else 2e-2)
        worst = max((rel(p.grad, grads_ref[n]), n) for n, p in m.named_parameters())
        print(f'{dtype}: worst relative gradient error {worst[0]:.3e} at {worst[1]}')
        assert worst[0] < tol, worst


@pytest.mark.parametrize('shape,heads,dh,ext', [
    ((2, 5, 16, 16), 1, 128, (3, 3, 3)),
    ((1, 9, 16, 16), 1, 128, (3, 1, 1)),
    ((1, 4, 16, 16), 2, 64, (1, 2, 3)),
    ((1, 3, 40, 16), 1, 32, (2, 2, 2)),        # H > 16: several workgroups per plane, both roles
    ((1, 2, 5, 16), 4, 32, (0, 1, 0)),
    ((2, 2, 1, 16), 4, 128, (0, 1, 16)),       # one row per plane (config 5's dense attention over 16 tokens)
    ((1, 2, 17, 16), 1, 64, (1, 2, 2)),        # H = 1 (mod 16)
])
def test_attention_backward_row16_fast_path(wmz, shape, heads, dh, ext):
    """bf16, W == 16: attn_bwd_row16.hip against torch.autograd over the oracle (2e-2 rel: P and dS are bf16 MFMA
    operands) -- dq, dk, dv."""
    torch.manual_seed(31)
    ops = wmz['ops']
    B, S, H, W = shape
    I = heads * dh
    q, k, v, do = (torch.randn(B, S, H, W, I).bfloat16() for _ in range(4))
    qr, kr, vr = (t.float().clone().requires_grad_(True) for t in (q, k, v))
    oat.local_attention(kr, vr, qr, ext, heads).backward(do.float())
    qd, kd, vd, dod = (t.cuda() for t in (q, k, v, do))
    out, lse, _ = ops.local3d_attention_fwd(qd, kd, vd, ext, heads, need_lse=True)
    dq, dkv = ops.local3d_attention_bwd(qd, kd, vd, out, lse, dod, ext, heads)
    assert rel(dq, qr.grad) < 2e-2
    assert rel(dkv[..., :I], kr.grad) < 2e-2
    assert rel(dkv[..., I:], vr.grad) < 2e-2
    # q | k | v as the column thirds of one [.., 3I] buffer (the fused to_qkv of config 5): the visiting tensors of the dk | dv pass
    # (q and dout) then have different row strides, and the gradients land in the thirds of one buffer
    qkv = torch.cat([qd, kd, vd], dim=-1)
    dqkv = torch.empty_like(qkv)
    ops.local3d_attention_bwd(qkv[..., :I], qkv[..., I:2 * I], qkv[..., 2 * I:], out, lse, dod, ext, heads, dqkv=dqkv)
    assert torch.equal(dqkv[..., :I], dq) and torch.equal(dqkv[..., I:], dkv)


@pytest.mark.parametrize('shape,heads,dh,ext', [
    ((1, 8, 8, 8), 1, 128, (3, 3, 3)),         # BASELINE configs[1] grid
    ((2, 6, 8, 8), 1, 128, (3, 1, 1)),         # the reference's geometry (main.py:394), published window
    ((1, 3, 8, 8), 2, 64, (1, 2, 2)),          # even row extent
    ((1, 2, 8, 8), 1, 32, (1, 0, 3)),          # eH = 0
    ((1, 3, 16, 8), 1, 128, (1, 3, 1)),        # 8 tile rows: two workgroups per plane
    ((1, 2, 24, 8), 1, 64, (0, 5, 2)),         # 12 tile rows: the 8-wave shapes
    ((1, 2, 6, 8), 1, 128, (1, 1, 1)),         # ragged chunk
    ((2, 3, 2, 8), 1, 128, (1, 1, 3)),         # two plane rows = one tile row
])
def test_attention_backward_8_wide_planes(wmz, shape, heads, dh, ext):
    """bf16, W == 8 (even H): the row kernels of attn_bwd_row16.hip on tile rows of 16 with rim masks, against torch.autograd over
    the oracle -- dq, dk, dv (2e-2 rel: P and dS are bf16 MFMA operands)."""
    torch.manual_seed(33)
    ops = wmz['ops']
    B, S, H, W = shape
    I = heads * dh
    q, k, v, do = (torch.randn(B, S, H, W, I).bfloat16() for _ in range(4))
    qr, kr, vr = (t.float().clone().requires_grad_(True) for t in (q, k, v))
    oat.local_attention(kr, vr, qr, ext, heads).backward(do.float())
    qd, kd, vd, dod = (t.cuda() for t in (q, k, v, do))
    out, lse, _ = ops.local3d_attention_fwd(qd, kd, vd, ext, heads, need_lse=True)
    dq, dkv = ops.local3d_attention_bwd(qd, kd, vd, out, lse, dod, ext, heads)
    e = (rel(dq, qr.grad), rel(dkv[..., :I], kr.grad), rel(dkv[..., I:], vr.grad))
    print(f'[8-wide bwd {shape} {ext}] dq {e[0]:.2e} dk {e[1]:.2e} dv {e[2]:.2e}')
    assert max(e) < 2e-2, e
