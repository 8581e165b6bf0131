"""Every width triple of csrc/chain_widths.h (the per-token chain kernels beside the default-width kernel): inference in bfloat16
and in the precise mode against the fp32 oracle, training forward + backward against the oracle's autograd and the op-by-op path.
The table is read from the header, so a triple added there is tested here.  Among them: the reference's own test() geometry
(local_3d_attention.py:166-174: dim 128, 3 heads of 64, mlp 256, extents (2,2,2), depth 4, grid (2,4,16,16) of a (10,16,16) model)."""
import os
import re

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import denoiser as oden          # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def width_table():
    text = open(os.path.join(ROOT, 'world_modelz_amd', 'csrc', 'chain_widths.h')).read()
    text = '\n'.join(ln for ln in text.splitlines() if not ln.lstrip().startswith('//'))
    rows = [tuple(int(v) for v in m.groups()) for m in re.finditer(r'X\((\d+), (\d+), (\d+), (\d+)\)', text)]
    assert len(rows) >= 12 and len(set(rows)) == len(rows)
    return rows


WIDTHS = width_table()


def heads_of(D, I):
    """heads x dim_head for the widths: 192 = 3 x 64 (the reference's test()), else heads of 128 -- but never ONE head as wide as
    the model (quirk Q6, local_3d_attention.py:50-53: to_out is then the identity, another stack): two heads of 64 there."""
    if I == 192:
        return 3, 64
    return (I // 128, 128) if I != D or I // 128 > 1 else (I // 64, 64)


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(scope='module')
def wmz():
    assert torch.cuda.is_available()
    from world_modelz_amd import config, fused, main, train
    return dict(config=config, fused=fused, main=main, train=train)


def test_table_matches_the_library(wmz):
    import ctypes
    from world_modelz_amd import _lib as L
    for D, I, M, MC in WIDTHS:
        mc = ctypes.c_int(0)
        assert L.lib().wmz_layer_chain_supported(D, I, M, ctypes.byref(mc)) == 1 and mc.value == MC
        assert (MC * D) % 8192 == 0 and M % MC == 0
    assert L.lib().wmz_layer_chain_supported(256, 128, 256, None) == 0          # the default widths: layer_fused.hip
    assert L.lib().wmz_layer_chain_supported(64, 64, 96, None) == 0


@pytest.mark.parametrize('D,I,M,MC', WIDTHS)
def test_inference_vs_oracle(wmz, D, I, M, MC):
    """Logits of a 3-layer model on the chain kernel: bfloat16 within the bf16 error of the other fused paths, the precise mode within
    north_star's 1e-3, the last-frame cone bit-identical to the full grid, and the entry points reached are the chain kernel's."""
    from conftest import chain_policy, recorded_calls
    cfg = wmz['config']
    heads, dh = heads_of(D, I)
    torch.manual_seed(D + I + M)
    shape = (4, 6, 16) if D % 128 else (3, 8, 8)          # a ragged 16-wide plane, or the reference's 8x8 latents
    m = wmz['main'].VqVideoDiffusionModel(data_shape=shape, dim=D, num_classes=300, extents=(2, 2, 2), depth=3, dim_head=dh,
                                          mlp_dim=M, heads=heads)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    z = torch.randint(0, 301, (3,) + shape)
    ref = oden.denoiser_forward(sd, z, (2, 2, 2), heads)
    m = m.cuda().eval()
    assert wmz['fused'].chain_widths(m.transformer) == (D, I, M, MC)
    with torch.no_grad(), chain_policy('always'):
        with cfg.compute_dtype(torch.bfloat16), recorded_calls() as seen_b:
            y_b = m(z.cuda())
            cfg.set_last_frame_cone(False)
            try:
                y_full = m(z.cuda())
            finally:
                cfg.set_last_frame_cone(True)
        with cfg.compute_dtype(torch.float16), recorded_calls() as seen_p:
            y_p = m(z.cuda())
    assert 'wmz_layer_chain_fwd_planes' in seen_b and 'wmz_layer_chain_fwd_planes_f16' in seen_p
    assert torch.equal(y_b, y_full)
    e_b, e_p = rel(y_b, ref), rel(y_p, ref)
    print(f'widths ({D}, {I}, {M}): logits vs fp32 oracle bf16 {e_b:.3e}, precise {e_p:.3e}')
    assert e_b < 1e-2 and e_p < 1e-3 and e_p < e_b / 4


@pytest.mark.parametrize('D,I,M,MC', WIDTHS)
def test_training_step_vs_oracle_autograd(wmz, D, I, M, MC):
    """Loss and every parameter gradient of one training step on the chain kernels (training forward, wmz_chain_ff_bwd,
    wmz_chain_qkv_bwd, batched weight gradients) against the fp32 oracle's autograd and against the op-by-op path."""
    from conftest import chain_policy, recorded_calls
    from oracle import train_step as ots
    cfg = wmz['config']
    heads, dh = heads_of(D, I)
    torch.manual_seed(7 + D + I + M)
    C = 64
    m = wmz['main'].VqVideoDiffusionModel(data_shape=(3, 16, 16), dim=D, num_classes=C, extents=(1, 2, 2), depth=2, dim_head=dh,
                                          mlp_dim=M, heads=heads)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    z = torch.randint(0, C + 1, (2, 3, 16, 16))
    target = torch.randint(0, C, (2, 16, 16))
    _, _, loss_ref, grads_ref = ots.step_grads(sd, z, target, (1, 2, 2), heads)
    m = m.cuda()
    with cfg.compute_dtype(torch.bfloat16), chain_policy('always'):
        tr = wmz['train'].DenoiserTrainer(m, C, lr=1e-3, warmup=0, max_steps=100, distributed=False)
        assert tr.chain_packs is not None
        got = {}
        for mode, fused_on in (('chain', True), ('ops', False)):
            cfg.set_fused_training(fused_on)
            try:
                tr.arena.zero_grad()
                with recorded_calls() as seen:
                    _, mean = tr.forward_backward(z.cuda(), target.cuda())
            finally:
                cfg.set_fused_training(True)
            on_chain = all(n in seen for n in ('wmz_layer_chain_fwd_train', 'wmz_chain_ff_bwd', 'wmz_chain_qkv_bwd'))
            assert on_chain == fused_on, (mode, sorted(seen))
            got[mode] = (float(mean), {n: p.grad.detach().clone() for n, p in m.named_parameters()})
    loss_c, g_c = got['chain']
    loss_o, g_o = got['ops']
    assert abs(loss_c - float(loss_ref)) < 2e-2 and abs(loss_c - loss_o) < 2e-2
    worst = max((float((g_c[n] - grads_ref[n].cuda()).norm() / (grads_ref[n].norm() + 1e-12)), n) for n in g_c)
    worst_o = max((float((g_c[n] - g_o[n]).norm() / (g_o[n].norm() + 1e-12)), n) for n in g_c)
    print(f'widths ({D}, {I}, {M}): gradients vs oracle {worst[0]:.3e} ({worst[1]}), vs op-by-op {worst_o[0]:.3e} ({worst_o[1]})')
    assert worst[0] < 6e-2 and worst_o[0] < 6e-2, (worst, worst_o)


def test_auto_policy_takes_the_chain_kernels_where_they_pay(wmz):
    """config.chain_policy 'auto' (the default; fused.chain_pays, profiles/r06/time_chain_tokens.txt): a model of 2.6 MB of weights
    per layer (dim 512 / mlp 1024) runs op by op on 1 536 tokens -- 12 chain workgroups would each stream all of it -- and on the
    chain kernel on 8 192; a 0.2 MB model (dim 96) takes the chain kernel on both; the precise mode takes the half chain kernel
    whatever the count (its alternative is the fp32 route).  Same logits either way within the two paths' rounding."""
    from conftest import chain_policy, recorded_calls
    cfg, fused = wmz['config'], wmz['fused']
    assert cfg.get_chain_policy() == 'auto'
    assert not fused.chain_pays((512, 128, 1024, 32), 1536, False) and fused.chain_pays((512, 128, 1024, 32), 8192, False)
    assert not fused.chain_pays((384, 128, 512, 64), 8192, True) and fused.chain_pays((384, 128, 512, 64), 16384, True)
    assert fused.chain_pays((96, 128, 256, 256), 512, True) and fused.chain_pays((96, 128, 256, 256), 512, False)
    torch.manual_seed(9)
    m = wmz['main'].VqVideoDiffusionModel(data_shape=(8, 16, 16), dim=512, num_classes=100, extents=(1, 1, 1), depth=2, dim_head=128,
                                          mlp_dim=1024, heads=1).cuda().eval()
    small, large = torch.randint(0, 101, (2, 3, 16, 16), device='cuda'), torch.randint(0, 101, (4, 8, 16, 16), device='cuda')
    with torch.no_grad(), cfg.compute_dtype(torch.bfloat16):
        with recorded_calls() as seen_small:
            y_small = m(small)
        with recorded_calls() as seen_large:
            m(large)
        with chain_policy('always'):
            y_small_chain = m(small)
    assert 'wmz_layer_chain_fwd_planes' not in seen_small and 'wmz_layer_chain_fwd_planes' in seen_large
    assert rel(y_small_chain, y_small) < 2e-2
    with torch.no_grad(), cfg.compute_dtype(torch.float16), recorded_calls() as seen_p:
        m(small)
    assert 'wmz_layer_chain_fwd_planes_f16' in seen_p


def test_reference_test_geometry_forward_backward(wmz):
    """The reference's test() verbatim (local_3d_attention.py:166-174): Local3dAttentionTransformer(data_shape=(10,16,16), dim=128,
    num_classes=1000, extents=(2,2,2), depth=4, mlp_dim=256, heads=3, dim_head=64) on x = randint(0, 99, (2,4,16,16));
    y = n(x); y.mean().backward() -- output and every gradient against the oracle (a bare module trains op by op: the chain
    kernels' training launches belong to a trainer's parameter arena, test_training_step_vs_oracle_autograd), and the same forward
    without gradients on the chain kernel."""
    from conftest import recorded_calls
    from world_modelz_amd.local_3d_attention import Local3dAttentionTransformer
    cfg = wmz['config']
    torch.manual_seed(0)
    n = Local3dAttentionTransformer(data_shape=(10, 16, 16), dim=128, num_classes=1000, extents=(2, 2, 2), depth=4, mlp_dim=256,
                                    heads=3, dim_head=64, dropout=.0)
    sd = {'transformer.' + k: v.clone().requires_grad_(True) for k, v in n.state_dict().items()}
    x = torch.randint(0, 99, (2, 4, 16, 16))
    y_ref = oden.transformer_forward(sd, x, (2, 2, 2), 3)
    y_ref.mean().backward()
    n = n.cuda()
    with cfg.compute_dtype(torch.bfloat16):
        y = n(x.cuda())
        y.mean().backward()
        with torch.no_grad(), recorded_calls() as seen:
            y_inf = n(x.cuda())
    assert y.shape == (2, 4, 16, 16, 128) and y.dtype == torch.float32
    e, e_inf = rel(y, y_ref), rel(y_inf, y_ref)
    assert 'wmz_layer_chain_fwd_planes' in seen
    worst = max((rel(p.grad, sd['transformer.' + k].grad), k) for k, p in n.named_parameters() if sd['transformer.' + k].grad is not None)
    print(f'reference test() geometry: y {e:.3e} (inference, chain kernel: {e_inf:.3e}), worst gradient {worst[0]:.3e} ({worst[1]})')
    assert e < 1e-2 and e_inf < 1e-2, (e, e_inf)
    assert worst[0] < 8e-2, worst
