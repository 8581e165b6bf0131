#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE's own Python modules
on CPU (build container only; /root/reference does not exist on the GPU box).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [--ref /root/reference]

Every .npz holds inputs AND expected outputs (fp32 / int64), so tests need nothing from the
reference tree.  wandb / torchvision / matplotlib / minerl are absent here; empty stand-in modules
are registered only so that `train_vqae.py` / `main.py` / `sparse_diffusion.py` import (their model
classes never touch those packages).  Fixture manifest: SURVEY.md appendix B.
"""
import argparse
import os
import sys
import types
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def import_reference(ref_root):
    os.environ.setdefault('MPLBACKEND', 'Agg')
    sys.dont_write_bytecode = True
    _stub('wandb')
    tv = _stub('torchvision')
    tv.transforms = _stub('torchvision.transforms')
    tv.datasets = _stub('torchvision.datasets')
    tv.utils = _stub('torchvision.utils')
    if 'matplotlib' not in sys.modules:
        try:
            import matplotlib.pyplot  # noqa: F401
        except Exception:
            mpl = _stub('matplotlib')
            mpl.pyplot = _stub('matplotlib.pyplot')
    sys.path.insert(0, os.path.join(ref_root, 'vq-video-diffusion'))
    import local_3d_attention, vq, autoencoder, train_vqae, main, importance_sampling, warmup_scheduler  # noqa
    return dict(l3a=local_3d_attention, vq=vq, ae=autoencoder, tv=train_vqae, main=main,
                isamp=importance_sampling, warm=warmup_scheduler)


def import_sparse(ref_root):
    """minecraft/sparse_diffusion.py needs `minerl` via buffered_traj_sampler; stub it."""
    for k in ['train_vqae', 'importance_sampling', 'warmup_scheduler', 'model_ema_v2', 'vq',
              'autoencoder', 'local_3d_attention', 'main']:
        sys.modules.pop(k, None)
    sys.path.insert(0, os.path.join(ref_root, 'minecraft'))
    _stub('minerl')
    _stub('buffered_traj_sampler', BufferedTrajSampler=object)
    import sparse_diffusion
    return sparse_diffusion


def npz(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **out)
    print(f'{name}.npz: {os.path.getsize(path) / 1024:.1f} KB')


def sd_arrays(sd, prefix='sd/'):
    return {prefix + k: v.detach().clone() for k, v in sd.items()}   # snapshot: buffers mutate later


def bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


# ---------------------------------------------------------------------------------------------

def gen_attn_core(R):
    L3 = R['l3a']
    from einops import rearrange, repeat
    cases = {
        'a': dict(grid=(2, 5, 6, 7), heads=1, dh=32, ext=(3, 3, 3), bf16=False),
        'b': dict(grid=(1, 8, 8, 8), heads=1, dh=32, ext=(3, 1, 1), bf16=False),
        'c': dict(grid=(1, 3, 4, 4), heads=3, dh=16, ext=(2, 2, 2), bf16=False),
        'd': dict(grid=(2, 5, 6, 7), heads=3, dh=16, ext=(0, 1, 2), bf16=False),
        'e': dict(grid=(1, 8, 8, 8), heads=1, dh=32, ext=(3, 3, 3), bf16=True),
        'f': dict(grid=(1, 4, 16, 16), heads=2, dh=64, ext=(1, 2, 3), bf16=True),
    }
    for tag, c in cases.items():
        torch.manual_seed(100 + ord(tag))
        B, S, H, W = c['grid']
        I = c['heads'] * c['dh']
        m = L3.Local3dAttention(c['ext'], dim=I, heads=c['heads'], dim_head=c['dh'], use_checkpointing=False)
        q, k, v = (torch.randn(B, S, H, W, I) for _ in range(3))
        if c['bf16']:
            q, k, v = bf16_round(q), bf16_round(k), bf16_round(v)
        out = m.local_attention(k, v, q)                       # [(bshw), heads, 1, dh]
        out = rearrange(out, 'b h n d -> b n (h d)').reshape(B, S, H, W, I)
        # masked logits re-derived with the reference's own pad/unfold/get_mask (:80-94)
        mask = m.get_mask(k.shape)
        ku = m.unfold(m.pad(k))
        qq = rearrange(q, 'b s h w (H d) -> (b s h w) H 1 d', H=c['heads'])
        ku = rearrange(ku, 'b s h w (H d) i j k -> (b s h w) H (i j k) d', H=c['heads'])
        dots = torch.matmul(qq, ku.transpose(-1, -2)) * m.scale
        mk = repeat(mask, '1 s h w i j k -> (b s h w) heads 1 (i j k)', b=B, heads=c['heads'])
        dots.masked_fill_(mk, -1e9)
        dots = dots.reshape(B, S, H, W, c['heads'], -1)
        npz(f'attn_core_{tag}', q=q, k=k, v=v, out=out, logits=dots,
            extents=np.array(c['ext']), heads=np.array(c['heads']))


def gen_attn_module(R):
    L3 = R['l3a']
    for tag, (heads, dh, D, ckpt) in {'a': (2, 16, 32, True), 'b': (1, 32, 32, False)}.items():
        torch.manual_seed(200 + ord(tag))
        m = L3.Local3dAttention((1, 2, 1), dim=D, heads=heads, dim_head=dh, use_checkpointing=ckpt)
        x = torch.randn(2, 3, 5, 4, D, requires_grad=True)
        q = torch.randn(2, 3, 5, 4, D, requires_grad=True)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            out = m(x, q=q)
            out.square().sum().backward()
        arrs = dict(x=x, q=q, out=out, dx=x.grad, dq=q.grad, extents=np.array((1, 2, 1)),
                    heads=np.array(heads))
        arrs.update(sd_arrays(m.state_dict()))
        arrs.update({'grad/' + n: p.grad for n, p in m.named_parameters()})
        npz(f'attn_module_{tag}', **arrs)


def gen_transformer(R):
    main = R['main']
    torch.manual_seed(300)
    ext = (1, 1, 1)
    model = main.VqVideoDiffusionModel(data_shape=(4, 5, 6), dim=32, num_classes=50, extents=ext, depth=2,
                                       dim_head=16, mlp_dim=48, heads=2)
    hidden = []
    hooks = [l[1].register_forward_hook(lambda mod, i, o: hidden.append(o))
             for l in model.transformer.layers]
    z = torch.randint(0, 51, (2, 4, 5, 6))
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        x_final = model.transformer(z)
        res = [x_final] + [h + 0 for h in hidden]  # ff outputs (before residual add)
        hidden.clear()
        logits = model(z)
        hidden.clear()
        z_short = z[:, :3]
        logits_short = model(z_short)
    for h in hooks:
        h.remove()
    arrs = dict(z=z, x_final=x_final, logits=logits, z_short=z_short, logits_short=logits_short,
                extents=np.array(ext), heads=np.array(2))
    arrs.update(sd_arrays(model.state_dict()))
    npz('transformer_tiny', **arrs)
    # a second tiny model exercising to_out == Identity (heads=1, dim_head==dim; quirk Q6)
    torch.manual_seed(301)
    model = main.VqVideoDiffusionModel(data_shape=(3, 4, 4), dim=16, num_classes=20, extents=(1, 0, 2), depth=1,
                                       dim_head=16, mlp_dim=24, heads=1)
    z = torch.randint(0, 21, (1, 3, 4, 4))
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        logits = model(z)
    arrs = dict(z=z, logits=logits, extents=np.array((1, 0, 2)), heads=np.array(1))
    arrs.update(sd_arrays(model.state_dict()))
    npz('transformer_identity_out', **arrs)


def gen_vq(R):
    VQ = R['vq']
    for C in (512, 1024, 8192):
        torch.manual_seed(400 + C)
        m = VQ.VectorQuantizerEMA(64, C)
        x = torch.randn(257, 64)
        # a genuine tie: duplicate one codebook row at a higher index and aim a sample at it
        m.embedding[0, C - 3] = m.embedding[0, 7]
        x[5] = m.embedding[0, 7] + 1e-3 * torch.randn(64)
        x[6] = m.embedding[0, 7]
        idx = m.encode(x)
        d = m.codebook_distance(x, normalize=False)
        dn = m.codebook_distance(x, normalize=True)
        dec = m.decode(idx)
        assert idx[5, 0] == 7 and idx[6, 0] == 7
        # keep the file small: store min / second-min distance and a strided slice of the full table
        top2 = torch.topk(d[:, 0], 2, dim=-1, largest=False).values
        npz(f'vq_encode_{C}', x=x, embedding=m.embedding, idx=idx, dist_min=top2[:, 0], dist_second=top2[:, 1],
            dist_rows=d[:8, 0], dist_rows_normalized=dn[:8, 0], decoded=dec)
    # odd sizes: E not a multiple of 8, C not a multiple of 64
    torch.manual_seed(450)
    m = VQ.VectorQuantizerEMA(20, 77)
    x = torch.randn(130, 20)
    npz('vq_encode_odd', x=x, embedding=m.embedding, idx=m.encode(x),
        dist=m.codebook_distance(x, normalize=False)[:, 0])


def gen_vq_forward(R):
    VQ = R['vq']
    torch.manual_seed(500)
    m = VQ.VectorQuantizerEMA(8, 32)
    arrs = {'embedding0': m.embedding.clone(), 'cluster_size0': m.cluster_size.clone()}

    def snap(tag):
        for b in ('embedding', 'cluster_size', 'activation_count', 'accumulated_error'):
            arrs[f'{tag}/{b}'] = getattr(m, b).clone()

    m.train()
    for t in range(3):
        x = (torch.randn(96, 8) * (1.0 + 0.5 * t)).requires_grad_(True)
        qz, enc, loss, ppl = m(x)
        (qz.square().sum() + 3.0 * loss).backward()
        arrs.update({f't{t}/x': x, f't{t}/quantized': qz, f't{t}/encodings_argmax': enc.argmax(-1),
                     f't{t}/loss': loss, f't{t}/perplexity': ppl, f't{t}/dx': x.grad})
        snap(f't{t}')
    m.eval()
    x = torch.randn(96, 8)
    qz, enc, loss, ppl = m(x)
    arrs.update({'e/x': x, 'e/quantized': qz, 'e/encodings_argmax': enc.argmax(-1), 'e/loss': loss,
                 'e/perplexity': ppl})
    snap('e')
    arrs['reused'] = np.array(m.reuse_inactive())
    snap('reuse')
    m.reset_stats()
    snap('reset')
    npz('vq_forward_train', **arrs)


def gen_ae(R):
    TV = R['tv']
    torch.manual_seed(600)
    model = TV.VqAutoEncoder(embedding_dim=16, num_embeddings=32, downscale_steps=2, hidden_planes=24, in_channels=3)
    # make BN affine / running stats non-trivial
    for mod in model.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.weight.data.uniform_(0.5, 1.5)
            mod.bias.data.uniform_(-0.3, 0.3)
            mod.running_mean.uniform_(-0.2, 0.2)
            mod.running_var.uniform_(0.5, 1.5)
    x = torch.rand(2, 3, 32, 32)
    arrs = dict(x=x)
    arrs.update(sd_arrays(model.state_dict(), 'sd0/'))
    model.eval()
    with torch.no_grad():
        idx = model.encode(x)
        rec = model.decode(idx)
        h = model.encoder(x)
        out, ll, ppl = model(x)
    arrs.update({'eval/idx': idx, 'eval/decoded': rec, 'eval/enc_out': h, 'eval/recon': out, 'eval/latent_loss': ll,
                 'eval/perplexity': ppl})
    model.train()
    with torch.no_grad():
        idx_t = model.encode(x)                      # Q3: BN batch statistics + running-stat update under no_grad
    arrs['train/idx'] = idx_t
    arrs.update(sd_arrays(model.state_dict(), 'sd1/'))
    xg = x.clone().requires_grad_(True)
    out, ll, ppl = model(xg)
    loss = torch.nn.functional.smooth_l1_loss(out, x) + 0.25 * ll
    loss.backward()
    arrs.update({'train/recon': out, 'train/latent_loss': ll, 'train/perplexity': ppl, 'train/loss': loss,
                 'train/dx': xg.grad})
    arrs.update({'train/grad/' + n: p.grad for n, p in model.named_parameters()})
    arrs.update(sd_arrays(model.state_dict(), 'sd2/'))
    npz('ae_roundtrip', **arrs)
    # config 1 of BASELINE.json: one 64x64 RGB frame, codebook 512, default sizes, eval mode
    torch.manual_seed(601)
    model = TV.VqAutoEncoder(embedding_dim=64, num_embeddings=512, downscale_steps=3, hidden_planes=128, in_channels=3)
    model.eval()
    x = torch.rand(1, 3, 64, 64)
    with torch.no_grad():
        idx = model.encode(x)
        rec = model.decode(idx)
    npz('ae_cfg1_meta', x=x, idx=idx, recon_mean=rec.mean(), recon_std=rec.std(), recon_corner=rec[0, :, :4, :4],
        seed=np.array(601))


def gen_step(R):
    main = R['main']
    torch.manual_seed(700)
    C, B = 32, 2
    ext = (1, 1, 1)
    model = main.VqVideoDiffusionModel(data_shape=(3, 4, 4), dim=16, num_classes=C, extents=ext, depth=2,
                                       dim_head=8, mlp_dim=24, heads=2)
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    batch_z = torch.randint(0, C, (B, 3, 4, 4))
    last = batch_z[:, -1]
    target = last.clone()
    r = torch.tensor([0.35, 0.8])
    mask_u = torch.rand(B, 16)
    mask = mask_u < r.view(B, 1)
    du = torch.ones(B, 16, C) / C
    dt = torch.nn.functional.one_hot(last.reshape(B, -1), num_classes=C).float()
    d = torch.lerp(dt, du, r.view(B, 1, 1) * 0.1)
    draw = torch.multinomial(d.view(-1, C), num_samples=1).view(B, -1)
    draw_raw = draw.clone()
    draw[mask] = C
    zc = batch_z.clone()
    zc[:, -1] = draw.view(last.shape)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, betas=(0.9, 0.999), weight_decay=1e-7, amsgrad=False)
    cos = torch.optim.lr_scheduler.CosineAnnealingLR(opt, 1000)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        sched = R['warm'].GradualWarmupScheduler(opt, multiplier=1.0, total_epoch=5, after_scheduler=cos)
        # force a non-zero lr for the pinned step: run 3 empty scheduler steps first
        lrs = [opt.param_groups[0]['lr']]
        for _ in range(3):
            opt.step()
            sched.step()
            lrs.append(opt.param_groups[0]['lr'])
        model.train()
        opt.zero_grad()
        y = model(zc)
        loss = torch.nn.functional.cross_entropy(y.reshape(-1, C), target.reshape(-1), reduction='none')
        per_sample = loss.view(B, -1).mean(dim=1)
        loss.mean().backward()
        gn = main.grad_norm(model.parameters())
        lr_used = opt.param_groups[0]['lr']
        opt.step()
        sched.step()
        # lr trajectory through warm-up into cosine
        for _ in range(8):
            lrs.append(opt.param_groups[0]['lr'])
            opt.step()
            sched.step()
    arrs = dict(batch_z=batch_z, r=r, mask_uniform=mask_u, draw=draw_raw, corrupted=zc, target=target, d_probs=d,
                logits=y, per_sample_loss=per_sample, loss=loss.mean(), grad_norm=np.array(gn), lr_used=np.array(lr_used),
                lr_trajectory=np.array(lrs), warmup=np.array(5), max_steps=np.array(1000), base_lr=np.array(1e-4),
                extents=np.array(ext), heads=np.array(2))
    arrs.update(sd_arrays(sd0, 'sd0/'))
    arrs.update({'grad/' + n: p.grad for n, p in model.named_parameters()})
    # weights after exactly ONE AdamW step from sd0 is not what `model` holds (3 empty steps with zero grads ran
    # first and only touch weight decay with lr=0..). Re-do a clean single step for the pin:
    model2 = main.VqVideoDiffusionModel(data_shape=(3, 4, 4), dim=16, num_classes=C, extents=ext, depth=2,
                                        dim_head=8, mlp_dim=24, heads=2)
    model2.load_state_dict(sd0)
    opt2 = torch.optim.AdamW(model2.parameters(), lr=3e-3, betas=(0.9, 0.999), weight_decay=1e-7, amsgrad=False)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        y2 = model2(zc)
        torch.nn.functional.cross_entropy(y2.reshape(-1, C), target.reshape(-1)).backward()
    opt2.step()
    arrs.update(sd_arrays(model2.state_dict(), 'sd1/'))
    arrs['adamw_lr'] = np.array(3e-3)
    # loss-aware sampler trace
    torch.manual_seed(701)
    s = R['isamp'].LossAwareSamplerEma(num_histogram_buckets=10, uniform_p=0.01, alpha=0.9, warmup=2)
    ts = torch.rand(64)
    ls = torch.rand(64) * 3
    s.update_with_losses(ts, ls)
    arrs.update({'sampler/ts': ts, 'sampler/losses': ls, 'sampler/weights_raw': s._weights.clone(),
                 'sampler/counts': s._counts.clone(), 'sampler/weights': s.weights(),
                 'sampler/warmed_up': np.array(bool(s.warmed_up()))})
    npz('step_tiny', **arrs)


def gen_sampler(R):
    """main.py:evaluate_model (:50-117) run by the reference itself with its two random draws replaced by injected uniform
    fields (oracle/sampler.py defines the inverse-CDF categorical draw both sides use).  Records the context tokens, every
    last frame fed to the model, the generated token frames and the decoded images."""
    import random
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import sampler as osamp
    main, TV = R['main'], R['tv']
    C, B, n_past, width, steps, iters = 16, 2, 2, 16, 2, 30
    for tag, topk in (('', -1), ('_topk', 4)):
        for seed in range(900, 960):
            torch.manual_seed(seed)
            random.seed(seed)
            ae = TV.VqAutoEncoder(embedding_dim=8, num_embeddings=C, downscale_steps=2, hidden_planes=8, in_channels=1)
            model = main.VqVideoDiffusionModel(data_shape=(n_past + 1, 4, 4), dim=16, num_classes=C, extents=(1, 1, 1),
                                               depth=2, dim_head=8, mlp_dim=24, heads=2)
            with torch.no_grad():                                   # make the logits peaked enough to matter
                model.logit_proj.weight.mul_(6.0)
            sd_model = {k: v.clone() for k, v in model.state_dict().items()}
            sd_ae = {k: v.clone() for k, v in ae.state_dict().items()}
            dataset = [np.random.RandomState(seed + j).rand(n_past + 1, width, width, 1).astype(np.float32) for j in range(5)]
            u_multi = torch.rand(steps, iters, B * 16)
            u_mask = torch.rand(steps, iters, B, 16)
            it_multi, it_mask = iter(u_multi.reshape(-1, B * 16)), iter(u_mask.reshape(-1, B, 16))
            cap = dict(z0=None, fed=[], tokens=[], margin=1.0)
            real_encode, real_decode, real_forward = ae.encode, ae.decode, model.forward

            def encode(x):
                z = real_encode(x)
                cap['z0'] = z.clone()
                return z

            def decode(z):
                cap['tokens'].append(z.clone())
                return real_decode(z)

            def forward(z):
                cap['fed'].append(z[:, -1].clone())
                return real_forward(z)

            def fake_multinomial(p, num_samples, replacement=False, **kw):
                u = next(it_multi)
                cap['margin'] = min(cap['margin'], osamp.icdf_margin(p, u))
                return osamp.multinomial_icdf(p, u).view(-1, 1)

            def fake_rand(*size, **kw):
                u = next(it_mask)
                assert tuple(u.shape) == tuple(size), (u.shape, size)
                return u

            ae.encode, ae.decode, model.forward = encode, decode, forward
            keep = torch.multinomial, torch.rand
            torch.multinomial, torch.rand = fake_multinomial, fake_rand
            try:
                with warnings.catch_warnings():
                    warnings.simplefilter('ignore')
                    images, _ = main.evaluate_model(device=torch.device('cpu'), model=model, decoder_model=ae,
                                                    num_embeddings=C, mask_token_index=C, batch_size=B, num_steps=steps,
                                                    n_past=n_past, image_width=width, dataset=dataset, sample_topk=topk)
            finally:
                torch.multinomial, torch.rand = keep
            if cap['margin'] > 2e-5:
                break
        else:
            raise RuntimeError('no seed with a safe inverse-CDF margin')
        z0 = cap['z0'].view(B, n_past + 1, 4, 4)
        arrs = dict(z0=z0, u_multi=u_multi, u_mask=u_mask, fed=torch.stack(cap['fed']), tokens=torch.stack(cap['tokens']),
                    images=images, margin=np.array(cap['margin']), seed=np.array(seed), topk=np.array(topk),
                    num_embeddings=np.array(C), extents=np.array((1, 1, 1)), heads=np.array(2))
        arrs.update(sd_arrays(sd_model, 'model/'))
        arrs.update(sd_arrays(sd_ae, 'ae0/'))                      # AE state BEFORE the run (train-mode BN mutates it, Q3)
        arrs.update(sd_arrays(ae.state_dict(), 'ae1/'))            # ... and after
        npz('sampler_tiny' + tag, **arrs)


def gen_checkpoint(R):
    """Reference-format checkpoints: the dicts train_vqae.py:168-179 and main.py:297-309 torch.save, holding
    the pickled argparse Namespace of the reference's own parsers, written after one real optimizer step each; plus what
    main.py:376-402 rebuilds from them (frozen AE in train mode -> tokens -> denoiser / EMA-denoiser logits)."""
    import copy
    main, TV = R['main'], R['tv']
    sys.modules.pop('model_ema_v2', None)
    from model_ema_v2 import ModelEmaV2
    argv = sys.argv
    try:
        sys.argv = ['train_vqae.py', '--embedding_dim', '8', '--num_embeddings', '16', '--downscale_steps', '2',
                    '--hidden_planes', '8', '--image_width', '16', '--name', 'vqvdvq_tiny', '--device', 'cpu']
        opt_ae = TV.parse_args()
        sys.argv = ['main.py', '--dim', '16', '--extents', '1,1,1', '--depth', '2', '--mlp_dim', '24', '--dim_head', '8',
                    '--heads', '2', '--n_past', '2', '--image_width', '16', '--decoder_model', 'ckpt_vqae_tiny.pth',
                    '--ema_decay', '0.9', '--name', 'vq_diffusion_tiny', '--device', 'cpu', '--batch_size', '2']
        opt = main.parse_args()
    finally:
        sys.argv = argv
    # ---- VQ-AE: train_vqae.main (:241) + one step of train (:139-150) + the save of :172-179
    torch.manual_seed(opt_ae.manual_seed)
    ae = TV.VqAutoEncoder(opt_ae.embedding_dim, opt_ae.num_embeddings, opt_ae.downscale_steps,
                          hidden_planes=opt_ae.hidden_planes, in_channels=1)
    optim_ae = torch.optim.AdamW(ae.parameters(), lr=opt_ae.lr, betas=(0.9, 0.999), weight_decay=opt_ae.weight_decay)
    sched_ae = torch.optim.lr_scheduler.StepLR(optim_ae, step_size=3, gamma=0.5)
    batch = torch.rand(8, 1, 16, 16)
    ae.train()
    rec, latent_loss, _ = ae(batch)
    loss = torch.nn.SmoothL1Loss(reduction='mean')(rec, batch) + opt_ae.latent_loss_weight * latent_loss
    optim_ae.zero_grad()
    loss.backward()
    optim_ae.step()
    torch.save({'step': 1, 'lr': sched_ae.get_last_lr(), 'model_state_dict': ae.state_dict(),
                'optimizer_state_dict': optim_ae.state_dict(), 'loss': {'train_recon_error': [loss.item()]}, 'opt': opt_ae},
               os.path.join(HERE, 'ckpt_vqae_tiny.pth'))
    # ---- denoiser: main.main (:376-402, :432-442) + one optimizer step + the save of :302-309
    torch.manual_seed(opt.manual_seed)
    decoder_data = torch.load(os.path.join(HERE, opt.decoder_model), map_location='cpu', weights_only=False)
    chk = decoder_data['opt']
    dec = TV.VqAutoEncoder(chk.embedding_dim, chk.num_embeddings, chk.downscale_steps, hidden_planes=chk.hidden_planes, in_channels=1)
    dec.load_state_dict(decoder_data['model_state_dict'])
    x = torch.rand(opt.n_past + 1, 1, 16, 16)
    z = dec.encode(x)                                                # (3, 4, 4) -> data_shape (:388-394)
    extents = [int(e) for e in opt.extents.split(',')]
    model = main.VqVideoDiffusionModel(data_shape=z.shape, dim=opt.dim, num_classes=chk.num_embeddings, extents=extents,
                                       depth=opt.depth, mlp_dim=opt.mlp_dim, dim_head=opt.dim_head, heads=opt.heads,
                                       dropout=opt.dropout)
    optim = torch.optim.AdamW(model.parameters(), lr=opt.lr, betas=(0.9, 0.999), weight_decay=opt.weight_decay, amsgrad=False)
    cos = torch.optim.lr_scheduler.CosineAnnealingLR(optim, opt.max_steps)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        sched = R['warm'].GradualWarmupScheduler(optim, multiplier=1.0, total_epoch=opt.warmup, after_scheduler=cos)
        ema = ModelEmaV2(model, decay=opt.ema_decay)
        zb = torch.randint(0, chk.num_embeddings + 1, (2, 3, 4, 4))
        for g_ in optim.param_groups:
            g_['lr'] = 1e-2                                        # warm-up starts at lr 0: force a visible step
        y = model(zb)
        torch.nn.functional.cross_entropy(y.reshape(-1, chk.num_embeddings), torch.randint(0, chk.num_embeddings, (32,))).backward()
        optim.step()
        sched.step()
        ema.update(model)
    torch.save({'step': 1, 'lr': sched.get_last_lr(), 'model_state_dict': model.state_dict(),
                'ema_model_state_dict': ema.module.state_dict(), 'optimizer_state_dict': optim.state_dict(), 'opt': opt},
               os.path.join(HERE, 'ckpt_denoiser_tiny.pth'))
    for f in ('ckpt_vqae_tiny.pth', 'ckpt_denoiser_tiny.pth'):
        print(f'{f}: {os.path.getsize(os.path.join(HERE, f)) / 1024:.1f} KB')
    # ---- what a user of main.py gets back from the two files (fresh objects, like a new process)
    dec2 = TV.VqAutoEncoder(chk.embedding_dim, chk.num_embeddings, chk.downscale_steps, hidden_planes=chk.hidden_planes, in_channels=1)
    dec2.load_state_dict(torch.load(os.path.join(HERE, 'ckpt_vqae_tiny.pth'), map_location='cpu', weights_only=False)['model_state_dict'])
    frames = torch.rand(2 * 3, 1, 16, 16)
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter('ignore')
        tokens = dec2.encode(frames).view(2, 3, 4, 4)              # train-mode BatchNorm, as main.py runs it (Q3)
        recon = dec2.decode(tokens.view(-1, 4, 4))
        logits = model(tokens)
        logits_ema = ema.module(tokens)
    npz('ckpt_tiny_expect', frames=frames, tokens=tokens, recon=recon, logits=logits, logits_ema=logits_ema,
        data_shape=np.array(tuple(z.shape)))


def gen_sparse(ref_root):
    SD = import_sparse(ref_root)
    torch.manual_seed(800)
    shape = (6, 4, 4)
    model = SD.VqSparseDiffusionModel(shape=shape, dim=32, num_classes=40, depth=2, dim_head=16, mlp_dim=48, heads=2)
    x = torch.randint(0, 41, (3, 16))
    idx = torch.stack([torch.randperm(6 * 4 * 4)[:16] for _ in range(3)])
    logits = model(x, idx)
    arrs = dict(x=x, indices=idx, logits=logits, shape=np.array(shape), heads=np.array(2))
    arrs.update(sd_arrays(model.state_dict()))
    npz('sparse_tiny', **arrs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ref', default='/root/reference')
    ap.add_argument('--only', default='')
    a = ap.parse_args()
    torch.set_num_threads(1)          # fixed reduction partitioning for reproducible fixtures
    R = import_reference(a.ref)
    gens = dict(attn_core=gen_attn_core, attn_module=gen_attn_module, transformer=gen_transformer, vq=gen_vq,
                vq_forward=gen_vq_forward, ae=gen_ae, step=gen_step, sampler=gen_sampler, checkpoint=gen_checkpoint)
    for name, fn in gens.items():
        if a.only and a.only != name:
            continue
        fn(R)
    if not a.only or a.only == 'sparse':
        gen_sparse(a.ref)


if __name__ == '__main__':
    main()
