"""Oracle (test infrastructure): the denoiser training-step body, fp32 CPU.

Restates the step body of vq-video-diffusion/main.py:train (:216-287) with every random draw
injected as an input, plus grad_norm (:188-193), AdamW as configured at :432-433 and the
warm-up + cosine schedule of :441-442 / warmup_scheduler.py:27-40, and the loss-aware noise
level sampler of importance_sampling.py:5-47.  Gradients come from torch.autograd over the
oracle's own functional forward (oracle/denoiser.py).
"""
import math
import torch
import torch.nn.functional as F

from .denoiser import denoiser_forward

P_MAX_UNIFORM = 0.1  # main.py:208


def corruption_probs(z_last, r, num_embeddings):
    """d = lerp(one_hot(z), 1/C, 0.1*r) (main.py:250-252): [B, HW, C] categorical weights."""
    B = z_last.shape[0]
    enc = z_last.reshape(B, -1)
    du = torch.ones(B, enc.shape[1], num_embeddings) / num_embeddings
    dt = F.one_hot(enc, num_classes=num_embeddings).float()
    return torch.lerp(dt, du, r.view(B, 1, 1) * P_MAX_UNIFORM)


def corrupt_last_frame(batch_z, r, mask_uniform, draw, num_embeddings):
    """main.py:240-259 with randomness injected.

    r: [B] noise level; mask_uniform: [B,HW] the `torch.rand` field of :249; draw: [B,HW] the
    `torch.multinomial` result of :254.  Returns (corrupted batch_z, target)."""
    B = batch_z.shape[0]
    target = batch_z[:, -1].clone()
    mask = mask_uniform < r.view(B, 1)
    draw = draw.clone().view(B, -1)
    draw[mask] = num_embeddings                       # mask token id == C (:211, :257)
    out = batch_z.clone()
    out[:, -1] = draw.view(target.shape)
    return out, target


def resample_tokens(z_last, r, u_switch, u_token, num_embeddings):
    """Distribution-equivalent closed form of `multinomial(lerp(one_hot, 1/C, a))`, a = 0.1*r:
    with probability a redraw uniformly over all C codes, else keep the token.  This is the
    form the fused HIP corruption kernel evaluates; equality of distributions
    with `corruption_probs` is checked in tests."""
    B = z_last.shape[0]
    a = (r.view(B, 1) * P_MAX_UNIFORM)
    flat = z_last.reshape(B, -1)
    uni = torch.clamp((u_token * num_embeddings).long(), max=num_embeddings - 1)
    return torch.where(u_switch < a, uni, flat)


def step_loss(params, batch_z, target, extents, heads):
    """Forward + CrossEntropyLoss(reduction='none') (main.py:266-274).
    Returns (logits, per_sample_loss[B], mean loss)."""
    y = denoiser_forward(params, batch_z, extents, heads)
    C = y.shape[-1]
    loss = F.cross_entropy(y.reshape(-1, C), target.reshape(-1), reduction='none')
    per_sample = loss.view(batch_z.shape[0], -1).mean(dim=1)
    return y, per_sample, loss.mean()


def step_grads(params, batch_z, target, extents, heads):
    """loss.backward() of main.py:278 -> {name: grad}."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    y, per_sample, loss = step_loss(leaves, batch_z, target, extents, heads)
    loss.backward()
    grads = {k: v.grad for k, v in leaves.items()}
    return y.detach(), per_sample.detach(), loss.detach(), grads


def grad_norm(grads):
    """main.py:188-193: sqrt(sum_p sum(grad**2)), accumulated in Python float."""
    sq = 0.0
    for g in grads.values():
        sq += (g ** 2).sum().item()
    return math.sqrt(sq)


def adamw_step(params, grads, opt_state, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-7):
    """torch.optim.AdamW semantics as configured at main.py:433 (amsgrad=False), in place."""
    b1, b2 = betas
    for k, p in params.items():
        g = grads[k]
        st = opt_state.setdefault(k, {'step': 0, 'm': torch.zeros_like(p), 'v': torch.zeros_like(p)})
        st['step'] += 1
        t = st['step']
        p.mul_(1 - lr * weight_decay)
        st['m'].mul_(b1).add_(g, alpha=1 - b1)
        st['v'].mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1 = 1 - b1 ** t
        bc2 = 1 - b2 ** t
        denom = (st['v'].sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(st['m'], denom, value=-lr / bc1)


def lr_at(step, base_lr, warmup, max_steps):
    """Learning rate used BY optimizer step number `step` (1-based) under
    GradualWarmupScheduler(multiplier=1, total_epoch=warmup, CosineAnnealingLR(T_max=max_steps))
    (main.py:441-442; warmup_scheduler.py:27-40, :58-66).  The scheduler is stepped once at
    construction, so optimizer step n runs with last_epoch = n-1."""
    e = step - 1
    if e <= warmup:
        return base_lr * (float(e) / warmup)
    t = e - warmup - 1  # cosine scheduler steps taken after `finished` flips
    return 0.5 * base_lr * (1 + math.cos(math.pi * max(t, 0) / max_steps))


class LossAwareSampler:
    """importance_sampling.py:5-47 with the multinomial / rand draws injectable."""

    def __init__(self, buckets=100, uniform_p=0.01, alpha=0.9, warmup=10):
        self.n, self.uniform_p, self.alpha, self.warmup = buckets, uniform_p, alpha, warmup
        self.w = torch.ones(buckets)
        self.counts = torch.zeros(buckets, dtype=torch.long)

    def warmed_up(self):
        return bool((self.counts > self.warmup).all())

    def weights(self):
        if not self.warmed_up():
            return torch.ones(self.n)
        w = self.w / self.w.sum()
        return (1 - self.uniform_p) * w + self.uniform_p / self.n

    def sample_from(self, bucket_idx, jitter_u):
        """:25-31 given the drawn bucket indices and the jitter uniforms."""
        return (bucket_idx.float() + jitter_u) / self.n

    def update(self, ts, losses):
        idx = (ts.view(-1) * self.n).long().clamp(0, self.n - 1)
        self.counts.scatter_add_(0, idx, torch.ones_like(idx))
        for i, j in enumerate(idx.tolist()):
            self.w[j] = self.w[j] * self.alpha + float(losses.view(-1)[i]) * (1 - self.alpha)
