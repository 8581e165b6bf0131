"""CPU oracle (TEST INFRASTRUCTURE, never imported by the product path): the iterative-unmasking sampler loop of
vq-video-diffusion/main.py::evaluate_model (:50-117), restated on token grids with the randomness INJECTED.

The reference draws with torch.multinomial(p, 1, True) (:85) and torch.rand(B, w*w) (:97-100).  Here both consume
caller-supplied uniform fields, so the loop is a deterministic function of (model, context tokens, uniforms):
  * categorical draw = inverse CDF: the first class whose cumulative probability exceeds u * total  (multinomial_icdf);
  * re-mask field    = u_mask > alpha, alpha = (i + 1) / num_eval_iterations                        (:88-100).
tests/golden/make_golden.py captures `sampler_tiny.npz` by running the REFERENCE's evaluate_model with torch.multinomial
/ torch.rand replaced by these same two definitions fed from the same fields; tests/test_oracle_golden.py pins this
restatement against that capture and the GPU test demands token-for-token equality of the generated frames.
"""
import torch
import torch.nn.functional as F


def multinomial_icdf(p, u):
    """p: [R, C] non-negative weights, u: [R] in [0, 1) -> int64 [R]: #classes whose inclusive cumulative weight is
    <= u * total (i.e. the first class whose cumulative weight exceeds it), clamped to C - 1."""
    cdf = p.cumsum(dim=-1)
    x = (u.to(cdf.dtype) * cdf[:, -1]).unsqueeze(-1)
    return (cdf <= x).sum(dim=-1).clamp(max=p.shape[-1] - 1)


def icdf_margin(p, u):
    """Smallest relative distance between u * total and a CDF step: how far the draw is from flipping."""
    cdf = p.double().cumsum(dim=-1)
    x = (u.double() * cdf[:, -1]).unsqueeze(-1)
    return float(((cdf - x).abs() / cdf[:, -1:]).min())


def top_k_logits(logits, k):
    """main.py:39-43: everything below the k-th largest logit of the row becomes -inf."""
    kth = torch.topk(logits, k, largest=True, sorted=True).values[:, [-1]]
    out = logits.clone()
    out[out < kth] = -float('inf')
    return out


def evaluate_tokens(model_fn, batch_z, num_embeddings, num_steps, u_multi, u_mask, num_eval_iterations=30, sample_topk=-1,
                    noise_schedule=None, consistent_masking=False):
    """main.py:61-115 on tokens.  model_fn(batch_z[B,S,H,W] int64) -> logits [B,H,W,C];
    u_multi: [num_steps, num_eval_iterations, B*H*W], u_mask: [num_steps, num_eval_iterations, B, H*W].
    Returns (list of generated token frames [B,H,W], final batch_z, list of the last frames fed to the model)."""
    batch_z = batch_z.clone()
    B, S, H, W = batch_z.shape
    mask_token = num_embeddings
    batch_z[:, -1] = mask_token                                    # :62 destroy all information in the last frame
    frames, fed = [], []
    for step in range(num_steps):
        logits = torch.zeros(B * H * W, num_embeddings)            # :71 flat start
        last_mask = torch.ones(B, H * W, dtype=torch.bool)
        for i in range(num_eval_iterations):
            logits = logits.reshape(-1, num_embeddings)
            if sample_topk > 0:
                logits = top_k_logits(logits, sample_topk)
            p = F.softmax(logits, dim=-1)
            denoised = multinomial_icdf(p, u_multi[step, i]).view(B, H, W)
            frac = (i + 1) / num_eval_iterations
            alpha = min(max(noise_schedule(frac) if noise_schedule is not None else frac, 0), 1)
            mask = u_mask[step, i] > alpha
            if consistent_masking:
                mask = last_mask & mask
                last_mask = mask
            batch_z[:, -1] = denoised                               # :107
            last = batch_z[:, -1].reshape(B, H * W)
            last[mask] = mask_token                                 # :109
            batch_z[:, -1] = last.view(B, H, W)
            fed.append(batch_z[:, -1].clone())
            logits = model_fn(batch_z)                              # :111
        frames.append(denoised.clone())
        batch_z[:, :-1] = batch_z[:, 1:].clone()                    # :115 shift frames
    return frames, batch_z, fed
