"""MI355X drop-in for the model side of minecraft/sparse_diffusion.py (config 5): VqSparseDiffusionModel :75-111 and
the position samplers :31-72.  The training script around them (MineRL loader, wandb) is out of scope."""
import math

import torch
import torch.nn as nn

from . import functional as Fw
from .transformer import Transformer


def _ranks_of_random_keys(rows, cols, device, generator=None):
    """[rows, cols] int64: row i is a uniform random permutation of 0..cols-1 (argsort of i.i.d. uniform keys)."""
    return torch.rand(rows, cols, device=device, generator=generator).argsort(dim=1)


def sample_flat_positions(batch_size, context_length, s, h, w, device):
    """batch_size * context_length flat grid positions, laid out as consecutive passes over the s*h*w grid, every pass a
    uniform permutation without replacement (semantics of reference :31-41).  One batched argsort instead of the
    reference's randperm-per-pass loop: pass k of the flattened request is row k of the permutation matrix."""
    grid, want = s * h * w, batch_size * context_length
    passes = -(-want // grid)
    perms = _ranks_of_random_keys(passes, grid, device)
    return perms.reshape(-1)[:want].reshape(batch_size, context_length)


def time_window(context_length, s, h, w, t, o=None):
    """Frame window (first frame, number of frames) per batch item for diffusion time t in [0, 1] (reference :53-64):
    the window spans at least ceil(context_length / (h w)) frames, grows linearly with t and is placed by o in [0, 1)
    (uniform when None).  Returns int64 tensors (first, frames)."""
    t = t.reshape(-1).clamp(0, 1)
    need = -(-context_length // (h * w))                     # frames that hold context_length positions
    assert context_length > 0 and need < s
    frames = (need + t * (s - need + 1)).floor().clamp(max=s - need)
    o = torch.rand_like(t) if o is None else o.reshape(-1).to(t.device).clamp(0, 1 - 1e-5)
    first = (o * (s - frames + 1)).floor()
    return first.long(), frames.long()


def sample_time_dependent(batch_size, context_length, s, h, w, t, device, o=None):
    """context_length distinct positions per item, uniform inside the item's frame window (reference :44-72).  The
    reference draws one randperm per batch item in a Python loop; here every item ranks i.i.d. uniform keys over the
    grid, keys outside its window pushed past 1, and keeps the context_length smallest -- the same law (uniform without
    replacement inside the window), one device op for the whole batch."""
    first, frames = time_window(context_length, s, h, w, t.to(device), o)
    grid = s * h * w
    inside = torch.arange(grid, device=device).unsqueeze(0) < (frames * (h * w)).unsqueeze(1)
    keys = torch.where(inside, torch.rand(batch_size, grid, device=device), torch.full((), 2.0, device=device))
    picked = keys.topk(context_length, dim=1, largest=False).indices
    return picked + (first * (h * w)).unsqueeze(1)


class VqSparseDiffusionModel(nn.Module):
    """tokens [B,n] (vocabulary num_classes + 1) at flat grid positions `indices` [B,n] -> logits [B,n,num_classes]."""

    def __init__(self, *, shape, dim, num_classes, depth, dim_head, mlp_dim, heads=1, dropout=0.0):
        super().__init__()
        self.shape = shape
        S, H, W = shape
        self.pos_emb_s = nn.Embedding(S, dim)
        self.pos_emb_h = nn.Embedding(H, dim)
        self.pos_emb_w = nn.Embedding(W, dim)
        self.embedding = nn.Embedding(num_classes + 1, dim)
        self.transformer = Transformer(dim=dim, depth=depth, heads=heads, dim_head=dim_head, mlp_dim=mlp_dim,
                                       dropout=dropout)
        self.logit_proj = nn.Linear(dim, num_classes)

    def pos_embedding_3d(self, indices):
        """fp32 position embedding of flat indices (reference :101-105); inspection helper, torch ops."""
        _, H, W = self.shape
        plane, in_plane = torch.div(indices, H * W, rounding_mode='floor'), indices.remainder(H * W)
        row, col = torch.div(in_plane, W, rounding_mode='floor'), in_plane.remainder(W)
        tables = (self.pos_emb_s.weight, self.pos_emb_h.weight, self.pos_emb_w.weight)
        return sum(tab[ix] for tab, ix in zip(tables, (plane, row, col)))

    def forward(self, x, indices):
        h = Fw.embed_tokens_indexed(x, indices, self.embedding.weight, self.pos_emb_s.weight, self.pos_emb_h.weight,
                                    self.pos_emb_w.weight, self.shape)
        h = self.transformer.forward_compute(h)
        return Fw.linear(h, self.logit_proj.weight, self.logit_proj.bias, out_f32=True)


# ------------------------------------------------------------------------------------------------ the sampler (reference :99-202)
def categorical_scatter(logits, indices, z_flat, seed, call_id=0, rank=0, counter=None):
    """One multinomial draw per row from softmax(logits) (reference :190-194), written into the clips at the rows' positions
    (:197 `full_z_flat.scatter_`): logits fp32 [B, n, C], indices int64 [B, n], z_flat int64 [B, G] (in place).  The draw's
    Philox stream: (seed, rank, call_id) -- or, with counter (device int64 [1]), the call number read on the device."""
    from . import _lib as L
    B, n, C = logits.shape
    assert logits.dtype == torch.float32 and logits.is_contiguous() and indices.is_contiguous() and z_flat.stride(1) == 1
    sid = (int(rank) << 40) | (0 if counter is not None else int(call_id) & ((1 << 40) - 1))
    L.call('wmz_categorical_scatter', L.ptr(logits), C, B * n, C, L.ptr(indices), L.ptr(z_flat), z_flat.stride(0), n, None,
           int(seed) & 0xFFFFFFFFFFFFFFFF, sid, L.ptr(counter), L.stream())


@torch.no_grad()
def sample_clips(model, batch_size, num_embeddings, sampling_type='neighbors', num_context=512, num_eval_iterations=100,
                 generator=None, seed=None, use_graph=True):
    """The token side of the reference's evaluate_model (:139-202): clips [batch_size, S, H, W] that start fully masked and are
    re-drawn num_eval_iterations times -- per iteration G // num_context + 1 contexts, each gathered from the clip, masked with
    probability 1 - i / (iterations - 1), denoised by the model, sampled from softmax(logits) and scattered back.  'neighbors' (the
    reference's default): a context = num_context positions of a frame window of width t = 1 - frac placed at the k-th of the
    iteration's shuffled offsets (sample_time_dependent with o given) -- window, positions, gather and masking are ONE launch
    (wmz_sparse_draw_context with p_uniform = 0), the draw and the scatter another (wmz_categorical_scatter); nothing returns to
    the host inside the loop, and with use_graph the whole sub-step (draw, forward, sample, scatter; the call number counted on
    the device) is ONE hipGraph launch -- same tokens as the eager launches, bit for bit.  'uniform': consecutive num_context-wide slices of one permutation per iteration (the reference
    slices at k * max_index, which is empty from k = 1 on; the evident intent -- k * num_context -- is what runs here)."""
    from . import _lib as L
    from .train import draw_sparse_context
    if sampling_type not in ('uniform', 'neighbors'):
        raise ValueError('Specified sampling_type not supported')                 # reference :182
    if num_eval_iterations < 2:
        raise ValueError('num_eval_iterations must be at least 2 (the schedule divides by iterations - 1)')
    S, H, W = (int(v) for v in model.shape)
    G = S * H * W
    dev = model.embedding.weight.device
    n = int(num_context)
    full_z = torch.full((batch_size, S, H, W), int(num_embeddings), dtype=torch.int64, device=dev)      # all mask tokens (:153-155)
    flat = full_z.view(batch_size, -1)
    if seed is None:
        seed = generator.initial_seed() if generator is not None else torch.initial_seed()
    fused = sampling_type == 'neighbors' and bool(L.lib().wmz_sparse_draw_context_supported(S, H * W, n))
    offset_count = G // n + 1
    was_training = model.training
    model.eval()
    graph = None
    if fused and use_graph:
        # one captured sub-step on static inputs: the noise level / window width r, the window placement o, the call counter
        r_s = torch.ones(batch_size, dtype=torch.float32, device=dev)
        o_s = torch.zeros(batch_size, dtype=torch.float32, device=dev)
        ctr = torch.zeros(1, dtype=torch.int64, device=dev)

        def substep():
            ctr.add_(1)
            indices, tokens, _ = draw_sparse_context(full_z, r_s, n, (S, H, W), num_embeddings, seed=seed, rank=0, counter=ctr, o=o_s,
                                                     p_uniform=0.0)
            logits = model(tokens, indices)
            categorical_scatter(logits.float().contiguous(), indices, flat, seed, counter=ctr)
        from . import config as _cfg
        side = _cfg.shared_stream('warmup')
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            substep()                                   # (allocations and operand caches settle outside the capture)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        from .graph import capture_mode
        with torch.cuda.graph(graph, capture_error_mode=capture_mode()):
            substep()
        full_z.fill_(int(num_embeddings))               # the warm-up wrote tokens: start from all masks, call number 0 (+ bit 38)
        ctr.fill_(1 << 38)
    try:
        for i in range(num_eval_iterations):
            frac = i / (num_eval_iterations - 1)
            order = torch.randperm(offset_count, generator=generator)                                    # :168 (host)
            if sampling_type == 'uniform':
                perm = torch.randperm(G, device=dev)
            r = torch.full((batch_size,), 1.0 - frac, dtype=torch.float32, device=dev)
            o_rows = (order.float() / (offset_count - 1)).to(dev)[:, None].expand(-1, batch_size).contiguous()
            if graph is not None:
                r_s.copy_(r)
            for k in range(offset_count):
                # the draw's Philox stream id: a function of the seed and the step (bit 38: apart from the trainers' eager call numbers)
                call = (1 << 38) + i * offset_count + k + 1
                if graph is not None:
                    o_s.copy_(o_rows[k])
                    graph.replay()
                    continue
                if sampling_type == 'uniform':
                    idx = perm[k * n:(k + 1) * n]
                    if idx.numel() == 0:
                        continue
                    indices = idx.unsqueeze(0).expand(batch_size, -1).contiguous()
                    tokens = torch.gather(flat, 1, indices)
                    tokens = torch.where(torch.rand(tokens.shape, device=dev) > frac, torch.full_like(tokens, num_embeddings), tokens)
                elif fused:
                    indices, tokens, _ = draw_sparse_context(full_z, r, n, (S, H, W), num_embeddings, seed=seed, rank=0, o=o_rows[k],
                                                             p_uniform=0.0, call_id=call)
                else:
                    indices = sample_time_dependent(batch_size, n, S, H, W, r, dev, o=o_rows[k])
                    tokens = torch.gather(flat, 1, indices)
                    tokens = torch.where(torch.rand(tokens.shape, device=dev) > frac, torch.full_like(tokens, num_embeddings), tokens)
                logits = model(tokens, indices)                                                         # [B, n, C] fp32
                categorical_scatter(logits.float().contiguous(), indices, flat, seed, call)
    finally:
        model.train(was_training)
    return full_z


def decode(decoder_model, batch, decode_N=16):
    """Token clips [B, S, H, W] -> frames [B, S, C, h, w] through the VQ auto-encoder, decode_N frames at a time (reference
    :115-136; tokens beyond the codebook -- left-over mask tokens -- decode as code 0)."""
    batch = batch.clone()
    batch[batch >= decoder_model.vq.num_embeddings] = 0
    shape = batch.shape
    flat = batch.view(-1, shape[2], shape[3])
    frames = torch.cat([decoder_model.decode(flat[i:i + decode_N]) for i in range(0, flat.shape[0], decode_N)])
    return frames.view(shape[0], -1, *frames.shape[1:])


@torch.no_grad()
def evaluate_model(device, batch_size, model, decoder_model, shape, sampling_type, num_context=512, num_eval_iterations=100):
    """The reference's evaluate_model (:139-202), same signature: sampled clips decoded to frames."""
    assert tuple(int(v) for v in shape) == tuple(int(v) for v in model.shape)
    z = sample_clips(model, batch_size, decoder_model.vq.num_embeddings, sampling_type, num_context, num_eval_iterations)
    return decode(decoder_model, z)
