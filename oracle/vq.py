"""Oracle (test infrastructure): VectorQuantizerEMA, fp32 CPU.

Restates vq-video-diffusion/vq.py:6-111 as functions over a dict of buffers
  state = {embedding[L,C,E], cluster_size[L,C], activation_count[L,C], accumulated_error[L,C]}
(L = num_latents; 1 everywhere in scope, train_vqae.py:31).

Distance arithmetic (vq.py:30, :79): dist[n,l,c] = sum_e (x[n,l,e]-emb[l,c,e])**2 in fp32,
evaluated with the same tensor expression as the reference (chunked over n only, which does
not change the per-(n,c) reduction), so the distances -- and hence argmin -- are bit-identical
to the reference on the same host.  Measured in the build container: ATen's CPU reduction for
this expression is the contiguous-inner `vectorized_inner_sum` (the broadcast output is laid out
E-innermost): 8 SIMD lanes x 4 interleaved accumulators,
  acc[k][j] += sq[8*(4*i+k)+j];  acc0 += leftover vectors;  t[j] = ((acc0+acc1)+acc2)+acc3;
  d = (((tail scalars + t0)+t1)+...)+t7
`distances_avx_order` below spells that order out; the HIP argmin kernel uses the same one.
"""
import torch


def new_state(embedding_dim, num_embeddings, num_latents=1, generator=None):
    """Buffers as registered in vq.py:16-20."""
    return {
        'embedding': torch.randn(num_latents, num_embeddings, embedding_dim, generator=generator),
        'cluster_size': torch.ones(num_latents, num_embeddings),
        'activation_count': torch.zeros(num_latents, num_embeddings),
        'accumulated_error': torch.zeros(num_latents, num_embeddings),
    }


def distances(x, embedding, chunk=4096):
    """codebook_distance(normalize=False) (vq.py:77-82): [N,L,C] fp32."""
    L, C, E = embedding.shape
    flat = x.reshape(-1, L, E)
    et = embedding.transpose(1, 2).unsqueeze(0)          # [1,L,E,C] (same strides as the reference)
    outs = []
    for n0 in range(0, flat.shape[0], chunk):
        f = flat[n0:n0 + chunk]
        outs.append((f.unsqueeze(-1) - et).pow(2).sum(dim=-2))
    return torch.cat(outs, dim=0) if outs else flat.new_zeros(0, L, C)


def distances_avx_order(x, embedding):
    """The explicit summation order ATen uses for `distances` on x86 (see module docstring).
    Small inputs only (materialises [N,C,E]); L must be 1.  Used by tests to pin the order the
    HIP kernel reproduces."""
    L, C, E = embedding.shape
    assert L == 1
    flat = x.reshape(-1, E)
    d = flat[:, None, :] - embedding[0][None, :, :]
    sq = d * d                                            # [N,C,E]
    V, ILP = 8, 4
    nv = E // V
    n_ilp = nv // ILP
    assert n_ilp < 16, "cascade levels of ATen's multi_row_sum not restated (E >= 512)"
    acc = [torch.zeros(sq.shape[0], C, V) for _ in range(ILP)]
    for i in range(n_ilp):
        for k in range(ILP):
            v0 = (i * ILP + k) * V
            acc[k] = acc[k] + sq[:, :, v0:v0 + V]
    t = acc[0]
    for i in range(n_ilp * ILP, nv):                      # leftover vectors join accumulator 0 first
        t = t + sq[:, :, i * V:(i + 1) * V]
    for k in range(1, ILP):
        t = t + acc[k]
    fin = torch.zeros(sq.shape[0], C)
    for e in range(nv * V, E):
        fin = fin + sq[:, :, e]
    for j in range(V):
        fin = fin + t[:, :, j]
    return fin.unsqueeze(1)


def encode(x, embedding, chunk=4096):
    """VectorQuantizerEMA.encode (vq.py:84-87): int64 [N,L]; ties -> lowest index."""
    L, C, E = embedding.shape
    flat = x.reshape(-1, L, E)
    et = embedding.transpose(1, 2).unsqueeze(0)
    outs = []
    for n0 in range(0, flat.shape[0], chunk):
        f = flat[n0:n0 + chunk]
        outs.append((f.unsqueeze(-1) - et).pow(2).sum(dim=-2).argmin(dim=-1))
    return torch.cat(outs, dim=0) if outs else torch.zeros(0, L, dtype=torch.int64)


def decode(indices, embedding):
    """VectorQuantizerEMA.decode (vq.py:89-94): gather rows, [*indices.shape, E]."""
    L, C, E = embedding.shape
    idx = indices.reshape(-1, L)
    offs = (torch.arange(L) * C).unsqueeze(0)
    rows = embedding.reshape(L * C, E)[(idx + offs).reshape(-1)]
    return rows.reshape(*indices.shape, E)


def forward(x, state, training, decay=0.99, eps=1e-5, assign=None):
    """VectorQuantizerEMA.forward (vq.py:25-75).  Mutates `state` in place exactly where the
    reference mutates its buffers (quirk Q4).  Returns (quantized, encodings, loss, perplexity);
    `quantized` carries the straight-through value (== gathered codebook rows numerically:
    x + (q - x), vq.py:70), computed with the same two fp32 roundings.
    `assign` (int64 [N, L], test knob, default None = the reference's own argmin): evaluate everything
    behind line :33 on GIVEN code assignments -- so that the gradients of a reduced-precision encoder,
    whose latents flip a few near-ties, can be compared on the same assignment."""
    emb = state['embedding']
    L, C, E = emb.shape
    flat = x.reshape(-1, L, E)
    dist = distances(flat, emb)
    idx = dist.argmin(dim=-1) if assign is None else assign.reshape(-1, L)   # :33
    quant = decode(idx, emb)                                             # :34
    err = ((quant - flat) ** 2).sum(dim=2).detach()                      # :35
    state['accumulated_error'].scatter_add_(-1, idx.t(), err.t())        # :36
    quant = quant.view_as(x)
    enc = torch.zeros_like(dist).scatter(-1, idx.unsqueeze(-1), 1)       # :39
    if training:
        with torch.no_grad():                                            # (the reference updates .data / buffers: no autograd history)
            counts = enc.sum(dim=0)                                          # :43
            state['activation_count'].add_(counts)                           # :44
            dw = enc.permute(1, 2, 0) @ flat.transpose(0, 1)                 # :46  [L,C,E]
            state['cluster_size'].mul_(decay).add_(counts, alpha=1 - decay)  # :53
            n = state['cluster_size'].sum(dim=-1, keepdim=True)              # :57
            cs = (state['cluster_size'] + eps) / (n + C * eps) * n           # :58
            dw = dw / cs.unsqueeze(-1)                                       # :64
            state['embedding'].mul_(decay).add_(dw, alpha=1 - decay)         # :65
    loss = torch.nn.functional.mse_loss(quant.detach(), x)               # :67  (gradient -> encoder only)
    st = x + (quant - x).detach()                                        # :70  straight-through estimator
    avg = enc.mean(dim=0)
    perplexity = torch.exp(-torch.sum(avg * torch.log(avg + 1e-10) / L))  # :73
    return st, enc, loss, perplexity


def reuse_inactive(state):
    """vq.py:96-107: pull never-activated codes toward the most active ones."""
    total = 0
    L = state['embedding'].shape[0]
    for i in range(L):
        dead = state['activation_count'][i] == 0
        nd = int(dead.count_nonzero())
        if nd > 0:
            _, j = state['activation_count'][i].topk(nd)
            state['embedding'][i][dead] = state['embedding'][i][dead] * 0.1 + state['embedding'][i][j] * 0.9
            total += nd
    return total


def reset_stats(state):
    """vq.py:109-111."""
    state['activation_count'].zero_()
    state['accumulated_error'].zero_()
