"""MI355X drop-in for vq-video-diffusion/vq.py::VectorQuantizerEMA (:6-111).

Same constructor, buffers (embedding / cluster_size persistent; latent_offsets / activation_count /
accumulated_error non-persistent) and methods.  Nearest-neighbour search, row gather, EMA statistics and the EMA
update are HIP kernels; the argmin reproduces the reference's fp32 summation order, so indices are bit-identical.
`VectorQuantizerEMA1` of the reference is dead code (never instantiated) and is not provided.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from ._lib import WmzError


class LazyOneHot:
    """The `encodings` return value of VectorQuantizerEMA.forward (reference vq.py:39: a dense fp32 one-hot [N, L, C]):
    268 MB at N = 65 536, C = 1024, and no caller in scope reads it (train_vqae.py:38 drops it).  This stands in for the
    tensor and builds it only when something actually looks at it: any torch function or tensor method applied to it
    materialises the dense one-hot first (`.indices` gives the int64 codes -- [N] for one latent, [N, L] otherwise -- without
    materialising anything)."""

    def __init__(self, indices, num_embeddings):
        self.indices = indices                      # int64 [N] or [N, L]
        self._C = num_embeddings
        self._dense = None

    @property
    def shape(self):
        n_lat = 1 if self.indices.dim() == 1 else self.indices.shape[1]
        return torch.Size((self.indices.shape[0], n_lat, self._C))

    dtype = torch.float32

    @property
    def device(self):
        return self.indices.device

    def materialize(self):
        if self._dense is None:
            N, n_lat, C = self.shape
            self._dense = torch.zeros(N, n_lat, C, device=self.indices.device).scatter_(-1, self.indices.view(N, n_lat, 1), 1.0)
        return self._dense

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        def dense(a):
            if isinstance(a, cls):
                return a.materialize()
            if isinstance(a, (list, tuple)):
                return type(a)(dense(b) for b in a)
            return a
        return func(*dense(args), **{k: dense(v) for k, v in (kwargs or {}).items()})

    def __getattr__(self, name):                    # tensor methods / attributes: .argmax(-1), .sum(0), .size() ...
        if name.startswith('__') and name.endswith('__'):
            raise AttributeError(name)
        return getattr(self.materialize(), name)

    def __getitem__(self, item):
        return self.materialize()[item]


def _binary(name):
    def op(self, *args):
        return getattr(self.materialize(), name)(*args)
    return op


for _n in ('add', 'radd', 'sub', 'rsub', 'mul', 'rmul', 'truediv', 'rtruediv', 'matmul', 'rmatmul', 'neg', 'eq', 'ne', 'lt',
           'le', 'gt', 'ge', 'len', 'float', 'bool'):
    setattr(LazyOneHot, f'__{_n}__', _binary(f'__{_n}__'))
LazyOneHot.__hash__ = object.__hash__


class VectorQuantizerEMA(nn.Module):
    def __init__(self, embedding_dim, num_embeddings, num_latents=1, decay=0.99, eps=1e-5):
        super().__init__()
        self.embedding_dim = embedding_dim
        self.num_embeddings = num_embeddings
        self.num_latents = num_latents
        self.decay = decay
        self.eps = eps
        self.register_buffer('embedding', torch.randn(num_latents, num_embeddings, embedding_dim))
        self.register_buffer('cluster_size', torch.ones(num_latents, num_embeddings))
        self.register_buffer('latent_offsets', torch.arange(num_latents).mul(num_embeddings).unsqueeze(0), persistent=False)
        self.register_buffer('activation_count', torch.zeros(num_latents, num_embeddings), persistent=False)
        self.register_buffer('accumulated_error', torch.zeros(num_latents, num_embeddings), persistent=False)
        self.simple_update = False
        self.laplace_smoothing = True
        # data-parallel training: set to a process group (or True for the default group) to all-reduce the EMA
        # statistics before the update, otherwise codebooks diverge across ranks (comm backend row)
        self.sync_stats = None

    # ---- helpers
    def _flat(self, x):
        """input -> fp32 [N, L, E] (reference :27: `input.reshape(-1, num_latents, embedding_dim)`)."""
        if not x.is_cuda:
            raise WmzError('VectorQuantizerEMA runs on the GPU only (no CPU fallback)')
        return x.reshape(-1, self.num_latents, self.embedding_dim).float()

    def codebook_distance(self, input, normalize=True):
        """[N, L, C] distances (reference :77-82).  Diagnostic API: materialises N*L*C floats, chunked over N."""
        flat = self._flat(input)
        outs = [(flat[i:i + 8192, :, None, :] - self.embedding[None]).pow(2).sum(-1) for i in range(0, flat.shape[0], 8192)]
        d = torch.cat(outs, 0)
        return d / self.embedding_dim if normalize else d

    def encode(self, input):
        """int64 [N, L] nearest-code indices (reference :84-87); every latent searches its own codebook."""
        flat = self._flat(input)
        return torch.stack([ops.vq_argmin(flat[:, l], self.embedding[l]) for l in range(self.num_latents)], dim=1)

    def decode(self, indices):
        """codebook rows, [*indices.shape, E] (reference :89-94: with several latents the last axis of `indices` walks them)."""
        if self.num_latents == 1:
            idx = indices.reshape(-1)
            return ops.vq_gather(idx, self.embedding[0]).reshape(*indices.shape, self.embedding_dim)
        if indices.shape[-1] != self.num_latents:
            raise WmzError(f'decode: the last axis of the indices ({indices.shape[-1]}) must be num_latents = {self.num_latents}')
        idx = indices.reshape(-1, self.num_latents)
        rows = torch.stack([ops.vq_gather(idx[:, l].contiguous(), self.embedding[l]) for l in range(self.num_latents)], dim=1)
        return rows.reshape(*indices.shape, self.embedding_dim)

    def forward(self, input):
        return self.forward_fused(input, input.dtype, False)

    def forward_fused(self, input, out_dtype, pad8):
        """forward() with the straight-through tensor written in `out_dtype` and, with pad8, its channels zero-padded to a multiple
        of 8 -- what the decoder's first conv consumes (VqAutoEncoder: no cast / pad launches between the two); `input` may be the
        encoder's activation dtype (the search runs on its fp32 rows either way)."""
        flat = self._flat(input)                                              # [N, L, E]
        N, C, n_lat = flat.shape[0], self.num_embeddings, self.num_latents
        fd = flat.detach()
        idx_l, q_l, counts = [], [], torch.zeros(n_lat, C, device=flat.device)
        dw = torch.zeros(n_lat, C, self.embedding_dim, device=flat.device) if self.training else None
        for l in range(n_lat):                                                # (one latent everywhere in scope: train_vqae.py:31)
            cb = self.embedding[l]
            idx = ops.vq_argmin(fd[:, l], cb)
            idx_l.append(idx)
            q_l.append(ops.vq_gather(idx, cb))                                # before the EMA update, like :34
            ops.vq_ema_stats(fd[:, l], idx, cb, counts[l], dw[l] if dw is not None else None, self.accumulated_error[l])   # :35-36 always, :43-46
        encodings = LazyOneHot(idx_l[0] if n_lat == 1 else torch.stack(idx_l, 1), C)   # returned (:39), dense only if it is read
        local_counts = counts
        if self.training:
            if self.sync_stats is not None and torch.distributed.is_initialized():
                group = None if self.sync_stats is True else self.sync_stats
                local_counts = counts.clone()
                torch.distributed.all_reduce(counts, group=group)
                torch.distributed.all_reduce(dw, group=group)
            for l in range(n_lat):
                ops.vq_ema_update(self.embedding[l], self.cluster_size[l], self.activation_count[l], counts[l], dw[l], self.decay, self.eps)
        E = self.embedding_dim
        Ep = -(-E // 8) * 8 if pad8 else E
        if n_lat == 1 and out_dtype in (torch.float32, torch.bfloat16) and input.dtype in (torch.float32, torch.bfloat16):
            # :67 commitment loss, :70 straight-through estimator, :72-73 perplexity: two launches (ops.vq_tail), one in the backward
            quantized, commitment_loss, perplexity = ops.vq_tail(input, fd[:, 0], q_l[0], local_counts[0], out_dtype, Ep)
            return quantized, encodings, commitment_loss, perplexity
        quantized = (q_l[0] if n_lat == 1 else torch.stack(q_l, 1)).view_as(input).to(input.dtype)
        commitment_loss = F.mse_loss(quantized.detach(), input)               # :67
        quantized = input + (quantized - input).detach()                      # straight-through (:70)
        avg_probs = local_counts / N                                          # == encodings.mean(0) (:72), this rank's batch
        perplexity = torch.exp(-torch.sum(avg_probs * torch.log(avg_probs + 1e-10) / self.num_latents))
        if Ep != E:
            quantized = F.pad(quantized, (0, Ep - E))
        return quantized.to(out_dtype), encodings, commitment_loss, perplexity

    def reuse_inactive(self):
        """Host-driven, rare (every 500 steps, train_vqae.py:160-164): kept in torch (reference :96-107)."""
        total = 0
        for i in range(self.num_latents):
            dead = self.activation_count[i] == 0
            nd = int(dead.count_nonzero().item())
            if nd > 0:
                _, j = self.activation_count[i].topk(nd)
                self.embedding[i][dead] = self.embedding[i][dead] * 0.1 + self.embedding[i][j] * 0.9
                total += nd
        return total

    def reset_stats(self):
        self.activation_count.zero_()
        self.accumulated_error.zero_()
