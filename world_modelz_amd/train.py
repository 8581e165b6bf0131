"""Build-owned counterpart of the step body of vq-video-diffusion/main.py:train (:216-287) for the MI355X path.

The reference's Python never travels; this restates the step with the same semantics:
token corruption (:246-259), CrossEntropyLoss(reduction='none') on the last frame (:266-274), loss-aware noise-level
sampler update (:271-272), grad_norm (:188-193), AdamW + warm-up/cosine schedule (:432-442) -- with the host syncs
removed: one flat fp32 arena for parameters / gradients / moments, ONE grad-norm launch, ONE AdamW launch, and the
gradient all-reduce of parallel.py overlapped with the backward.
"""
import math

import torch
import torch.nn.functional as F

from . import _cast, ops
from . import _lib as L
from .parallel import BucketedAllReduce, FlatArena, broadcast_parameters, layer_bucket_key

P_MAX_UNIFORM = 0.1     # main.py:208


_corrupt_calls = 0


def _dp_rank():
    d = torch.distributed
    return d.get_rank() if d.is_available() and d.is_initialized() else 0


def corrupt_last_frame(batch_z, r, num_embeddings, generator=None, seed=None, rank=None, counter=None):
    """main.py:240-259 on the GPU without the [B,HW,C] one-hot / lerp / multinomial temporaries: ONE kernel.

    multinomial(lerp(one_hot(z), 1/C, a)) with a = 0.1 r has the closed form "with probability a redraw uniformly
    over all C codes, else keep z" (checked against the reference's categorical law in tests); then positions with
    rand < r become the mask token C.  Same distribution as the reference, not the same RNG stream (in-kernel
    Philox keyed by `seed` -- default torch's initial seed -- and a stream id made of the data-parallel rank and a
    per-process call counter, so that ranks seeded alike still draw different masks for their different clips)."""
    global _corrupt_calls
    assert batch_z.is_cuda and batch_z.dtype == torch.int64
    B, S = batch_z.shape[:2]
    HW = batch_z[0, 0].numel()
    src = batch_z.contiguous()                         # the kernel indexes [B, S, HW] densely: never keep foreign strides
    out = src.clone()
    target = torch.empty((B,) + tuple(batch_z.shape[2:]), dtype=torch.int64, device=batch_z.device)
    r = r.to(batch_z.device, torch.float32).contiguous()
    if seed is None:
        seed = generator.initial_seed() if generator is not None else torch.initial_seed()
    rank = _dp_rank() if rank is None else int(rank)
    if counter is not None:
        # hipGraph replay: the per-call part of the stream id is a device counter (uint64 in an int64 tensor) that the
        # caller advances inside the graph
        L.call('wmz_corrupt_tokens_dev', src.data_ptr() + (S - 1) * HW * 8, S * HW, L.ptr(r), out.data_ptr() + (S - 1) * HW * 8,
               S * HW, L.ptr(target), B, HW, int(num_embeddings), int(seed) & 0xFFFFFFFFFFFFFFFF, rank << 40, L.ptr(counter),
               L.stream())
        return out, target
    _corrupt_calls += 1
    stream_id = (rank << 40) | (_corrupt_calls & ((1 << 40) - 1))
    L.call('wmz_corrupt_tokens', src.data_ptr() + (S - 1) * HW * 8, S * HW, L.ptr(r), out.data_ptr() + (S - 1) * HW * 8,
           S * HW, L.ptr(target), B, HW, int(num_embeddings), int(seed) & 0xFFFFFFFFFFFFFFFF, stream_id, L.stream())
    return out, target


def corrupt_tokens(tokens, r, num_embeddings, generator=None, seed=None, rank=None, counter=None):
    """The same corruption law on a [B, n] token matrix (minecraft/sparse_diffusion.py:440-449: every gathered context token
    is perturbed).  Returns (corrupted [B, n], target [B, n]).  counter: device int64 [1] holding the per-call part of the Philox
    stream id (hipGraph replays: the caller advances it inside the graph)."""
    global _corrupt_calls
    assert tokens.is_cuda and tokens.dtype == torch.int64 and tokens.dim() == 2
    B, n = tokens.shape
    src = tokens.contiguous()
    out = torch.empty_like(src)
    target = torch.empty_like(src)
    r = r.reshape(-1).to(tokens.device, torch.float32).contiguous()
    if seed is None:
        seed = generator.initial_seed() if generator is not None else torch.initial_seed()
    rank = _dp_rank() if rank is None else int(rank)
    if counter is not None:
        L.call('wmz_corrupt_tokens_dev', L.ptr(src), n, L.ptr(r), L.ptr(out), n, L.ptr(target), B, n, int(num_embeddings),
               int(seed) & 0xFFFFFFFFFFFFFFFF, rank << 40, L.ptr(counter), L.stream())
        return out, target
    _corrupt_calls += 1
    L.call('wmz_corrupt_tokens', L.ptr(src), n, L.ptr(r), L.ptr(out), n, L.ptr(target), B, n, int(num_embeddings),
           int(seed) & 0xFFFFFFFFFFFFFFFF, (rank << 40) | (_corrupt_calls & ((1 << 40) - 1)), L.stream())
    return out, target


def draw_sparse_context(z, r, num_context, shape, num_embeddings, generator=None, seed=None, rank=None, counter=None, o=None,
                        p_uniform=0.1, call_id=None):
    """Config 5's step prologue by ONE launch (wmz_sparse_draw_context: minecraft/sparse_diffusion.py:44-72 sample_time_dependent,
    :437 gather, :440-449 perturbation & masking): z [B, S, H, W] token clips, r [B] noise levels -> (indices, corrupted tokens,
    target), each [B, num_context].  o [B]: the window placements (else drawn in the kernel).  p_uniform: the redraw probability
    per unit of r (:447 p_max_uniform = 0.1; the sampler masks only: 0).  Random stream as corrupt_tokens:
    (seed, rank, per-call counter -- on the device when `counter` is given: hipGraph replays)."""
    global _corrupt_calls
    assert z.is_cuda and z.dtype == torch.int64
    B = z.shape[0]
    S, H, W = (int(v) for v in shape)
    zf = z.reshape(B, -1)
    assert zf.shape[1] == S * H * W and zf.stride(1) == 1
    n = int(num_context)
    dev = z.device
    indices = torch.empty((B, n), dtype=torch.int64, device=dev)
    tokens, target = torch.empty_like(indices), torch.empty_like(indices)
    r = r.reshape(-1).to(dev, torch.float32).contiguous()
    o = None if o is None else o.reshape(-1).to(dev, torch.float32).contiguous()
    if seed is None:
        seed = generator.initial_seed() if generator is not None else torch.initial_seed()
    rank = _dp_rank() if rank is None else int(rank)
    if counter is not None:
        stream_id = rank << 40
    elif call_id is not None:                     # (the caller numbers its own calls: a reproducible sequence for one seed)
        stream_id = (rank << 40) | (int(call_id) & ((1 << 40) - 1))
    else:
        _corrupt_calls += 1
        stream_id = (rank << 40) | (_corrupt_calls & ((1 << 40) - 1))
    L.call('wmz_sparse_draw_context', L.ptr(zf), zf.stride(0), L.ptr(r), L.ptr(o), L.ptr(indices), L.ptr(tokens), L.ptr(target),
           B, S, H * W, n, int(num_embeddings), float(p_uniform), int(seed) & 0xFFFFFFFFFFFFFFFF, stream_id, L.ptr(counter), L.stream())
    return indices, tokens, target


from .graph import gc_quiet as _gc_quiet

CE_DIRECT = True      # development knobs of _LinearCrossEntropy (tools/time_train_step.py): parameter gradients straight into the arena;
CE_SIDE = None        # ... and, in a captured step, on the weight-gradient side branch (None: as the caller asks)


class _LinearCrossEntropy(torch.autograd.Function):
    """mean CrossEntropy(x W^T + b, target) WITHOUT the [R, C] logits in memory (config 5: R = 3072 rows per GPU x C = 8192
    classes x fp32 = 100 MB, 800 MB for the global batch on one device): the rows are walked in chunks whose logits stay
    cache-resident -- GEMM (fp32 out) -> row log-sum-exp and loss -> softmax - one_hot (already scaled by 1/R, bf16) -> its
    dgrad and wgrad -- so the gradient w.r.t. x, W and b is complete when the forward returns; backward() only scales it."""

    @staticmethod
    def forward(ctx, x, w, b, target, chunk, grad_scale, grad_on, side_branch):
        from . import backward as Bk
        R, K = x.shape
        C = w.shape[0]
        dt = x.dtype
        Cp = -(-C // 8) * 8
        if Cp != C:
            # a class count that is no multiple of 8 (the GEMMs' 16-byte granule: the reference takes any --num_embeddings): the
            # operands carry Cp - C padding classes -- zero weight rows and a bias of -1e30, whose probability is exactly 0, so
            # loss, log-sum-exp and every gradient of the real classes are untouched and the padding's gradient rows are zero
            pad = Cp - C
            w_c = _cast.operand((w,), dt, 'ce_pad', lambda a: F.pad(a, (0, 0, 0, pad)))
            wT = _cast.operand((w,), dt, 'ce_padT', lambda a: F.pad(a, (0, 0, 0, pad)).t())
            b_c = (_cast.operand((b,), torch.float32, 'ce_padb', lambda a: F.pad(a, (0, pad), value=-1e30)) if b is not None
                   else torch.cat([torch.zeros(C, device=x.device), torch.full((pad,), -1e30, device=x.device)]))
        else:
            w_c, wT = _cast.operand(w, dt), Bk._wt(w, dt, 'wT')
            b_c = None if b is None else b.detach()
        loss = torch.empty(R, dtype=torch.float32, device=x.device)
        dx = torch.empty_like(x) if R > chunk else None
        wbuf, bbuf = getattr(w, '_wmz_grad', None), getattr(b, '_wmz_grad', None) if b is not None else None
        # (the arena shortcut is only taken when a backward will follow: under no_grad, or for callers that do not ask for the
        #  parameter gradients, nothing may be accumulated into the arena behind autograd's back.  grad_on is the CALLER's grad mode:
        #  inside Function.forward autograd has switched it off -- round 6: read here, it was always False and the shortcut never
        #  taken: zero fills of [C, K] and [C], three scalings and two accumulations per step, 12 graph nodes at config 5)
        direct = (CE_DIRECT and wbuf is not None and (b is None or bbuf is not None) and bool(grad_on) and ctx.needs_input_grad[1]
                  and (b is None or ctx.needs_input_grad[2]) and Cp == C)
        dw = wbuf if direct else torch.zeros((Cp, K), dtype=torch.float32, device=x.device)
        db = (bbuf if direct else torch.zeros(Cp, dtype=torch.float32, device=x.device)) if b is not None else None
        # d(grad_scale * mean loss) / d(row loss): the scale of gradient accumulation (main.py:274-278) is applied HERE, the
        # parameter gradients are final when forward() returns
        ones = torch.full((min(chunk, R),), float(grad_scale) / R, dtype=torch.float32, device=x.device)
        target = target.contiguous()
        for r0 in range(0, R, chunk):
            r1 = min(R, r0 + chunk)
            xs = x[r0:r1]
            lg = ops.linear_fwd(xs, w_c, bias=b_c, out_f32=True)
            lse = torch.empty(r1 - r0, dtype=torch.float32, device=x.device)
            d = torch.empty((r1 - r0, Cp), dtype=dt, device=x.device)
            L.call('wmz_ce_fwd_bwd', L.ptr(lg), lg.stride(0), L.ptr(target[r0:r1]), L.ptr(loss[r0:r1]), L.ptr(lse), L.ptr(ones),
                   L.ptr(d), r1 - r0, Cp, L.dtype_code(dt), L.stream())
            if r0 == 0 and r1 == R:
                dx = ops.linear_dgrad(d, wT)               # one chunk (the denoiser's last frame): no staging copy
            else:
                dx[r0:r1] = ops.linear_dgrad(d, wT)
            if direct and (side_branch if CE_SIDE is None else CE_SIDE):
                # straight into the arena, and -- in a captured step -- on the weight-gradient side branch: nothing of the forward
                # or of the backward's chain reads it (joined when the backward pass ends: backward() below asks for that).  The
                # caller's choice: it pays where the step is a chain of small launches with a side branch of its own (config 5:
                # 2.94 -> 2.79 ms); beside chip-filling launches the fork and join cost more than they hide (config 4: 1.79 -> 1.91)
                with ops.arena_fill():
                    ops.linear_wgrad(d, xs, dw, db, join_later=True)
            else:
                ops.linear_wgrad(d, xs, dw, db)
        if Cp != C:
            dw = dw[:C]
            db = db[:C] if db is not None else None
        ctx.direct = direct
        ctx.params = (w, b)
        ctx.save_for_backward(dx, None if direct else dw, None if (direct or db is None) else db)
        ctx.mark_non_differentiable(loss)
        return loss.mean(), loss

    @staticmethod
    def backward(ctx, g, _g_rows):
        dx, dw, db = ctx.saved_tensors
        w, b = ctx.params
        if ctx.direct:
            ops.wgrad_join_at_end()
            # the arena already holds grad_scale * d(mean loss)/dW, and dx is scaled the same way: `mean.backward()` is called
            # as it is on this path (g == 1 by contract: a scale belongs in grad_scale, where it reaches every gradient)
            for p in (w, b):
                ready = getattr(p, '_wmz_ready', None) if p is not None else None
                if ready is not None:
                    ready()
            return dx, None, None, None, None, None, None, None
        return dx * g.to(dx.dtype), dw * g, (db * g if db is not None else None), None, None, None, None, None


def linear_cross_entropy(x, w, b, target, chunk=1024, grad_scale=1.0, side_branch=False):
    """(mean loss, per-row loss [R], detached) of CrossEntropy(x W^T + b, target) over rows x: [R, K].  The gradients it
    produces are those of grad_scale * mean loss (gradient accumulation): call `.backward()` on the returned mean as it is.
    side_branch: in a captured step the parameter gradients leave on the weight-gradient side stream (ops.side_branch)."""
    return _LinearCrossEntropy.apply(x, w, b, target, chunk, grad_scale, torch.is_grad_enabled(), bool(side_branch))


class _CrossEntropyRows(torch.autograd.Function):
    """CrossEntropyLoss(reduction='none') (main.py:268, :444) on fp32 logits; the gradient leaves in the GEMM dtype."""

    @staticmethod
    def forward(ctx, logits, target, grad_dtype):
        R, C = logits.shape
        assert logits.dtype == torch.float32 and logits.stride(1) == 1
        loss = torch.empty(R, dtype=torch.float32, device=logits.device)
        lse = torch.empty(R, dtype=torch.float32, device=logits.device)
        target = target.contiguous()
        L.call('wmz_ce_fwd', L.ptr(logits), logits.stride(0), L.ptr(target), L.ptr(loss), L.ptr(lse), R, C, L.stream())
        ctx.save_for_backward(logits, target, lse)
        ctx.grad_dtype = grad_dtype
        return loss

    @staticmethod
    def backward(ctx, dloss):
        logits, target, lse = ctx.saved_tensors
        R, C = logits.shape
        d = torch.empty((R, C), dtype=ctx.grad_dtype, device=logits.device)
        L.call('wmz_ce_bwd', L.ptr(logits), logits.stride(0), L.ptr(target), L.ptr(lse), L.ptr(dloss.float().contiguous()),
               L.ptr(d), R, C, L.dtype_code(ctx.grad_dtype), L.stream())
        return d, None, None


def cross_entropy_rows(logits, target):
    """Per-row CE on fp32 logits.  (autograd hands a gradient back in the dtype of `logits`, so the fused backward
    writes fp32 here; wmz_ce_bwd can emit bf16 for callers that fuse it with the projection's backward.)"""
    return _CrossEntropyRows.apply(logits, target, torch.float32)


class LossAwareSamplerEma:
    """importance_sampling.py:5-47 (100-bucket EMA-of-loss histogram over the noise level), host side."""

    def __init__(self, num_histogram_buckets=100, uniform_p=0.01, alpha=0.9, warmup=10, jitter=True):
        assert num_histogram_buckets > 1
        self.n, self.uniform_p, self.alpha, self.warmup, self.jitter = num_histogram_buckets, uniform_p, alpha, warmup, jitter
        self._weights = torch.ones(self.n)
        self._counts = torch.zeros(self.n, dtype=torch.long)

    def warmed_up(self):
        return bool((self._counts > self.warmup).all())

    def weights(self):
        if not self.warmed_up():
            return torch.ones(self.n)
        w = self._weights / self._weights.sum()
        return (1 - self.uniform_p) * w + self.uniform_p / self.n

    def sample(self, batch_size, generator=None):
        w = torch.multinomial(self.weights(), batch_size, replacement=True, generator=generator).float()
        if self.jitter:
            return (w + torch.rand(w.shape, generator=generator)) / self.n
        return w / (self.n - 1)

    def update_with_losses(self, ts, losses):
        # On numpy views of the two state tensors, in the reference's float32 arithmetic and order (a bucket hit twice in one batch
        # is updated twice).  This runs between a step's read-back and the next step's launch, with the GPU idle: as element-wise
        # torch indexing it cost 140 us per step at 6 samples (tools/time_replay_host.py), 5 % of config 5's step.
        import numpy as np
        ts = ts.detach().reshape(-1).to('cpu', torch.float32).numpy()
        ls = losses.detach().reshape(-1).to('cpu', torch.float32).tolist()
        idx = np.clip((ts * np.float32(self.n)).astype(np.int64), 0, self.n - 1).tolist()      # (.long(): truncation, like astype)
        w, c = self._weights.numpy(), self._counts.numpy()
        a = np.float32(self.alpha)
        for j, l in zip(idx, ls):
            c[j] += 1
            w[j] = w[j] * a + np.float32(l * (1 - self.alpha))


def lr_at(step, base_lr, warmup, max_steps):
    """Learning rate of optimizer step `step` (1-based) under GradualWarmupScheduler(multiplier 1, total_epoch=warmup)
    handing over to CosineAnnealingLR(T_max=max_steps) (main.py:441-442)."""
    e = step - 1
    if e <= warmup:
        return base_lr * (float(e) / warmup) if warmup > 0 else base_lr
    return 0.5 * base_lr * (1 + math.cos(math.pi * max(e - warmup - 1, 0) / max_steps))


class _AdamState:
    """Checkpoint view of a trainer's flat-arena optimizer state and weight EMA (model, arena, m, v, step_count, betas, eps,
    wd are the host class's)."""

    # ------------------------------------------------------------------------------------------------ checkpoint state
    def optimizer_state_dict(self):
        """The AdamW state of the flat arena in torch.optim.AdamW's state_dict layout (what main.py:302-309 /
        train_vqae.py:172-179 store as `optimizer_state_dict`): per parameter `step`, `exp_avg`, `exp_avg_sq` in
        `model.parameters()` order, one param group with this trainer's hyper-parameters.  CPU tensors."""
        a = self.arena
        state = {}
        if self.step_count > 0:            # (torch's AdamW has an EMPTY state before its first step)
            for i, (p, o) in enumerate(zip(a.params, a.offsets)):
                n = p.numel()
                state[i] = {'step': torch.tensor(float(self.step_count)),
                            'exp_avg': self.m[o:o + n].view_as(p).detach().cpu().clone(),
                            'exp_avg_sq': self.v[o:o + n].view_as(p).detach().cpu().clone()}
        group = {'lr': float(self._current_lr()), 'initial_lr': float(self.lr),      # initial_lr: what a scheduler resumes from
                 'betas': tuple(self.betas), 'eps': self.eps, 'weight_decay': self.wd, 'amsgrad': False, 'maximize': False,
                 'foreach': None, 'capturable': False, 'differentiable': False, 'fused': None,
                 'params': list(range(len(a.params)))}
        return {'state': state, 'param_groups': [group]}

    def load_optimizer_state_dict(self, sd):
        """Inverse of optimizer_state_dict(): moments and step count from a torch.optim.AdamW state_dict (a reference checkpoint's
        `optimizer_state_dict`, or one written by this trainer)."""
        a = self.arena
        steps = set()
        for i, (p, o) in enumerate(zip(a.params, a.offsets)):
            st = sd['state'].get(i)
            if st is None:
                continue
            n = p.numel()
            self.m[o:o + n].view_as(p).copy_(st['exp_avg'])
            self.v[o:o + n].view_as(p).copy_(st['exp_avg_sq'])
            steps.add(int(float(st['step'])))
        if len(steps) > 1:
            raise ValueError('per-parameter step counts differ: not an AdamW state this trainer can resume')
        if steps:
            self.step_count = steps.pop()

    def enable_ema(self, decay):
        """Weight EMA with ModelEmaV2's law e = decay e + (1 - decay) w after every optimizer step (model_ema_v2.py:33-41,
        main.py:286-287): one lerp over the flat arena, captured with the step when it is graphed.  ModelEmaV2 averages EVERY
        state_dict value, so the model's persistent buffers (BatchNorm running statistics, the VQ codebook and cluster sizes of
        the VQ auto-encoder; the denoiser has none) get shadow copies under the same law.  Call before enable_graph."""
        self.ema_decay = float(decay)
        self.ema_flat = self.arena.flat_param.detach().clone()
        named = {id(p) for p in self.model.parameters()}
        self._ema_live = {k: v for k, v in self.model.state_dict(keep_vars=True).items() if id(v) not in named}
        self.ema_bufs = {k: v.detach().clone() for k, v in self._ema_live.items()}
        return self

    def _ema_update(self):
        if getattr(self, 'ema_flat', None) is not None:
            self.ema_flat.lerp_(self.arena.flat_param, 1.0 - self.ema_decay)
            for k, e in self.ema_bufs.items():
                live = self._ema_live[k]
                if e.is_floating_point():
                    e.lerp_(live.detach(), 1.0 - self.ema_decay)
                else:                      # (num_batches_tracked: the reference computes the average in float and copies it back)
                    e.copy_(self.ema_decay * e + (1.0 - self.ema_decay) * live)

    def ema_state_dict(self):
        """name -> EMA values (CPU), the reference's `ema_model_state_dict`: parameters from the flat shadow arena, persistent
        buffers from their own shadow copies (ModelEmaV2._update walks the whole state_dict, model_ema_v2.py:33-41)."""
        if getattr(self, 'ema_flat', None) is None:
            return None
        a = self.arena
        by_id = {id(p): self.ema_flat[o:o + p.numel()].view_as(p) for p, o in zip(a.params, a.offsets)}
        named = dict(self.model.named_parameters())
        out = {}
        for k, v in self.model.state_dict().items():
            p = named.get(k)
            if p is not None and id(p) in by_id:
                out[k] = by_id[id(p)].detach().cpu().clone()
            else:
                out[k] = self.ema_bufs.get(k, v).detach().cpu().clone()
        return out


class _TrainerBase(_AdamState):
    """One object = model + flat arenas + AdamW state + (optional) data-parallel reducer; subclasses supply the step body
    (DenoiserTrainer: main.py:216-287 on token grids; SparseDenoiserTrainer: minecraft/sparse_diffusion.py:398-467)."""

    def __init__(self, model, num_embeddings, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-7, warmup=500,
                 max_steps=200 * 1000, distributed=None, bucket_bytes=None, accumulation_steps=1):
        """bucket_bytes None: one all-reduce bucket per transformer layer (plus the embeddings and the head), so the
        collective of layer l overlaps the backward of layer l-1; a number: consecutive parameters up to that size.
        accumulation_steps: micro-batches per optimizer step (main.py:221, :275-276: each micro-loss is divided by it)."""
        self.model = model
        self.C = num_embeddings
        self.arena = FlatArena(model)
        self.m = torch.zeros_like(self.arena.flat_param)
        self.v = torch.zeros_like(self.arena.flat_param)
        self.sq = torch.zeros(1, dtype=torch.float32, device=self.arena.flat_param.device)
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.warmup, self.max_steps = warmup, max_steps
        self.step_count = 0
        if distributed is None:
            distributed = torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1
        self.acc_steps = int(accumulation_steps)
        assert self.acc_steps >= 1
        self.reducer = BucketedAllReduce(self.arena, bucket_bytes=bucket_bytes,
                                         group_of=None if bucket_bytes is not None else layer_bucket_key,
                                         rounds=self.acc_steps, always=True) if distributed else None
        if distributed:
            broadcast_parameters(self.arena)
        self.rank = _dp_rank()
        # every rank draws its own noise levels: a generator of its own, keyed by the rank, so that ranks
        # started from the same torch seed do not all apply one r / one mask pattern to their different clips
        self.sampler_gen = torch.Generator()
        self.sampler_gen.manual_seed((torch.initial_seed() + 0x9E3779B97F4A7C15 * (self.rank + 1)) & 0x7FFFFFFFFFFFFFFF)
        self.sampler = LossAwareSamplerEma(num_histogram_buckets=100, uniform_p=0.01, alpha=0.9, warmup=10)
        self.packs = None                  # fused.PackSet: the fused kernels' weight streams, rebuilt with the operands
        self.chain_packs = None            # fused.ChainPackSet: the same for the widths of csrc/layer_chain.hip (training forward)
        self.operands = self._register_operands()
        self._refresh_operands()
        self._graph = None                 # captured training step (enable_graph)

    def _register_operands(self):
        """Every operand copy the step asks _cast.operand() for (bf16 casts, dgrad transposes, k|v concatenations), so
        that ONE launch rebuilds them all after the optimizer step."""
        from .config import get_compute_dtype
        dt = get_compute_dtype()
        bulk = _cast.BulkOperands()
        if dt == torch.float32 or not self.arena.flat_param.is_cuda:
            return bulk                                  # parity mode: weights are used as they are, transposes per use
        tr = self.model.transformer
        from . import config, fused
        # the default-width denoiser trains on the fused per-token kernels in both directions: they read their own packed
        # weight streams (fused._layer_pack / _layer_pack_bwd), none of the per-layer operand copies below
        layers = list(tr.layers)
        if hasattr(tr, 'pos_emb_s') and fused.supported(tr, dt) and config.get_fused_training() and config.fused_backward():
            layers = []
            self.packs = fused.PackSet(tr)
        elif hasattr(tr, 'pos_emb_s') and fused.chain_supported(tr, dt) and config.get_fused_training():
            # the reference's published widths: training FORWARD on the chain kernel (one gather rebuilds every launch's weight
            # stream); the backward runs op by op and keeps reading the per-layer operand copies registered below
            try:
                self.chain_packs = fused.ChainPackSet(tr, self.arena)
            except fused.ChainLayoutError:
                self.chain_packs = None                  # (op by op: the per-layer operand copies below serve both directions)
        for attn, ff in layers:
            a, f = attn.fn, ff.fn
            if hasattr(a, 'to_qkv'):                     # config 5: lucidrains ViT block with one fused projection
                bulk.add((a.to_qkv.weight,), dt, 'w')
                bulk.add((a.to_qkv.weight,), dt, 'wqkvT', transpose=True)
            else:
                bulk.add((a.to_q.weight,), dt, 'w')
                bulk.add((a.to_q.weight,), dt, 'wqT', transpose=True)
                bulk.add((a.to_k.weight, a.to_v.weight), dt, 'kv')
                bulk.add((a.to_k.weight, a.to_v.weight), dt, 'kvT', transpose=True)
                bulk.add((a.to_v.bias,), torch.float32, 'bkv', zero_first=True)
            if not isinstance(a.to_out, torch.nn.Identity):
                bulk.add((a.to_out[0].weight,), dt, 'w')
                bulk.add((a.to_out[0].weight,), dt, 'woutT', transpose=True)
            bulk.add((f.net[0].weight,), dt, 'w')
            bulk.add((f.net[0].weight,), dt, 'w1T', transpose=True)
            bulk.add((f.net[3].weight,), dt, 'w')
            bulk.add((f.net[3].weight,), dt, 'w2T', transpose=True)
        bulk.add((self.model.logit_proj.weight,), dt, 'w')
        bulk.add((self.model.logit_proj.weight,), dt, 'wT', transpose=True)
        return bulk

    def optimizer_step(self, lr=None):
        """Finish the all-reduce, grad-norm (device scalar, no sync), AdamW.  Returns the squared grad-norm tensor."""
        ops.wgrad_join()
        scale = self.reducer.finish() if self.reducer is not None else 1.0
        self.step_count += 1
        if lr is None:
            lr = lr_at(self.step_count, self.lr, self.warmup, self.max_steps)
        a = self.arena
        st = L.stream()
        self.sq.zero_()
        L.call('wmz_grad_sqnorm', L.ptr(a.flat_grad), a.numel, float(scale), L.ptr(self.sq), st)
        L.call('wmz_adamw_step', L.ptr(a.flat_param), L.ptr(a.flat_grad), L.ptr(self.m), L.ptr(self.v), a.numel,
               float(lr), self.betas[0], self.betas[1], self.eps, self.wd, self.step_count, float(scale), st)
        self._ema_update()
        _cast.invalidate(self.arena.params)            # the kernel rewrote the arena behind torch's version counters
        self._refresh_operands()      # ... and every operand copy of the weights is rebuilt by one launch
        return self.sq

    def _current_lr(self):
        return lr_at(max(self.step_count, 1), self.lr, self.warmup, self.max_steps)

    def _refresh_operands(self):
        self.operands.refresh()
        if self.packs is not None:
            self.packs.refresh()
        if self.chain_packs is not None:
            self.chain_packs.refresh()


    # ------------------------------------------------------------------------------------------------ hipGraph (every trainer)
    @_gc_quiet
    def enable_graph(self, example_batch, warmup=3, keep_warmup_updates=False):
        """Capture corrupt -> forward -> CE -> backward -> [gradient all-reduce] -> grad-norm -> AdamW -> operand re-pack as ONE
        hipGraph and replay it from train_step() (single micro-batch).  With a data-parallel reducer the per-layer RCCL
        all-reduces are captured too: the reducer's side stream forks from the capturing stream when a bucket's last gradient
        has landed and joins it again in finish(), so inside the graph every collective is a node whose only dependencies are
        the backward kernels that produced its bucket -- the overlap of the eager path, without the host between the launches.
        What changes per step lives in device memory: the clips and their noise levels (static input tensors), the
        corruption's stream counter (advanced inside the graph), the learning rate and AdamW bias corrections (`hyper`);
        torch's device RNG (config 5's position sampling) is capture-aware and advances per replay by itself.
        The warm-up steps run the real step body on `example_batch` (RCCL creates its communicator and the kernels their
        caches outside the capture); unless keep_warmup_updates, the weights, moments and step count they moved are restored,
        so a run starts from the same state whether or not it is graphed."""
        assert self.acc_steps == 1, 'the graphed step is one micro-batch per optimizer step'
        dev = self.arena.flat_param.device
        self._g_z = example_batch.contiguous().clone()
        # the step's host-written scalars live in ONE device buffer fed from ONE pinned host buffer: noise levels | lr, bias corrections
        nb = example_batch.shape[0]
        self._g_in = torch.zeros(nb + 3, dtype=torch.float32, device=dev)
        self._g_in_host = torch.zeros(nb + 3, dtype=torch.float32).pin_memory()
        self._g_in_np = self._g_in_host.numpy()
        self._g_r = self._g_in[:nb]
        # the per-call part of the corruption's Philox stream id: graph replays count on the device, eager calls on the host
        # (_corrupt_calls).  Bit 39 keeps the two ranges apart, so a run that mixes replays with eager fallbacks (another batch
        # shape) or with corrupt_tokens never draws one stream twice.
        # (created once: capturing again -- another batch shape, a caller's second enable_graph -- must go on counting, or the
        #  steps after it would replay the noise and mask streams of the run's first steps)
        if getattr(self, '_g_ctr', None) is None:
            self._g_ctr = torch.full((1,), 1 << 39, dtype=torch.int64, device=dev)
            self._g_seed = torch.initial_seed()
        self._g_hyper = self._g_in[nb:]
        snap = None if keep_warmup_updates else (self.arena.flat_param.clone(), self.m.clone(), self.v.clone(), self.step_count,
                                                 self.sampler_gen.get_state(),
                                                 None if getattr(self, 'ema_flat', None) is None else self.ema_flat.clone())
        ops.wgrad_reset()                  # nothing queued by an earlier (failed) pass may flush into this capture
        from . import config as _cfg
        side = _cfg.shared_stream('warmup')
        side.wait_stream(torch.cuda.current_stream())
        try:
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    self._set_step_inputs(None)
                    self._graph_body()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self._set_step_inputs(None)
            g = torch.cuda.CUDAGraph()
            from .graph import capture_mode
            mode = capture_mode()              # ('thread_local' next to a live process group: see there)
            with torch.cuda.graph(g, capture_error_mode=mode):
                self._g_out = self._graph_body()
        except BaseException:
            ops.wgrad_reset()
            raise
        self.step_count -= 1               # capturing records the launches, it does not run a step
        if snap is not None:
            if snap[5] is not None:
                self.ema_flat.copy_(snap[5])
            self.arena.flat_param.copy_(snap[0])
            self.m.copy_(snap[1])
            self.v.copy_(snap[2])
            self.step_count = snap[3]
            self.sampler_gen.set_state(snap[4])
            _cast.invalidate(self.arena.params)
            self._refresh_operands()
        self._graph = g
        # the graph bakes in the addresses of the library workspaces it was captured with (split-K partial tiles, counting-sort
        # counters): it HOLDS them.  When a later, larger eager call makes ops replace one, the eager path moves on to the new
        # allocation and the graph keeps replaying on the old one, which stays alive (and self-consistent: only the graph uses it)
        # through this reference.  Nothing is captured again behind the caller's back: a rank-local re-capture would run warm-up
        # steps with real all-reduces while its peers replay one step.
        self._g_ws = ops.workspace_snapshot(dev)
        return self

    def _set_step_inputs(self, r):
        """Host side of one graphed step: next noise levels and the optimizer scalars of step t into device memory."""
        B = self._g_z.shape[0]
        if r is None:
            r = self.sampler.sample(B, generator=self.sampler_gen)
        self._g_r_host = r
        self.step_count += 1
        lr = lr_at(self.step_count, self.lr, self.warmup, self.max_steps)
        bc1 = 1.0 - self.betas[0] ** self.step_count
        bc2 = 1.0 - self.betas[1] ** self.step_count
        h = self._g_in_host
        ev = getattr(self, '_g_in_ev', None)
        if ev is not None:
            ev.synchronize()               # the last copy out of the pinned buffer is done (a no-op behind a step's read-back)
        hn = self._g_in_np                                 # (a view of the pinned buffer: element writes without tensor indexing --
        hn[:B] = r.detach().reshape(-1).to('cpu', torch.float32).numpy()     # this runs with the GPU idle between two replays)
        hn[B], hn[B + 1], hn[B + 2] = lr, bc1, math.sqrt(bc2)
        self._g_in.copy_(h, non_blocking=True)
        if ev is None:
            ev = self._g_in_ev = torch.cuda.Event()
        ev.record()

    def _graph_body(self):
        a = self.arena
        a.flat_grad.zero_()
        # whatever is derived from the weights is rebuilt INSIDE the graph, every replay: the bulk operand copies by one
        # launch here, the fused kernels' weight streams by the forward (their cache entries are stale by construction)
        _cast.invalidate(self.arena.params)
        self._refresh_operands()
        per_sample, mean = self._graph_step(self._g_z, self._g_r)      # subclass: corruption (device counter) -> forward / backward
        self._g_ctr += 1
        ops.wgrad_join()                   # (already joined by the autograd pass's end-of-backward callback)
        # data parallel: buckets the backward did not launch itself, then the compute stream joins the reducer's side stream
        # (under capture: the fork / join edges of the graph); the 1/world of the gradient MEAN rides in the AdamW pass
        scale = self.reducer.finish() if self.reducer is not None else 1.0
        st = L.stream()
        self.sq.zero_()
        L.call('wmz_adamw_step_dev', L.ptr(a.flat_param), L.ptr(a.flat_grad), L.ptr(self.m), L.ptr(self.v), a.numel,
               L.ptr(self._g_hyper), self.betas[0], self.betas[1], self.eps, self.wd, float(scale), L.ptr(self.sq), st)   # + grad norm
        self._ema_update()
        return torch.cat([mean.reshape(1), self.sq.reshape(1), per_sample.reshape(-1)])     # the step's one read-back, packed in the graph


    def _replay(self, batch_z, r):
        """One graphed step: inputs into the static buffers, one hipGraph launch, ONE host read-back (loss, grad-norm, per-sample
        losses for the loss-aware sampler).  Returns (mean loss, grad norm)."""
        self._g_z.copy_(batch_z, non_blocking=True)
        self._set_step_inputs(r)
        self._graph.replay()
        _cast.invalidate(self.arena.params)             # the replay rewrote the weights: eager consumers rebuild their operand copies
        out = self._g_out.cpu()                                                  # the step's one host sync
        self.sampler.update_with_losses(self._g_r_host, out[2:])
        return float(out[0]), math.sqrt(float(out[1]))


class DenoiserTrainer(_TrainerBase):
    """The step body of vq-video-diffusion/main.py:train on [B,S,H,W] token clips (last frame corrupted and predicted)."""

    def forward_backward(self, batch_z, target, loss_scale=1.0):
        """Forward, per-sample CE over the last frame, backward of loss.mean() * loss_scale (gradient accumulation:
        main.py:274-278).  Returns (per_sample_loss[B], mean loss) on device."""
        from . import config, fused
        m = self.model
        tr = m.transformer
        dt = config.get_compute_dtype()
        if (batch_z.is_cuda and hasattr(tr, 'pos_emb_s') and config.get_fused_training() and config.fused_backward()
                and fused.supported(tr, dt) and batch_z.numel() % 32 == 0):
            # fused stack in both directions: only the last plane leaves it (main.py:37), its logits and cross-entropy are
            # one chunked linear + CE whose gradient is complete when the forward returns
            tr.check_grid(batch_z)
            last = fused.transformer_forward_train(tr, batch_z, last_only=True)              # [B, H, W, D]
            mean, rows = linear_cross_entropy(last.reshape(-1, last.shape[-1]), m.logit_proj.weight, m.logit_proj.bias,
                                              target.reshape(-1), chunk=4096, grad_scale=loss_scale)
            mean.backward()                                # (the accumulation scale is inside the fused gradient)
            return rows.view(batch_z.shape[0], -1).mean(dim=1), mean.detach()
        if (batch_z.is_cuda and self.chain_packs is not None and config.get_fused_training()
                and fused.chain_pays(self.chain_packs.widths, batch_z.numel(), True)):
            tr.check_grid(batch_z)
            last = fused.transformer_forward_chain_train(tr, self.chain_packs, batch_z, last_only=True)    # [B, H, W, D]
            mean, rows = linear_cross_entropy(last.reshape(-1, last.shape[-1]), m.logit_proj.weight, m.logit_proj.bias,
                                              target.reshape(-1), chunk=4096, grad_scale=loss_scale)
            mean.backward()
            return rows.view(batch_z.shape[0], -1).mean(dim=1), mean.detach()
        y = m(batch_z)
        loss = cross_entropy_rows(y.reshape(-1, self.C), target.reshape(-1))
        per_sample = loss.view(batch_z.shape[0], -1).mean(dim=1)
        mean = loss.mean()
        (mean if loss_scale == 1.0 else mean * loss_scale).backward()
        return per_sample.detach(), mean.detach()

    def _graph_step(self, z, r):
        zc, target = corrupt_last_frame(z, r, self.C, seed=self._g_seed, rank=self.rank, counter=self._g_ctr)
        return self.forward_backward(zc, target)

    def train_step(self, batch_z, r=None, generator=None):
        """corrupt -> forward/backward (all-reduce overlapped) -> grad-norm -> AdamW; sampler update on the host.
        batch_z: one micro-batch [B,S,H,W], or a list of `accumulation_steps` of them (main.py:221-280: gradients
        accumulate over the micro-batches, each micro-loss scaled by 1/acc_steps, loss_sum is their sum)."""
        if self._graph is not None and not isinstance(batch_z, (list, tuple)) and batch_z.shape == self._g_z.shape:
            return self._replay(batch_z, r)
        micro = list(batch_z) if isinstance(batch_z, (list, tuple)) else [batch_z]
        assert len(micro) == self.acc_steps, f'expected {self.acc_steps} micro-batches, got {len(micro)}'
        rs = list(r) if isinstance(r, (list, tuple)) else [r] * len(micro)
        self.arena.zero_grad()
        loss_sum, seen = 0.0, []
        for z, rr in zip(micro, rs):
            if rr is None:
                rr = self.sampler.sample(z.shape[0], generator=self.sampler_gen)
            zc, target = corrupt_last_frame(z, rr, self.C, generator, rank=self.rank)
            per_sample, mean = self.forward_backward(zc, target, 1.0 / self.acc_steps)
            loss_sum = loss_sum + mean / self.acc_steps
            seen.append((rr, per_sample))
        sq = self.optimizer_step()
        for rr, per_sample in seen:
            self.sampler.update_with_losses(rr, per_sample)   # the step's host sync (reference: ~50 per micro-batch)
        return float(loss_sum), math.sqrt(float(sq))


class SparseDenoiserTrainer(_TrainerBase):
    """The step body of minecraft/sparse_diffusion.py:train (:398-467) for VqSparseDiffusionModel (config 5): per clip,
    `num_context` grid positions are drawn (uniformly, or from a frame window that widens with the noise level: :403-408),
    their tokens gathered and corrupted with the same mask + uniform-redraw law (:440-449), the model predicts every
    gathered token, CE over all of them (:455-462).  The 8192-way logits never exist in memory (linear_cross_entropy)."""

    def __init__(self, model, num_embeddings, num_context=512, sampling_type='neighbors', **kw):
        super().__init__(model, num_embeddings, **kw)
        self.num_context = int(num_context)
        if sampling_type not in ('uniform', 'neighbors'):
            raise ValueError('Specified sampling_type not supported')          # sparse_diffusion.py:408
        self.sampling_type = sampling_type
        self.use_fused_context = True      # (False: positions / gather / corruption by the torch ops of sparse_diffusion.py)

    def sample_positions(self, B, r, device):
        from .sparse_diffusion import sample_flat_positions, sample_time_dependent
        S, H, W = self.model.shape
        if self.sampling_type == 'uniform':
            return sample_flat_positions(B, self.num_context, S, H, W, device)
        return sample_time_dependent(B, self.num_context, S, H, W, r, device)

    def forward_backward(self, tokens, indices, target, loss_scale=1.0):
        """tokens / indices / target: [B, n].  Returns (per-sample loss [B], mean loss) on device."""
        from . import functional as Fw
        m = self.model
        h = Fw.embed_tokens_indexed(tokens, indices, m.embedding.weight, m.pos_emb_s.weight, m.pos_emb_h.weight,
                                    m.pos_emb_w.weight, m.shape)
        h = m.transformer.forward_compute(h)
        # (config 5: 3 072 rows per GPU = ONE chunk -- 100 MB of fp32 logits is nothing on a 288 GB card, and a 1 024-row chunk
        #  leaves the 8 192-deep dgrad on 64 workgroups)
        mean, rows = linear_cross_entropy(h.reshape(-1, h.shape[-1]), m.logit_proj.weight, m.logit_proj.bias, target.reshape(-1),
                                          chunk=4096, grad_scale=loss_scale, side_branch=True)
        mean.backward()                                    # (the accumulation scale is inside the fused gradient)
        return rows.view(tokens.shape[0], -1).mean(dim=1), mean.detach()

    def _graph_step(self, z, r):
        """The device side of one step on static inputs (captured by enable_graph): position sampling (torch's capture-aware
        device RNG), gather, corruption from the in-kernel Philox stream counted on the device, forward / backward."""
        B = z.shape[0]
        if self.fused_context(z):
            # one launch instead of ~45 graph nodes (window arithmetic on B-element tensors, top-k, sort, gather, corruption)
            indices, tokens, target = draw_sparse_context(z, r, self.num_context, self.model.shape, self.C, seed=self._g_seed,
                                                          rank=self.rank, counter=self._g_ctr)
            return self.forward_backward(tokens, indices, target)
        indices = self.sample_positions(B, r, z.device)
        gathered = torch.gather(z.reshape(B, -1), 1, indices)
        tokens, target = corrupt_tokens(gathered, r, self.C, seed=self._g_seed, rank=self.rank, counter=self._g_ctr)
        return self.forward_backward(tokens, indices, target)

    def fused_context(self, z):
        """The step prologue runs as wmz_sparse_draw_context: the reference's default sampler ('neighbors'), a grid the kernel
        holds (<= 65 536 positions, <= 512 drawn), `self.use_fused_context` left on."""
        S, H, W = self.model.shape
        return (self.use_fused_context and self.sampling_type == 'neighbors' and z.is_cuda
                and bool(L.lib().wmz_sparse_draw_context_supported(int(S), int(H) * int(W), self.num_context)))

    def train_step(self, batch_z, r=None, indices=None, generator=None):
        """batch_z: [B,S,H,W] token clips (the frozen VQ-AE's output).  Returns (mean loss, grad norm)."""
        if self._graph is not None and indices is None and batch_z.shape == self._g_z.shape:
            return self._replay(batch_z, r)
        B = batch_z.shape[0]
        if r is None:
            r = self.sampler.sample(B, generator=self.sampler_gen)
        self.arena.zero_grad()
        if indices is None and self.fused_context(batch_z):
            indices, tokens, target = draw_sparse_context(batch_z, r, self.num_context, self.model.shape, self.C, generator,
                                                          rank=self.rank)
        else:
            if indices is None:
                indices = self.sample_positions(B, r, batch_z.device)
            gathered = torch.gather(batch_z.reshape(B, -1), 1, indices)              # :437
            tokens, target = corrupt_tokens(gathered, r, self.C, generator, rank=self.rank)
        per_sample, mean = self.forward_backward(tokens, indices, target)
        sq = self.optimizer_step()
        self.sampler.update_with_losses(r, per_sample)
        return float(mean), math.sqrt(float(sq))


class VqaeTrainer(_AdamState):
    """The step body of vq-video-diffusion/train_vqae.py:train (:125-164) for the drop-in VqAutoEncoder: reconstruction loss
    (SmoothL1 / MSE / L1, :264-271) + latent_loss_weight x commitment loss (:148), AdamW (lr 2e-4, weight decay 0: :254-256)
    with the per-epoch StepLR(step_size 3, gamma 0.5) schedule (:262), and every `vq_reuse_interval` steps the dead-code
    revival `vq.reuse_inactive()` + `vq.reset_stats()` (:160-164).  Parameters / gradients / moments live in one flat arena
    (one grad-norm-free AdamW launch per step); under data parallelism the gradient buckets are all-reduced as for the
    denoiser and the VQ EMA statistics are all-reduced inside VectorQuantizerEMA.forward (vq.sync_stats)."""

    LOSSES = {'SmoothL1': F.smooth_l1_loss, 'MSE': F.mse_loss, 'MAE': F.l1_loss, 'L1': F.l1_loss}

    def __init__(self, model, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, loss_fn='SmoothL1',
                 latent_loss_weight=0.01, vq_reuse_interval=500, steps_per_epoch=None, distributed=None):
        if loss_fn not in self.LOSSES:
            raise RuntimeError('Unsupported loss function type specified.')     # train_vqae.py:271
        self.model, self.loss_fn, self.loss_name = model, self.LOSSES[loss_fn], loss_fn
        self.latent_loss_weight, self.vq_reuse_interval = latent_loss_weight, vq_reuse_interval
        self.arena = FlatArena(model)
        self.m = torch.zeros_like(self.arena.flat_param)
        self.v = torch.zeros_like(self.arena.flat_param)
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.steps_per_epoch = steps_per_epoch
        self.step_count = 0
        if distributed is None:
            distributed = torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1
        self.reducer = BucketedAllReduce(self.arena, bucket_bytes=1 << 20, always=True) if distributed else None
        if distributed:
            broadcast_parameters(self.arena)
            model.vq.sync_stats = True
        self.reused = 0
        # (this trainer zeroes the arena every step and runs one backward per step: BatchNorm's column sums may land straight
        #  in the parameters' gradient slots, autoencoder._BnActFn.backward)
        for mod in model.modules():
            if isinstance(mod, torch.nn.BatchNorm2d) and mod.weight is not None:
                mod.weight._wmz_single_use = True
        self.fused_losses = True       # (False: the reconstruction as an NCHW fp32 tensor and torch's loss on it)
        self._conv_ops = {}            # compute dtype -> _cast.ConvOperands (every conv layer's GEMM operands, one launch a step)

    def _refresh_conv_operands(self):
        from . import config
        dt = config.get_compute_dtype()
        co = self._conv_ops.get(dt)
        if co is None:
            convs = [m for m in self.model.modules() if isinstance(m, torch.nn.Conv2d)]
            co = self._conv_ops[dt] = _cast.ConvOperands(convs, dt)
        co.refresh()

    def _current_lr(self):
        return self.lr_now()

    def lr_now(self):
        """StepLR(step_size=3, gamma=0.5) stepped once per epoch (train_vqae.py:192, :262)."""
        if not self.steps_per_epoch:
            return self.lr
        return self.lr * 0.5 ** ((self.step_count // self.steps_per_epoch) // 3)

    def _forward_backward(self, batch):
        """model -> reconstruction + commitment loss -> backward (gradients land in the arena).  Returns the four scalars of the
        step as ONE device tensor [loss, reconstruction loss, latent loss, perplexity]."""
        if (self.fused_losses and batch.is_cuda and batch.dtype == torch.float32 and hasattr(self.model, 'training_losses')
                and self.loss_name in ops.RECON_LOSS_KINDS):
            # the reconstruction loss on the decoder's NHWC output in place (one launch pair forward, one backward)
            r_loss, latent_loss, perplexity = self.model.training_losses(batch, self.loss_name)
        else:
            recon, latent_loss, perplexity = self.model(batch)
            r_loss = self.loss_fn(recon, batch)
        loss = r_loss + self.latent_loss_weight * latent_loss
        loss.backward()
        return torch.stack([loss.detach(), r_loss.detach(), latent_loss.detach().reshape(()), perplexity.detach().reshape(())])

    @_gc_quiet
    def enable_graph(self, example_batch, warmup=2):
        """Capture zero-grad -> encoder -> VectorQuantizerEMA (incl. the in-place EMA codebook update, vq.py:42-65) -> decoder ->
        losses -> backward -> AdamW as ONE hipGraph, replayed by train_step() for batches of this shape: ~460 launches with the
        autograd bookkeeping between them become one launch, the step's four scalars come back in one read.  The learning rate
        and AdamW's bias corrections live in device memory (wmz_adamw_step_dev); the dead-code revival (train_vqae.py:160-164,
        host-driven, every vq_reuse_interval steps) runs between replays.  The warm-up steps are real training steps."""
        # data parallel: the gradient buckets' all-reduces and the VQ EMA statistics' (counts [C], dw [C, E]: vq.sync_stats)
        # are captured with the step -- RCCL collectives are capturable, gloo's are not
        if self.reducer is not None:
            import torch.distributed as _dist
            if _dist.get_backend() != 'nccl':
                raise RuntimeError('the graphed data-parallel VQ-AE step needs the RCCL (`nccl`) backend: gloo collectives cannot '
                                   'be captured into a hipGraph (train eagerly, or drop enable_graph)')
        ops.wgrad_reset()
        dev = self.arena.flat_param.device
        self.model.train()
        self._g_x = example_batch.contiguous().clone()
        self._g_hyper = torch.zeros(3, dtype=torch.float32, device=dev)
        self._g_hyper_host = torch.zeros(3, dtype=torch.float32).pin_memory()       # (staged through pinned memory: an asynchronous copy)
        self._g_hyper_np, self._g_hyper_ev = self._g_hyper_host.numpy(), None
        self._g_sq = torch.zeros(1, dtype=torch.float32, device=dev)
        from . import config as _cfg
        side = _cfg.shared_stream('warmup')
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._set_hyper()
                self._graph_body()
                self._after_step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        from .graph import capture_mode
        with torch.cuda.graph(g, capture_error_mode=capture_mode()):
            self._g_out = self._graph_body()
        self._graph = g
        self._g_ws = ops.workspace_snapshot(dev)
        return self

    def _set_hyper(self):
        lr = self.lr_now()
        self.step_count += 1
        bc1 = 1.0 - self.betas[0] ** self.step_count
        bc2 = 1.0 - self.betas[1] ** self.step_count
        if self._g_hyper_ev is not None:
            self._g_hyper_ev.synchronize()     # the last copy out of the pinned buffer is done (a no-op behind a step's read-back)
        self._g_hyper_np[:] = (lr, bc1, math.sqrt(bc2))
        self._g_hyper.copy_(self._g_hyper_host, non_blocking=True)
        if self._g_hyper_ev is None:
            self._g_hyper_ev = torch.cuda.Event()
        self._g_hyper_ev.record()

    def _graph_body(self):
        a = self.arena
        a.flat_grad.zero_()
        _cast.invalidate(self.arena.params)                 # (the conv / codebook operand copies are rebuilt from the weights inside the graph)
        self._refresh_conv_operands()
        out = self._forward_backward(self._g_x)
        ops.wgrad_join()                   # (the conv weight gradients' side branch; already joined when the autograd pass ended)
        # data parallel: buckets the backward did not launch itself, then the compute stream joins the reducer's side stream;
        # the 1 / world of the gradient mean is folded into AdamW's gradient scale
        scale = self.reducer.finish() if self.reducer is not None else 1.0
        self._g_sq.zero_()
        L.call('wmz_adamw_step_dev', L.ptr(a.flat_param), L.ptr(a.flat_grad), L.ptr(self.m), L.ptr(self.v), a.numel,
               L.ptr(self._g_hyper), self.betas[0], self.betas[1], self.eps, self.wd, float(scale), L.ptr(self._g_sq), L.stream())
        self._ema_update()
        return out

    def _after_step(self):
        _cast.invalidate(self.arena.params)
        if self.vq_reuse_interval and self.step_count % self.vq_reuse_interval == 0:
            self.reused = self.model.vq.reuse_inactive()
            self.model.vq.reset_stats()

    def train_step(self, batch):
        """batch: [B, C, H, W] frames on the GPU.  Returns (loss, reconstruction loss, latent loss, perplexity) as floats."""
        self.model.train()
        if getattr(self, '_graph', None) is not None and batch.shape == self._g_x.shape:      # (the graph holds its workspaces)
            self._g_x.copy_(batch, non_blocking=True)
            self._set_hyper()
            self._graph.replay()
            vals = self._g_out.cpu()                                  # the step's one host sync
            self._after_step()
            return tuple(float(v) for v in vals)
        self.arena.zero_grad()
        self._refresh_conv_operands()
        out = self._forward_backward(batch)
        scale = self.reducer.finish() if self.reducer is not None else 1.0
        lr = self.lr_now()
        self.step_count += 1
        a = self.arena
        L.call('wmz_adamw_step', L.ptr(a.flat_param), L.ptr(a.flat_grad), L.ptr(self.m), L.ptr(self.v), a.numel, float(lr),
               self.betas[0], self.betas[1], self.eps, self.wd, self.step_count, float(scale), L.stream())
        self._ema_update()
        self._after_step()
        return tuple(float(v) for v in out.cpu())
