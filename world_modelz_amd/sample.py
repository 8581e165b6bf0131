"""Iterative-unmasking sampler (the inference caller of the denoiser): build-owned counterpart of
vq-video-diffusion/main.py:evaluate_model (:50-117) -- SURVEY 8f N2.

Per generated frame: the last latent frame starts fully masked; 30 denoise iterations of
{top-k filter -> softmax -> multinomial -> re-mask a (1 - frac) share -> write into the last frame -> model.forward};
then the frames shift by one.  Everything stays on the GPU: the forward is ONE hipGraph launch per iteration
(GraphedForward), sampling uses device RNG, and there is no host sync inside a frame.
"""
import torch
import torch.nn.functional as F

from .graph import GraphedForward


def top_k_logits(logits, k):
    """Keep the k largest logits per row, -inf elsewhere (main.py:39-43)."""
    v, _ = torch.topk(logits, k, largest=True, sorted=True)
    return logits.masked_fill(logits < v[:, [-1]], -float('inf'))


def categorical_from_uniform(p, u):
    """Inverse-CDF categorical draw: the first class whose cumulative weight exceeds u * total (u in [0, 1), one per row).
    This is the definition the sampler-parity fixtures are captured with (torch.multinomial replaced by it inside the
    reference's evaluate_model, tests/golden/make_golden.py), so with the same uniforms the draws are the reference's."""
    cdf = p.cumsum(dim=-1)
    x = (u.to(cdf.dtype) * cdf[:, -1]).unsqueeze(-1)
    return (cdf <= x).sum(dim=-1).clamp(max=p.shape[-1] - 1)


@torch.no_grad()
def sample_frames(model, batch_z, num_embeddings, num_frames, num_eval_iterations=30, sample_topk=-1, noise_schedule=None,
                  consistent_masking=False, generator=None, use_graph=True, uniforms=None, trace=None):
    """batch_z: int64 [B,S,H,W] context tokens on the GPU (the last frame is overwritten).  Returns the list of generated
    latent frames [B,H,W] (decode them with VqAutoEncoder.decode) and the final batch_z.

    uniforms = (u_multi [num_frames, num_eval_iterations, B*H*W], u_mask [num_frames, num_eval_iterations, B, H*W]):
    injected randomness -- the categorical draw becomes categorical_from_uniform and the re-mask field u_mask > alpha
    (main.py:85, :97-100), which makes the loop a deterministic function of its inputs (parity tests); default: device RNG.
    trace: optional list that receives every last frame fed to the model."""
    assert batch_z.is_cuda
    B, S, H, W = batch_z.shape
    mask_token = num_embeddings
    batch_z = batch_z.clone()
    batch_z[:, -1] = mask_token                                   # destroy all information in the last frame (:62)
    if use_graph and uniforms is None and trace is None and num_embeddings <= 2048:
        return _sample_frames_fused(model, batch_z, num_embeddings, num_frames, num_eval_iterations, sample_topk, noise_schedule,
                                    consistent_masking, generator)
    fwd = GraphedForward(model, batch_z) if use_graph else None
    if fwd is not None:
        batch_z = fwd.static_in                                   # the loop edits the graph's own input buffer: no staging copy
    dev = batch_z.device
    if uniforms is not None:
        u_multi, u_mask = (t.to(dev) for t in uniforms)
        assert u_multi.shape[:2] == (num_frames, num_eval_iterations) and u_mask.shape[:2] == (num_frames, num_eval_iterations)
    out = []
    for f in range(num_frames):
        logits = torch.zeros(B * H * W, num_embeddings, device=dev)      # flat start (:71)
        last_mask = torch.ones(B, H * W, dtype=torch.bool, device=dev)
        for i in range(num_eval_iterations):
            if sample_topk > 0:
                logits = top_k_logits(logits, sample_topk)
            p = F.softmax(logits, dim=-1)
            if uniforms is not None:
                denoised = categorical_from_uniform(p, u_multi[f, i].reshape(-1)).view(B, H * W)
            else:
                denoised = torch.multinomial(p, 1, True, generator=generator).view(B, H * W)
            frac = (i + 1) / num_eval_iterations
            alpha = min(max(noise_schedule(frac) if noise_schedule is not None else frac, 0.0), 1.0)
            u = u_mask[f, i].view(B, H * W) if uniforms is not None else torch.rand(B, H * W, device=dev, generator=generator)
            mask = u > alpha
            if consistent_masking:
                mask = last_mask & mask
                last_mask = mask
            frame = torch.where(mask, torch.full_like(denoised, mask_token), denoised)
            batch_z[:, -1] = frame.view(B, H, W)
            if trace is not None:
                trace.append(frame.view(B, H, W).clone())
            if i + 1 < num_eval_iterations:                       # (the reference runs the model behind a frame's last draw too, :111,
                #                                                    and never reads those logits)
                logits = (fwd(batch_z) if fwd is not None else model(batch_z)).reshape(B * H * W, num_embeddings).float()
        out.append(denoised.view(B, H, W).clone())
        batch_z[:, :-1] = batch_z[:, 1:].clone()                  # shift frames (:115)
    return out, (batch_z.clone() if fwd is not None else batch_z)   # (never hand out the graph's own buffer)


MAX_SESSIONS = 2          # captured sampler steps kept per model (least recently used beyond that are dropped: a graph, its
#                           buffers and its references to the operand copies are ~tens of MB each)


class _Sessions(dict):
    """The model's captured sampler steps, most recently used last.  They belong to THIS module object on THIS device: a copy of the
    model (copy.deepcopy, pickling the module) starts without them instead of failing on the hipGraph inside."""

    def __deepcopy__(self, memo):
        return _Sessions()

    def __reduce__(self):
        return (_Sessions, ())

    def touch(self, key, make):
        ses = self.pop(key, None)
        if ses is None:
            ses = make()
        self[key] = ses                                   # (re-inserted: dicts keep insertion order)
        while len(self) > MAX_SESSIONS:
            del self[next(iter(self))]
        return ses


class _WeakModel:
    """What a session's graph runner holds instead of the model: a weak reference.  model -> sessions -> session -> runner -> model
    was a reference cycle -- a dropped model, its hipGraphs and their private pools were freed only by the cyclic collector,
    whenever that ran (possibly in the middle of another stream capture); now dropping the model frees them at once."""

    def __init__(self, model):
        import weakref
        self._ref = weakref.ref(model)

    def _model(self):
        m = self._ref()
        if m is None:
            raise RuntimeError('the model of this sampler session no longer exists')
        return m

    def parameters(self):
        return self._model().parameters()

    def buffers(self):
        return self._model().buffers()

    def __call__(self, z):
        return self._model()(z)


def release_sampler_sessions(model):
    """Drop the captured sampler steps kept with `model` (their graphs, device buffers and operand references) now."""
    ses = model.__dict__.pop('_wmz_sampler_sessions', None)
    if ses is not None:
        ses.clear()


_SEED_KEY = 0x9E3779B97F4A7C15                   # the captured kernels' Philox key (a kernel ARGUMENT, i.e. baked into the graph):
#                                                   what varies per call is the device-side counter's starting value
# (the sessions live ON the model -- `model._wmz_sampler_sessions`, {configuration: captured sampler step + its device buffers},
#  at most MAX_SESSIONS of them -- and die with it: the graph runner of a session holds the model weakly (_WeakModel))


class _Session:
    """One captured sampler step (forward + draw of the next iteration + counter) and the device state it works on, kept with the
    model across sample_frames calls: capturing costs ~4 ms, a frame of 30 iterations ~11 (GraphedForward re-captures by itself
    when the weights, the compute dtype or the run-time configuration changed)."""

    def __init__(self, model, batch_z, C, n_iter, sample_topk, consistent_masking):
        from . import ops
        B, S, H, W = batch_z.shape
        dev = batch_z.device
        R = B * H * W
        self.alphas = torch.zeros(n_iter, dtype=torch.float32, device=dev)
        self.logits = torch.zeros(R, (C + 3) // 4 * 4, dtype=torch.float32, device=dev)[:, :C]     # rows 16-byte aligned whatever C is
        self.denoised = torch.zeros(R, dtype=torch.int64, device=dev)
        self.counter = torch.zeros(1, dtype=torch.int64, device=dev)
        self.last_mask = torch.ones(R, dtype=torch.uint8, device=dev) if consistent_masking else None
        aligned = C % 4 == 0
        holder = {}

        def draw(src, z):
            ops.sample_tokens(src, sample_topk, self.alphas, C, z[:, -1], self.denoised, self.counter, _SEED_KEY, self.last_mask)
            self.counter.add_(1)

        def pre(z):
            holder['z'] = z

        def post(y):
            src = y.reshape(R, C)
            if not aligned:
                self.logits.copy_(src)
                src = self.logits
            draw(src, holder['z'])
        self.draw = draw
        self.fwd = GraphedForward(_WeakModel(model), batch_z, pre=pre, post=post)


def _sample_frames_fused(model, batch_z, num_embeddings, num_frames, num_eval_iterations, sample_topk, noise_schedule,
                         consistent_masking, generator):
    """The same loop with NO host work inside a frame but one graph launch per iteration: the draw + re-mask step is one
    kernel (wmz_sample_tokens_dev: top-k, softmax, inverse-CDF draw and re-mask per row, uniforms from in-kernel Philox indexed
    by a device-side iteration counter that starts, per call, at a draw from the caller's generator) captured BEHIND the forward
    whose logits it reads, the counter increment behind it.  One graph replay = the forward on the current grid + the draw of the
    NEXT iteration from its logits (read where the graph leaves them: no hand-over copy when the rows are 16-byte aligned).  A
    frame's first draw (from the flat start, :71) is one eager launch in front of its replays, and the forward behind a frame's
    LAST draw -- whose logits the reference computes (:111) and never reads -- is not run: n - 1 forwards per frame instead of n.
    The captured step stays with the model between calls (_Session).  Same distribution as the torch path (which stays for
    injected uniforms: the parity fixtures), different random stream."""
    B, S, H, W = batch_z.shape
    n = num_eval_iterations
    key = (B, S, H, W, num_embeddings, n, int(sample_topk), bool(consistent_masking), batch_z.device)
    per_model = model.__dict__.setdefault('_wmz_sampler_sessions', _Sessions())
    ses = per_model.touch(key, lambda: _Session(model, batch_z, num_embeddings, n, sample_topk, consistent_masking))
    ses.alphas.copy_(torch.tensor([min(max(noise_schedule((i + 1) / n) if noise_schedule is not None else (i + 1) / n, 0.0), 1.0)
                                   for i in range(n)], dtype=torch.float32))
    # The Philox stream is indexed by the counter; a call starts it at a 62-bit draw from the caller's generator (the global CPU
    # generator when none is passed) rounded to a multiple of n (the kernel takes alpha = alphas[counter % n]), which advances
    # that generator like the reference's torch.multinomial / torch.rand do: two calls in a row see different noise, reseeding
    # the generator reproduces a call.  (A CUDA generator costs one device sync per call here.)
    base = int(torch.randint(0, 1 << 62, (1,), generator=generator,
                             device=generator.device if generator is not None else 'cpu').item()) // n * n
    fwd = ses.fwd
    fwd.refresh()                                                 # (a re-capture draws: before the call's state is set up, not inside it)
    z = fwd.static_in
    z.copy_(batch_z)                                              # (the capture / the previous call drew into the last frame)
    ses.counter.fill_(base)
    out = []
    for f in range(num_frames):
        ses.logits.zero_()                                        # flat start (:71)
        if ses.last_mask is not None:
            ses.last_mask.fill_(1)
        ses.draw(ses.logits, z)                                   # iteration 0
        for i in range(1, n):
            fwd(z)                                                # forward of iteration i - 1, draw of iteration i
        out.append(ses.denoised.view(B, H, W).clone())
        z[:, :-1] = z[:, 1:].clone()                              # shift frames (:115)
    return out, z.clone()
