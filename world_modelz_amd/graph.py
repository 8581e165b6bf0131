"""hipGraph capture of a forward denoise step (launch-bound at ~17 kernels per step otherwise).

    runner = GraphedForward(model, example_tokens)     # warm-up + capture on a side stream
    logits = runner(tokens)                            # copy into the static input, one hipGraphLaunch

The C ABI never allocates or synchronises and launches on torch's current stream, so the whole step is
capturable; torch's caching allocator provides the graph-private pool for intermediates.
"""
import torch

from . import _cast, config


def capture_mode():
    """capture_error_mode for torch.cuda.graph: with a process group alive its watchdog thread issues HIP calls (event queries) of
    its own while this thread captures; under the default 'global' mode those interact with the capture (an intermittent crash
    inside capture_end() with an RCCL world of one, twice in ~10 full test-suite runs).  'thread_local' confines the capture's
    legality checks to the capturing thread -- the recipe for captures next to NCCL / RCCL."""
    import torch.distributed as dist
    return 'thread_local' if dist.is_available() and dist.is_initialized() else 'global'


def gc_quiet(fn):
    """Decorator for functions that warm up and capture a hipGraph: one collection up front, the cyclic collector off until the
    function returns -- a collection inside would destroy dead graph holders (and free their pools) in the middle of the capture."""
    import functools
    import gc

    @functools.wraps(fn)
    def wrapped(*a, **k):
        gc.collect()
        was_enabled = gc.isenabled()
        gc.disable()
        try:
            return fn(*a, **k)
        finally:
            if was_enabled:
                gc.enable()
    return wrapped


class GraphedForward:
    """The captured graph bakes in the device addresses of the operand copies of the weights (bf16 casts, packed weight
    streams: _cast).  It therefore (i) holds strong references to every operand tensor that existed at capture time, so
    none of them can be freed and recycled under a retained graph, and (ii) stamps the parameter versions / addresses,
    the _cast epoch and the run-time configuration it was captured under, and re-captures when any of them changed
    (optimizer step, load_state_dict, EMA copy, .to(), another compute dtype) instead of replaying stale weights."""

    def __init__(self, model, example, warmup=3, pre=None, post=None):
        """pre(static_in) / post(static_out): optional device-only work captured in front of / behind the forward (the
        sampler's draw + re-mask step and its logits hand-over: sample.py), replayed with it."""
        self.model = model
        self.static_in = example.clone()
        self.warmup = warmup
        self.pre, self.post = pre, post
        self.recaptures = 0
        self._capture()

    def _stamp(self):
        # (the tensor list is collected once per capture: walking the module tree costs ~70 us of host time per call, the
        # version / address reads of ~90 tensors ~20; parameters replaced by NEW objects are caught by _capture's list
        # being rebuilt whenever anything else in the stamp changes -- and by load_state_dict / .to() / optimizers, which
        # all write the existing objects)
        return (_cast.epoch_of(self._tensors), config.get_mode_dtype(), config.get_last_frame_cone(), config.get_clip_streams(),
                tuple((t._version, t.data_ptr()) for t in self._tensors))

    @gc_quiet
    def _capture(self):
        self._tensors = list(self.model.parameters()) + list(self.model.buffers())
        s = config.shared_stream('warmup')
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s), torch.no_grad():
            for _ in range(self.warmup):
                if self.pre is not None:
                    self.pre(self.static_in)
                y = self.model(self.static_in)
                if self.post is not None:
                    self.post(y)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph, capture_error_mode=capture_mode()):
            if self.pre is not None:
                self.pre(self.static_in)
            self.static_out = self.model(self.static_in)
            if self.post is not None:
                self.post(self.static_out)
        self._operands = [h[1] for h in _cast._cache.values()]      # strong references (tensors or tuples of tensors)
        self.stamp = self._stamp()

    def refresh(self):
        """Re-capture now if the weights / configuration moved since the capture (a caller whose pre / post work keeps device
        state -- the sampler's counter and last frame -- calls this BEFORE it sets that state up: a capture runs pre / post)."""
        if self._stamp() != self.stamp:
            self.recaptures += 1
            self._capture()

    def __call__(self, tokens):
        self.refresh()
        if tokens is not self.static_in:
            self.static_in.copy_(tokens, non_blocking=True)
        self.graph.replay()
        return self.static_out


class _EncodeAdaptor:
    """VqAutoEncoder.encode behind the interface GraphedForward captures (a callable with parameters() / buffers())."""

    def __init__(self, ae):
        self.ae = ae

    def parameters(self):
        return self.ae.parameters()

    def buffers(self):
        return self.ae.buffers()

    def __call__(self, frames):
        return self.ae.encode(frames)


class GraphedEncoder(GraphedForward):
    """The frozen VQ auto-encoder's frame encoder (main.py:236 `decoder_model.encode(frames)`: conv encoder with BatchNorm in
    training mode -- quirk Q3, the running statistics move on every call, inside the graph -- + codebook search) as ONE hipGraph
    launch per batch of frames:

        enc = GraphedEncoder(ae, example_frames)
        tokens = enc(frames)                               # [B, h, w] int64 (the runner's static output)
    """

    def __init__(self, ae, example_frames, warmup=2):
        super().__init__(_EncodeAdaptor(ae), example_frames, warmup=warmup)
