"""Single-node data parallelism for the denoiser training step (no counterpart in the reference: SURVEY 8e).

One process per GPU; clips are sharded across ranks, so the only collective is the gradient SUM all-reduce
(RCCL over xGMI: backend "nccl" on ROCm).  Gradients live in ONE flat fp32 arena cut into per-layer buckets; as
soon as autograd has accumulated the last gradient of a bucket, the bucket's slice is all-reduced on a side HIP
stream, so the collective of layer l overlaps the backward of layer l-1.  The 1/world scaling is folded into the
consumers (wmz_grad_sqnorm / wmz_adamw_step), not applied as an extra pass.

Message sizes here are small (6.4 MB for the default model), xGMI is point-to-point (7 links x ~153 GB/s): a few
per-layer buckets keep every collective in RCCL's low-latency regime while still overlapping.
"""
import torch
import torch.distributed as dist


class FlatArena:
    """All parameters of `module` re-homed as views into one contiguous fp32 buffer, gradients likewise."""

    def __init__(self, module, align=64):
        params = [p for p in module.parameters() if p.requires_grad]
        assert params, 'module has no trainable parameters'
        dev = params[0].device
        self.params = params
        self.names = {id(p): n for n, p in module.named_parameters()}
        offs, n = [], 0
        for p in params:
            offs.append(n)
            n += (p.numel() + align - 1) // align * align
        self.offsets, self.numel = offs, n
        self.flat_param = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(n, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, o in zip(params, offs):
                assert p.dtype == torch.float32
                self.flat_param[o:o + p.numel()].view_as(p).copy_(p)
                p.data = self.flat_param[o:o + p.numel()].view_as(p)
                p.grad = self.flat_grad[o:o + p.numel()].view_as(p)
                p._wmz_grad = p.grad          # backward kernels accumulate straight into the arena (backward._emit)

    def zero_grad(self):
        self.flat_grad.zero_()
        for p, o in zip(self.params, self.offsets):          # keep .grad pointing into the arena
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * o:
                p.grad = self.flat_grad[o:o + p.numel()].view_as(p)
                p._wmz_grad = p.grad


def layer_bucket_key(name):
    """Bucket key of a denoiser parameter name: one bucket per transformer layer (`...layers.<l>.`), one for whatever
    precedes the layers (embeddings) and one for what follows (the logit head) -- consecutive in arena order."""
    parts = name.split('.')
    if 'layers' in parts:
        i = parts.index('layers')
        if i + 1 < len(parts) and parts[i + 1].isdigit():
            return 'layer' + parts[i + 1]
    return 'pre' if ('emb' in name) else 'post'


class BucketedAllReduce:
    """Overlapped gradient all-reduce over a FlatArena.

    buckets: consecutive parameter ranges (in arena order), cut either where `group_of(parameter name)` changes (one
    bucket per transformer layer: layer_bucket_key) or at `bucket_bytes`; parameters are registered in module order, the
    backward produces them roughly in reverse, so buckets complete back to front and the collective of layer l runs on
    the side stream under the backward of layer l-1.  rounds: backward passes per optimizer step (gradient
    accumulation) -- a bucket is reduced when its last gradient of the LAST round has landed.
    `always`: register the hooks and run the collectives even in a world of one (exercises the side-stream path on a
    single GPU; RCCL's all-reduce over one rank is an in-place no-op copy)."""

    def __init__(self, arena, process_group=None, bucket_bytes=4 << 20, group_of=None, rounds=1, always=False):
        self.arena = arena
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (always and dist.is_initialized())
        self.rounds = int(rounds)
        self.cuda = arena.flat_grad.is_cuda
        from . import config
        self.stream = config.shared_stream('allreduce', arena.flat_grad.device) if self.cuda else None     # (one per device: see there)
        self.buckets = []                      # (start_elem, end_elem, [param indices])
        self.order = []                        # bucket ids in the order their collectives are enqueued this step
        self.last_order = []                   # ... and were in the step finish() closed last
        cur, start, key = [], 0, None
        for i, (p, o) in enumerate(zip(arena.params, arena.offsets)):
            k = group_of(arena.names[id(p)]) if group_of is not None else None
            cut = (k != key) if group_of is not None else ((o + p.numel() - start) * 4 > bucket_bytes)
            if cur and cut:
                self.buckets.append((start, o, cur))
                cur, start = [], o
            cur.append(i)
            key = k
        self.buckets.append((start, arena.numel, cur))
        self.bucket_of = {}
        for b, (_, _, idxs) in enumerate(self.buckets):
            for i in idxs:
                self.bucket_of[i] = b
        self.pending = [0] * len(self.buckets)
        self.handles = []
        self.launched = []
        self._hooks = []
        self._timing = None                    # enable_timing(): [(start, end) events of every collective], [(w0, w1) of finish()]
        self._epoch = 0                        # one per reducing step (reset()): scopes the kernel-announced marks below
        self._direct = [-1] * len(arena.params)
        if self.active:
            for i, p in enumerate(arena.params):
                from_autograd, from_kernel = self._make_hooks(i)
                self._hooks.append(p.register_post_accumulate_grad_hook(from_autograd))   # gradients arriving through autograd
                p._wmz_ready = from_kernel                                             # ... and those written in place
        self.reset()

    def reset(self):
        self.pending = [len(idxs) * self.rounds for (_, _, idxs) in self.buckets]
        self.handles = []
        self.launched = [False] * len(self.buckets)
        self._epoch += 1                       # (marks of the step just closed no longer match: nothing carries over)

    def _make_hooks(self, i):
        """A parameter's gradient is counted ONCE per backward pass, whichever way it arrives.  A backward kernel that wrote it
        straight into the arena says so itself (`_wmz_ready`, inside the autograd node's backward) and hands autograd None --
        and torch (2.10) then still runs the parameter's post-accumulate hooks for that undefined gradient when the node
        returns: that echo must not count a second time (with several autograd nodes per bucket -- the op-by-op path: a
        layer's feed-forward node, then its attention node -- the doubled counts of the first node's parameters launched the
        bucket's all-reduce before the second node had written its gradients: replicas diverged; found by the fp32 two-rank
        test of round 4).  The mark is the step's epoch, not a flag the echo clears: whether torch sends the echo is version
        dependent, and a flag left standing by a missing echo would swallow a genuine autograd gradient of a later step."""
        def count():
            b = self.bucket_of[i]
            self.pending[b] -= 1
            if self.pending[b] == 0:
                self._launch(b)

        def from_autograd(_param):
            if self._direct[i] == self._epoch:     # the echo of a gradient the kernel announced in THIS step
                return
            count()

        def from_kernel():
            self._direct[i] = self._epoch
            count()
        return from_autograd, from_kernel

    def _launch(self, b):
        s, e, _ = self.buckets[b]
        view = self.arena.flat_grad[s:e]
        self.launched[b] = True
        self.order.append(b)
        if self.cuda:
            self.stream.wait_stream(torch.cuda.current_stream())      # the bucket's gradients are complete
            from . import ops
            ops.wgrad_side_wait(self.stream)                          # ... including those launched on the wgrad side stream
            with torch.cuda.stream(self.stream):
                # (under hipGraph capture the side stream has just joined the capture through wait_stream: the collective
                #  becomes a graph node behind the bucket's last backward kernel; timing events are an eager-mode probe)
                if self._timing is not None and not torch.cuda.is_current_stream_capturing():
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg)
                    e1.record()
                    self._timing[0].append((e0, e1))
                else:
                    dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg)
        else:
            self.handles.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def finish(self):
        """Call after backward(): reduces any bucket whose hooks did not all fire (unused parameters) and makes the
        compute stream wait for the side stream.  Returns the factor consumers must apply to the summed gradient."""
        if self.active:
            for b in range(len(self.buckets)):
                if not self.launched[b]:
                    self._launch(b)
            if self.cuda:
                if self._timing is not None and not torch.cuda.is_current_stream_capturing():
                    w0, w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    w0.record()
                    torch.cuda.current_stream().wait_stream(self.stream)
                    w1.record()
                    self._timing[1].append((w0, w1))
                else:
                    torch.cuda.current_stream().wait_stream(self.stream)
            else:
                for h in self.handles:
                    h.wait()
        self.last_order, self.order = self.order, []
        self.reset()
        return 1.0 / self.world

    def enable_timing(self, on=True):
        """Record HIP events around every collective (side stream) and around finish()'s wait (compute stream)."""
        self._timing = ([], []) if on else None

    def timing_summary(self):
        """After a device synchronise: {'allreduce_ms': time the collectives took on the side stream, 'exposed_ms': time the
        compute stream stood waiting for them in finish(), 'overlap_fraction': share of the collective time hidden under
        the backward}, summed over the steps since enable_timing()."""
        if not self._timing:
            return None
        total = sum(a.elapsed_time(b) for a, b in self._timing[0])
        exposed = sum(a.elapsed_time(b) for a, b in self._timing[1])
        return {'allreduce_ms': total, 'exposed_ms': exposed, 'collectives': len(self._timing[0]),
                'overlap_fraction': (1.0 - min(exposed, total) / total) if total > 0 else None}

    def remove_hooks(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
        for p in self.arena.params:
            if hasattr(p, '_wmz_ready'):
                del p._wmz_ready


def broadcast_parameters(arena, src=0, process_group=None):
    """Make every rank start from rank `src`'s weights (one broadcast of the flat arena)."""
    if dist.is_initialized() and dist.get_world_size(process_group) > 1:
        dist.broadcast(arena.flat_param, src=src, group=process_group)
