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


class BucketedAllReduce:
    """Overlapped gradient all-reduce over a FlatArena.

    buckets: consecutive parameter ranges (in arena order) of at most `bucket_bytes`; parameters are registered in
    module order, the backward produces them roughly in reverse, so buckets complete back to front."""

    def __init__(self, arena, process_group=None, bucket_bytes=4 << 20):
        self.arena = arena
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.cuda = arena.flat_grad.is_cuda
        self.stream = torch.cuda.Stream() if self.cuda else None
        self.buckets = []                      # (start_elem, end_elem, [param indices])
        cur, start = [], 0
        for i, (p, o) in enumerate(zip(arena.params, arena.offsets)):
            if cur and (o + p.numel() - start) * 4 > bucket_bytes:
                self.buckets.append((start, o, cur))
                cur, start = [], o
            cur.append(i)
        self.buckets.append((start, arena.numel, cur))
        self.bucket_of = {}
        for b, (_, _, idxs) in enumerate(self.buckets):
            for i in idxs:
                self.bucket_of[i] = b
        self.pending = [0] * len(self.buckets)
        self.handles = []
        self.launched = []
        self._hooks = []
        if self.world > 1:
            for i, p in enumerate(arena.params):
                hook = self._make_hook(i)
                self._hooks.append(p.register_post_accumulate_grad_hook(hook))     # gradients arriving through autograd
                p._wmz_ready = (lambda h=hook, q=p: h(q))                          # ... and those written in place
        self.reset()

    def reset(self):
        self.pending = [len(idxs) for (_, _, idxs) in self.buckets]
        self.handles = []
        self.launched = [False] * len(self.buckets)

    def _make_hook(self, i):
        def hook(_param):
            b = self.bucket_of[i]
            self.pending[b] -= 1
            if self.pending[b] == 0:
                self._launch(b)
        return hook

    def _launch(self, b):
        s, e, _ = self.buckets[b]
        view = self.arena.flat_grad[s:e]
        self.launched[b] = True
        if self.cuda:
            self.stream.wait_stream(torch.cuda.current_stream())      # the bucket's gradients are complete
            with torch.cuda.stream(self.stream):
                dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg)
        else:
            self.handles.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def finish(self):
        """Call after backward(): reduces any bucket whose hooks did not all fire (unused parameters) and makes the
        compute stream wait for the side stream.  Returns the factor consumers must apply to the summed gradient."""
        if self.world > 1:
            for b in range(len(self.buckets)):
                if not self.launched[b]:
                    self._launch(b)
            if self.cuda:
                torch.cuda.current_stream().wait_stream(self.stream)
            else:
                for h in self.handles:
                    h.wait()
        self.reset()
        return 1.0 / self.world

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
