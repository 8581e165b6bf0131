"""CPU (gloo, world_size 2): the flat-arena bucketed gradient all-reduce of world_modelz_amd/parallel.py.
The reducer is device-agnostic; on the GPU box the same code runs over RCCL with the side-stream overlap."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model():
    torch.manual_seed(7)
    return torch.nn.Sequential(torch.nn.Linear(12, 40), torch.nn.GELU(), torch.nn.Linear(40, 24), torch.nn.LayerNorm(24),
                               torch.nn.Linear(24, 5))


def _worker(rank, world, port, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from world_modelz_amd.parallel import BucketedAllReduce, FlatArena, broadcast_parameters
    torch.set_num_threads(1)
    model = _model()
    if rank == 1:                                       # perturb: broadcast must restore rank 0's weights
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)
    arena = FlatArena(model)
    broadcast_parameters(arena)
    red = BucketedAllReduce(arena, bucket_bytes=1024)   # tiny buckets -> several collectives per step
    assert len(red.buckets) >= 3
    torch.manual_seed(100)
    x = torch.randn(8, 12)
    y = torch.randint(0, 5, (8,))
    for it in range(2):                                 # two steps: state resets between them
        arena.zero_grad()
        n = 8 // world
        xs, ys = x[rank * n:(rank + 1) * n], y[rank * n:(rank + 1) * n]
        torch.nn.functional.cross_entropy(model(xs), ys).backward()
        scale = red.finish()
        g = (arena.flat_grad * scale).clone()
    ret[rank] = (g, arena.flat_param.clone(), [b[:2] for b in red.buckets])
    dist.destroy_process_group()


def test_bucketed_allreduce_world2():
    mp.set_start_method('spawn', force=True)
    with mp.Manager() as mgr:
        ret = mgr.dict()
        port = _free_port()
        mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
        g0, p0, buckets = ret[0]
        g1, p1, _ = ret[1]
    assert torch.equal(p0, p1)                          # broadcast made the replicas identical
    assert torch.equal(g0, g1)                          # and every rank holds the same reduced gradient
    # single-process reference: mean loss over the union of both ranks' shards
    from world_modelz_amd.parallel import FlatArena
    model = _model()
    arena = FlatArena(model)
    torch.manual_seed(100)
    x = torch.randn(8, 12)
    y = torch.randint(0, 5, (8,))
    loss = 0.5 * (torch.nn.functional.cross_entropy(model(x[:4]), y[:4]) + torch.nn.functional.cross_entropy(model(x[4:]), y[4:]))
    loss.backward()
    assert torch.allclose(g0, arena.flat_grad, rtol=1e-5, atol=1e-7)
    assert buckets[0][0] == 0 and buckets[-1][1] == arena.numel


def test_bucketed_allreduce_world4():
    """The same over FOUR ranks (rank counts above two had never run any of this code: VERDICT r04): identical replicas after the
    broadcast, the same reduced gradient on every rank, equal to the single-process gradient of the mean loss over the union."""
    mp.set_start_method('spawn', force=True)
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(4, _free_port(), ret), nprocs=4, join=True)
        res = [ret[r] for r in range(4)]
    for g, p_, _ in res[1:]:
        assert torch.equal(p_, res[0][1]) and torch.equal(g, res[0][0])
    from world_modelz_amd.parallel import FlatArena
    model = _model()
    arena = FlatArena(model)
    torch.manual_seed(100)
    x = torch.randn(8, 12)
    y = torch.randint(0, 5, (8,))
    loss = sum(torch.nn.functional.cross_entropy(model(x[2 * r:2 * r + 2]), y[2 * r:2 * r + 2]) for r in range(4)) / 4
    loss.backward()
    assert torch.allclose(res[0][0], arena.flat_grad, rtol=1e-5, atol=1e-7)


class _DirectLinear(torch.autograd.Function):
    """y = x W^T whose backward writes dW straight into the parameter's arena slice and announces it (`_wmz_ready`), returning
    None to autograd -- what the HIP backward kernels do (backward._emit)."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x)
        ctx.w = w
        return x @ w.t()

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        w = ctx.w
        w._wmz_grad.add_(dy.t() @ x)
        ready = getattr(w, '_wmz_ready', None)
        if ready is not None:
            ready()
        return dy @ w.detach(), None


class _TwoNodeBlock(torch.nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(11)
        self.a = torch.nn.Parameter(torch.randn(16, 8) * 0.3)
        self.b = torch.nn.Parameter(torch.randn(16, 16) * 0.3)
        self.c = torch.nn.Parameter(torch.randn(4, 16) * 0.3)

    def forward(self, x):
        h = torch.tanh(_DirectLinear.apply(x, self.a))
        h = torch.tanh(_DirectLinear.apply(h, self.b))
        return _DirectLinear.apply(h, self.c)


def _worker_direct(rank, world, port, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from world_modelz_amd.parallel import BucketedAllReduce, FlatArena
    torch.set_num_threads(1)
    model = _TwoNodeBlock()
    arena = FlatArena(model)
    red = BucketedAllReduce(arena, group_of=lambda name: 'all')         # ONE bucket fed by three autograd nodes
    assert len(red.buckets) == 1
    torch.manual_seed(100)
    x = torch.randn(8, 8)
    launched_with = []
    orig = red._launch
    red._launch = lambda b: (launched_with.append([float(p._wmz_grad.abs().sum()) > 0 for p in arena.params]), orig(b))[1]
    for it in range(2):
        arena.zero_grad()
        model(x[rank * 4:(rank + 1) * 4]).square().mean().backward()
        scale = red.finish()
    ret[rank] = ((arena.flat_grad * scale).clone(), launched_with)
    dist.destroy_process_group()


def _worker_echo(rank, world, port, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from world_modelz_amd.parallel import BucketedAllReduce, FlatArena
    torch.set_num_threads(1)
    model = _TwoNodeBlock()
    arena = FlatArena(model)
    red = BucketedAllReduce(arena, group_of=lambda name: 'all', always=True)
    log = []
    orig = red._launch
    red._launch = lambda b: (log.append('launch'), orig(b))[1]
    hooks = [red._make_hooks(i) for i in range(len(arena.params))]
    # step 1: every gradient announced by its kernel, and NO echo from autograd (the other torch behaviour)
    for _, from_kernel in hooks:
        from_kernel()
    assert log == ['launch']                                    # the bucket left when its last gradient was announced
    red.finish()
    log.append('finish')
    # step 2: the same parameters arrive through autograd -- none may be mistaken for an echo of step 1
    for from_autograd, _ in hooks:
        from_autograd(None)
    ok2 = log == ['launch', 'finish', 'launch']
    red.finish()
    # step 3: kernel announcements WITH their echoes, in between and behind
    n0 = len(log)
    for j, (from_autograd, from_kernel) in enumerate(hooks):
        from_kernel()
        if j % 2 == 0:
            from_autograd(None)
    launched_early = len(log) == n0 + 1
    for j, (from_autograd, _) in enumerate(hooks):
        if j % 2 == 1:
            from_autograd(None)
    ret[rank] = (ok2, launched_early, len(log) == n0 + 1)
    red.finish()
    dist.destroy_process_group()


def test_reducer_counts_a_gradient_once_whether_or_not_autograd_echoes_a_kernel_announcement():
    """parallel.BucketedAllReduce: a gradient a backward kernel wrote in place is announced by the kernel; torch 2.10 then still runs the
    parameter's post-accumulate hook for the None the node returned (an echo that must not count), other versions may not.  The
    dedup mark is scoped to the reducing step: without an echo nothing stale survives into the next step (a genuine autograd gradient
    there is counted, the bucket leaves before finish()), with echoes -- early or late -- the bucket leaves exactly once."""
    mp.set_start_method('spawn', force=True)
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker_echo, args=(1, _free_port(), ret), nprocs=1, join=True)
        assert ret[0] == (True, True, True), ret[0]


def test_bucket_waits_for_every_autograd_node_that_writes_in_place():
    """A bucket whose gradients are written in place by SEVERAL autograd nodes (the op-by-op HIP backward: feed-forward node, then
    attention node of one layer) is reduced once ALL of them have written: torch runs a parameter's post-accumulate hooks even
    for the None an in-place node returns, and counting that echo launched the collective after the first node (replicas
    diverged on the GPU in fp32; this is the CPU restatement)."""
    mp.set_start_method('spawn', force=True)
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker_direct, args=(2, _free_port(), ret), nprocs=2, join=True)
        (g0, l0), (g1, _) = ret[0], ret[1]
    assert torch.equal(g0, g1)
    assert len(l0) == 2 and all(all(flags) for flags in l0), l0          # one launch per step, every gradient already written
    model = _TwoNodeBlock()
    from world_modelz_amd.parallel import FlatArena
    arena = FlatArena(model)
    torch.manual_seed(100)
    x = torch.randn(8, 8)
    (0.5 * (model(x[:4]).square().mean() + model(x[4:]).square().mean())).backward()
    assert torch.allclose(g0, arena.flat_grad, rtol=1e-5, atol=1e-7)


def test_flat_arena_views_and_zero_grad():
    from world_modelz_amd.parallel import FlatArena
    model = _model()
    ref = [p.detach().clone() for p in model.parameters()]
    arena = FlatArena(model)
    for p, r in zip(model.parameters(), ref):
        assert torch.equal(p, r)
        assert p.data_ptr() >= arena.flat_param.data_ptr()
        assert p.grad is not None and p.grad.data_ptr() >= arena.flat_grad.data_ptr()
    model(torch.randn(3, 12)).sum().backward()
    assert float(arena.flat_grad.abs().sum()) > 0
    arena.zero_grad()
    assert float(arena.flat_grad.abs().sum()) == 0
    with torch.no_grad():
        arena.flat_param.add_(1.0)                      # arena writes are visible through the parameters
    for p, r in zip(model.parameters(), ref):
        assert torch.allclose(p, r + 1.0)


def test_lr_schedule_and_sampler_match_golden():
    import math
    from conftest import load_golden
    from world_modelz_amd.train import LossAwareSamplerEma, lr_at
    g = load_golden('step_tiny')
    traj = g['lr_trajectory'].tolist()
    mine = [lr_at(s, float(g['base_lr']), int(g['warmup']), int(g['max_steps'])) for s in range(1, len(traj) + 1)]
    assert all(math.isclose(a, b, rel_tol=1e-6, abs_tol=1e-12) for a, b in zip(mine, traj))
    s = LossAwareSamplerEma(num_histogram_buckets=10, uniform_p=0.01, alpha=0.9, warmup=2)
    s.update_with_losses(g['sampler/ts'], g['sampler/losses'])
    assert torch.equal(s._counts, g['sampler/counts'])
    assert torch.allclose(s._weights, g['sampler/weights_raw'], rtol=1e-6)
    assert torch.allclose(s.weights(), g['sampler/weights'], rtol=1e-6)
    r = s.sample(16)
    assert r.shape == (16,) and float(r.min()) >= 0 and float(r.max()) < 1


def test_c_abi_exports_every_declared_symbol():
    """include/wmz.h <-> libwmz_hip.so: the library loads on a CPU-only host and exports every entry point."""
    import re
    from world_modelz_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, 'include', 'wmz.h')).read()
    declared = set(re.findall(r'\b(wmz_[a-z0-9_]+)\s*\(', hdr))
    assert len(declared) >= 15
    lib = _lib.lib()
    assert lib.wmz_version() >= 100
    missing = [n for n in sorted(declared) if not hasattr(lib, n)]
    assert not missing, missing
    for n in declared - {'wmz_version', 'wmz_last_error'}:
        assert n in _lib.SIGNATURES, f'{n} has no ctypes signature'


def test_cone_planes_schedule():
    """Host logic of the last-frame dependence cone: brute-force reachability over the temporal windows."""
    from world_modelz_amd.fused import cone_planes
    for S in (1, 2, 5, 9, 32):
        for eS in (0, 1, 3):
            for depth in (1, 2, 4, 6):
                need, src = cone_planes(S, eS, depth)
                live = {S - 1}                                   # planes of layer `depth-1` output that matter
                for l in range(depth - 1, -1, -1):
                    assert need[l] == S - min(live), (S, eS, depth, l)
                    live = {k for s in live for k in range(max(0, s - eS), min(S - 1, s + eS) + 1)}
                    assert src[l] == S - min(live)


def test_untracked_loads_stay_untouched_until_their_wait():
    """layer_fused.hip fetches the residual rows with inline-asm loads that hipcc does not track (so that it does not drain
    the weight ring at their first use).  That is only safe if no instruction touches their destination registers before
    the counted s_waitcnt: tools/check_untracked.py compiles the kernel to ISA and scans exactly that span."""
    import os, shutil, subprocess, sys
    if shutil.which('hipcc') is None:
        import pytest
        pytest.skip('hipcc not on PATH')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'check_untracked.py')], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count('touches in between: 0') >= 2, r.stdout


def test_direct_conv_kernels_never_touch_loads_in_flight():
    """conv_direct.hip (convr_kernel) fetches its weight fragments with inline-asm global loads retired by counted vmcnt waits and
    its patch fragments with inline-asm LDS reads retired by counted lgkmcnt waits: nothing may read or write a destination
    register before the wait that retires it (hipcc hands the register of a DEAD asm result to the next instruction -- the round-5
    memory fault).  tools/check_untracked_conv.py compiles the file to ISA and scans every instantiation."""
    import os, shutil, subprocess, sys
    if not os.path.exists('/opt/rocm/bin/hipcc') and shutil.which('hipcc') is None:
        import pytest
        pytest.skip('hipcc not available')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'check_untracked_conv.py')], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count('touches before their waits: 0') == 14, r.stdout


def test_scoped_invalidate_names_whose_weights_moved():
    """_cast.invalidate(params) moves every operand copy's epoch (rebuilt at next use) but only the graph stamps of holders of
    those parameters (graph.GraphedForward stamps _cast.epoch_of(its tensors)): the frozen auto-encoder's captured encoder
    must survive the denoiser's optimizer steps (main.py:229-287)."""
    from world_modelz_amd import _cast
    a, b = torch.nn.Parameter(torch.zeros(3)), torch.nn.Parameter(torch.zeros(3))
    ea, eb, glob = _cast.epoch_of([a]), _cast.epoch_of([b]), _cast._epoch
    _cast.invalidate([a])
    assert _cast.epoch_of([a]) != ea and _cast.epoch_of([b]) == eb and _cast._epoch == glob + 1
    _cast.invalidate()
    assert _cast.epoch_of([b]) != eb


def test_fused_backward_kernels_are_straight_line_and_never_touch_loads_in_flight():
    """layer_fused_bwd.hip retires its compiler-untracked operand loads with counted s_waitcnt vmcnt(n): that is only safe
    for straight-line code without scratch traffic in which no instruction touches a load's destination registers before
    the wait that retires it.  tools/check_untracked_bwd.py compiles the file to ISA and simulates the vmcnt queue."""
    import os, shutil, subprocess, sys
    if shutil.which('hipcc') is None:
        import pytest
        pytest.skip('hipcc not on PATH')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'check_untracked_bwd.py')], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count('violations: 0') == 8, r.stdout


def test_lazy_one_hot_behaves_like_the_dense_encodings():
    """VectorQuantizerEMA.forward returns `encodings` (reference vq.py:39) as a stand-in that materialises the dense one-hot
    only when it is read: torch functions, tensor methods, indexing."""
    from world_modelz_amd.vq import LazyOneHot
    idx = torch.tensor([2, 0, 3, 3, 1])
    e = LazyOneHot(idx, 4)
    assert e.shape == (5, 1, 4) and e.dtype == torch.float32 and e._dense is None     # nothing built yet
    dense = torch.nn.functional.one_hot(idx, 4).float().unsqueeze(1)
    assert torch.equal(e.argmax(-1), idx.unsqueeze(1))
    assert torch.equal(torch.sum(e, dim=0), dense.sum(0))
    assert torch.equal(e.mean(dim=0), dense.mean(0)) and torch.equal(e[1], dense[1])
    x = torch.randn(5, 8)
    assert torch.allclose(torch.matmul(e.squeeze(1).t(), x), dense.squeeze(1).t() @ x)
    assert torch.equal(e * 2.0, dense * 2.0) and float(e.sum()) == 5.0


# ------------------------------------------------------------------------------------------------ VQ EMA statistics under DDP
class _TorchVqOps:
    """torch restatement of the four VQ entry points the module calls (the HIP kernels need a GPU); what this test exercises is
    the MODULE's data-parallel logic: VectorQuantizerEMA.forward all-reducing counts / dw before the EMA update (sync_stats)."""

    @staticmethod
    def vq_argmin(x, cb):
        return (x[:, None, :] - cb[None]).pow(2).sum(-1).argmin(-1)

    @staticmethod
    def vq_gather(idx, cb):
        return cb[idx]

    @staticmethod
    def vq_ema_stats(x, idx, cb, counts=None, dw=None, sqerr=None):
        counts.index_add_(0, idx, torch.ones_like(idx, dtype=counts.dtype))
        if dw is not None:
            dw.index_add_(0, idx, x)
        if sqerr is not None:
            sqerr.index_add_(0, idx, (cb[idx] - x).pow(2).sum(-1))

    @staticmethod
    def vq_tail(inp, flat, q, counts, out_dtype, Ep):
        """vq.py:67-73 (commitment loss, straight-through estimator, perplexity): what ops.vq_tail launches on the GPU."""
        qv = q.view_as(inp).to(inp.dtype)
        loss = torch.nn.functional.mse_loss(qv.detach(), inp)
        st = inp + (qv - inp).detach()
        p = counts / flat.shape[0]
        ppl = torch.exp(-torch.sum(p * torch.log(p + 1e-10)))
        if Ep != st.shape[-1]:
            st = torch.nn.functional.pad(st, (0, Ep - st.shape[-1]))
        return st.to(out_dtype), loss, ppl

    @staticmethod
    def vq_ema_update(embedding, cluster_size, activation_count, counts, dw, decay, eps):
        C = embedding.shape[-2]                                     # vq.py:44, :53-65 (laplace smoothing, batch sum / EMA count);
        activation_count += counts                                  # one latent's slices of the buffers, as the module hands them over
        cluster_size.mul_(decay).add_(counts, alpha=1 - decay)
        n = cluster_size.sum()
        cs = (cluster_size + eps) / (n + C * eps) * n
        embedding.mul_(decay).add_(dw / cs.unsqueeze(-1), alpha=1 - decay)


def _vq_on_cpu():
    from world_modelz_amd import vq as vq_mod
    vq_mod.ops = _TorchVqOps
    vq_mod.VectorQuantizerEMA._flat = lambda self, x: x.reshape(-1, self.num_latents, self.embedding_dim).float()      # (the module's reshape without its GPU-only check)
    return vq_mod


def _vq_inputs():
    g = torch.Generator().manual_seed(3)
    return [torch.randn(2, 6, 4, 8, generator=g) for _ in range(3)]           # three steps of [B=2, 6, 4, E=8] latents


def _vq_worker(rank, world, port, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    vq_mod = _vq_on_cpu()
    torch.manual_seed(11)
    q = vq_mod.VectorQuantizerEMA(8, 16)
    q.sync_stats = True
    q.train()
    perp = []
    for x in _vq_inputs():
        out = q(x[rank:rank + 1])                                    # each rank quantises its own clip
        perp.append(float(out[3]))
    ret[rank] = ({k: v.clone() for k, v in q.state_dict().items()}, q.activation_count.clone(), q.accumulated_error.clone(), perp)
    dist.destroy_process_group()


def test_vq_ema_statistics_allreduce_world2():
    """under data parallelism VectorQuantizerEMA.forward must all-reduce the batch statistics (counts, dw) before
    the EMA update of vq.py:53-65, or the codebooks diverge silently.  Two gloo ranks with sync_stats, each quantising half of
    the batch: both end with IDENTICAL codebooks, equal to the single-process module fed the whole batch."""
    mp.set_start_method('spawn', force=True)
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_vq_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
        (sd0, act0, err0, perp0), (sd1, act1, err1, perp1) = ret[0], ret[1]
    for k in sd0:
        assert torch.equal(sd0[k], sd1[k]), k                        # replicas identical, bit for bit
    assert torch.equal(act0, act1)
    from world_modelz_amd import vq as vq_real
    saved = (vq_real.ops, vq_real.VectorQuantizerEMA._flat)          # (this process's module is patched only for the test)
    try:
        _check_vq_against_single_process(sd0, act0, err0, err1, perp0, perp1)
    finally:
        vq_real.ops, vq_real.VectorQuantizerEMA._flat = saved


def _check_vq_against_single_process(sd0, act0, err0, err1, perp0, perp1):
    vq_mod = _vq_on_cpu()
    torch.manual_seed(11)
    q = vq_mod.VectorQuantizerEMA(8, 16)
    q.train()
    for x in _vq_inputs():
        q(x)                                                         # the union batch in one process, no process group
    assert torch.allclose(sd0['embedding'], q.embedding, rtol=1e-5, atol=1e-6)
    assert torch.allclose(sd0['cluster_size'], q.cluster_size, rtol=1e-6, atol=1e-7)
    assert torch.equal(act0, q.activation_count)                     # global usage counts on every rank (dead-code revival agrees)
    assert torch.allclose(err0 + err1, q.accumulated_error, rtol=1e-5, atol=1e-6)   # the error statistic stays per rank
    assert perp0 != perp1                                            # perplexity is this rank's batch (vq.py:72-73)
    # and WITHOUT the all-reduce the replicas would have diverged: the check above is not vacuous
    torch.manual_seed(11)
    qa, qb = vq_mod.VectorQuantizerEMA(8, 16), vq_mod.VectorQuantizerEMA(8, 16)
    qb.load_state_dict(qa.state_dict())
    for x in _vq_inputs():
        qa(x[0:1]); qb(x[1:2])
    assert not torch.equal(qa.embedding, qb.embedding)


def test_operand_cache_drops_the_entries_of_deleted_parameters():
    """_cast caches operand copies per parameter object; a deleted model's entries (~10 MB per 3 M parameters on the GPU) go when the
    cache has doubled since the last sweep (at least 256 entries) -- not only past 2048 entries as before round 4."""
    import gc
    from world_modelz_amd import _cast
    _cast.clear()
    _cast._sweep_at = 256
    keep = [torch.nn.Parameter(torch.randn(4, 4)) for _ in range(10)]
    for p in keep:
        _cast.operand(p, torch.bfloat16)
    for _ in range(6):
        dead = [torch.nn.Parameter(torch.randn(4, 4)) for _ in range(100)]
        for p in dead:
            _cast.operand(p, torch.bfloat16)
        del dead, p
        gc.collect()
    live = sum(1 for h in _cast._cache.values() if all(r() is not None for r in h[2]))
    assert live >= 10 and len(_cast._cache) < 400, (live, len(_cast._cache))          # 610 were inserted
    for p in keep:                                                                # the living ones still hit
        v = _cast.operand(p, torch.bfloat16)
        assert v is _cast.operand(p, torch.bfloat16)
    _cast.clear()

