"""Config 5's captured training step, timed without a profiler: the fused context draw (wmz_sparse_draw_context) against the torch-op
prologue, kernels per replay counted by torch's graph debug dump being unavailable -> see tools/prof_sparse_graph.py for counts."""
import sys, time, torch
sys.path.insert(0, '.')
from world_modelz_amd import config
from world_modelz_amd.sparse_diffusion import VqSparseDiffusionModel
from world_modelz_amd.train import SparseDenoiserTrainer
config.set_compute_dtype(torch.bfloat16)
zs = torch.randint(0, 8192, (6, 64, 16, 16), device='cuda')
rs = torch.full((6,), 0.5)
for fused in (True, False, True, False):
    torch.manual_seed(43)
    sm = VqSparseDiffusionModel(shape=(64, 16, 16), dim=512, num_classes=8192, depth=8, dim_head=128, mlp_dim=1024, heads=4).cuda()
    st = SparseDenoiserTrainer(sm, 8192, num_context=512, lr=1e-4, warmup=500, distributed=False)
    st.use_fused_context = fused
    st.enable_graph(zs)
    for _ in range(5):
        st.train_step(zs, r=rs)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(20):
            st.train_step(zs, r=rs)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 20)
    print(f'fused context draw {fused}: {best * 1e3:.3f} ms per captured step', flush=True)
    del st, sm

# where a replay's time goes: host launch call, device span (events around the launch), back-to-back replays without read-back
torch.manual_seed(43)
sm = VqSparseDiffusionModel(shape=(64, 16, 16), dim=512, num_classes=8192, depth=8, dim_head=128, mlp_dim=1024, heads=4).cuda()
st = SparseDenoiserTrainer(sm, 8192, num_context=512, lr=1e-4, warmup=500, distributed=False)
st.enable_graph(zs)
for _ in range(5):
    st.train_step(zs, r=rs)
torch.cuda.synchronize()
g = st._graph
host, span = [], []
for _ in range(20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    t0 = time.perf_counter()
    g.replay()
    host.append(time.perf_counter() - t0)
    e1.record()
    torch.cuda.synchronize()
    span.append(e0.elapsed_time(e1))
print(f'graph.replay() host call {min(host) * 1e3:.3f} ms (median {sorted(host)[10] * 1e3:.3f}), device span between events {min(span):.3f} ms '
      f'(median {sorted(span)[10]:.3f})')
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    g.replay()
torch.cuda.synchronize()
print(f'20 replays back to back, no read-back: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms each')
