"""Timing-only experiment for VERDICT r03 item 6 (two launch chains for the training step): two trainers with half the batch each,
their whole step bodies captured into ONE hipGraph on two streams, against one trainer with the full batch.  The two chains share
the library's scratch buffers here (racy numerics, the timing is what is read): an upper bound on what splitting the data chains
of one trainer could buy (that form would still run the weight gradients over the union batch)."""
import sys, time, torch
sys.path.insert(0, '.')
from world_modelz_amd import config
from world_modelz_amd.main import VqVideoDiffusionModel
from world_modelz_amd.train import DenoiserTrainer
config.set_compute_dtype(torch.bfloat16)
def make(B):
    torch.manual_seed(42)
    m = VqVideoDiffusionModel(data_shape=(32, 16, 16), dim=256, num_classes=1024, extents=(3, 3, 3), depth=4, dim_head=128, mlp_dim=256, heads=1).cuda().train()
    tr = DenoiserTrainer(m, 1024, lr=1e-4, warmup=500, max_steps=200000)
    z = torch.randint(0, 1024, (B, 32, 16, 16), device='cuda')
    tr.enable_graph(z)
    return tr, z
def timeit(fn, n=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
full, z8 = make(8)
r8 = torch.full((8,), 0.5)
t_full = timeit(lambda: full.train_step(z8, r=r8))
print(f'one trainer, B = 8, one graph: {t_full:.3f} ms/step', flush=True)
ta, z4a = make(4)
tb, z4b = make(4)
r4 = torch.full((4,), 0.5)
t_seq = timeit(lambda: (ta._graph.replay(), tb._graph.replay()))
print(f'two trainers, B = 4 each, their graphs one after the other: {t_seq:.3f} ms per pair', flush=True)
g = torch.cuda.CUDAGraph()
sb = torch.cuda.Stream()
torch.cuda.synchronize()
with torch.cuda.graph(g):
    cur = torch.cuda.current_stream()
    sb.wait_stream(cur)
    with torch.cuda.stream(sb):
        tb._graph_body()
    ta._graph_body()
    cur.wait_stream(sb)
t_par = timeit(lambda: g.replay())
print(f'two trainers, B = 4 each, both step bodies in ONE graph on two streams: {t_par:.3f} ms per pair ({(1 - t_par / t_full) * 100:+.1f} % vs one trainer)', flush=True)
