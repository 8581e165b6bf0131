"""Config-5 (sparse model) training step, graph replay and eager (tools/prof_sparse.py is the rocprofv3 driver)."""
import sys, time, torch
sys.path.insert(0, '.')
from world_modelz_amd import config
from world_modelz_amd.sparse_diffusion import VqSparseDiffusionModel
from world_modelz_amd.train import SparseDenoiserTrainer
torch.manual_seed(43)
config.set_compute_dtype(torch.bfloat16)
sm = VqSparseDiffusionModel(shape=(64, 16, 16), dim=512, num_classes=8192, depth=8, dim_head=128, mlp_dim=1024, heads=4).cuda()
st = SparseDenoiserTrainer(sm, 8192, num_context=512, lr=1e-4, warmup=500, distributed=False)
zs = torch.randint(0, 8192, (6, 64, 16, 16), device='cuda')
rs = torch.full((6,), 0.5)
for mode in ('eager', 'graph'):
    if mode == 'graph':
        st.enable_graph(zs)
    for _ in range(3): st.train_step(zs, r=rs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): st.train_step(zs, r=rs)
    torch.cuda.synchronize()
    print(f'config 5 {mode}: {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms/step  (wgrad side stream {config.get_wgrad_stream()})', flush=True)

# where a replayed step's wall time goes: device time between two events around the replay vs the wall period
g = st._graph
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
orig = g.replay
acc = {'dev': 0.0, 'launch': 0.0, 'n': 0}
def replay():
    ev0.record()
    t = time.perf_counter()
    orig()
    acc['launch'] += time.perf_counter() - t
    ev1.record()
g.replay = replay
import types
t0 = time.perf_counter()
for _ in range(20):
    st.train_step(zs, r=rs)
    acc['dev'] += ev0.elapsed_time(ev1); acc['n'] += 1
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 20 * 1e3
print(f'config 5 replay: wall {wall:.2f} ms/step, device time inside the replay {acc["dev"] / acc["n"]:.2f} ms, host time inside hipGraphLaunch {acc["launch"] / acc["n"] * 1e3:.2f} ms', flush=True)
