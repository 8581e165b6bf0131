"""Where the host side of a captured training step goes (config 5; the same _replay serves every trainer): wall clock per statement
of _TrainerBase._replay over 200 steps, medians.  The device is idle from the step's last kernel to the next step's first one:
read-back latency + host statements + launch latency."""
import sys, time, math, statistics, torch
sys.path.insert(0, '.')
from world_modelz_amd import config, _cast
from world_modelz_amd.sparse_diffusion import VqSparseDiffusionModel
from world_modelz_amd.train import SparseDenoiserTrainer
config.set_compute_dtype(torch.bfloat16)
torch.manual_seed(43)
sm = VqSparseDiffusionModel(shape=(64, 16, 16), dim=512, num_classes=8192, depth=8, dim_head=128, mlp_dim=1024, heads=4).cuda()
st = SparseDenoiserTrainer(sm, 8192, num_context=512, lr=1e-4, warmup=500, distributed=False)
zs = torch.randint(0, 8192, (6, 64, 16, 16), device='cuda')
rs = torch.full((6,), 0.5)
st.enable_graph(zs)
for _ in range(5):
    st.train_step(zs, r=rs)
torch.cuda.synchronize()
T = {k: [] for k in ('copy_z', 'set_inputs', 'replay', 'invalidate', 'readback', 'sampler', 'total')}
for _ in range(200):
    t0 = time.perf_counter()
    st._g_z.copy_(zs, non_blocking=True)
    t1 = time.perf_counter()
    st._set_step_inputs(rs)
    t2 = time.perf_counter()
    st._graph.replay()
    t3 = time.perf_counter()
    _cast.invalidate(st.arena.params)
    t4 = time.perf_counter()
    out = st._g_out.cpu()
    t5 = time.perf_counter()
    st.sampler.update_with_losses(st._g_r_host, out[2:])
    res = float(out[0]), math.sqrt(float(out[1]))
    t6 = time.perf_counter()
    for k, v in zip(T, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, t6 - t0)):
        T[k].append(v * 1e6)
for k, v in T.items():
    print(f'{k:12s} {statistics.median(v):8.1f} us')
