"""Soak of the data-parallel step in an RCCL world of one: graphed DenoiserTrainer(distributed=True) (per-layer all-reduce buckets
captured on the shared side stream), the sampler on the same model every 25 steps, a fresh capture (enable_graph) every 50, a
second trainer + graph runner made and dropped every 50 -- the ingredients of the round-4 stream-pool crash, in one process."""
import os, sys, gc, torch
sys.path.insert(0, '.')
import torch.distributed as dist
from world_modelz_amd import config, sample
from world_modelz_amd.graph import GraphedForward
from world_modelz_amd.main import VqVideoDiffusionModel
from world_modelz_amd.train import DenoiserTrainer
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29537'); os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
config.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
C = 128
def model():
    return VqVideoDiffusionModel(data_shape=(4, 16, 16), dim=256, num_classes=C, extents=(1, 1, 1), depth=3, dim_head=128, mlp_dim=256, heads=1).cuda()
m = model()
tr = DenoiserTrainer(m, C, lr=3e-4, warmup=10, max_steps=100000, distributed=True)
assert tr.reducer is not None and tr.reducer.active
z = torch.randint(0, C, (4, 4, 16, 16), device='cuda')
tr.enable_graph(z)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
losses = []
for it in range(steps):
    loss, gn = tr.train_step(z)
    losses.append(loss)
    assert loss == loss and gn == gn, (it, loss, gn)
    if it % 25 == 24:
        m.eval()
        frames, _ = sample.sample_frames(m, z[:2], C, num_frames=1, num_eval_iterations=6, sample_topk=20)
        m.train()
    if it % 50 == 49:
        tr.enable_graph(z, keep_warmup_updates=True)           # a fresh RCCL-capturing graph
        m2 = model()
        t2 = DenoiserTrainer(m2, C, distributed=True)
        t2.enable_graph(z)
        t2.train_step(z)
        g2 = GraphedForward(m2.eval(), z)
        g2(z)
        t2._graph = None
        del t2, g2, m2
        gc.collect(); torch.cuda.synchronize()
        print(f'step {it + 1}: loss {sum(losses[-25:]) / 25:.4f}, allocated {torch.cuda.memory_allocated() >> 20} MiB, shared streams {sorted(k[0] for k in config._shared_streams)}', flush=True)
assert sum(losses[-25:]) < sum(losses[:25])
tr._graph = None
gc.collect(); torch.cuda.synchronize()
print('SOAK OK', flush=True)
os._exit(0)
