"""Phases of a captured denoiser training step by wall-clock markers inside the hipGraph (wmz_debug_stamp; see stamp_vqae_step.py):
forward + loss done, backward chain done on the compute stream, weight-gradient side branch done, AdamW done.
    python3 tools/stamp_train_step.py config4 | config3 | dim96 | dim384 | refgeo"""
import sys, time, torch
sys.path.insert(0, '.')
from world_modelz_amd import config, ops
from world_modelz_amd import _lib as L
from world_modelz_amd.main import VqVideoDiffusionModel
from world_modelz_amd.train import DenoiserTrainer
which = sys.argv[1] if len(sys.argv) > 1 else 'dim384'
config.set_compute_dtype(torch.bfloat16)
buf = torch.zeros(16, dtype=torch.int64, device='cuda')


def stamp(slot):
    L.call('wmz_debug_stamp', L.ptr(buf), slot, L.stream())


orig_join = ops.wgrad_join


def join():
    if any(ent[1] for ent in ops._wgrad_side.values()):
        ops._flush_deferred()
        stamp(2)
        for ent in ops._wgrad_side.values():
            if ent[1]:
                with torch.cuda.stream(ent[0]):
                    stamp(3)
    orig_join()


ops.wgrad_join = join


class Stamped(DenoiserTrainer):
    def _graph_step(self, z, r):
        out = super()._graph_step(z, r)
        stamp(4)
        return out

    def _graph_body(self):
        stamp(0)
        out = super()._graph_body()
        stamp(5)
        return out


grid = {'config4': (8, 32, 16, 16), 'config3': (16, 16, 16, 16), 'dim96': (8, 32, 16, 16), 'dim384': (8, 32, 16, 16), 'refgeo': (64, 6, 8, 8)}[which]
dim, mlp, depth, ext = {'config4': (256, 256, 4, (3, 3, 3)), 'config3': (256, 256, 4, (3, 3, 3)), 'dim96': (96, 256, 12, (3, 1, 1)),
                        'dim384': (384, 512, 20, (3, 1, 1)), 'refgeo': (384, 512, 20, (3, 1, 1))}[which]
torch.manual_seed(42)
m = VqVideoDiffusionModel(data_shape=grid[1:], dim=dim, num_classes=1024, extents=ext, depth=depth, dim_head=128, mlp_dim=mlp, heads=1).cuda()
t = Stamped(m, 1024, lr=1e-4, warmup=500, max_steps=200000, distributed=False)
z = torch.randint(0, 1024, grid, device='cuda')
r = torch.full((grid[0],), 0.5)
t.enable_graph(z)
rows = []
for _ in range(25):
    t.train_step(z, r=r)
    torch.cuda.synchronize()
    rows.append(buf.cpu().clone())
s = torch.stack(rows[5:]).double()
med = ((s[:, 1:6] - s[:, :1]) / 100.0).median(dim=0).values.tolist()
for n, v in zip(['(unused)', 'backward chain done (compute stream)', 'side branch done', 'forward + backward returned (joined)', 'AdamW done'], med):
    if n != '(unused)':
        print(f'{v:9.1f} us  {n}')
best = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    for _ in range(10):
        t.train_step(z, r=r)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / 10)
print(f'{which}: wall clock {best * 1e3:.3f} ms per step')
