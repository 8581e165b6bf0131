"""Soak: the reference's outer loop in miniature (main.py: train, every N steps evaluate by sampling frames) -- a graphed trainer,
and every 25 steps the sampler on the SAME model (its session re-captures: the weights moved) -- for a few hundred steps: loss finite
and falling on a fixed batch, device memory flat after the first cycles, no new streams, sampler output valid."""
import sys, torch
sys.path.insert(0, '.')
from world_modelz_amd import config, sample
from world_modelz_amd.main import VqVideoDiffusionModel
from world_modelz_amd.train import DenoiserTrainer
config.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
C = 256
m = VqVideoDiffusionModel(data_shape=(8, 16, 16), dim=256, num_classes=C, extents=(3, 3, 3), depth=4, dim_head=128, mlp_dim=256, heads=1).cuda()
tr = DenoiserTrainer(m, C, lr=3e-4, warmup=10, max_steps=100000)
z = torch.randint(0, C, (8, 8, 16, 16), device='cuda')
tr.enable_graph(z)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
mem, losses = [], []
for it in range(steps):
    loss, gn = tr.train_step(z)
    losses.append(loss)
    assert loss == loss and gn == gn, (it, loss, gn)
    if it % 25 == 24:
        m.eval()
        frames, _ = sample.sample_frames(m, z[:2], C, num_frames=1, num_eval_iterations=8, sample_topk=50)
        m.train()
        assert int(frames[0].min()) >= 0 and int(frames[0].max()) < C
        torch.cuda.synchronize()
        mem.append(torch.cuda.memory_allocated() >> 20)
        ses = next(iter(m._wmz_sampler_sessions.values()))
        print(f'step {it + 1}: loss {sum(losses[-25:]) / 25:.4f}, allocated {mem[-1]} MiB, reserved {torch.cuda.memory_reserved() >> 20} MiB, '
              f'sampler re-captures {ses.fwd.recaptures}, shared streams {len(config._shared_streams)}', flush=True)
assert sum(losses[-25:]) < sum(losses[:25]), 'loss did not fall on a fixed batch'
assert max(mem[3:]) - min(mem[3:]) <= 64, f'allocated memory moves: {mem}'
print('SOAK OK')
