"""VQ-AE training step (train_vqae.py's step body) on 64 frames of 64x64 as hipGraph replays: wall, device time inside the
replay, host time inside hipGraphLaunch.  WMZ_WGRAD_STREAM=0: the conv weight gradients on the compute stream (A/B)."""
import sys, time, torch
sys.path.insert(0, '.')
from world_modelz_amd import config
from world_modelz_amd.train_vqae import VqAutoEncoder
from world_modelz_amd.train import VqaeTrainer
torch.manual_seed(7)
config.set_compute_dtype(torch.bfloat16)
ae = VqAutoEncoder(embedding_dim=64, num_embeddings=1024, downscale_steps=2, hidden_planes=128).cuda()
tr = VqaeTrainer(ae, distributed=False)
frames = torch.rand(64, 3, 64, 64, device='cuda')
tr.enable_graph(frames)
g = tr._graph
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
orig = g.replay
acc = {'dev': 0.0, 'launch': 0.0}
def replay():
    ev0.record()
    t = time.perf_counter()
    orig()
    acc['launch'] += time.perf_counter() - t
    ev1.record()
g.replay = replay
for _ in range(3):
    tr.train_step(frames)
acc['dev'] = acc['launch'] = 0.0
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 20
for _ in range(n):
    tr.train_step(frames)
    acc['dev'] += ev0.elapsed_time(ev1)
torch.cuda.synchronize()
print(f'VQ-AE step (wgrad side stream {config.get_wgrad_stream()}): wall {(time.perf_counter() - t0) / n * 1e3:.2f} ms, device time inside the replay '
      f'{acc["dev"] / n:.2f} ms, host time inside hipGraphLaunch {acc["launch"] / n * 1e3:.2f} ms', flush=True)
