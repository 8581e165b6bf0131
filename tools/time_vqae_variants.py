"""The captured VQ-AE training step under the scheduling variants of its weight-gradient side branch, same box, one process each
variant (best of 3 x 20 replays): product, no side branch (config.wgrad_stream 0), a marker launch on the compute stream where a
batch of weight gradients is issued (tools/stamp_vqae_step.py main)."""
import sys, time, torch
sys.path.insert(0, '.')
from world_modelz_amd import config, ops
from world_modelz_amd import _lib as L
from world_modelz_amd.train_vqae import VqAutoEncoder
from world_modelz_amd.train import VqaeTrainer
config.set_compute_dtype(torch.bfloat16)
frames = torch.rand(64, 3, 64, 64, device='cuda')
buf = torch.zeros(8, dtype=torch.int64, device='cuda')
orig_conv, orig_lin = ops._issue_pending_conv, ops._issue_pending


def marked(orig):
    def f():
        L.call('wmz_debug_stamp', L.ptr(buf), 0, L.stream())
        orig()
    return f


for name in sys.argv[1:] or ['product', 'noside', 'marker', 'product', 'noside', 'marker']:
    config.set_wgrad_stream(0 if name == 'noside' else 1)
    ops.WGRAD_BATCH = int(name[5:]) if name.startswith('batch') else 6          # batch<k>: k weight gradients to a side-branch launch
    ops._wgrad_side.clear()
    if name.startswith('prio'):                                                 # prio-1 / prio0: the side stream's priority (-1 = high)
        dev = torch.device('cuda', torch.cuda.current_device())
        ops._wgrad_side[dev] = [torch.cuda.Stream(device=dev, priority=int(name[4:])), False, None, []]
    ops._issue_pending_conv = marked(orig_conv) if name == 'marker' else orig_conv
    torch.manual_seed(7)
    ae = VqAutoEncoder(embedding_dim=64, num_embeddings=1024, downscale_steps=2, hidden_planes=128).cuda()
    tr = VqaeTrainer(ae, distributed=False)
    tr.enable_graph(frames)
    for _ in range(5):
        tr.train_step(frames)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(20):
            tr.train_step(frames)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 20)
    print(f'{name}: {best * 1e3:.3f} ms per captured step', flush=True)
    del tr, ae
