"""Phases of the captured VQ-AE training step by wall-clock markers inside the hipGraph (wmz_debug_stamp: no profiler, whose per-node
cost distorts the overlap of the weight-gradient side branch with the backward chain): start, forward + losses done, backward chain
done on the compute stream, side branch done, after the join + AdamW.  Median over 30 replays, microseconds from the step's start."""
import sys, torch
sys.path.insert(0, '.')
from world_modelz_amd import config, ops
from world_modelz_amd import _lib as L
from world_modelz_amd.train_vqae import VqAutoEncoder
from world_modelz_amd.train import VqaeTrainer
config.set_compute_dtype(torch.bfloat16)
buf = torch.zeros(64, dtype=torch.int64, device='cuda')


def stamp(slot):
    L.call('wmz_debug_stamp', L.ptr(buf), slot, L.stream())


orig_join = ops.wgrad_join


def join():
    pending = any(ent[1] for ent in ops._wgrad_side.values())
    if pending:
        ops._flush_deferred()
        stamp(3)                                   # the backward chain's end on the compute stream
        for ent in ops._wgrad_side.values():
            if ent[1]:
                with torch.cuda.stream(ent[0]):
                    stamp(4)                       # the side branch's end
    orig_join()


ops.wgrad_join = join

# every batch of side-branch weight gradients: the compute stream's clock where it is issued (its last fork point has passed),
# the side stream's clock when it starts and when it ends
batch_no = [0]
orig_issue_conv = ops._issue_pending_conv


def issue_conv():
    if not ops._pending_conv:
        return
    k = batch_no[0]
    batch_no[0] += 1
    n = min(len(ops._pending_conv), ops.WGRAD_BATCH)
    ent = ops._pending_conv[0]['ent']
    if 'main' in MODE:
        stamp(8 + 4 * k)                           # compute stream: issue point
    if 'head' in MODE:
        ent[0].wait_event(ops._pending_conv[n - 1]['fork'])
        with torch.cuda.stream(ent[0]):
            stamp(9 + 4 * k)                       # side stream: the batch may start
    rest = ops._pending_conv[n:]
    ops._pending_conv = ops._pending_conv[:n]
    orig_issue_conv()
    if 'tail' in MODE:
        with torch.cuda.stream(ent[0]):
            stamp(10 + 4 * k)                      # side stream: the batch is done
    ops._pending_conv = rest
    if rest:
        issue_conv()


MODE = sys.argv[1] if len(sys.argv) > 1 else ''      # any of main / head / tail, e.g. main+head+tail
if MODE:
    ops._issue_pending_conv = issue_conv


class Stamped(VqaeTrainer):
    def _forward_backward(self, batch):
        stamp(1)
        r_loss, latent_loss, perplexity = self.model.training_losses(batch, self.loss_name)
        loss = r_loss + self.latent_loss_weight * latent_loss
        stamp(2)
        loss.backward()
        stamp(5)
        return torch.stack([loss.detach(), r_loss.detach(), latent_loss.detach().reshape(()), perplexity.detach().reshape(())])

    def _graph_body(self):
        batch_no[0] = 0
        stamp(0)
        out = super()._graph_body()
        stamp(6)
        return out


torch.manual_seed(7)
ae = VqAutoEncoder(embedding_dim=64, num_embeddings=1024, downscale_steps=2, hidden_planes=128).cuda()
tr = Stamped(ae, distributed=False)
frames = torch.rand(64, 3, 64, 64, device='cuda')
tr.enable_graph(frames)
rows = []
for _ in range(35):
    tr.train_step(frames)
    torch.cuda.synchronize()
    rows.append(buf.cpu().clone())
t = torch.stack(rows[5:]).double()
rel = (t[:, 1:7] - t[:, :1]) / 100.0               # 100 MHz -> us
med = rel.median(dim=0).values.tolist()
names = ['zero-grad + operand refresh done', 'forward + losses done', 'backward chain done (compute stream)', 'side branch done',
         'joined, backward() returned', 'AdamW done']
for n, v in zip(names, med):
    print(f'{v:9.1f} us  {n}')
for k in range(batch_no[0]):
    v = ((t[:, 8 + 4 * k:11 + 4 * k] - t[:, :1]) / 100.0).median(dim=0).values.tolist()
    print(f'weight-gradient batch {k}: issued at {v[0]:8.1f} us (compute stream), starts {v[1]:8.1f}, done {v[2]:8.1f} (side stream)')
import time
torch.cuda.synchronize()
best = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    for _ in range(20):
        tr.train_step(frames)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / 20)
print(f'wall clock: {best * 1e3:.3f} ms per step (train_step incl. its host side)')
