"""Phases of config 5's captured training step by wall-clock markers inside the hipGraph (wmz_debug_stamp; see stamp_vqae_step.py):
context drawn, forward done, linear + cross-entropy done, backward chain done on the compute stream, side branch done, AdamW done."""
import sys, time, torch
sys.path.insert(0, '.')
from world_modelz_amd import config, ops, train
from world_modelz_amd import _lib as L
from world_modelz_amd import functional as Fw
from world_modelz_amd.sparse_diffusion import VqSparseDiffusionModel
config.set_compute_dtype(torch.bfloat16)
buf = torch.zeros(16, dtype=torch.int64, device='cuda')


def stamp(slot):
    L.call('wmz_debug_stamp', L.ptr(buf), slot, L.stream())


orig_join = ops.wgrad_join


def join():
    if any(ent[1] for ent in ops._wgrad_side.values()):
        ops._flush_deferred()
        stamp(4)
        for ent in ops._wgrad_side.values():
            if ent[1]:
                with torch.cuda.stream(ent[0]):
                    stamp(5)
    orig_join()


ops.wgrad_join = join


class Stamped(train.SparseDenoiserTrainer):
    def forward_backward(self, tokens, indices, target, loss_scale=1.0):
        stamp(1)
        m = self.model
        h = Fw.embed_tokens_indexed(tokens, indices, m.embedding.weight, m.pos_emb_s.weight, m.pos_emb_h.weight, m.pos_emb_w.weight, m.shape)
        h = m.transformer.forward_compute(h)
        stamp(2)
        mean, rows = train.linear_cross_entropy(h.reshape(-1, h.shape[-1]), m.logit_proj.weight, m.logit_proj.bias, target.reshape(-1),
                                                chunk=4096, grad_scale=loss_scale, side_branch=True)
        stamp(3)
        mean.backward()
        stamp(6)
        return rows.view(tokens.shape[0], -1).mean(dim=1), mean.detach()

    def _graph_body(self):
        stamp(0)
        out = super()._graph_body()
        stamp(7)
        return out


torch.manual_seed(43)
sm = VqSparseDiffusionModel(shape=(64, 16, 16), dim=512, num_classes=8192, depth=8, dim_head=128, mlp_dim=1024, heads=4).cuda()
st = Stamped(sm, 8192, num_context=512, lr=1e-4, warmup=500, distributed=False)
zs = torch.randint(0, 8192, (6, 64, 16, 16), device='cuda')
rs = torch.full((6,), 0.5)
st.enable_graph(zs)
rows = []
for _ in range(35):
    st.train_step(zs, r=rs)
    torch.cuda.synchronize()
    rows.append(buf.cpu().clone())
t = torch.stack(rows[5:]).double()
med = ((t[:, 1:8] - t[:, :1]) / 100.0).median(dim=0).values.tolist()
for n, v in zip(['zero-grad, operand refresh, context drawn', 'forward done', 'linear + cross-entropy done', 'backward chain done (compute stream)',
                 'side branch done', 'joined, backward() returned', 'AdamW done'], med):
    print(f'{v:9.1f} us  {n}')
best = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    for _ in range(20):
        st.train_step(zs, r=rs)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / 20)
print(f'wall clock: {best * 1e3:.3f} ms per step')
