"""Where does a graphed forward step lose time between a bare graph replay loop and GraphedForward.__call__?"""
import sys, time, torch
sys.path.insert(0, '.')
from world_modelz_amd import config
from world_modelz_amd.main import VqVideoDiffusionModel
from world_modelz_amd.graph import GraphedForward
torch.manual_seed(42)
config.set_compute_dtype(torch.bfloat16); config.set_last_frame_cone(False)
m = VqVideoDiffusionModel(data_shape=(32, 16, 16), dim=256, num_classes=1024, extents=(3, 3, 3), depth=4, dim_head=128, mlp_dim=256, heads=1).cuda().eval()
z = torch.randint(0, 1025, (8, 32, 16, 16), device='cuda')
r = GraphedForward(m, z)
def timeit(fn, tag, n=50):
    with torch.no_grad():
        for _ in range(150): fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n): fn()
        t_enq = time.perf_counter() - t0
        torch.cuda.synchronize()
        print(f'{tag}: {(time.perf_counter() - t0) / n * 1e3:.4f} ms/step (enqueue {t_enq / n * 1e3:.4f})', flush=True)
timeit(lambda: r.graph.replay(), 'bare replay          ')
timeit(lambda: r(r.static_in), 'runner(static_in)    ')
timeit(lambda: r(z), 'runner(z) (copy)     ')
timeit(lambda: (r._stamp(), r.graph.replay()), 'stamp + replay       ')
timeit(lambda: r.graph.replay(), 'bare replay again    ')
