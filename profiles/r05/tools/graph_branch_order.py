"""Experiment (development): when does hipGraph start the nodes of a side branch?  A main chain of short kernels on the capture
stream; after main kernel 10 a side stream forks and runs `nside` kernels; joined at the end.  Variants: the side launch recorded
BEFORE or AFTER the main chain's next kernel.  Prints the start offset of the first side kernel and the span (HIP events /
s_memrealtime are not needed: run under rocprofv3 --kernel-trace and read the trace; here: wall time per replay)."""
import sys, time, torch
x = torch.zeros(1 << 20, device='cuda')
ys = [torch.zeros(1 << 24, device='cuda') for _ in range(8)]
side = torch.cuda.Stream()


def build(nmain, nside, before, fork_at):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            x.add_(1.0)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        cur = torch.cuda.current_stream()
        for i in range(nmain):
            if i == fork_at:
                ev = torch.cuda.Event()
                ev.record(cur)
                if before:
                    side.wait_event(ev)
                    with torch.cuda.stream(side):
                        for y in ys[:nside]:
                            y.mul_(1.0001)
                    x.add_(1.0)
                else:
                    x.add_(1.0)
                    side.wait_event(ev)
                    with torch.cuda.stream(side):
                        for y in ys[:nside]:
                            y.mul_(1.0001)
            else:
                x.add_(1.0)
        cur.wait_stream(side)
    return g


for nmain in (50, 200):
    for before in (False, True):
        g = build(nmain, 6, before, 10)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            g.replay()
        torch.cuda.synchronize()
        print(f'main {nmain} side-recorded-{"before" if before else "after"}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms per replay')
