"""Feasibility probe (round 4): how much of an HBM-bound side-stream kernel can the VALU-bound launches of the headline step
absorb?  A copy of X MB is launched on a second stream at the start of every step; the main stream joins it at the step's end.
    python tools/_overlap_probe.py            (GPU box)
"""
import sys, time, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch
import bench

dev = torch.device("cuda", 0)
ctx = bench.build(dev, 0, 1)
for i in range(40):
    bench.run_step(ctx, i)
torch.cuda.synchronize()
side = torch.cuda.Stream()
main = torch.cuda.current_stream()


def timed(mb, steps=300, where="start"):
    n = max(1, int(mb * 1e6 / 4))
    src = torch.empty(n, device=dev).normal_()
    dst = torch.empty_like(src)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        if mb > 0:
            ev = torch.cuda.Event()
            ev.record(main)
            side.wait_event(ev)
            with torch.cuda.stream(side):
                dst.copy_(src)
                ev2 = torch.cuda.Event()
                ev2.record(side)
        bench.run_step(ctx, 1000 + i)
        if mb > 0:
            main.wait_event(ev2)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def alone(mb, steps=300):
    n = max(1, int(mb * 1e6 / 4))
    src = torch.empty(n, device=dev).normal_()
    dst = torch.empty_like(src)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        dst.copy_(src)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for rep in range(2):
    for mb in (0, 20, 40, 80, 160):
        print(f"copy of {mb} MB (x2 traffic): step {timed(mb):.4f} ms   copy alone {alone(mb) if mb else 0:.4f} ms", flush=True)
