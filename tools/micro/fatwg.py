"""python tools/micro/fatwg.py : launch cost of 256 fat workgroups (see fatwg.hip): event time per launch, and from the waves' own
cycle stamps: entry -> first barrier.
Round 5 (MI355X): 3.2-3.5 us per empty launch, ~450 cycles to the barrier; clearing 128 KB of LDS first: 4.4 us, 4.6 k cycles."""
import ctypes, os, subprocess, torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "fatwg.so")
if not os.path.exists(so):
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(here, "fatwg.hip")])
lib = ctypes.CDLL(so)
lib.fat_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda:0")
out = torch.zeros(256 * 16 * 3, dtype=torch.int64, device=dev)
st = torch.cuda.current_stream().cuda_stream
for lds in (0, 66 * 1024, 130 * 1024):
    for touch in (0, 1):
        if touch and lds < 128 * 1024:
            continue
        for _ in range(20): lib.fat_launch(out.data_ptr(), touch, 256, lds, st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): lib.fat_launch(out.data_ptr(), touch, 256, lds, st)
        e1.record(); torch.cuda.synchronize()
        t = out.view(-1, 3).cpu()
        t0, t1 = t[:, 0], t[:, 1]
        print(f"lds={lds >> 10} KB touch={touch}: {e0.elapsed_time(e1) * 5:.2f} us per launch (back to back); "
              f"entry -> barrier passed: mean {(t1 - t0).float().mean().item():.0f} max {(t1 - t0).max().item()} cycles")
