"""python tools/micro/rowrmw.py : random-row read-modify-write bandwidth by row layout (see rowrmw.hip)."""
import ctypes, os, subprocess, sys, torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "rowrmw.so")
if not os.path.exists(so):
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(here, "rowrmw.hip")])
lib = ctypes.CDLL(so)
lib.rmw_launch.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda:0")
for N, D, n in ((14541, 2000, 2000), (123182, 1000, 2500), (40943, 1000, 2500)):
    for n_arrays in (4, 2, 1):
        width = D * 4 // n_arrays  # floats per row of one array
        arrs = [torch.zeros(N, width, device=dev) for _ in range(n_arrays)]
        ptr = [a.data_ptr() for a in arrs] + [0] * (4 - n_arrays)
        g = torch.Generator(device="cpu").manual_seed(1)
        idsets = [torch.randperm(N, generator=g)[:n].to(dev) for _ in range(8)]
        st = torch.cuda.current_stream().cuda_stream
        def run(k):
            lib.rmw_launch(*ptr, n_arrays, width // 4, idsets[k % 8].data_ptr(), n, st)
        for k in range(10): run(k)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(200): run(k)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / 200
        mb = n * D * 4 * 4 * 2 / 1e6
        print(f"N={N} D={D} rows={n} arrays={n_arrays} contiguous={width * 4 / 1024:.0f}KB: {us:.1f} us  {mb / us:.2f} TB/s")
