"""Per-workgroup timeline of pool_bwd_x at the headline shape (debug build with -DMKB_TRACE_WG).

    python tools/wgtrace.py build          (container; writes variants/lib_trace.so)
    python tools/wgtrace.py run [fwd|bwd_q|bwd_x]     (GPU box)

Each workgroup records wall_clock64() (100 MHz) at entry, after the row-list build, after the pair loop and at exit,
plus HW_ID / XCC_ID, so the schedule (which CU ran what, when) can be reconstructed.
"""
import ctypes
import os
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
VDIR = ROOT / "variants"
sys.path.insert(0, str(ROOT))


def build():
    from mkb_amd.csrc import build as hb
    VDIR.mkdir(exist_ok=True)
    hb.build(force=True, verbose=False, extra_flags=["-DMKB_TRACE_WG"], out=VDIR / "lib_trace.so", objdir=VDIR / "obj_trace")


def run(kind="bwd_x", step=10, n_traced=30):
    step, n_traced = int(step), int(n_traced)
    kind_id = {"fwd": 0, "bwd_q": 1, "bwd_x": 2, "all": 3, "bwd1": 4}[kind]
    os.environ["MKB_HIP_LIB"] = str(VDIR / "lib_trace.so")
    import numpy as np
    import torch

    import bench
    from mkb_amd import _hip

    ctx = bench.build(torch.device("cuda", 0), 0, 1)
    for i in range(step):
        bench.run_step(ctx, i)
    torch.cuda.synchronize()
    buf = torch.zeros(8 * 4096 * 3, dtype=torch.int64, device="cuda")
    lib = _hip.lib()
    lib.mkb_debug_set_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.mkb_debug_set_trace.restype = None
    lib.mkb_debug_set_trace(buf.data_ptr(), kind_id)
    for i in range(n_traced):  # steady state (the host runs ahead of the device); the last step's records win
        bench.run_step(ctx, step + 2 * i)
    torch.cuda.synchronize()
    lib.mkb_debug_set_trace(None, kind_id)
    if kind == "bwd1":  # per-wave cycle accounts of the single-pass backward (wave loop, hand-off waits, run setup)
        t = buf.cpu().numpy().reshape(-1, 8)
        t = t[t[:, 3] != 0].astype(np.float64)
        print(f"waves {len(t)}  loop cycles mean {t[:, 0].mean():.0f} (min {t[:, 0].min():.0f} max {t[:, 0].max():.0f})  "
              f"hand-off wait mean {t[:, 1].mean():.0f} max {t[:, 1].max():.0f}  run setup mean {t[:, 2].mean():.0f}  "
              f"positions per wave mean {t[:, 4].mean():.1f} max {t[:, 4].max():.0f}")
        w3 = t[:, 3].astype(np.int64)
        p2_slots, p2_rows, p2_batch = (w3 >> 1) & 127, (w3 >> 8) & 65535, w3 >> 24
        print(f"fringe dx half per wave: slots mean {p2_slots.mean():.2f} max {p2_slots.max()}  rows mean {p2_rows.mean():.1f} max {p2_rows.max()}  "
              f"cycles in the row batches mean {p2_batch.mean():.0f} max {p2_batch.max()}")
        hv = np.argsort(-t[:, 7])[:8]
        print("  slowest waves (fringe dx cycles, slots, rows, batch cycles):", [(int(t[i, 7]), int(p2_slots[i]), int(p2_rows[i]), int(p2_batch[i])) for i in hv])
        pro = np.floor(t[:, 4] / 65536.0)
        t[:, 4] = t[:, 4] - pro * 65536.0
        print(f"prologue (entry -> first dense pass) mean {pro.mean():.0f}  fringe dx part mean {t[:, 7].mean():.0f} max {t[:, 7].max():.0f}  dense pass mean {t[:, 6].mean():.0f}")
        for w in range(16):
            m = t[:, 5] == w
            print(f"  wave {w:2d}: loop {t[m, 0].mean():8.0f}  hand-off {t[m, 1].mean():8.0f}  setup {t[m, 2].mean():7.0f}  items {t[m, 4].mean():5.1f}  dense {t[m, 6].mean():8.0f}")
        return
    if kind == "all":  # the three kernels of one step on one clock: where does the time between them go?
        raw = buf.cpu().numpy().reshape(3, 4096, 8)
        base = None
        for k, nm in enumerate(("fwd", "bwd_q", "bwd_x")):
            t = raw[k][raw[k][:, 3] != 0]
            base = t[:, 0].min() if base is None else base
            print(f"{nm:6s} first start {(t[:, 0].min() - base) / 100:7.1f}  last start {(t[:, 0].max() - base) / 100:7.1f}  "
                  f"first end {(t[:, 3].min() - base) / 100:7.1f}  last end {(t[:, 3].max() - base) / 100:7.1f}  us   ({len(t)} workgroups)")
        return
    t = buf.cpu().numpy().reshape(-1, 8)
    t = t[t[:, 3] != 0]
    t0 = t[:, 0].min()
    us = lambda x: (x - t0) / 100.0
    start, tl, tp, end = us(t[:, 0]), us(t[:, 1]), us(t[:, 2]), us(t[:, 3])
    hw, xcc, rows, bid = t[:, 4], t[:, 5] & 0xF, t[:, 6], t[:, 7]
    cu = (hw >> 8) & 0xF
    sh = (hw >> 12) & 0x1
    se = (hw >> 13) & 0x7
    key = xcc * 1000 + se * 100 + sh * 10 + cu  # not unique digits, fine for grouping
    key = (xcc << 12) | (se << 8) | (sh << 4) | cu
    print(f"{kind}: workgroups {len(t)}  kernel span {end.max():.1f} us   distinct CUs {len(set(key.tolist()))}")
    heavy = rows >= rows.max() * 0.9
    for name, m in (("heavy", heavy), ("light", ~heavy & (rows > 0)), ("empty", rows == 0)):
        if m.sum() == 0:
            continue
        print(f"{name:6s} n={m.sum():4d} start {start[m].min():6.1f}..{start[m].max():6.1f}  end {end[m].min():6.1f}..{end[m].max():6.1f}  "
              f"dur mean {np.mean(end[m] - start[m]):6.1f} max {np.max(end[m] - start[m]):6.1f} | list {np.mean(tl[m] - start[m]):5.1f} "
              f"loop {np.mean(tp[m] - tl[m]):6.1f} flush {np.mean(end[m] - tp[m]):5.1f}  rows {rows[m].mean():.0f}")
    # co-location of heavy workgroups
    from collections import Counter
    c = Counter(key[heavy].tolist())
    print("heavy workgroups per CU histogram:", sorted(Counter(c.values()).items()))
    per_xcc = Counter(xcc[heavy].tolist())
    print("heavy per XCC:", sorted(per_xcc.items()))
    # duration of heavy workgroups vs co-residents
    for k in (1, 2, 3):
        sel = [kk for kk, v in c.items() if v == k]
        if sel:
            m = heavy & np.isin(key, sel)
            print(f"  CUs with {k} heavy: mean heavy dur {np.mean(end[m] - start[m]):.1f} us, last end {end[m].max():.1f}")
    # every workgroup: how many share a CU, and when each CU / XCC goes idle
    call = Counter(key.tolist())
    print("workgroups per CU histogram:", sorted(Counter(call.values()).items()))
    cu_end = {}
    for kk, e in zip(key.tolist(), end.tolist()):
        cu_end[kk] = max(cu_end.get(kk, 0.0), e)
    for n in sorted(set(call.values())):
        ends = [cu_end[kk] for kk, v in call.items() if v == n]
        print(f"  CUs with {n} workgroups: {len(ends)}  idle at mean {np.mean(ends):.1f} us (min {np.min(ends):.1f}, max {np.max(ends):.1f})")
    for x in sorted(set(xcc.tolist())):
        m = xcc == x
        print(f"  XCC {x}: {m.sum()} workgroups, CUs {len(set(key[m].tolist()))}, mean dur {np.mean(end[m] - start[m]):.1f}, last end {end[m].max():.1f}, "
              f"work {rows[m].sum()}")
    order = np.argsort(end)[-8:]
    print("last finishers:", [(int(bid[i]), int(rows[i]), round(float(start[i]), 1), round(float(end[i]), 1)) for i in order])


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    else:
        run(*sys.argv[2:5])
