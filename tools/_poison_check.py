"""Do results depend on what freed device memory holds?  Runs the fused step of each model at the headline shape twice in one
process: first in a fresh process (the allocator hands out zero pages), then after every cached block has been filled with NaN
(and with a huge finite value) and freed.  Loss and gradients must be bitwise the same for the deterministic parts (loss) and
within atomics' reordering noise for the gradients.
    python tools/_poison_check.py        (GPU box)"""
import os, sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent / "tests"))
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import numpy as np, torch
import test_gpu_gemm_bf16x3 as G


def poison(val):
    blocks = []
    try:
        for _ in range(24):
            blocks.append(torch.full((256 << 20,), val, dtype=torch.float32, device="cuda"))  # 1 GB each
    except RuntimeError:
        pass
    torch.cuda.synchronize()
    del blocks
    torch.cuda.synchronize()


CASES = [(sys.argv[1], bool(int(sys.argv[2])))] if len(sys.argv) > 2 else \
    [(n, f) for n in ("DistMult", "ComplEx", "RotatE", "TransE", "pRotatE") for f in ((False, True) if n in ("DistMult", "ComplEx") else (False,))]
SHAPE = tuple(int(v) for v in sys.argv[3:6]) if len(sys.argv) > 5 else (1000, 1024, 256)
for name, no_mfma in CASES:
    if True:
        ref = {}
        for tag, val in (("fresh", None), ("nan", float("nan")), ("huge", 3e38)):
            if val is not None:
                poison(val)
            out = {}
            for mode in ("head-batch", "tail-batch"):
                l, e, r = G._grads(name, *SHAPE, "1", mode, no_mfma=no_mfma)
                out[mode] = (np.float32(l), e, r)
            if tag == "fresh":
                ref = out
                continue
            for mode in out:
                l0, e0, r0 = ref[mode]
                l, e, r = out[mode]
                bad = (not np.isfinite(e).all()) or (not np.isfinite(r).all())
                print(f"{name:9s} no_mfma={int(no_mfma)} {mode:10s} after {tag:5s}: loss {l!r} vs {l0!r} {'SAME' if l.tobytes() == l0.tobytes() else 'DIFF'}"
                      f"  max|dE| {np.abs(e - e0).max():.3e} (scale {np.abs(e0).max():.3e})  max|dR| {np.abs(r - r0).max():.3e}  {'NON-FINITE' if bad else ''}", flush=True)
