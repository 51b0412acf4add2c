"""Per-GPU compute time of the dimension-sharded step WITHOUT the collective: build rank 0's shard of a `world`-way
split of the headline model and run its kernels on the global batch (world x 1024 rows) on one GPU.
    python tools/shard_emulate.py 8"""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import bench  # noqa: E402
from mkb_amd import _hip, models, optim, parallel, sampling  # noqa: E402
from mkb_amd.datasets.base import subsampling_weights  # noqa: E402


def main(world):
    train_np, n_ent, n_rel = bench.load_fb15k237()
    ents, rels = {i: i for i in range(n_ent)}, {i: i for i in range(n_rel)}
    torch.manual_seed(42)
    full = models.RotatE(hidden_dim=bench.HIDDEN, entities=ents, relations=rels, gamma=bench.GAMMA)
    m = parallel.shard_dims(full, 0, world, "cuda")
    m._dim_shard = (0, 1, m._dim_shard[2], m._dim_shard[3])  # no collective
    ns = sampling.NegativeSampling(size=bench.K, train_triples=train_np, entities=ents, relations=rels, seed=42)
    opt = optim.Adam([p for p in m.parameters() if p.requires_grad and p is not m.modulus], lr=bench.LR, lazy_rows=True, draw_ahead=ns, defer_step=True)
    step = parallel.DimShardedStep(m, bench.ALPHA)
    train = torch.as_tensor(train_np, device="cuda")
    w = subsampling_weights(train_np).cuda()
    gb = world * bench.B
    kinds = list(_hip.PROF_KINDS)

    def run(i):
        lo = (i * gb) % (len(train_np) - gb)
        s, ww = train[lo: lo + gb], w[lo: lo + gb]
        mode = "head-batch" if i % 2 == 0 else "tail-batch"
        step.sampled(s, ww, ns, mode)
        opt.step(); opt.zero_grad()

    for i in range(10):
        run(i)
    torch.cuda.synchronize()
    for k in kinds:
        _hip.profile_enable(k, True)
    t0 = time.perf_counter()
    n = 40
    for i in range(n):
        run(10 + i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"world={world}: per-GPU step without the all-reduce {dt * 1e3:.3f} ms  "
          f"(single-GPU-equivalent rate {gb * (bench.K + 1) / dt / 1e6:.0f} M triples/s per {world} GPUs)")
    print({k: round(_hip.profile_read(k)[1] / n * 1e3, 1) for k in kinds})


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 8)
