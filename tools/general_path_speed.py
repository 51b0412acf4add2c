"""Throughput of the reference's low-level loop (README.md:448-474) at the headline shape with negatives that carry NO
shared-pool description (a plain LongTensor), autograd and the dense Adam kernel: once with the pool scan that
mkb_amd.models applies to such negatives (PoolInfo.discover -> pooled kernels), once on the general kernels.
    python tools/general_path_speed.py"""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from mkb_amd import datasets, losses, models, optim, sampling  # noqa: E402

ds = datasets.Fb15k237(batch_size=1024, shuffle=False, seed=42, num_workers=0)
torch.manual_seed(42)
m = models.RotatE(hidden_dim=1000, entities=ds.entities, relations=ds.relations, gamma=9.0).cuda()
ns = sampling.NegativeSampling(size=256, train_triples=ds.train, entities=ds.entities, relations=ds.relations, seed=42)
opt = optim.Adam([p for p in m.parameters() if p.requires_grad], lr=5e-5)
loss_fn = losses.Adversarial(alpha=1.0)
train = torch.as_tensor(ds.train, dtype=torch.int64).cuda()
w = torch.ones(1024, device="cuda")


def step(i):
    s = train[(i * 1024) % 200000: (i * 1024) % 200000 + 1024]
    mode = "head-batch" if i % 2 == 0 else "tail-batch"
    neg = ns.generate(s, mode).clone()  # a plain LongTensor: the general kernels score it slot by slot
    err = loss_fn(m(s), m(s, neg, mode), w)
    err.backward()
    opt.step()
    opt.zero_grad()


import mkb_amd.models.base as model_base  # noqa: E402

for auto in (True, False):
    model_base.AUTO_POOL = auto
    for i in range(5):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 30
    for i in range(n):
        step(5 + i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"plain negatives, {'pool scan + pooled kernels' if auto else 'general kernels'}: {dt * 1e3:.3f} ms/step = "
          f"{1024 * 257 / dt / 1e6:.0f} M scored triples/s")
