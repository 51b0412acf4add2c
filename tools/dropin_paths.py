"""The literal drop-in paths at the headline configuration (FB15k-237, RotatE hidden 1000, K 256, batch 1024), one line each:

    python tools/dropin_paths.py [--epochs-host 1] > profiles/rNN_dropin_paths.txt

  readme-loop/torch-adam     the unchanged README loop (reference README.md:448-474): model(sample), sampling.generate,
                             model(sample, neg, mode), loss, backward into dense .grad, the USER's torch.optim.Adam, zero_grad;
                             batches index-selected on the device (the loop itself, without a batch producer)
  readme-loop/host-dataset   the same loop iterating the drop-in `datasets.Fb15k237` (two host DataLoaders with a worker
                             process, reference mkb/datasets/dataset.py:297-303), H2D copy per batch
  pipeline/host+torch-adam   compose.Pipeline.learn on the host DataLoader dataset with torch.optim.Adam
  pipeline/host+mkb-adam     ... with mkb_amd.optim.Adam(lazy_rows=True)
  pipeline/device+mkb-adam   the opted-in form (DeviceBatches + row-lazy Adam): what tools/pipeline_speed.py times
"""
import argparse
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from mkb_amd import compose, datasets, evaluation, losses, models, optim, sampling  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--workers", type=int, default=1, help="num_workers of the host DataLoaders (the reference's default is 1)")
args = ap.parse_args()
B, K, HID = 1024, 256, 1000
TRIPLES = B * (K + 1)


def fresh(ds, which):
    torch.manual_seed(42)
    m = models.RotatE(hidden_dim=HID, entities=ds.entities, relations=ds.relations, gamma=9.0).cuda()
    ns = sampling.NegativeSampling(size=K, train_triples=ds.train, entities=ds.entities, relations=ds.relations, seed=42)
    ps = [p for p in m.parameters() if p.requires_grad]
    opt = torch.optim.Adam(ps, lr=5e-5) if which == "torch" else optim.Adam(ps, lr=5e-5, lazy_rows=True)
    return m, ns, opt


def line(name, dt, extra=""):
    print(f"{name:28s} {dt * 1e3:8.3f} ms/step = {TRIPLES / dt / 1e6:7.0f} M scored triples/s {extra}", flush=True)


def readme_loop(ds, batches, n):
    m, ns, opt = fresh(ds, "torch")
    loss_fn = losses.Adversarial(alpha=1.0)
    it = iter(batches)

    def step():
        data = next(it)
        sample, weight, mode = data["sample"].to("cuda"), data["weight"].to("cuda"), data["mode"]
        negative_sample = ns.generate(sample=sample, mode=mode).to("cuda")
        positive_score = m(sample)
        negative_score = m(sample=sample, negative_sample=negative_sample, mode=mode)
        error = loss_fn(positive_score, negative_score, weight)
        error.backward()
        opt.step()
        opt.zero_grad()

    for _ in range(10):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def device_batches(ds):
    train = torch.as_tensor(ds.train, dtype=torch.int64).cuda()
    w = torch.ones(B, device="cuda")
    i = 0
    while True:
        lo = (i * B) % 200000
        yield {"sample": train[lo: lo + B], "weight": w, "mode": "head-batch" if i % 2 == 0 else "tail-batch"}
        i += 1


def endless(ds):
    while True:
        yield from ds


def pipeline(ds, batches, which, epochs):
    m, ns, opt = fresh(ds, which)
    ev = evaluation.Evaluation(true_triples=ds.true_triples, entities=ds.entities, relations=ds.relations, batch_size=1024,
                               device="cuda", num_workers=0)
    ds.valid, ds.test = [], []  # the training loop alone
    loss_fn = losses.Adversarial(alpha=1.0)
    compose.Pipeline(epochs=1, eval_every=100, device="cuda").learn(model=m, dataset=batches, sampling=ns, optimizer=opt,
                                                                    loss=loss_fn, evaluation=ev)  # warm-up epoch
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    compose.Pipeline(epochs=epochs, eval_every=100, device="cuda").learn(model=m, dataset=batches, sampling=ns, optimizer=opt,
                                                                         loss=loss_fn, evaluation=ev)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (epochs * len(batches))


print(f"# headline configuration: FB15k-237, RotatE hidden {HID}, K {K}, batch {B}; host DataLoaders with num_workers={args.workers}")
ds = datasets.Fb15k237(batch_size=B, shuffle=True, seed=42, num_workers=args.workers)
line("readme-loop/torch-adam", readme_loop(ds, device_batches(ds), args.steps), "(batches sliced on the device)")
line("readme-loop/host-dataset", readme_loop(ds, endless(ds), args.steps), "(drop-in datasets.Fb15k237 iterated as in the README)")
line("pipeline/host+torch-adam", pipeline(ds, ds, "torch", 1))
ds = datasets.Fb15k237(batch_size=B, shuffle=True, seed=42, num_workers=args.workers)
line("pipeline/host+mkb-adam", pipeline(ds, ds, "mkb", 1))
ds = datasets.Fb15k237(batch_size=B, shuffle=True, seed=42, num_workers=0)
line("pipeline/device+mkb-adam", pipeline(ds, datasets.DeviceBatches(ds, "cuda", seed=42), "mkb", 8))
