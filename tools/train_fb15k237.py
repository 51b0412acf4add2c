"""End-to-end training run of the headline configuration on one MI355X, reporting filtered MRR / HITS:
FB15k-237 + RotatE hidden_dim=1000, K=256, batch 1024, Adversarial alpha=1, gamma=9, Adam.

    python tools/train_fb15k237.py [--epochs 200] [--lr 5e-5] [--eval-every 50] [--model RotatE]

Uses the public mkb_amd API only (models / sampling / losses / compose.Pipeline-equivalent loop / evaluation) plus
the device batch producer.  Prints one JSON line per evaluation."""
import argparse
import json
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from mkb_amd import datasets, evaluation, losses, models, optim, sampling  # noqa: E402
from mkb_amd.fused import FusedTrainStep  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=200)
    ap.add_argument("--lr", type=float, default=5e-5)
    ap.add_argument("--eval-every", type=int, default=50)
    ap.add_argument("--model", default="RotatE")
    ap.add_argument("--hidden", type=int, default=1000)
    ap.add_argument("--gamma", type=float, default=9.0)
    ap.add_argument("--alpha", type=float, default=1.0)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--dataset", default="Fb15k237")
    ap.add_argument("--no-defer", action="store_true", help="apply the optimizer step at once (one more launch per step)")
    args = ap.parse_args()

    ds = getattr(datasets, args.dataset)(batch_size=args.batch, shuffle=True, seed=42, num_workers=0)
    batches = datasets.DeviceBatches(ds, "cuda", seed=42)
    torch.manual_seed(42)
    model = getattr(models, args.model)(hidden_dim=args.hidden, entities=ds.entities, relations=ds.relations,
                                        gamma=args.gamma).cuda()
    sampler = sampling.NegativeSampling(size=args.size, train_triples=ds.train, entities=ds.entities,
                                        relations=ds.relations, seed=42)
    opt = optim.Adam([p for p in model.parameters() if p.requires_grad], lr=args.lr, lazy_rows=True, draw_ahead=sampler,
                     defer_step=not args.no_defer)  # gradients are cleared through opt.zero_grad() only: the step may wait
    step = FusedTrainStep(model, args.alpha)
    ev = evaluation.Evaluation(true_triples=ds.true_triples, entities=ds.entities, relations=ds.relations,
                               batch_size=1024, device="cuda", num_workers=0)
    n_steps, t_train = 0, 0.0
    for epoch in range(args.epochs):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for data in batches:
            loss = step.sampled(data["sample"], data["weight"], sampler, data["mode"])
            opt.step()
            opt.zero_grad()
            n_steps += 1
        torch.cuda.synchronize()
        t_train += time.perf_counter() - t0
        sampler.check()
        if (epoch + 1) % args.eval_every == 0 or epoch + 1 == args.epochs:
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            valid = ev.eval(model=model, dataset=ds.valid)
            test = ev.eval(model=model, dataset=ds.test)
            torch.cuda.synchronize()
            print(json.dumps({"epoch": epoch + 1, "steps": n_steps, "loss": float(loss.item()),
                              "train_seconds": round(t_train, 2),
                              "triples_per_s": round(n_steps * args.batch * (args.size + 1) / t_train),
                              "eval_seconds": round(time.perf_counter() - t1, 2), "valid": valid, "test": test}), flush=True)


if __name__ == "__main__":
    main()
