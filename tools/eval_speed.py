"""python tools/eval_speed.py [Model ...]: seconds of one filtered evaluation (evaluation.Evaluation.eval, both corruption sides) of
FB15k-237's test split -- 40,932 queries x 14,541 entities, hidden 1000, random tables -- per model.  A/B switches of mkb_rank:
MKB_RANK_TILE=0 (RotatE / TransE without the register tile), MKB_RANK_GEMM=0 (ComplEx / DistMult without the matrix cores),
MKB_RANK_WIDE=1 (16 waves x 1 unit per lane)."""
import sys, time, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from mkb_amd import datasets, evaluation, models
ds = datasets.Fb15k237(batch_size=1024, shuffle=False, seed=42, num_workers=0)
ev = evaluation.Evaluation(true_triples=ds.true_triples, entities=ds.entities, relations=ds.relations, batch_size=1024, device="cuda", num_workers=0)
for name in (sys.argv[1:] or ["TransE", "RotatE", "ComplEx", "DistMult", "pRotatE"]):
    torch.manual_seed(1)
    m = getattr(models, name)(hidden_dim=1000, entities=ds.entities, relations=ds.relations, gamma=9.0).cuda().eval()
    ev.eval(model=m, dataset=ds.test[:2048])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = ev.eval(model=m, dataset=ds.test)
    torch.cuda.synchronize()
    print(f"{name:9s} {time.perf_counter() - t0:.3f} s  {out}", flush=True)
