"""Throughput of the drop-in `compose.Pipeline.learn` loop itself (device batch producer, fused step, row-lazy Adam)
at the headline configuration: python tools/pipeline_speed.py"""
import sys, time, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from mkb_amd import compose, datasets, evaluation, losses, models, optim, sampling
ds = datasets.Fb15k237(batch_size=1024, shuffle=True, seed=42, num_workers=0)
db = datasets.DeviceBatches(ds, "cuda", seed=42)
torch.manual_seed(42)
m = models.RotatE(hidden_dim=1000, entities=ds.entities, relations=ds.relations, gamma=9.0).cuda()
ns = sampling.NegativeSampling(size=256, train_triples=ds.train, entities=ds.entities, relations=ds.relations, seed=42)
opt = optim.Adam([p for p in m.parameters() if p.requires_grad], lr=5e-5, lazy_rows=True)
ev = evaluation.Evaluation(true_triples=ds.true_triples, entities=ds.entities, relations=ds.relations, batch_size=1024,
                           device="cuda", num_workers=0)
ds.valid, ds.test = [], []  # time the training loop alone (the final evaluation is measured by bench.py)
EPOCHS = 8
compose.Pipeline(epochs=1, eval_every=100, device="cuda").learn(  # warm-up epoch: module load, workspace, allocator
    model=m, dataset=db, sampling=ns, optimizer=opt, loss=losses.Adversarial(alpha=1.0), evaluation=ev)
torch.cuda.synchronize(); t0 = time.perf_counter()
pipe = compose.Pipeline(epochs=EPOCHS, eval_every=100, device="cuda")
pipe.learn(model=m, dataset=db, sampling=ns, optimizer=opt, loss=losses.Adversarial(alpha=1.0), evaluation=ev)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
steps = EPOCHS * len(db)
print(f"PIPE steps {steps} in {dt:.2f}s = {dt/steps*1e3:.3f} ms/step = {steps*1024*257/dt/1e6:.0f} M triples/s")
