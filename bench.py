"""bench.py -- scored triples/sec of the mkb training hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], the headline): FB15k-237 (14,541 entities, 237 relations, 272,115 training
triples) + RotatE hidden_dim=1000 (entity rows 2000 fp32), K=256 negatives, batch 1024 rows PER GPU, Adversarial
alpha=1, gamma=9, Adam lr 5e-5.  One step = what compose/pipeline.py:206-240 does for one batch:
    filtered negative draw (on device, bit-exact) -> positive forward -> negative forward -> Adversarial ->
    backward into dense gradients -> dense Adam step (+ zero_grad)
and scores B*(K+1) = 263,168 triples per GPU.  Inputs (training triples, subsampling weights, tables, optimizer
state) are resident in HBM before the timed region; batches are index-selected on the device.

Multi-GPU (N > 1), weak scaling with a global batch of N*1024 rows scored against ONE candidate pool (replicated
MT19937 state).  Default --parallelism table-rows, the partitioning BASELINE.json's north_star names: the entity table, its
gradient and its Adam state are sharded by ROW (owner = id % N, mkb_amd/table_rows.py); pool rows by one all-reduce,
positive rows by all-to-all, gradients back the same way, row-lazy Adam per shard.  The same run then measures the two
alternatives (fewer steps) and BASELINE configs[4] (YAGO3-10 RotatE-500, the config north_star shards) and reports them
under "other_partitionings" / "config5" of the same JSON line:
--parallelism dims: the embedding DIMENSION is sharded (rank g holds 1/N of every table column-wise, 1/N of the optimizer
state); every rank scores all N*1024 rows on its dims, ONE RCCL all-reduce of the partial scores [N*1024, 2K+1] per step,
then loss / backward / Adam are local -- no gradient exchange.
--parallelism rows: batch-row data parallel with replicated tables and a sparse all-reduce of the touched gradient
rows (mkb_amd.parallel.SparseGradExchange).

Prints ONE JSON line (rank 0).  Extra objects: "roofline" for the dominant kernel (HIP-event timed inside the
timed region) and "cpu_baseline" (the oracle restatement of the reference's PyTorch-CPU path on a bounded sample).
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HIDDEN, K, B, GAMMA, ALPHA, LR = 1000, 256, 1024, 9.0, 1.0, 5e-5
MODEL, DATASET = "RotatE", "fb15k237"
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
FP32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 vector = fp32 matrix peak (256 CU x 4 SIMD x 64 flop/clk x 2.4 GHz)
TRANS_PEAK_TOPS = 19.66   # quarter-rate transcendentals: 1024 SIMDs x 8 results/clk x 2.4 GHz
BF16_PEAK_TFLOPS = 2516.6  # MI355X_MICROARCH.md: dense bf16 matrix peak (256 CU x 4 SIMD x 1024 flop/clk x 2.4 GHz)

# BASELINE.json configs; the default (and the only one the driver runs) is the headline, configs[2]
CONFIGS = {
    "headline": dict(MODEL="RotatE", DATASET="fb15k237", HIDDEN=1000, K=256, B=1024, GAMMA=9.0, ALPHA=1.0, LR=5e-5),
    "umls-transe": dict(MODEL="TransE", DATASET="umls", HIDDEN=64, K=16, B=256, GAMMA=6.0, ALPHA=1.0, LR=1e-3),
    "wn18rr-rotate": dict(MODEL="RotatE", DATASET="wn18rr", HIDDEN=500, K=128, B=1024, GAMMA=6.0, ALPHA=0.5, LR=5e-5),
    "fb15k237-complex": dict(MODEL="ComplEx", DATASET="fb15k237", HIDDEN=1000, K=256, B=1024, GAMMA=9.0, ALPHA=1.0, LR=5e-5),
    "fb15k237-transe": dict(MODEL="TransE", DATASET="fb15k237", HIDDEN=1000, K=256, B=1024, GAMMA=9.0, ALPHA=1.0, LR=5e-5),
    "yago310-rotate": dict(MODEL="RotatE", DATASET="yago310", HIDDEN=500, K=256, B=1024, GAMMA=6.0, ALPHA=0.5, LR=5e-5),
    "fb15k237-distmult": dict(MODEL="DistMult", DATASET="fb15k237", HIDDEN=1000, K=256, B=1024, GAMMA=9.0, ALPHA=1.0, LR=5e-5),
}


def load_fb15k237():
    if DATASET == "yago310":  # train.csv is absent from the reference mount: synthetic triples (datasets.Yago310)
        from mkb_amd import datasets

        ds = datasets.Yago310(batch_size=B, shuffle=False, seed=42, num_workers=0)
        return np.asarray(ds.train, dtype=np.int64), ds.n_entity, ds.n_relation
    z = np.load(os.path.join(ROOT, "mkb_amd", "datasets", "data", f"{DATASET}.npz"))
    tr = z["train"].astype(np.int64)
    n_ent = int(max(z["train"][:, [0, 2]].max(), z["valid"][:, [0, 2]].max(), z["test"][:, [0, 2]].max())) + 1
    n_rel = int(max(z["train"][:, 1].max(), z["valid"][:, 1].max(), z["test"][:, 1].max())) + 1
    if DATASET == "fb15k237":
        n_ent, n_rel = 14541, 237
    return tr, n_ent, n_rel


def build(device, rank, world, seed=42, parallelism="table-rows", force=False):
    from mkb_amd import models, optim, parallel, sampling
    from mkb_amd.datasets.base import subsampling_weights
    from mkb_amd.fused import FusedTrainStep

    train_np, n_ent, n_rel = load_fb15k237()
    ents, rels = {i: i for i in range(n_ent)}, {i: i for i in range(n_rel)}
    torch.manual_seed(seed)
    model = getattr(models, MODEL)(hidden_dim=HIDDEN, entities=ents, relations=rels, gamma=GAMMA)
    dims = world > 1 and parallelism == "dims"
    trows = (world > 1 or force) and parallelism == "table-rows"  # force: the sharded code path on one GPU (no collective runs)
    table = rel_rep = None
    if trows:  # entity table, its gradient and its Adam state sharded by ROW (owner = id % world); relation table replicated
        from mkb_amd.table_rows import TableRowShardedStep, shard_table_rows

        table, rel_rep = shard_table_rows(model, device=device)
    else:
        model = parallel.shard_dims(model, rank, world, device) if dims else model.to(device)
    sampler = sampling.NegativeSampling(size=K, train_triples=train_np, entities=ents, relations=rels, seed=seed)
    # dense-Adam semantics, evaluated row-lazily (bit-identical to the dense kernel, tests/test_gpu_general.py);
    # MKB_BENCH_DENSE_ADAM=1 selects the plain dense streaming kernel instead
    lazy = os.environ.get("MKB_BENCH_DENSE_ADAM", "0") != "1"
    if trows:
        # the shard steps like the single-GPU table: dense-Adam semantics evaluated row-lazily, real step deferred into the
        # next step's catch-up launch (which also draws the sampler's next pool); no optimizer communication
        opt = optim.Adam([table.data, rel_rep], lr=LR, lazy_rows=lazy,
                         draw_ahead=sampler if os.environ.get("MKB_BENCH_NO_DRAW_AHEAD", "0") != "1" else None,
                         defer_step=lazy and os.environ.get("MKB_BENCH_NO_DEFER", "0") != "1")
        step = TableRowShardedStep(table, rel_rep, ALPHA, model_cls=getattr(models, MODEL), hidden_dim=HIDDEN, gamma=GAMMA)
    else:
        opt = optim.Adam([p for p in model.parameters() if p.requires_grad and (MODEL != "RotatE" or p is not model.modulus)],
                         lr=LR, lazy_rows=lazy, draw_ahead=sampler if os.environ.get("MKB_BENCH_NO_DRAW_AHEAD", "0") != "1" else None,
                         # what compose.Pipeline configures for its fused loop (MKB_BENCH_NO_DEFER=1: the separate step launch)
                         defer_step=os.environ.get("MKB_BENCH_NO_DEFER", "0") != "1")
        step = parallel.DimShardedStep(model, ALPHA) if dims else FusedTrainStep(model, ALPHA)
    train = torch.as_tensor(train_np, device=device)
    weights = subsampling_weights(train_np).to(device)
    g = torch.Generator(device="cpu").manual_seed(seed)
    perm = torch.randperm(len(train_np), generator=g).to(device)
    train, weights = train[perm].contiguous(), weights[perm].contiguous()  # shuffled once (per epoch in a real loop):
    return dict(ride=os.environ.get("MKB_BENCH_NO_RIDE", "0") != "1", model=model, sampler=sampler, opt=opt, step=step, train=train, weights=weights, perm=perm, rank=rank,  # batches are views
                world=world, n_train=len(train_np), exchange=None, dims=dims, trows=trows, table=table)


def run_step(ctx, i):
    """One training step for this rank's rows of global batch i."""
    n, world, rank = ctx["n_train"], ctx["world"], ctx["rank"]
    mode = "head-batch" if i % 2 == 0 else "tail-batch"
    Bl = ctx.get("rows_per_rank", B)  # rows per rank (weak scaling: B; strong scaling: B / world)
    if ctx.get("trows"):  # row-sharded entity table: this rank's rows of the global batch, ONE shared pool (replicated RNG)
        lo = ((i * world + rank) * Bl) % (n - Bl)
        sample, weight = ctx["train"][lo: lo + Bl], ctx["weights"][lo: lo + Bl]
        # the next two batches: their routing (route kernel + id exchange, side stream) is prepared while this step runs
        ahead = [ctx["train"][nlo: nlo + Bl] for nlo in ((((i + d) * world + rank) * Bl) % (n - Bl) for d in (1, 2))]
        loss = ctx["step"].sampled(sample, weight, ctx["sampler"], mode, next_sample=ahead[: ctx.get("lookahead", 2)])
        ctx["opt"].step()
        ctx["opt"].zero_grad()
        return loss
    if ctx["dims"]:  # dimension sharding: every rank scores ALL world*B rows of the global batch on its 1/world of the dims
        gb = world * Bl
        lo = (i * gb) % (n - gb)
        sample, weight = ctx["train"][lo: lo + gb], ctx["weights"][lo: lo + gb]
        if ctx.get("ride", True):
            loss = ctx["step"].sampled(sample, weight, ctx["sampler"], mode)  # one all-reduce of the partial scores inside
        else:
            loss = ctx["step"](sample, weight, ctx["sampler"].generate(sample, mode), mode)
        ctx["opt"].step()
        ctx["opt"].zero_grad()
        return loss
    lo = ((i * world + rank) * Bl) % (n - Bl)
    if os.environ.get("MKB_BENCH_FIXED_BATCH"):  # diagnostic: the SAME rows and the SAME negatives every step (two modes: two batches)
        lo = (i % 2) * Bl
        sample, weight = ctx["train"][lo: lo + Bl], ctx["weights"][lo: lo + Bl]
        fixed = ctx.setdefault("_fixed_neg", {})
        if mode not in fixed:
            fixed[mode] = ctx["sampler"].generate(sample, mode)
        loss = ctx["step"](sample, weight, fixed[mode], mode)
        ctx["opt"].step()
        ctx["opt"].zero_grad()
        return loss
    sample = ctx["train"][lo: lo + Bl]
    weight = ctx["weights"][lo: lo + Bl]
    ex = ctx["exchange"]
    wsum = ex.weight_sum(weight) if ex is not None else None   # global-batch normaliser (all-reduced scalar)
    if ctx.get("ride", True):  # sampler folded into the optimizer's catch-up launch (identical negatives)
        loss = ctx["step"].sampled(sample, weight, ctx["sampler"], mode, weight_sum=wsum)
        neg = ctx["step"].negative_sample
    else:
        neg = ctx["sampler"].generate(sample, mode)
        loss = ctx["step"](sample, weight, neg, mode, weight_sum=wsum)
    if ex is not None:
        ex(sample, neg)                                           # sparse all-reduce of the touched gradient rows
    ctx["opt"].step()
    ctx["opt"].zero_grad()
    return loss


def train_and_rank(ctx, epochs, first_step):
    """The MRR half of BASELINE.json's metric: keep training the bench's model for `epochs` passes over the training
    set (outside the timed region), then filtered link-prediction ranking of the real FB15k-237 test set against all
    entities on the device (mkb_rank)."""
    from mkb_amd import evaluation

    z = np.load(os.path.join(ROOT, "mkb_amd", "datasets", "data", "fb15k237.npz"))
    true = np.concatenate([z["train"], z["valid"], z["test"]]).astype(np.int64)
    steps = epochs * 2 * (-(-ctx["n_train"] // B))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    decay_at = steps * 2 // 3  # the RotatE paper's recipe in small: the rate drops tenfold for the last third of the run
    for i in range(steps):     # (tools/mrr_attribution.py: 200 epochs with the drop at 100 reach test MRR 0.335)
        if i == decay_at:
            ctx["opt"].lr = LR / 10.0
        run_step(ctx, first_step + i)
    torch.cuda.synchronize()
    t_train = time.perf_counter() - t0
    m = ctx["model"]
    ev = evaluation.Evaluation(true_triples=true, entities=m.entities, relations=m.relations, batch_size=1024,
                               device="cuda", num_workers=0)
    t1 = time.perf_counter()
    res = ev.eval(model=m, dataset=z["test"].astype(np.int64))
    torch.cuda.synchronize()
    return {"test": res, "epochs": epochs, "steps": steps, "train_seconds": round(t_train, 2),
            "eval_seconds": round(time.perf_counter() - t1, 2), "lr": f"{LR} for 2/3 of the steps, then {LR / 10.0}",
            "note": "filtered ranking of 20,466 test triples x 2 sides against all 14,541 entities (mkb_rank); "
                    "literature RotatE on FB15k-237: MRR ~0.338; this code with the paper's schedule (200 epochs, rate / 10 after 100): 0.335 "
                    "(profiles/r03_mrr_attribution.jsonl)"}


def cpu_baseline(rows=64, rows_sqrt=128, seed=42, warmup=2, timed=3):
    """The oracle (torch-CPU restatement of the reference path) timed on the host cores, by BASELINE.md section 3's
    protocol on a bounded sample: `warmup` untimed + `timed` timed steps over `rows` rows of a headline batch (full
    tables, same K / dims), each phase timed on its own -- sample (the plain-C sampler restatement), fwd (positive +
    negative forward), loss, bwd (autograd into dense gradients) -- and the dense torch-Adam step over the FULL tables
    timed separately (it does not scale with the rows).  A 1024-row step is then
        t_step = (t_sample + t_fwd + t_loss + t_bwd) * 1024 / rows + t_opt,     value = 1024 * (K + 1) / t_step.
    Two forms of the RotatE forward: the reference-faithful stack -> norm(dim=0) (a torch-CPU pathology) and
    sqrt(re^2 + im^2), so that the CPU figure is not inflated by that."""
    import ctypes

    from mkb_amd.datasets.base import subsampling_weights
    from mkb_amd.sampling.negative_sampling import _filter_csr
    from oracle import scoring

    train_np, n_ent, n_rel = load_fb15k237()
    torch.manual_seed(seed)
    tb = scoring.init_tables("RotatE", n_ent, n_rel, HIDDEN, GAMMA)
    w_all = subsampling_weights(train_np)
    (hk, ho, hv, _), _tail = _filter_csr(train_np, n_ent, n_rel)
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    lib.orc_generate.restype = ctypes.c_int
    st = ctypes.create_string_buffer(4 * 624 + 4)
    lib.orc_mt_seed(st, ctypes.c_uint32(seed))
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    cores = torch.get_num_threads()
    pick = np.random.RandomState(7)

    def one_step(n_rows, fast_norm):
        """-> seconds per phase of one step over n_rows rows (head-batch)."""
        idx = pick.randint(len(train_np), size=n_rows)
        smp = np.ascontiguousarray(train_np[idx])
        t = {}
        t0 = time.perf_counter()
        neg = np.zeros((n_rows, K), dtype=np.int64)
        pool = np.zeros(2 * K, dtype=np.int64)
        rc = lib.orc_generate(st, ctypes.c_int64(n_ent), ctypes.c_int64(K), p(smp), ctypes.c_int64(n_rows), ctypes.c_int(1),
                              p(hk), ctypes.c_int64(len(hk)), p(ho), p(hv), ctypes.c_int64(n_ent), p(neg), p(pool))
        assert rc == 0
        t["sample"] = time.perf_counter() - t0
        ent = tb.ent.detach().clone().requires_grad_(True)
        rel = tb.rel.detach().clone().requires_grad_(True)
        s_t, n_t, w_t = torch.as_tensor(smp), torch.as_tensor(neg), w_all[idx]
        t0 = time.perf_counter()
        pos = scoring.score(tb, s_t, ent=ent, rel=rel, fast_norm=fast_norm)
        ngs = scoring.score(tb, s_t, n_t, "head-batch", ent=ent, rel=rel, fast_norm=fast_norm)
        t["fwd"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        loss = scoring.adversarial(pos, ngs, w_t, ALPHA)
        t["loss"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        loss.backward()
        t["bwd"] = time.perf_counter() - t0
        return t, ent.grad, rel.grad

    def protocol(n_rows, fast_norm):
        acc = {"sample": 0.0, "fwd": 0.0, "loss": 0.0, "bwd": 0.0}
        g = None
        for i in range(warmup + timed):
            t, ge, gr = one_step(n_rows, fast_norm)
            if i >= warmup:
                for k in acc:
                    acc[k] += t[k] / timed
            g = (ge, gr)
        return acc, g

    split, grads = protocol(rows, False)
    split_sqrt, _ = protocol(rows_sqrt, True)
    # dense Adam over the full tables (zero_grad included), once warm + `timed` timed
    state = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in (("ent", tb.ent), ("rel", tb.rel))}
    t_opt = 0.0
    for i in range(warmup + timed):
        t0 = time.perf_counter()
        scoring.adam_update(tb.ent, grads[0], *state["ent"], i + 1, lr=LR)
        scoring.adam_update(tb.rel, grads[1], *state["rel"], i + 1, lr=LR)
        grads[0].zero_(), grads[1].zero_()
        if i >= warmup:
            t_opt += (time.perf_counter() - t0) / timed
    full = lambda sp, n: sum(sp.values()) * B / n + t_opt
    ms = lambda sp: {k: round(v * 1e3, 2) for k, v in sp.items()}
    return {"value": B * (K + 1) / full(split, rows), "unit": "scored triples/s", "cores": cores, "kind": "port",
            "value_sqrt_form": B * (K + 1) / full(split_sqrt, rows_sqrt),
            "ms_per_phase": dict(ms(split), opt_full_tables=round(t_opt * 1e3, 2), rows=rows),
            "ms_per_phase_sqrt_form": dict(ms(split_sqrt), opt_full_tables=round(t_opt * 1e3, 2), rows=rows_sqrt),
            "sample": f"{warmup} warm-up + {timed} timed steps over {rows} of the 1024 rows of a headline batch (FB15k-237 RotatE "
                      f"hidden={HIDDEN} K={K}, full tables, reference-faithful stack->norm forward, C sampler), phases timed "
                      f"separately; dense torch Adam over the full tables timed on its own; value = {B}*(K+1) / ((sample+fwd+loss+bwd)"
                      f"*{B}/{rows} + opt); value_sqrt_form = the same protocol with sqrt(re^2+im^2) on {rows_sqrt} rows"}


def step_variants(ctx, steps=60):
    """the step with and without the optimizer / the sampler, outside the main timed region (N = 1).
    'no_optimizer': sampler + forward + loss + backward into dense gradients that keep accumulating (no Adam, no zero_grad);
    'no_sampler_no_optimizer': the same with ONE pre-drawn negative batch reused.  Scored triples/s each."""
    from mkb_amd.fused import FusedTrainStep

    m, sampler = ctx["model"], ctx["sampler"]
    ctx["opt"].flush()
    step = FusedTrainStep(m, ALPHA)
    from mkb_amd import _links
    owner = _links.owner(m.entity_embedding)
    _links.detach(m.entity_embedding)  # plain dense-gradient path for these two loops
    out = {}
    try:
        sample, weight = ctx["train"][:B], ctx["weights"][:B]
        neg = sampler.generate(sample, "head-batch")
        for name, draw in (("no_optimizer", True), ("no_sampler_no_optimizer", False)):
            for i in range(5):
                step(sample, weight, sampler.generate(sample, "head-batch") if draw else neg, "head-batch")
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                lo = (i * B) % (ctx["n_train"] - B)
                sm, wt = (ctx["train"][lo: lo + B], ctx["weights"][lo: lo + B]) if draw else (sample, weight)
                mode = "head-batch" if (i % 2 == 0 or not draw) else "tail-batch"
                step(sm, wt, sampler.generate(sm, mode) if draw else neg, mode)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            out[name] = {"value": B * (K + 1) / dt, "ms_per_step": dt * 1e3}
    finally:
        if owner is not None:
            _links.attach(m.entity_embedding, owner)
        for q in (m.entity_embedding, m.relation_embedding):
            if q.grad is not None:
                q.grad.zero_()
    return out


# rocprofv3 kernel-name fragments of the profiled classes (the in-run HBM traffic measurement matches on them)
KERNEL_NAMES = {"pool_fwd": ("pool_fwd", "pair_fwd", "gemm"), "pool_bwd_q": ("pool_bwd1", "pair_bwd", "pool_bwd_kernel", "gemm"),
                "adam": ("adam_rows_catchup_kernel", "adam_kernel"), "pool_bwd_x": ("gemm",)}


def measure_traffic(prof_kind, config):
    """HBM bytes per launch of the profiled kernel class AND of every mkb:: kernel of a step, measured NOW: two short rocprofv3
    PMC passes (FETCH_SIZE, then WRITE_SIZE: the TCC has 4 slots, the two counters need 5) over this same script, as
    MI355X_MICROARCH.md's HBM section prescribes; FETCH_SIZE doubled (gfx950 tallies a 128-byte request as 64 bytes for wide
    coalesced reads: calibrated on the dense Adam kernel in round 1), KiB -> bytes.
    Returns (bytes of the class or None, note, rocprof name of the class's kernel or None, per-step table or None)."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not on PATH", None, None
    frags = KERNEL_NAMES.get(prof_kind)
    if not frags:
        return None, f"no kernel name known for class {prof_kind}", None, None
    out, notes, names, tables = {}, [], [], {}
    tmp = tempfile.mkdtemp(prefix="mkb_traffic_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", MKB_BENCH_INNER="1")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            cmd = [exe, "--kernel-trace", "--pmc", counter, "-d", d, "-o", "run", "--", sys.executable, os.path.abspath(__file__),
                   "--steps", "12", "--warmup", "4", "--config", config, "--no-cpu-baseline", "--mrr-epochs", "0", "--no-variants",
                   "--profile-kernel", "none", "--no-traffic"]
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=240)
            dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
            if r.returncode != 0 or not dbs:
                return None, f"rocprofv3 --pmc {counter} failed (rc {r.returncode}): {r.stderr[-200:]}", None, None
            cur = sqlite3.connect(dbs[0]).cursor()
            per = {}
            for name, val in cur.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)):
                a = per.setdefault(name, [0, 0.0])
                a[0] += 1
                a[1] += val
            tables[counter] = per
            best = None
            for frag in frags:  # first fragment that matches; the heaviest kernel of that name
                hit = [(tot / n, n, name) for name, (n, tot) in per.items() if frag in name]
                if hit:
                    best = max(hit)
                    break
            if best is None:
                return None, f"no kernel matching {frags} in the {counter} pass", None, None
            out[counter] = best[0] * 1024.0  # KiB per dispatch -> bytes
            names.append(best[2])
            notes.append(f"{counter} {best[0] * 1024 / 1e6:.1f} MB x {best[1]} launches of {best[2][:60]}")
    except Exception as e:  # noqa: BLE001 -- the profiler is optional equipment: never lose the bench line over it
        return None, f"traffic pass failed: {type(e).__name__}: {e}", None, None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    total = 2.0 * out["FETCH_SIZE"] + out["WRITE_SIZE"]
    # the whole step: every mkb:: kernel of the passes, bytes per launch x launches per step (a step = one row_fwd launch;
    # flush / set-up launches that run once per process are left out: fewer than one launch per two steps)
    step_table = None
    try:
        short = lambda n: n.split("(")[0].replace("void ", "")
        fe, wr = tables["FETCH_SIZE"], tables["WRITE_SIZE"]
        n_steps = sum(n for name, (n, _) in fe.items() if "row_fwd_kernel" in name) or sum(n for name, (n, _) in fe.items() if "query_build" in name)
        per_kernel, per_step = {}, 0.0
        for name in fe:
            if "mkb::" not in name or n_steps == 0 or fe[name][0] * 2 < n_steps:
                continue
            f_b = 2.0 * fe[name][1] * 1024.0 / n_steps
            w_b = wr.get(name, [0, 0.0])[1] * 1024.0 / n_steps
            k = short(name)
            per_kernel[k] = round(per_kernel.get(k, 0.0) + (f_b + w_b) / 1e6, 2)
            per_step += f_b + w_b
        step_table = {"measured_hbm_bytes_per_step": per_step, "per_kernel_MB_per_step": per_kernel, "steps_in_the_pass": n_steps,
                      "note": "sum over the mkb:: kernels of a step of (2 x FETCH_SIZE + WRITE_SIZE), head- and tail-batch "
                              "instantiations averaged by their launch counts; TCC counters: L2 <-> fabric traffic, of which the "
                              "256 MB Infinity Cache absorbs an unknown share before HBM"}
    except Exception:  # noqa: BLE001
        step_table = None
    return total, ("measured in this run: rocprofv3 --pmc, separate passes over 12 steps; 2 x FETCH_SIZE + WRITE_SIZE, mean per launch: "
                   + "; ".join(notes)), (names[0].split("(")[0].replace("void ", "") if names else None), step_table


def measure(args, device, rank, world, config, parallelism, steps, warmup, profile=True, force=False, windows=1):
    """Build `config` under `parallelism`, warm up, time exactly `steps` steps between barriers (max over ranks).
    -> dict with the context, the timing and the HIP-event timing of the dominant kernel class."""
    import torch.distributed as dist
    from mkb_amd import _hip

    globals().update(CONFIGS[config])
    if os.environ.get("MKB_BENCH_LR"):  # diagnostic (e.g. 0: the tables never move): not a benchmark setting
        globals()["LR"] = float(os.environ["MKB_BENCH_LR"])
    if os.environ.get("MKB_BENCH_PREWARM_OTHER"):  # diagnostic: that many steps on ANOTHER model instance (kept alive: other memory) first
        other = build(device, rank, world, parallelism=parallelism, force=force)
        for i in range(int(os.environ["MKB_BENCH_PREWARM_OTHER"])):
            run_step(other, i)
        other["opt"].flush()
        torch.cuda.synchronize()
        globals()["_PREWARM_KEEP"] = other
    ctx = build(device, rank, world, parallelism=parallelism, force=force)
    ctx["rows_per_rank"] = B if (args.scaling == "weak" or world == 1) else max(8, B // world)
    if world > 1 and not ctx["dims"] and not ctx["trows"]:
        from mkb_amd import parallel

        ctx["exchange"] = parallel.SparseGradExchange(ctx["model"], equal_batches=True)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    kinds = ["pool_fwd", "pool_bwd_q", "pool_bwd_x", "adam", "sampler", "loss", "general_fwd", "general_bwd"]
    if getattr(args, "preheat_ms", 0.0) > 0:
        a_ = torch.randn(4096, 4096, device=device)
        torch.cuda.synchronize()
        t_ = time.perf_counter()
        while (time.perf_counter() - t_) * 1e3 < args.preheat_ms:
            for _ in range(8):
                a_ @ a_
            torch.cuda.synchronize()
        del a_
    for i in range(warmup):
        run_step(ctx, i)
    barrier()
    prof_kind = args.profile_kernel if profile else "none"
    first = warmup  # index of the first timed batch: batches are consumed in order (the row-sharded step plans the NEXT batch's
    #                 routes while it runs: a gap in the sequence is an error there, not a harmless skip)
    if prof_kind == "auto":  # pick the dominant kernel class on a short probe
        first = warmup + 8
        for k in kinds:
            _hip.profile_enable(k, True)
        for i in range(8):
            run_step(ctx, warmup + i)
        torch.cuda.synchronize()
        tot = {}
        for k in kinds:
            n, ms = _hip.profile_read(k)
            tot[k] = ms
            _hip.profile_enable(k, False)
        prof_kind = max(("pool_fwd", "pool_bwd_q", "pool_bwd_x", "adam"), key=lambda k: tot[k])
        if args.breakdown and rank == 0:
            print("probe ms/step by kernel class:", {k: round(v / 8, 4) for k, v in tot.items()}, file=sys.stderr)
    if prof_kind != "none":
        # every 5th launch of the class is bracketed with HIP events inside the timed region (odd stride: head- and tail-batch
        # steps are both sampled); each bracket costs ~12 us of stream time, so bracketing every launch would tax every step
        _hip.profile_enable(prof_kind, 5)

    # Everything the timed windows launch has run once before the first of them: the flush below is the process's first
    # launch of the flush form of the advance kernel (bit-exact at any time), and a full garbage collection now + gc.freeze()
    # takes the long-lived host objects (filter dictionaries, triples, modules) out of the collector's later passes -- a
    # generation-2 pass over them lasts tens of milliseconds, ten times a 20-step window (see DESIGN.md section 0, r05's stall).
    ctx["opt"].flush()
    barrier()
    gc_log = []
    if not os.environ.get("MKB_BENCH_NO_GC_FREEZE"):
        gc.collect()
        gc.freeze()
    gc_t = [0.0]

    def gc_cb(phase, info):  # every collection that happens from here on is reported in the line, with its duration
        if phase == "start":
            gc_t[0] = time.perf_counter()
        else:
            gc_log.append({"generation": info["generation"], "ms": round((time.perf_counter() - gc_t[0]) * 1e3, 3), "at": time.perf_counter()})

    gc.callbacks.append(gc_cb)
    # R windows of exactly `steps` steps each, every one between barriers; ms_per_step / value = the MEDIAN window, all of them
    # are reported (windows_ms).  Per window: the host's enqueue time, the device's own time between two events on the step's
    # stream, and the largest gap between two consecutive step submissions -- enough to tell a host stall from a device stall.
    wins = []
    sclk = torch.zeros(windows, device=device) if os.environ.get("MKB_BENCH_SCLK") else None
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for w in range(windows):
        barrier()
        ev0.record()
        t0 = time.perf_counter()
        prev, gap, gap_at = t0, 0.0, -1
        for i in range(steps):
            loss = run_step(ctx, first + w * steps + i)
            now = time.perf_counter()
            if now - prev > gap:
                gap, gap_at = now - prev, i
            prev = now
        ctx["opt"].flush()  # pending zero-gradient Adam steps of rows not touched lately are part of the timed work (no-op if dense)
        t_host = time.perf_counter() - t0  # host enqueue time of the timed steps (the device may still be running)
        ev1.record()
        if sclk is not None:  # the shader clock right behind the window's last kernel (one wave, ~20 us: outside the host's enqueue time)
            _hip.check(_hip.lib().mkb_debug_sclk_mhz(_hip.ptr(sclk[w: w + 1]), _hip.stream_ptr()), "mkb_debug_sclk_mhz")
        barrier()
        dt = time.perf_counter() - t0
        t1 = t0 + dt
        if world > 1:
            t = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = t.item()
        wins.append({"ms": dt * 1e3, "host_enqueue_ms": t_host * 1e3, "device_ms": ev0.elapsed_time(ev1),
                     "largest_submit_gap_ms": gap * 1e3, "largest_submit_gap_at_step": gap_at,
                     "gc_ms": sum(e["ms"] for e in gc_log if t0 <= e["at"] <= t1)})
        if sclk is not None:
            wins[-1]["sclk_mhz_after"] = float(sclk[w].item())
    gc.callbacks.remove(gc_cb)
    order_ = sorted(range(windows), key=lambda j: wins[j]["ms"])
    med = wins[order_[(windows - 1) // 2]]  # (lower median for an even count: an actual window, not a mean of two)
    dt, t_host = med["ms"] / 1e3, med["host_enqueue_ms"] / 1e3
    if args.breakdown and rank == 0:
        print(f"host enqueue {t_host / steps * 1e3:.4f} ms/step of {dt / steps * 1e3:.4f} ms/step", file=sys.stderr)
    sampler_note = None
    try:
        ctx["sampler"].check()
    except RuntimeError as e:  # a row whose true set covers the whole pool (dense toy graphs): the reference would hang
        if config == "headline":
            raise
        sampler_note = f"sampler: {e}"
    assert torch.isfinite(loss).item() or sampler_note
    if world > 1:  # rows: replicas must hold identical tables; dims / table-rows: every rank must report the same global loss
        probe = (loss.detach().double().reshape(1) if (ctx["dims"] or ctx["trows"])
                 else ctx["model"].entity_embedding.detach()[::97].double().sum().reshape(1))
        lo_, hi_ = probe.clone(), probe.clone()
        dist.all_reduce(lo_, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi_, op=dist.ReduceOp.MAX)
        # the all-reduced scores / loss are bitwise equal on every rank with ring / tree all-reduce; allow for algorithms
        # that are not (rel. 1e-5) instead of aborting the run.  rows: replicas apply identical updates.
        assert abs(hi_.item() - lo_.item()) <= 1e-5 * max(1.0, abs(hi_.item())), "data-parallel replicas diverged"
    launches, kms = (0, 0.0)
    if prof_kind != "none":
        launches, kms = _hip.profile_read(prof_kind)
        _hip.profile_enable(prof_kind, False)
    return dict(ctx=ctx, dt=dt, steps=steps, warmup=warmup, loss=float(loss.item()), launches=launches, kms=kms, prof_kind=prof_kind,
                sampler_note=sampler_note, config=config, t_host=t_host, windows=wins,
                gc=[{k: e[k] for k in ("generation", "ms")} for e in gc_log],
                value=world * ctx["rows_per_rank"] * (K + 1) * steps / dt, ms_per_step=dt / steps * 1e3)


def parallelism_label(ctx, world):
    if world == 1 and not ctx["trows"]:
        return "single"
    if ctx["dims"]:
        return f"dims{world} (embedding dimension sharded, 1 score all-reduce/step)"
    if ctx["trows"]:
        return (f"table-rows{world} (entity table + gradient + row-lazy Adam state sharded by row; pool rows all-reduce, positive "
                "rows all-to-all, routes planned one batch ahead)")
    return f"dp{world} (rows, sparse grad all-reduce)"


def roofline_of(res, world, want_traffic):
    """The `roofline` object of the JSON line for the kernel class that was HIP-event timed inside the timed region."""
    ctx, prof_kind, launches = res["ctx"], res["prof_kind"], res["launches"]
    if not launches:
        return None
    m_ = ctx["model"]
    De, Dr, N, R = m_.entity_dim, m_.relation_dim, m_.n_entity, m_.n_relation  # per rank (dims: 1/world of the row)
    Bk = world * ctx["rows_per_rank"] if ctx["dims"] else ctx["rows_per_rank"]  # rows each rank's kernels see
    avg_s = res["kms"] / launches / 1e3
    traffic, traffic_note, kname, step_table = (None, "not measured (--no-traffic, N > 1, or a non-default run)", None, None)
    if want_traffic:
        traffic, traffic_note, kname, step_table = measure_traffic(prof_kind, res["config"])
    res["step_traffic"] = step_table
    # the rocprofv3 name of the class's kernel: taken from the PMC pass when there was one, else spelled from the configuration
    # (template arguments: model id, head-batch, units per lane, dense pass); the internal class name rides along as `class`
    mid = {"TransE": 0, "RotatE": 1, "ComplEx": 2, "DistMult": 3, "pRotatE": 4}.get(MODEL, "?")
    paired = (MODEL in ("ComplEx", "DistMult") and os.environ.get("MKB_GEMM_NO_PAIR") is None and os.environ.get("MKB_GEMM_NO128") is None
              and os.environ.get("MKB_GEMM_BF16X3", "1") != "0")  # the two backward products of the bilinear models in ONE launch (gemm_mfma.h)
    guess = {"pool_bwd_q": (f"mkb::pool_bwd1_kernel<{mid}, true|false, .., ..>" if MODEL not in ("ComplEx", "DistMult") else
                            ("mkb::gemm128_bf16x3_pair_kernel<true, false, 0, .., false, false, 0, ..> (dQ = G . X and dX = G^T . Q)" if paired
                             else "mkb::gemm128_bf16x3_mfma_kernel<true, false, 0, ..> (dQ = G . X)")),
             "pool_fwd": (f"mkb::pool_fwd_tile_kernel<{mid}, true|false, ..>" if MODEL in ("RotatE", "TransE") and HIDDEN >= 300 else f"mkb::pool_fwd_kernel<{mid}, ..>"),
             "pool_bwd_x": "mkb::gemm128_bf16x3_mfma_kernel<false, false, 0, ..> (dX = G^T . Q)", "adam": "mkb::adam_rows_catchup_kernel"}.get(prof_kind, prof_kind)
    kernel_name = kname or guess
    info = ctx["sampler"].generate(ctx["train"][:Bk], "head-batch")._mkb_pool
    if prof_kind == "adam":
        lazy_rows = getattr(ctx["opt"], "lazy_rows", False)
        if lazy_rows:
            # row-lazy advance launch: the DISTINCT rows the batch touches x (p, m, v, g read + p, m, v, g written) x 4 B,
            # plus the relation table's dense step that rides it
            ids = info.touched if info.touched is not None else torch.cat([info.pool, ctx["train"][:Bk, 0], ctx["train"][:Bk, 2]])
            rows = int(torch.unique(ids).numel())
            alg = 8 * 4 * (rows * De + R * Dr)
            note = (f"row-lazy Adam advance launch: {rows} distinct touched rows x {De} floats x (p, m, v, g in + p, m, v, 0 -> g out) "
                    f"+ the dense relation-table step riding it")
        else:
            alg = 8 * 4 * (N * De + R * Dr) / 2.0  # one launch per parameter tensor, averaged over the two
            note = "dense Adam (+zero_grad) kernel: 8 x 4 B per parameter element, averaged over the ent/rel launches"
        ach = alg / avg_s / 1e9
        return {"bound": "hbm", "kernel": kernel_name, "class": prof_kind, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": traffic_note, "avg_kernel_us": avg_s * 1e6,
                "launches": launches, "algorithmic_bytes_per_launch": alg, "note": note}
    # The pooled kernels reuse every candidate row from registers / LDS / L2 (PMC traffic is ~1/30 of the logical gather
    # bytes), so HBM does not bound them: the fp32 VALU + the quarter-rate transcendental pipe do.
    # achieved = ALGORITHMIC flop per launch (every used (row, pool position, dim) pair term evaluated ONCE; flop per
    # term from the instruction sequence in model_math.h) / mean launch duration, against the 157.3 TFLOP/s fp32
    # vector peak of MI355X_MICROARCH.md.  The MFMA route of the bilinear models is priced as its GEMM over the USED part of
    # the pool (positions some row uses), not over the padded launch.
    used = (info.cnt.to(torch.int32) > 0)
    pairs = int(used.sum().item())                      # (row, pool position) pairs the batch really uses
    p_used = int(used.any(dim=0).sum().item())          # pool positions at least one row uses (P' of SURVEY 8d)
    units = m_.hidden_dim if MODEL == "RotatE" else De  # pair terms per (row, position)
    bwd = prof_kind != "pool_fwd"
    # (TransE backward: z = q - x, then dq -= g sign(z), dx += g sign(z): one subtraction and two multiply-adds = 5 flop per term)
    per_term = {"RotatE": ((6, 1), (15, 1)), "TransE": ((2, 0), (5, 0)), "pRotatE": ((4, 1), (9, 2)),
                "ComplEx": ((2, 0), (4, 0)), "DistMult": ((2, 0), (4, 0))}[MODEL][1 if bwd else 0]
    mfma = MODEL in ("ComplEx", "DistMult")
    if mfma:  # S = Q.X^T | dQ = G.X | dX = G^T.Q over [B, P'] -- algorithmic: the columns somebody uses
        flop, trans = 2.0 * Bk * p_used * De, 0.0
        if paired and prof_kind == "pool_bwd_q":
            flop *= 2.0  # (both backward products ride the launch that is timed)
    else:
        flop, trans = float(pairs) * units * per_term[0], float(pairs) * units * per_term[1]
    ach = flop / avg_s / 1e12
    label = "pool_bwd (single pass: dq and dx from one evaluation of every pair term)" if bwd else "pool_fwd"
    roof = {"bound": "mfma" if mfma else "valu", "kernel": kernel_name, "class": prof_kind, "achieved": ach, "peak": FP32_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": ach / FP32_PEAK_TFLOPS, "traffic": traffic, "traffic_note": traffic_note,
            "avg_kernel_us": avg_s * 1e6, "launches": launches, "algorithmic_flop_per_launch": flop,
            "note": (f"{label}: " + ((f"two MFMA GEMMs (dQ = G . X, dX = G^T . Q) in one launch, 2 * B * P' * De each with P' = {p_used} used pool positions of {2 * K}"
                                      if paired and prof_kind == "pool_bwd_q" else f"MFMA GEMM 2 * B * P' * De with P' = {p_used} used pool positions of {2 * K}")
                                     if mfma else f"{pairs} used (row, pool position) pairs x {units} terms x {per_term[0]} flop")
                     + "; peak = fp32 vector / matrix rate of MI355X")}
    if mfma and os.environ.get("MKB_GEMM_BF16X3", "1") != "0":
        # the product runs on the bf16 matrix pipe: fp32 operands split three ways, six bf16 instructions per fp32 product
        # (gemm_mfma.h): the algorithmic (fp32-equivalent) rate above is what the step gets; the EXECUTED bf16 rate is 6x it
        roof["executed_bf16_TFLOPs"] = 6.0 * ach
        roof["executed_frac_of_bf16_peak"] = 6.0 * ach / BF16_PEAK_TFLOPS
        roof["note"] += (f"; executed on the bf16 matrix pipe as 6 bf16 products per fp32 product (three-way operand split, results "
                         f"within ~2e-7 |a||b| of fp32): {6.0 * ach:.0f} TFLOP/s executed = {6.0 * ach / BF16_PEAK_TFLOPS:.3f} of the "
                         f"{BF16_PEAK_TFLOPS:.0f} TFLOP/s dense bf16 peak")
    if MODEL == "RotatE":
        # ceiling at the instruction rates MEASURED on this part (tools/ubench/trans_rate.hip, profiles/r03_instruction_rates_and_
        # tile_ubench.txt: v_pk_*_f32 2.17 ns, v_sqrt / v_rsq 3.58 ns per wave64 instruction and SIMD, SIMDs saturated): the pair
        # body is 5 packed + 2 v_sqrt per two terms forward, 9 packed + 2 v_rsq backward, on 1024 SIMDs
        body_ns = (9 if bwd else 5) * 2.17 + 2 * 3.58
        floor_us = float(pairs) * units / 2.0 / 64.0 / 1024.0 * body_ns / 1e3
        roof["issue_floor_us"] = floor_us
        roof["frac_of_issue_floor"] = floor_us / (avg_s * 1e6)
    if MODEL == "TransE":
        # issue floor at the measured rate of a plain fp32 VALU instruction (1.17 ns per wave64 instruction and SIMD): forward
        # sub + |.|-add = 2 per term, backward sub + v_med3_i32 + v_cvt + 2 fma = 5 per term (model_math.h), on 1024 SIMDs
        floor_us = float(pairs) * units / 64.0 / 1024.0 * (5 if bwd else 2) * 1.17 / 1e3
        roof["issue_floor_us"] = floor_us
        roof["frac_of_issue_floor"] = floor_us / (avg_s * 1e6)
        roof["bound_note"] = ("2-5 VALU operations per 4-byte pair term: next to the pair math the kernel moves operands (LDS images of the "
                              "candidate rows, the LDS read-modify-write of dx) and keeps its per-position books; the fraction of the "
                              "fp32 peak counts the pair math alone")
    if trans:
        t_rate = trans / avg_s / 1e12
        roof.update({"transcendental_rate_Tops": t_rate, "transcendental_peak_Tops": TRANS_PEAK_TOPS,
                     "transcendental_frac": t_rate / TRANS_PEAK_TOPS,
                     # issue-slot view: fp32 ops 2 cycles per wave and 64 results (SIMD-32), v_rsq / v_sqrt 8
                     "issue_frac": (flop / 2.0 / 32.0 + trans / 8.0) / (1024 * 2.4e9) / avg_s})
    # secondary: SURVEY 8(d)'s logical gather bytes (what the reference formulation would move) -- a REUSE figure, may
    # exceed the HBM peak; never a roofline fraction
    roof["logical_gather_GBps"] = (2 if bwd else 1) * Bk * K * De * 4 / avg_s / 1e9
    return roof


def step_roofline(res, world):
    """Whole-step view: the step's algorithmic flop and its compulsory HBM bytes over ms_per_step."""
    ctx = res["ctx"]
    m_ = ctx["model"]
    De = m_.entity_dim
    Bk = ctx["rows_per_rank"]
    info = ctx["sampler"].generate(ctx["train"][:Bk], "head-batch")._mkb_pool
    ids = info.touched if info.touched is not None else torch.cat([info.pool, ctx["train"][:Bk, 0], ctx["train"][:Bk, 2]])
    rows = int(torch.unique(ids).numel())
    flop_per = {"RotatE": 16 * m_.hidden_dim, "TransE": 6 * De, "pRotatE": 13 * De, "ComplEx": 6 * De, "DistMult": 6 * De}[MODEL]
    flop = float(Bk) * K * flop_per
    # compulsory: every distinct touched row is read once by the step (forward) and moves once through the optimizer
    # (p, m, v, g in; p, m, v, cleared g out): 9 row passes
    byts = 9.0 * rows * De * 4
    s = res["ms_per_step"] / 1e3
    measured = {}
    if res.get("step_traffic"):
        st = res["step_traffic"]
        measured = {"measured_hbm_bytes_per_step": st["measured_hbm_bytes_per_step"],
                    "measured_over_compulsory": st["measured_hbm_bytes_per_step"] / byts,
                    "measured_per_kernel_MB_per_step": st["per_kernel_MB_per_step"], "measured_note": st["note"]}
    return {**measured, "algorithmic_flop_per_step": flop, "achieved_TFLOPs": flop / s / 1e12, "frac_of_fp32_peak": flop / s / 1e12 / FP32_PEAK_TFLOPS,
            "compulsory_hbm_bytes_per_step": byts, "compulsory_GBps": byts / s / 1e9, "frac_of_hbm_peak": byts / s / 1e9 / HBM_PEAK_GBS,
            "distinct_touched_rows": rows,
            "note": f"SURVEY 8(d): B*K*{flop_per} flop per step (fwd + bwd, per rank); compulsory bytes = {rows} distinct touched rows x "
                    f"{De * 4} B x 9 passes (1 read by the step + p, m, v, g in / out through the optimizer); both over ms_per_step"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--windows", type=int, default=int(os.environ.get("MKB_BENCH_WINDOWS", "15")),
                    help="timed windows of --steps steps each, back to back, every one between barriers; ms_per_step / value are "
                         "the MEDIAN window and all of them are printed (windows_ms)")
    ap.add_argument("--preheat-ms", type=float, default=0.0,
                    help="diagnostic: keep the device busy with unrelated torch matmuls for that long right before the warm-up steps "
                         "(tells a clock / power ramp of the device from a warm-up effect of this code's own state)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rows", type=int, default=64, help="rows of the cpu_baseline sample (reference-faithful form)")
    ap.add_argument("--no-variants", action="store_true", help="skip the with/without optimizer & sampler step variants")
    ap.add_argument("--no-traffic", action="store_true", help="skip the in-run rocprofv3 PMC passes behind roofline.traffic")
    ap.add_argument("--no-extras", action="store_true", help="N>1: skip the other partitionings and config 5")
    ap.add_argument("--profile-kernel", default="auto", help="kernel class bracketed with HIP events (or 'none')")
    ap.add_argument("--breakdown", action="store_true", help="also print per-phase timings (stderr)")
    ap.add_argument("--scaling", default=os.environ.get("MKB_BENCH_SCALING", "weak"), choices=["weak", "strong"],
                    help="N>1: weak = 1024 rows per GPU (global batch N*1024); strong = global batch fixed at 1024 rows")
    ap.add_argument("--parallelism", default=os.environ.get("MKB_BENCH_PARALLELISM", "table-rows"), choices=["dims", "rows", "table-rows"],
                    help="N>1: 'table-rows' (default) = entity table + gradient + Adam state sharded by row (north_star's / BASELINE "
                         "config 5's partitioning): pool rows by all-reduce, positive rows by all-to-all, gradients back the same "
                         "way; 'dims' = shard the embedding dimension (one all-reduce of partial scores per step, no gradient "
                         "exchange); 'rows' = batch-row data parallel with sparse gradient all-reduce")
    ap.add_argument("--force-parallelism", action="store_true",
                    help="N=1: run the table-rows code path on the one GPU (world 1, no collective): the per-rank compute of that path")
    ap.add_argument("--dirty-memory", type=float, default=0.0, metavar="GB",
                    help="fill that much device memory with NaN and free it before anything is built: the run then works on dirty "
                         "allocator blocks instead of a fresh process's zero pages (loss / MRR must not change)")
    ap.add_argument("--rccl-world1", action="store_true",
                    help="N=1: the table-rows code path with a world-1 `nccl` process group and MKB_ROWS_FORCE_COLLECTIVES=1 -- every "
                         "collective of the step is issued through RCCL instead of being short-circuited (1-GPU boxes only)")
    ap.add_argument("--config", default="headline", choices=sorted(CONFIGS),
                    help="BASELINE.json configuration (default: the headline; the others are for profiles/)")
    ap.add_argument("--mrr-epochs", type=int, default=30,
                    help="after the timed region (N=1 only): train this many more epochs, then filtered MRR on the test set")
    args = ap.parse_args()
    if args.config != "headline":
        args.mrr_epochs, args.no_cpu_baseline = 0, True

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py --gpus N")
    # MKB_BENCH_ONE_DEVICE=1: functional check of the N>1 path on a single-GPU box (all ranks on cuda:0, gloo)
    one_dev = os.environ.get("MKB_BENCH_ONE_DEVICE", "0") == "1"
    dev_index = 0 if one_dev else local_rank
    torch.cuda.set_device(dev_index)
    if args.dirty_memory > 0:
        blocks = [torch.full((256 << 20,), float("nan"), dtype=torch.float32, device=f"cuda:{dev_index}") for _ in range(max(1, int(args.dirty_memory)))]
        torch.cuda.synchronize()
        del blocks
    device = torch.device("cuda", dev_index)
    dist = None
    if world == 1 and args.rccl_world1:
        import socket

        import torch.distributed as dist

        os.environ["MKB_ROWS_FORCE_COLLECTIVES"] = "1"
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        args.force_parallelism, args.parallelism = True, "table-rows"
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=device)
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if one_dev:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    # ---- the line's own measurement.  N > 1: the north_star partitioning first; should it fail on every rank alike (an
    # RCCL feature missing on the box), fall back rather than lose the line -- the failure is reported in the line.
    order = [args.parallelism] + ([p for p in ("dims", "rows") if p != args.parallelism] if world > 1 else [])
    res, failures = None, {}
    # N > 1 on real GPUs: the row-sharded step's collectives are issued by libmkb_hip.so through its own RCCL communicators, a
    # form no round could run on more than one GPU.  So a SHORT measurement with the collectives in torch.distributed (the
    # round-4 form: plain all_reduce / all_to_all_single) comes first, and a watchdog prints THAT line if the library-issued
    # form has not finished in time -- a hang must not cost the driver its line (and the box its GPUs).
    safety, dog_main = None, None
    if (world > 1 and (not one_dev or os.environ.get("MKB_BENCH_FORCE_SAFETY", "0") == "1")  # (forced: the one-device functional check)
            and order[0] == "table-rows" and os.environ.get("MKB_ROWS_PY_COLLECTIVES", "0") != "1"
            and os.environ.get("MKB_BENCH_NO_SAFETY", "0") != "1"):
        import threading

        os.environ["MKB_ROWS_PY_COLLECTIVES"] = "1"
        try:
            r0 = measure(args, device, rank, world, args.config, "table-rows", min(args.steps, 40), min(args.warmup, 5), profile=False,
                         force=args.force_parallelism)
            safety = {"value": r0["value"], "ms_per_step": r0["ms_per_step"], "loss": r0["loss"], "steps": min(args.steps, 40),
                      "warmup": min(args.warmup, 5), "rows_per_rank": r0["ctx"]["rows_per_rank"],
                      "parallelism": parallelism_label(r0["ctx"], world)}
            del r0
        except Exception as e:  # noqa: BLE001
            print(f"[bench] rank {rank}: torch.distributed form of table-rows failed: {type(e).__name__}: {str(e)[:200]}", file=sys.stderr)
        finally:
            os.environ["MKB_ROWS_PY_COLLECTIVES"] = "0"

        def bail_main():
            if rank == 0:
                line = {"metric": "scored triples/sec (pos+K neg), FB15k-237 RotatE d=1000" if args.config == "headline"
                        else f"scored triples/sec (pos+K neg), {args.config}", "value": safety["value"] if safety else None,
                        "unit": "triples/s", "n_gpus": world, "steps": safety["steps"] if safety else 0,
                        "warmup": safety["warmup"] if safety else 0, "ms_per_step": safety["ms_per_step"] if safety else None,
                        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "see the N=1 line",
                        "config": {"workload": args.config, "global_batch": world * safety["rows_per_rank"] if safety else None,
                                   "negatives": K, "parallelism": safety["parallelism"] if safety else "table-rows"},
                        "loss": safety["loss"] if safety else None, "roofline": None, "cpu_baseline": None,
                        "error": "the library-issued RCCL collectives did not finish in time; this line is the short measurement "
                                 "with the collectives issued by torch.distributed (MKB_ROWS_PY_COLLECTIVES=1)"}
                print(json.dumps(line), flush=True)
            os._exit(0 if safety else 3)

        dog_main = threading.Timer(float(os.environ.get("MKB_BENCH_MAIN_TIMEOUT", "240")), bail_main)
        dog_main.daemon = True
        dog_main.start()
    for par in order:
        try:
            res = measure(args, device, rank, world, args.config, par, args.steps, args.warmup, force=args.force_parallelism,
                          windows=max(1, args.windows))
            break
        except Exception as e:  # noqa: BLE001
            if world == 1:
                raise
            failures[par] = f"{type(e).__name__}: {str(e)[:300]}"
            print(f"[bench] rank {rank}: parallelism {par} failed: {failures[par]}", file=sys.stderr)
            torch.cuda.synchronize()
    if dog_main is not None:
        dog_main.cancel()
    if res is None:
        raise SystemExit(f"every partitioning failed: {failures}")
    ctx = res["ctx"]

    out = None
    if rank == 0:
        if res["sampler_note"]:
            print(res["sampler_note"], file=sys.stderr)
        default_run = world == 1 and not os.environ.get("MKB_BENCH_INNER")
        out = {
            "metric": "scored triples/sec (pos+K neg), FB15k-237 RotatE d=1000" if args.config == "headline"
            else f"scored triples/sec (pos+K neg), {args.config}", "value": res["value"], "unit": "triples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": res["ms_per_step"],
            "higher_is_better": True, "scaling": args.scaling if world > 1 else "weak", "vs_baseline": None, "dtype": "f32",
            "data": (f"{DATASET} train triples (packaged asset" + ("; SYNTHETIC triples: train.csv absent upstream" if DATASET == "yago310" else "")
                     + "), random-init tables (torch.manual_seed(42)), synthetic batch order"),
            "config": {"workload": ("BASELINE configs[2]: " if args.config == "headline" else f"{args.config}: ")
                                   + f"datasets.{DATASET} + models.{MODEL} hidden_dim={HIDDEN}, K={K}, batch {B}/GPU, "
                                   f"Adversarial alpha={ALPHA}, gamma={GAMMA}, dense Adam lr={LR} (row-lazy exact evaluation); "
                                   "step = sampler + pos/neg forward + loss + backward + Adam",
                       "global_batch": world * ctx["rows_per_rank"], "negatives": K,
                       "parallelism": parallelism_label(ctx, world)},
            "loss": res["loss"],
            # the timed region: `windows` back-to-back windows of `steps` steps, each between barriers; ms_per_step and value
            # are the MEDIAN window; nothing is dropped -- every window is listed
            "windows": len(res["windows"]), "windows_ms": [round(w["ms"], 4) for w in res["windows"]],
            "windows_ms_per_step": [round(w["ms"] / args.steps, 5) for w in res["windows"]],
            "windows_spread": (max(w["ms"] for w in res["windows"]) - min(w["ms"] for w in res["windows"])) / res["dt"] / 1e3,
            "t_host_ms_per_step": res["t_host"] / args.steps * 1e3,
            "windows_detail": [{k: round(v, 4) if isinstance(v, float) else v for k, v in w.items()} for w in res["windows"]],
            "gc_collections_during_windows": res["gc"],
            "roofline": roofline_of(res, world, want_traffic=default_run and not args.no_traffic and res["prof_kind"] != "none"),
        }
        if out["roofline"] is not None:
            out["roofline"]["step"] = step_roofline(res, world)
        if failures:
            out["partitioning_failures"] = failures
        if safety is not None:
            out["torch_distributed_form"] = {k: safety[k] for k in ("value", "ms_per_step", "steps", "warmup", "loss")}
        if ctx["trows"]:
            from mkb_amd.table_rows import _collectives_run, _Route

            comm = getattr(ctx["step"], "_comm", None)
            out["table_rows"] = {"collectives_issued": bool(_collectives_run(world)), "backend": dist.get_backend() if dist is not None and dist.is_initialized() else None,
                                 "issued_by": "libmkb_hip.so (own RCCL communicators: mkb_rows_comm_*)" if comm else "torch.distributed",
                                 "steps_counted": args.warmup + 8 + args.steps * len(res["windows"])}
            if comm:
                st = comm.stats()
                out["table_rows"].update(st, host_waits_that_blocked=st["waited_with_idle_stream"],
                                         note="split sizes reach the host through a mailbox the plan's last kernel writes (planned two batches "
                                              "ahead on a side stream); takes_that_waited = the host asked before the plan had executed (it "
                                              "ran ahead of the device); host_waits_that_blocked = those of them that found the step's "
                                              "stream EMPTY, i.e. the device without work")
            else:
                out["table_rows"].update(host_waits_that_blocked=_Route.host_waits,
                                         note="host_waits_that_blocked = steps whose split sizes (read back one batch ahead on a side "
                                              "stream) were not there yet when the step needed them")

    # ---- N > 1: the other partitionings and BASELINE configs[4] in the same run (fewer steps; same barrier + max-over-ranks
    # timing).  A watchdog prints the line without them if they hang: the headline number must not be lost to an extra.
    if world > 1 and not args.no_extras and not failures:
        import threading

        def bail():
            if rank == 0:
                out["extras_error"] = "extras timed out"
                print(json.dumps(out), flush=True)
            os._exit(0)

        dog = threading.Timer(float(os.environ.get("MKB_BENCH_EXTRAS_TIMEOUT", "300")), bail)
        dog.daemon = True
        dog.start()
        main_par = "table-rows" if ctx["trows"] else ("dims" if ctx["dims"] else "rows")
        del ctx
        res["ctx"] = None
        e_steps, e_warm = min(args.steps, 60), min(args.warmup, 10)
        extras, cfg5 = {}, None
        try:
            for par in ("table-rows", "dims", "rows"):
                if par == main_par:
                    continue
                r = measure(args, device, rank, world, args.config, par, e_steps, e_warm, profile=False)
                extras[par] = {"value": r["value"], "ms_per_step": r["ms_per_step"], "steps": e_steps, "warmup": e_warm, "loss": r["loss"]}
                r["ctx"] = None
            if args.config == "headline":
                r = measure(args, device, rank, world, "yago310-rotate", "table-rows", e_steps, e_warm, profile=False)
                cfg5 = {"workload": "BASELINE configs[4]: datasets.Yago310 (123,182 entities; SYNTHETIC train triples, train.csv absent "
                                    f"upstream) + RotatE hidden_dim=500, K=256, batch 1024/GPU, entity table row-sharded over {world} GPUs",
                        "value": r["value"], "unit": "triples/s", "ms_per_step": r["ms_per_step"], "steps": e_steps, "warmup": e_warm,
                        "loss": r["loss"], "parallelism": parallelism_label(r["ctx"], world)}
                r["ctx"] = None
        except Exception as e:  # noqa: BLE001
            if rank == 0:
                out["extras_error"] = f"{type(e).__name__}: {str(e)[:300]}"
        dog.cancel()
        globals().update(CONFIGS[args.config])
        if rank == 0:
            out["other_partitionings"] = extras
            if cfg5 is not None:
                out["config5"] = cfg5

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    ctx = res["ctx"]
    if world == 1 and args.mrr_epochs > 0 and not ctx["trows"]:
        out["mrr"] = train_and_rank(ctx, args.mrr_epochs, args.warmup + 8 + args.steps * len(res["windows"]))
    if world == 1 and args.config == "headline" and not args.no_variants and not ctx["trows"]:
        out["step_variants"] = step_variants(ctx)
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(rows=args.cpu_rows)
    print(json.dumps(out))
    if dist is not None and dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
