"""CPU, world_size 2, gloo: the data-parallel gradient exchange (mkb_amd.parallel) and the sharding math.

The compute of each rank is done by the ORACLE here (there is no GPU in this container): what is under test
is the host-side distributed logic -- that sharding a global batch over ranks with the global normaliser W and
summing only the touched rows reproduces the single-process gradient of the whole batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, fn, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def _run(fn, world=2):
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), fn, ret), nprocs=world, join=True)
    return [ret[r] for r in range(world)]


def _touched_rows_case(rank, world):
    from mkb_amd.parallel import allreduce_touched_rows

    g = torch.Generator().manual_seed(5)
    N, D = 200, 6
    full = [torch.zeros(N, D) for _ in range(world)]
    ids = [torch.randint(N, (17 + 5 * r,), generator=g) for r in range(world)]   # unequal counts, duplicates
    for r in range(world):
        full[r][ids[r]] = torch.randn(ids[r].numel(), D, generator=g)
    rel = [torch.randn(4, 3, generator=g) for _ in range(world)]
    sc = [torch.randn(1, generator=g) for _ in range(world)]
    grad, e1, e2 = full[rank].clone(), rel[rank].clone(), sc[rank].clone()
    moved = allreduce_touched_rows(grad, ids[rank], [e1, e2])
    want = sum(full)
    ok = torch.allclose(grad, want) and torch.allclose(e1, sum(rel)) and torch.allclose(e2, sum(sc))
    # dense fallback gives the same answer
    grad2, e3 = full[rank].clone(), rel[rank].clone()
    moved2 = allreduce_touched_rows(grad2, ids[rank], [e3], dense_threshold=0.0)
    ok = ok and torch.allclose(grad2, want) and moved2 == N and 0 < moved < N
    return bool(ok)


def test_allreduce_touched_rows_world2():
    assert all(_run(_touched_rows_case))


def _sharded_step_case(rank, world):
    """Global batch of 12 rows split over 2 ranks == single-process step on the 12 rows (oracle as compute)."""
    from mkb_amd.parallel import allreduce_touched_rows, shard_rows
    from oracle import scoring

    torch.manual_seed(0)
    tb = scoring.init_tables("RotatE", 60, 5, 8, 6.0)
    g = torch.Generator().manual_seed(1)
    B, K = 12, 9
    sample = torch.stack([torch.randint(60, (B,), generator=g), torch.randint(5, (B,), generator=g),
                          torch.randint(60, (B,), generator=g)], 1)
    pool = torch.randint(60, (2 * K,), generator=g)
    neg = pool[torch.stack([torch.randperm(2 * K, generator=g)[:K] for _ in range(B)])]   # rows draw from ONE pool
    w = torch.rand(B, generator=g) + 0.1
    ref = scoring.train_step_grads(tb, sample, neg, w, "tail-batch", 1.0, fast_norm=True)

    lo, hi = shard_rows(B, rank, world)
    s, n, wl = sample[lo:hi], neg[lo:hi], w[lo:hi]
    W = w.sum()                                    # == all-reduce of the local sums
    wsum = wl.sum().reshape(1)
    dist.all_reduce(wsum)
    assert torch.allclose(wsum, W.reshape(1))
    # local step with the GLOBAL normaliser: scale the locally-normalised result by W_local / W_global
    loc = scoring.train_step_grads(tb, s, n, wl, "tail-batch", 1.0, fast_norm=True)
    scale = wl.sum() / W
    g_ent, g_rel, loss = loc["g_ent"] * scale, loc["g_rel"] * scale, (loc["loss"] * scale).reshape(1)
    ids = torch.cat([s[:, 0], s[:, 2], pool])
    allreduce_touched_rows(g_ent, ids, [g_rel, loss], equal_counts=True)
    ok = (torch.allclose(g_ent, ref["g_ent"], atol=1e-6) and torch.allclose(g_rel, ref["g_rel"], atol=1e-5)
          and torch.allclose(loss, ref["loss"].reshape(1), atol=1e-6))
    return bool(ok)


def test_sharded_global_batch_equals_single_process():
    assert all(_run(_sharded_step_case))


def test_shard_rows_partition():
    from mkb_amd.parallel import shard_rows

    for n in (0, 1, 7, 1024):
        for world in (1, 2, 3, 8):
            parts = [shard_rows(n, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))


def test_scores_are_sums_over_dimension_shards():
    """The identity dimension sharding rests on (oracle closed form, no distributed runtime needed): for every model
    the score is c0 + sum over shards of the shard's partial sum, with the phase divisor taken from the GLOBAL dims."""
    from oracle import closed, scoring

    rs = np.random.RandomState(0)
    N, R, d, B, K, gamma, world = 30, 4, 12, 5, 6, 6.0, 3
    sample = np.stack([rs.randint(N, size=B), rs.randint(R, size=B), rs.randint(N, size=B)], 1)
    neg = rs.randint(N, size=(B, K))
    k = closed.emb_range_over_pi(gamma, d)
    for model in scoring.MODELS:
        de, dr = scoring.dims(model, d)
        ent, rel = rs.randn(N, de) * 0.3, rs.randn(R, dr) * 0.3
        for mode in ("head-batch", "tail-batch"):
            head = mode == "head-batch"
            full = closed.scores(model, ent, rel, sample, neg, mode, gamma, d, modulus=np.array([[0.7]]))
            total = np.zeros_like(full)
            for g in range(world):
                own = np.arange(g * d // world, (g + 1) * d // world)
                ec = np.concatenate([own, d + own]) if model in ("RotatE", "ComplEx") else own
                rc = np.concatenate([own, d + own]) if model == "ComplEx" else own
                q = closed.build_query(model, ent[:, ec], rel[:, rc], sample, head, k)
                total += closed.pair_scores(model, q, ent[:, ec][neg], head, 0.0, k, 0.7)   # partial: gamma = 0
            c0 = float(np.float32(gamma)) if model in ("TransE", "RotatE", "pRotatE") else 0.0
            np.testing.assert_allclose(total + c0, full, rtol=0, atol=1e-12)


# ------------------------------------------------------------------------------------------------ row-sharded entity table
def _table_rows_case(rank, world):
    """mkb_amd.table_rows (BASELINE config 5's partitioning): entity rows e % world == rank live on
    ``rank`` together with their gradient and Adam state; pool rows are gathered with one all-reduce, positive rows
    through an all-to-all; the oracle computes each rank's step on the COMPACT table.  Three steps (head / tail / head)
    with dense Adam on the shards must reproduce the single-process run on the full table."""
    from types import SimpleNamespace

    from mkb_amd.table_rows import RowShardedTable, TableRowShardedStep, gather_table_rows
    from oracle import scoring
    from row_ops_torch import TorchRowOps

    name, N, R, hidden, gamma, alpha = "RotatE", 61, 5, 8, 6.0, 0.5   # 61 rows: uneven shards
    B, K = 6 * world, 7
    torch.manual_seed(0)
    full = scoring.init_tables(name, N, R, hidden, gamma)
    g = torch.Generator().manual_seed(2)

    def batch():
        sample = torch.stack([torch.randint(N, (B,), generator=g), torch.randint(R, (B,), generator=g),
                              torch.randint(N, (B,), generator=g)], 1)
        pool = torch.randint(N, (2 * K,), generator=g)                       # duplicates allowed, like the reference's pool
        pos = torch.stack([torch.randperm(2 * K, generator=g)[:K] for _ in range(B)])
        cnt = torch.zeros(B, 2 * K, dtype=torch.int32).scatter_add_(1, pos, torch.ones_like(pos, dtype=torch.int32))
        return sample, pool, pos, cnt, torch.rand(B, generator=g) + 0.1

    def oracle_compute(ent, rel, sample, weight, info, mode, weight_sum):
        tb = scoring.Tables(name, hidden, gamma, ent, rel.detach(), full.modulus)
        r = scoring.train_step_grads(tb, sample, info.pos.long(), weight, mode, alpha, fast_norm=True)
        scale = weight.sum() / weight_sum            # global normaliser W (adversarial.py:28-29)
        return r["loss"] * scale, r["g_ent"] * scale, r["g_rel"] * scale

    table = RowShardedTable.from_full(full.ent, ops=TorchRowOps())  # (CPU: the protocol; the HIP row kernels run in -m gpu)
    rel = torch.nn.Parameter(full.rel.clone())
    step = TableRowShardedStep(table, rel, alpha, compute=oracle_compute)
    ref_ent, ref_rel = full.ent.clone(), full.rel.clone()
    st = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in (("e", ref_ent), ("r", ref_rel))}
    st_loc = {"e": (torch.zeros_like(table.data), torch.zeros_like(table.data)), "r": (torch.zeros_like(rel), torch.zeros_like(rel))}
    ok = table.data.shape[0] == len(range(rank, N, world))
    for it, mode in enumerate(["head-batch", "tail-batch", "head-batch"]):
        sample, pool, pos, cnt, w = batch()
        # single process, full table
        tb = scoring.Tables(name, hidden, gamma, ref_ent, ref_rel, full.modulus)
        ref = scoring.train_step_grads(tb, sample, pool[pos], w, mode, alpha, fast_norm=True)
        # this rank's rows of the same global batch
        lo, hi = rank * B // world, (rank + 1) * B // world
        neg = pool[pos][lo:hi]
        neg._mkb_pool = SimpleNamespace(pool=pool, pos=pos[lo:hi].to(torch.int32), cnt=cnt[lo:hi], size=K, mode_id=0)
        table.data.grad, rel.grad = None, None
        loss = step(sample[lo:hi], w[lo:hi], neg, mode)
        ok = ok and torch.allclose(loss, ref["loss"], atol=1e-6)
        ok = ok and torch.allclose(table.data.grad, ref["g_ent"][rank::world], atol=1e-6)
        ok = ok and torch.allclose(rel.grad, ref["g_rel"], atol=1e-5)
        # dense Adam: the full table in one process, the shard here
        scoring.adam_update(ref_ent, ref["g_ent"], *st["e"], it + 1, lr=1e-2)
        scoring.adam_update(ref_rel, ref["g_rel"], *st["r"], it + 1, lr=1e-2)
        with torch.no_grad():
            scoring.adam_update(table.data, table.data.grad, *st_loc["e"], it + 1, lr=1e-2)
            scoring.adam_update(rel, rel.grad, *st_loc["r"], it + 1, lr=1e-2)
        ok = ok and torch.allclose(gather_table_rows(table), ref_ent, atol=1e-6) and torch.allclose(rel.detach(), ref_rel, atol=1e-6)
    return bool(ok)


@pytest.mark.parametrize("world", [2, 4])
def test_row_sharded_entity_table_equals_single_process(world):
    assert all(_run(_table_rows_case, world=world))


def _private_gather_case(rank, world):
    from mkb_amd.table_rows import RowShardedTable, gather_table_rows
    from row_ops_torch import TorchRowOps

    torch.manual_seed(3)
    full = torch.randn(37, 4)
    t = RowShardedTable.from_full(full, ops=TorchRowOps())
    g = torch.Generator().manual_seed(10 + rank)
    ids = torch.randint(37, (5 + 3 * rank,), generator=g)                 # different counts per rank, duplicates
    rows, route = t.gather_private(ids)
    ok = torch.equal(rows, full[ids]) and torch.equal(gather_table_rows(t), full)
    shared = torch.randint(37, (11,), generator=torch.Generator().manual_seed(99))
    ok = ok and torch.equal(t.gather_shared(shared), full[shared])
    grads = torch.randn(ids.numel(), 4, generator=g)
    t.scatter_add_private(route, grads)
    want = torch.zeros_like(full)
    all_ids = [torch.empty(5 + 3 * r, dtype=torch.int64) for r in range(world)]
    all_g = [torch.empty(5 + 3 * r, 4) for r in range(world)]
    for r in range(world):  # broadcast every rank's contribution to build the expected dense gradient
        src_ids, src_g = (ids, grads) if r == rank else (all_ids[r], all_g[r])
        dist.broadcast(src_ids, src=r)
        dist.broadcast(src_g, src=r)
        want.index_add_(0, src_ids, src_g)
    return bool(ok and torch.allclose(t.data.grad, want[rank::world], atol=1e-6))


def test_private_row_gather_and_gradient_return_world3():
    assert all(_run(_private_gather_case, world=3))


def _plan_case(rank, world):
    """Routes planned one batch ahead (mkb_amd.table_rows): a plan is keyed on the batch as the CALLER holds it, so a
    non-contiguous view planned and then stepped hits the plan on every rank alike (the copy made for the kernels has another
    address: that used to be a miss on that rank only -- one collective more than its peers: a hang); a step for ANOTHER batch
    than the one planned raises on every rank instead of re-planning silently; drop_plan() discards a plan."""
    from types import SimpleNamespace

    from mkb_amd.table_rows import RowShardedTable, TableRowShardedStep
    from oracle import scoring
    from row_ops_torch import TorchRowOps

    name, N, R, hidden, gamma, alpha, K = "TransE", 40, 3, 6, 6.0, 1.0, 4
    B = 4 * world
    torch.manual_seed(0)
    full = scoring.init_tables(name, N, R, hidden, gamma)
    g = torch.Generator().manual_seed(4)
    calls = []

    def compute(ent, rel, sample, weight, info, mode, weight_sum):
        calls.append(1)
        tb = scoring.Tables(name, hidden, gamma, ent, rel.detach(), full.modulus)
        r = scoring.train_step_grads(tb, sample, info.pos.long(), weight, mode, alpha, fast_norm=True)
        scale = weight.sum() / weight_sum
        return r["loss"] * scale, r["g_ent"] * scale, r["g_rel"] * scale

    table = RowShardedTable.from_full(full.ent, ops=TorchRowOps())
    rel = torch.nn.Parameter(full.rel.clone())
    step = TableRowShardedStep(table, rel, alpha, compute=compute)
    wide = torch.stack([torch.randint(N, (3 * B,), generator=g), torch.randint(R, (3 * B,), generator=g),
                        torch.randint(N, (3 * B,), generator=g), torch.zeros(3 * B, dtype=torch.int64)], 1)
    pool = torch.randint(N, (2 * K,), generator=g)
    pos = torch.stack([torch.randperm(2 * K, generator=g)[:K] for _ in range(B)])
    cnt = torch.zeros(B, 2 * K, dtype=torch.int32).scatter_add_(1, pos, torch.ones_like(pos, dtype=torch.int32))
    lo, hi = rank * B // world, (rank + 1) * B // world

    def neg_of():
        neg = pool[pos][lo:hi]
        neg._mkb_pool = SimpleNamespace(pool=pool, pos=pos[lo:hi].to(torch.int32), cnt=cnt[lo:hi], size=K, mode_id=0)
        return neg

    w = torch.ones(hi - lo)
    views = [wide[i * B + lo: i * B + hi, :3] for i in range(3)]   # NON-contiguous [b, 3] views of a [3B, 4] table
    assert not views[0].is_contiguous()
    step(views[0], w, neg_of(), "head-batch", next_sample=views[1])        # plans batch 1 while stepping batch 0
    planned = len(step._plans) == 1
    step(wide[B + lo: B + hi, :3], w, neg_of(), "tail-batch")              # a fresh view of the same storage: the plan is taken
    took = not step._plans
    step.plan(views[2], 2 * K)
    raised = False
    try:
        step(views[0], w, neg_of(), "head-batch")                          # not the batch that was planned
    except RuntimeError:
        raised = True
    step.plan(views[2], 2 * K)
    step.drop_plan()
    step(views[0], w, neg_of(), "head-batch")                              # plans inline (every rank alike)
    # several batches ahead: plans are consumed in the order they were made
    step(views[0], w, neg_of(), "head-batch", next_sample=[views[1], views[2]])
    deep = len(step._plans) == 2
    step(views[1], w, neg_of(), "tail-batch", next_sample=[views[2]])      # already planned: nothing new
    deep = deep and len(step._plans) == 1
    step(views[2], w, neg_of(), "tail-batch")
    deep = deep and not step._plans
    return bool(planned and took and raised and deep and len(calls) == 6)


def test_row_sharded_routes_planned_ahead_are_keyed_alike_on_every_rank():
    assert all(_run(_plan_case, world=2))
