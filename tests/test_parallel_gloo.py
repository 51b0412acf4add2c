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
