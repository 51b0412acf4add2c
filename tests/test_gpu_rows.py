"""-m gpu: the row-movement kernels of the row-sharded entity table (mkb_amd/csrc/rows.hip through
mkb_amd.table_rows.HipRowOps) against their torch restatement (tests/row_ops_torch.py), and the sharded form of the
row-lazy optimizer's catch-up against the plain one on a materialised id list (bit for bit)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("b", [5, 1024, 1500, 2048, 2049, 3000])  # (2 b requests: merged through the LDS table up to 4096 of them)
def test_route_groups_requests_by_owner_in_request_order(world, b):
    from mkb_amd.table_rows import HipRowOps
    from row_ops_torch import TorchRowOps

    g = torch.Generator().manual_seed(b * 31 + world)
    sample = torch.stack([torch.randint(100000, (b,), generator=g), torch.randint(37, (b,), generator=g),
                          torch.randint(100000, (b,), generator=g)], 1)
    sample[: b // 3, 0] = sample[0, 0]  # a hot entity: one owner gets far more than its share
    got = HipRowOps().route(sample.cuda(), world, row0=777, sample_layout=True)
    ref = TorchRowOps().route(sample, world, row0=777, sample_layout=True, merge=2 * b <= 4096)
    for a, r in zip(got, ref):
        assert torch.equal(a.cpu(), r), (world, b)
    flat = torch.randint(5000, (2 * b + 1,), generator=g)
    got = HipRowOps().route(flat.cuda(), world)
    ref = TorchRowOps().route(flat, world, merge=2 * b + 1 <= 4096)
    assert got[3] is None and all(torch.equal(a.cpu(), r) for a, r in zip(got[:3], ref[:3]))


@pytest.mark.parametrize("D", [1000, 64, 37])
def test_gather_and_scatter_add_equal_the_torch_restatement(D):
    from mkb_amd.table_rows import HipRowOps
    from row_ops_torch import TorchRowOps

    g = torch.Generator().manual_seed(D)
    world, rank, n_local = 4, 1, 900
    shard = torch.randn(n_local, D, generator=g)
    pool = torch.randint(n_local * world, (512,), generator=g)      # global ids: ~1/4 are this rank's
    want = torch.randint(n_local, (700,), generator=g)              # shard indices, with duplicates
    weight = torch.rand(1024, generator=g)
    outs = {}
    for tag, ops, dev in (("hip", HipRowOps(), "cuda"), ("ref", TorchRowOps(), "cpu")):
        a, b = torch.full((512, D), 7.0, device=dev), torch.full((700, D), 7.0, device=dev)
        loc = torch.full((512,), 99, dtype=torch.int64, device=dev)
        wsum = torch.zeros(1, device=dev)
        junk = torch.ones(333 * 4 + 3, device=dev)  # not a multiple of 16 bytes
        ops.gather(shard.to(dev), [(pool.to(dev), a, world, rank, loc), (want.to(dev), b, 0, 0, None)], weight=weight.to(dev),
                   weight_sum=wsum, zero=junk)
        outs[tag] = [t.cpu() for t in (a, b, loc, wsum, junk)]
    for x, y in zip(outs["hip"][:3], outs["ref"][:3]):
        assert torch.equal(x, y)
    np.testing.assert_allclose(outs["hip"][3].numpy(), outs["ref"][3].numpy(), rtol=1e-6)
    assert not outs["hip"][4].any()
    # scatter: duplicates add (atomics: compare to fp32 tolerance), rows other ranks own are skipped, the dense rider adds
    rows_a, rows_b = torch.randn(512, D, generator=g), torch.randn(700, D, generator=g)
    dd, ds = torch.randn(4000, generator=g), torch.randn(4000, generator=g)
    res = {}
    for tag, ops, dev in (("hip", HipRowOps(), "cuda"), ("ref", TorchRowOps(), "cpu")):
        grad, dst = torch.zeros(n_local, D, device=dev), dd.clone().to(dev)
        ops.scatter_add(grad, [(pool.to(dev), rows_a.to(dev), world, rank, None), (want.to(dev), rows_b.to(dev), 0, 0, None)],
                        dense_dst=dst, dense_src=ds.to(dev))
        res[tag] = (grad.cpu(), dst.cpu())
    np.testing.assert_allclose(res["hip"][0].numpy(), res["ref"][0].numpy(), rtol=0, atol=1e-5)
    assert torch.equal(res["hip"][1], res["ref"][1])
    # with the occurrence counts of a gather over the same segments: rows listed once are added without atomics, same sums;
    # the scatter resets the counts, so a second step starts from zero
    occ = torch.zeros(n_local, dtype=torch.int32, device="cuda")
    grad = torch.zeros(n_local, D, device="cuda")
    for pl, wt in [(pool, want), (pool.flip(0), want[:300])]:
        a, b = torch.empty(pl.numel(), D, device="cuda"), torch.empty(wt.numel(), D, device="cuda")
        segs = [(pl.cuda(), a, world, rank, None), (wt.cuda(), b, 0, 0, None)]
        HipRowOps().gather(shard.cuda(), segs, occ=occ)
        mine = (pl % world) == rank
        cnt = torch.bincount(torch.cat([pl[mine] // world, wt]), minlength=n_local)
        assert torch.equal(occ.cpu().long(), cnt)
        ra, rb = rows_a[: pl.numel()].cuda(), rows_b[: wt.numel()].cuda()
        HipRowOps().scatter_add(grad, [(pl.cuda(), ra, world, rank, None), (wt.cuda(), rb, 0, 0, None)], occ=occ)
        assert not occ.any()
    want_grad = torch.zeros(n_local, D)
    TorchRowOps().scatter_add(want_grad, [(pool, rows_a, world, rank, None), (want, rows_b, 0, 0, None)])
    TorchRowOps().scatter_add(want_grad, [(pool.flip(0), rows_a, world, rank, None), (want[:300], rows_b[:300], 0, 0, None)])
    np.testing.assert_allclose(grad.cpu().numpy(), want_grad.numpy(), rtol=0, atol=2e-5)


@pytest.mark.parametrize("defer", [True, False])
def test_sharded_catch_up_is_the_plain_catch_up_on_the_materialised_list(defer):
    """mkb_adam_rows_advance_sharded(global ids filtered by ownership | shard indices) == mkb_adam_rows_advance / _catchup on
    the list [e // world for owned e] + shard indices: the same rows, the same replay -> identical bits."""
    from mkb_amd import _links, optim

    world, rank, n_local, D = 4, 2, 6000, 96
    outs = []
    for sharded in (True, False):
        g = torch.Generator().manual_seed(1)
        p = torch.nn.Parameter(torch.randn(n_local, D, generator=g).cuda())
        p.grad = torch.zeros_like(p)
        opt = optim.Adam([p], lr=1e-2, lazy_rows=True, defer_step=defer)
        for it in range(9):
            pool = torch.randint(n_local * world, (256,), generator=g).cuda()
            want = torch.randint(n_local, (300,), generator=g).cuda()
            mine = (pool % world) == rank
            ids = torch.cat([torch.where(mine, pool // world, torch.full_like(pool, -1)), want])
            if sharded:
                opt.catch_up_sharded(p, pool, world, rank, want)
                opt._state(p)["caught_up"] = (ids, opt._state(p)["n"])
            else:
                opt.catch_up(p, ids)   # (negative entries are skipped by the kernels)
            rows = torch.unique(ids[ids >= 0])
            p.grad[rows] += torch.randn(rows.numel(), D, generator=g).cuda()
            _links.mark_touched(p, ids)
            opt.step()
            opt.zero_grad()
        opt.flush()
        outs.append((p.detach().clone(), opt.state[p]["m"].clone(), opt.state[p]["v"].clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize("world,b", [(2, 96), (3, 50), (8, 128), (8, 1024)])
def test_plan_blocks_played_for_several_ranks_on_one_device(world, b):
    """The device side of ``mkb_rows_comm_plan`` at world > 1.  RCCL refuses two ranks on one device and only 1-GPU boxes can be
    reached, so the plan's transport cannot run here with several ranks -- but its two kernels can: every rank's route + pack is
    played in this process, the blocks are handed over the way the send / receive group would (rank r receives block r of every
    rank), and unpack must give every owner the right ``want`` list and both count vectors.  Then the rows travel the same way
    (owners gather ``want``, requesters take their segments) and every request must come back with its row of the full table."""
    import ctypes

    from mkb_amd import _hip
    from mkb_amd.table_rows import HipRowOps

    lib = _hip.lib()
    g = torch.Generator().manual_seed(100 * world + b)
    N, D, cap = 5000, 12, 2 * b
    full = torch.randn(N, D, generator=g).cuda()
    ops = HipRowOps()
    ranks = []
    for r in range(world):
        hubs = torch.randint(N, (5,), generator=g)
        ent = torch.where(torch.rand(2 * b, generator=g) < 0.3, hubs[torch.randint(5, (2 * b,), generator=g)], torch.randint(N, (2 * b,), generator=g))
        sample = torch.stack([ent[:b], torch.randint(7, (b,), generator=g), ent[b:]], 1).cuda()
        send_ids, slot, counts, compact = ops.route(sample, world, 0, sample_layout=True)
        blocks = torch.full((world, 1 + cap), -7, dtype=torch.int64, device="cuda")
        _hip.check(lib.mkb_rows_blocks_pack(_hip.ptr(counts), _hip.ptr(send_ids), _hip.ptr(blocks), world, cap, _hip.stream_ptr()), "pack")
        ranks.append(dict(sample=sample, send_ids=send_ids, slot=slot, counts=counts, blocks=blocks))
    counts_all = torch.stack([x["counts"] for x in ranks]).cpu()  # [requester, owner]
    for r, me in enumerate(ranks):
        recv = torch.stack([ranks[j]["blocks"][r] for j in range(world)]).contiguous()  # what the group delivers to owner r
        want = torch.full((world * cap,), -9, dtype=torch.int64, device="cuda")
        mail = torch.zeros(1 + 2 * 64, dtype=torch.int64, device="cuda")
        bad = torch.zeros(1, dtype=torch.int32, device="cuda")
        _hip.check(lib.mkb_rows_blocks_unpack(_hip.ptr(recv), _hip.ptr(me["counts"]), _hip.ptr(want), want.numel(), _hip.ptr(mail), 41 + r,
                                             _hip.ptr(bad), world, cap, _hip.stream_ptr()), "unpack")
        m = mail.cpu()
        assert int(bad.item()) == 0 and int(m[0]) == 41 + r
        assert torch.equal(m[1: 1 + world], counts_all[r]) and torch.equal(m[65: 65 + world], counts_all[:, r])
        expect = []
        for j in range(world):
            lo = int(counts_all[j, :r].sum())
            expect.append(ranks[j]["send_ids"][lo: lo + int(counts_all[j, r])])
        expect = torch.cat(expect)
        assert torch.equal(want[: expect.numel()], expect)
        me["want"], me["wanted"] = want[: expect.numel()], counts_all[:, r]
        # the owner reads what it was asked for (shard index s of rank r is entity s * world + r)
        me["reply"] = full[me["want"] * world + r]
    for j, me in enumerate(ranks):  # rows back to the requesters, owner after owner: the all-to-all of the step
        got = []
        for r in range(world):
            lo = int(counts_all[:j, r].sum())
            got.append(ranks[r]["reply"][lo: lo + int(counts_all[j, r])])
        got = torch.cat(got)
        ent = torch.cat([me["sample"][:, 0], me["sample"][:, 2]])
        assert torch.equal(got[me["slot"].long()], full[ent])
