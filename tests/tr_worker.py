"""Worker for tests/test_gpu_pool.py::test_row_sharded_table_training_equals_single_device: launched with
torch.distributed.run (gloo, every rank on cuda:0).  Every rank owns the entity rows e % world == rank (table, gradient,
Adam state), runs a few row-sharded steps (mkb_amd.table_rows: pool rows by all-reduce, positive rows by all-to-all, the
fused HIP step on the compact table), and rank 0 compares the reassembled tables with a single-process run."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from mkb_amd import datasets, models, optim, sampling
    from mkb_amd.fused import FusedTrainStep
    from mkb_amd.table_rows import TableRowShardedStep, gather_table_rows, shard_table_rows

    name, hidden, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    size = sys.argv[4] if len(sys.argv) > 4 else "small"
    yago = size == "yago"          # BASELINE config 5's table (123,182 entities) at its batch size: 1024 rows per rank
    wide = size == "wide"          # FB15k-237 with 2200 rows per rank: 4400 row requests, more than the route kernel merges (4096)
    big = size == "big" or yago or wide    # FB15k-237: shards of > 4096 rows step ROW-LAZILY, real step deferred
    # MKB_TR_BACKEND=nccl: the collectives go through RCCL (one rank per GPU; at world 1 together with
    # MKB_ROWS_FORCE_COLLECTIVES=1, which keeps the step from short-circuiting them: tests/test_gpu_rccl_world1.py)
    backend = os.environ.get("MKB_TR_BACKEND", "gloo")
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) if backend == "nccl" else 0)
    __import__("util_gpu").dirty_device_memory(0.5)  # (MKB_TEST_DIRTY_MEMORY: NaN-filled freed memory instead of zero pages)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", torch.cuda.current_device()))
    else:
        dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    ds = (datasets.Yago310 if yago else datasets.Fb15k237 if big else datasets.Umls)(batch_size=64, shuffle=False, seed=42, num_workers=0)
    train = torch.as_tensor(np.asarray(ds.train, dtype=np.int64)).cuda()
    B = (1024 if yago else 2200 if wide else 64 if big else 24) * world
    adam_kw = dict(lazy_rows=True, defer_step=True) if big else {}

    def batches():
        g = torch.Generator().manual_seed(9)
        for i in range(5):
            idx = torch.randint(len(train), (B,), generator=g).cuda()
            yield train[idx], (torch.rand(B, generator=g) + 0.1).cuda(), "head-batch" if i % 2 == 0 else "tail-batch"

    def run(sharded):
        torch.manual_seed(5)
        full = getattr(models, name)(hidden_dim=hidden, entities=ds.entities, relations=ds.relations, gamma=6.0)
        ns = sampling.NegativeSampling(size=K, train_triples=ds.train, entities=ds.entities, relations=ds.relations, seed=3)
        losses = []
        if sharded:
            table, rel = shard_table_rows(full, device="cuda")
            mod = torch.nn.Parameter(full.modulus.detach().clone().cuda()) if name == "pRotatE" else None
            step = TableRowShardedStep(table, rel, 0.5, model_cls=getattr(models, name), hidden_dim=hidden, gamma=6.0,
                                       modulus=mod)
            opt = optim.Adam([table.data, rel] + ([mod] if mod is not None else []), lr=2e-3, **adam_kw)
            lo, hi = rank * B // world, (rank + 1) * B // world
            todo = [(s[lo:hi].contiguous(), w[lo:hi].contiguous(), mode) for s, w, mode in batches()]
            for i, (sl, wl, mode) in enumerate(todo):
                nxt = todo[i + 1][0] if i + 1 < len(todo) and i != 2 else None  # routes planned one batch ahead (and once not)
                if nxt is not None and os.environ.get("MKB_TR_LOOKAHEAD", "1") == "2" and i == 0:
                    nxt = [todo[1][0], todo[2][0]]  # ... and once two batches ahead (plans are consumed in order)
                if big and i % 2 == 1:  # the sampler riding the shard's optimizer launch (identical negatives)
                    losses.append(step.sampled(sl, wl, ns, mode, next_sample=nxt).item())
                else:
                    neg = ns.generate(sl, mode)  # one pool draw per call on every rank: identical pools, own rows filtered
                    losses.append(step(sl, wl, neg, mode, next_sample=nxt).item())
                opt.step()
                opt.zero_grad()
            opt.flush()
            step.check()
            run.lib = step._comm.stats() if step._comm else None
            return losses, gather_table_rows(table), rel.detach().clone(), None if mod is None else mod.detach().clone()
        model = full.cuda()
        opt = optim.Adam([model.entity_embedding, model.relation_embedding] + ([model.modulus] if name == "pRotatE" else []),
                         lr=2e-3, **adam_kw)
        step = FusedTrainStep(model, 0.5)
        for s, w, mode in batches():
            losses.append(step(s, w, ns.generate(s, mode), mode).item())
            opt.step()
            opt.zero_grad()
        opt.flush()
        return losses, model.entity_embedding.detach(), model.relation_embedding.detach(), \
            (model.modulus.detach() if name == "pRotatE" else None)

    l1, e1, r1, m1 = run(True)
    if rank == 0:
        l0, e0, r0, m0 = run(False)
        if m0 is not None:  # the replicated trainable modulus took the same steps
            assert float(m0) != float(getattr(models, name)(hidden_dim=hidden, entities=ds.entities, relations=ds.relations,
                                                            gamma=6.0).modulus)
            np.testing.assert_allclose(m1.cpu().numpy(), m0.cpu().numpy(), rtol=0, atol=3e-5)
        # Adam's update is scale-free (m / sqrt(v)): where a gradient element is ~0, the last bits of the fp32 atomics' order --
        # which differs between the sharded and the single-process run -- decide a step of up to lr.  At YAGO3-10's size
        # (123 M table elements, 5 steps at lr 2e-3) a handful of elements land a few 1e-5 apart: tolerance 3e-4 there, and
        # the bulk must still agree to 3e-5
        tol = 3e-4 if yago else 3e-5
        np.testing.assert_allclose(l1, l0, rtol=0, atol=3e-5)
        d = (e1 - e0).abs()
        assert float(d.max()) <= tol, float(d.max())
        assert float((d > 3e-5).float().mean()) <= 1e-6, float((d > 3e-5).float().mean())
        np.testing.assert_allclose(r1.cpu().numpy(), r0.cpu().numpy(), rtol=0, atol=tol)
        from mkb_amd.table_rows import _collectives_run, _Route
        print("TR_OK", name, world, backend, "collectives_run", _collectives_run(world), "host_waits", _Route.host_waits,
              "lib_collectives", run.lib is not None, run.lib)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
