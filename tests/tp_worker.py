"""Worker for tests/test_gpu_pool.py::test_dim_sharded_training_equals_single_device: launched with
torch.distributed.run (gloo, every rank on cuda:0).  Trains a few steps with the dimension-sharded step and lazy Adam,
reassembles the tables, and rank 0 compares them with a single-process run of the same batches."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from mkb_amd import datasets, models, optim, parallel, sampling
    from mkb_amd.fused import FusedTrainStep

    name, hidden, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    big = len(sys.argv) > 4 and sys.argv[4] == "big"  # 14,541 entities: the table steps row-lazily, the sharded run defers
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    __import__("util_gpu").dirty_device_memory(0.5)  # (MKB_TEST_DIRTY_MEMORY: NaN-filled freed memory instead of zero pages)
    ds = (datasets.Fb15k237 if big else datasets.Umls)(batch_size=64, shuffle=False, seed=42, num_workers=0)
    train = torch.as_tensor(np.asarray(ds.train, dtype=np.int64)).cuda()

    def run(sharded):
        torch.manual_seed(5)
        full = getattr(models, name)(hidden_dim=hidden, entities=ds.entities, relations=ds.relations, gamma=6.0)
        if name == "pRotatE":
            with torch.no_grad():
                full.modulus.fill_(0.7)
        model = parallel.shard_dims(full, rank, world, "cuda") if sharded else full.cuda()
        ns = sampling.NegativeSampling(size=K, train_triples=ds.train, entities=ds.entities, relations=ds.relations, seed=3)
        opt = optim.Adam([p for p in model.parameters() if p.requires_grad], lr=2e-3, lazy_rows=True,
                         draw_ahead=ns if (big and sharded) else None, defer_step=bool(big and sharded))
        step = parallel.DimShardedStep(model, 0.5) if sharded else FusedTrainStep(model, 0.5)
        losses = []
        g = torch.Generator().manual_seed(9)
        for i in range(6):
            idx = torch.randint(len(train), (96,), generator=g).cuda()
            s, w = train[idx], (torch.rand(96, generator=g) + 0.1).cuda()
            mode = "head-batch" if i % 2 == 0 else "tail-batch"
            if big and sharded:  # the sampler rides the optimizer's advance launch, the real step waits in the gradient rows
                losses.append(step.sampled(s, w, ns, mode).item())
            else:
                losses.append(step(s, w, ns.generate(s, mode), mode).item())
            opt.step()
            opt.zero_grad()
        if sharded:
            assert not big or opt.state[model.entity_embedding].get("defer")
            ent, rel = parallel.gather_dims(model)
        else:
            opt.flush()
            ent, rel = model.entity_embedding.detach(), model.relation_embedding.detach()
        mod = model.modulus.detach().clone() if name == "pRotatE" else None
        return losses, ent, rel, mod

    l1, e1, r1, m1 = run(True)
    if rank == 0:
        l0, e0, r0, m0 = run(False)
        np.testing.assert_allclose(l1, l0, rtol=0, atol=3e-5)
        np.testing.assert_allclose(e1.cpu().numpy(), e0.cpu().numpy(), rtol=0, atol=3e-5)
        np.testing.assert_allclose(r1.cpu().numpy(), r0.cpu().numpy(), rtol=0, atol=3e-5)
        if m0 is not None:
            np.testing.assert_allclose(m1.cpu().numpy(), m0.cpu().numpy(), rtol=0, atol=3e-5)
        print("TP_OK", name, world)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
