"""Host-side profile of the row-sharded step with every collective through RCCL at world size 1 (where the step is host-bound):
    python tools/_prof_tr_host.py [config]        (GPU box)"""
import cProfile, os, pstats, socket, sys, time, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch
import torch.distributed as dist
import bench

cfg = sys.argv[1] if len(sys.argv) > 1 else "headline"
os.environ["MKB_ROWS_FORCE_COLLECTIVES"] = "1"
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
bench.__dict__.update(bench.CONFIGS[cfg])
ctx = bench.build(dev, 0, 1, parallelism="table-rows", force=True)
ctx["rows_per_rank"] = bench.B
for i in range(30):
    bench.run_step(ctx, i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(200):
    bench.run_step(ctx, 30 + i)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"{cfg}: host enqueue {t_host / 200 * 1e3:.4f} ms/step, wall {(time.perf_counter() - t0) / 200 * 1e3:.4f} ms/step")
pr = cProfile.Profile()
pr.enable()
for i in range(200):
    bench.run_step(ctx, 230 + i)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumulative").print_stats(30)
dist.destroy_process_group()
