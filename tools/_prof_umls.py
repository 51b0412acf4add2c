import cProfile, pstats, sys, torch
sys.path.insert(0, ".")
import bench
bench.__dict__.update(bench.CONFIGS["umls-transe"])
ctx = bench.build(torch.device("cuda", 0), 0, 1)
ctx["rows_per_rank"] = bench.B
for i in range(50): bench.run_step(ctx, i)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for i in range(400): bench.run_step(ctx, 50 + i)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
