import cProfile, pstats, sys, torch
sys.path.insert(0, "/root/repo")
import bench, recsys_benchmark_amd as pkg
dev = torch.device("cuda", 0)
dims = list(bench.CRITEO_KAGGLE_26)
torch.manual_seed(0)
m = pkg.DeepFM(dims, 16, [400, 400, 400], p_dropout=0.5, use_batchnorm=True, embedding_config={"name": "vanilla", "sparse": True}, fc_sparse=True).to(dev)
m.pack_tables(); m.eval()
x, _ = bench.synth_batch(dims, 64, 1, dev)
with torch.no_grad():
    for _ in range(20): m(x)
    torch.cuda.synchronize()
    import time
    t = time.perf_counter()
    for _ in range(200): m(x)
    torch.cuda.synchronize()
    print("eager us/forward", (time.perf_counter() - t) / 200 * 1e6)
    pr = cProfile.Profile(); pr.enable()
    for _ in range(200): m(x)
    torch.cuda.synchronize()
    pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
