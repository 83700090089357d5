"""TT-Rec lookup timing at the reference's DeepFM TT config (configs/deepfm/tt_rec.yaml: ranks [128, 96]):
per-launch dispatch times of one forward + backward, and wall time per pass (includes the torch planning ops)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recsys_benchmark_amd import _kernels
from recsys_benchmark_amd.embeddings import TTRecTorch
from recsys_benchmark_amd.profiling import KernelTimer
dev = torch.device("cuda")
N, D, B, F = 1086810, 16, 4096, 39
emb = TTRecTorch(N, D, [128, 96], weight_dist="normal").to(dev)
print("p", [int(v) for v in emb.tt_p_shapes], "q", [int(v) for v in emb.tt_q_shapes], "r", emb.tt_ranks)
idx = torch.randint(0, N, (B, F), device=dev)
G = torch.randn(B, F, D, device=dev)
def fb():
    emb.zero_grad()
    (emb(idx) * G).sum().backward()
def wall(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
for mode in ("grouped", "per-lookup"):
    if mode == "per-lookup":
        _kernels._TT_GROUPED_MIN = 1 << 60
    with torch.no_grad():
        t_f = wall(lambda: emb(idx))
    t_fb = wall(fb, 5)
    with KernelTimer(256) as kt:
        fb(); torch.cuda.synchronize()
    tot = sum(us for _, us in kt.records)
    print(f"[{mode}] wall fwd {t_f:.3f} ms, fwd+bwd {t_fb:.3f} ms; library kernels of one fwd+bwd: {tot:.0f} us")
    for k, us in kt.records:
        print(f"    {k:18s} {us:8.1f} us")
