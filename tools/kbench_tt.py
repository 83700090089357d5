"""TT-Rec lookup timing at the reference's DeepFM TT config (configs/deepfm/tt_rec.yaml: ranks [128, 96])."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recsys_benchmark_amd.embeddings import TTRecTorch
from recsys_benchmark_amd.profiling import KernelTimer
dev = torch.device("cuda")
N, D, B, F = 1086810, 16, 4096, 39
emb = TTRecTorch(N, D, [128, 96], weight_dist="normal").to(dev)
print("p", emb.tt_p_shapes, "q", emb.tt_q_shapes, "r", emb.tt_ranks, "params", emb.get_num_params())
idx = torch.randint(0, N, (B, F), device=dev)
G = torch.randn(B, F, D, device=dev)
def fb():
    emb.zero_grad()
    (emb(idx) * G).sum().backward()
for _ in range(2): fb()
torch.cuda.synchronize()
with KernelTimer(64) as kt:
    for _ in range(3): fb()
    torch.cuda.synchronize()
for k, v in kt.summary().items():
    print(k, f"{v['avg_us']:.0f} us  ({B*F/v['avg_us']:.2f} lookups/us)")
