"""LightGCN step around the propagation at Yelp2018 shape (U=31 668, I=38 048, D=64): validation scoring tail
(2048 users per batch, k=20, ~36 train items per user) and the BPR loss over 2048 triples; library kernels
by dispatch events, wall time vs the same step in stock torch ops on the GPU."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recsys_benchmark_amd.lightgcn import score_topk
from recsys_benchmark_amd.losses import bpr_loss_rows
from recsys_benchmark_amd.profiling import KernelTimer
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
U, I, D, B, k = 31668, 38048, 64, 2048, 20
ue, ie = torch.randn(U, D, generator=g).to(dev), torch.randn(I, D, generator=g).to(dev)
lens = torch.randint(5, 70, (U,), generator=g)
crow = torch.zeros(U + 1, dtype=torch.int64); crow[1:] = torch.cumsum(lens, 0)
col = torch.randint(0, I, (int(crow[-1]),), generator=g)
csr = (crow.to(dev), col.to(dev))
users = torch.randint(0, U, (B,), generator=g).to(dev)
def wall(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
def torch_tail():
    s = ue[users] @ ie.T
    # the device-side equivalent of the reference's ind0/ind1 lists (built here without the Python loop)
    cnt = csr[0][users + 1] - csr[0][users]
    ind0 = torch.repeat_interleave(torch.arange(B, device=dev), cnt)
    start = torch.repeat_interleave(csr[0][users], cnt)
    off = torch.arange(ind0.numel(), device=dev) - torch.repeat_interleave(torch.cumsum(cnt, 0) - cnt, cnt)
    s[ind0, csr[1][start + off]] = float("-inf")
    return torch.topk(s, k)[1]
a, b = score_topk(ue, ie, users, k, csr), torch_tail()
print("agreement with torch ops:", float((a == b).float().mean()))
print(f"score_topk wall {wall(lambda: score_topk(ue, ie, users, k, csr)):.0f} us; torch ops {wall(torch_tail):.0f} us")
with KernelTimer(64) as kt:
    score_topk(ue, ie, users, k, csr); torch.cuda.synchronize()
for name, us in kt.records: print(f"    {name:14s} {us:8.1f} us")
E = torch.randn(U + I, D, generator=g).to(dev).requires_grad_(True)
pos, neg = torch.randint(0, I, (B,), generator=g).to(dev), torch.randint(0, I, (B,), generator=g).to(dev)
def fused():
    E.grad = None
    au, ai = torch.split(E, [U, I])
    bpr_loss_rows(au, ai, users, pos, neg).backward()
def stock():
    E.grad = None
    au, ai = torch.split(E, [U, I])
    u, p, n = au.index_select(0, users), ai.index_select(0, pos), ai.index_select(0, neg)
    (-torch.nn.functional.logsigmoid((u * p).sum(1) - (u * n).sum(1)).mean()).backward()
print(f"bpr fwd+bwd wall: fused {wall(fused):.0f} us; torch ops {wall(stock):.0f} us")
with KernelTimer(64) as kt:
    fused(); torch.cuda.synchronize()
for name, us in kt.records: print(f"    {name:14s} {us:8.1f} us")
# contrastive loss on the batch's distinct user + positive rows (src/trainer/lightgcn.py:215-229)
from recsys_benchmark_amd.losses import info_nce
n = 3900
V = torch.randn(n, D, generator=g).to(dev).requires_grad_(True)
def nce_fused():
    V.grad = None
    info_nce(V, V, 0.2).backward()
def nce_stock():
    V.grad = None
    v = torch.nn.functional.normalize(V, dim=1)
    (-torch.diag(torch.nn.functional.log_softmax(v @ v.T / 0.2, dim=1)).mean()).backward()
nce_fused(); a = V.grad.clone(); nce_stock()
print("info_nce grad max |diff| vs torch ops:", float((a - V.grad).abs().max()))
print(f"info_nce fwd+bwd (n={n}) wall: fused {wall(nce_fused):.0f} us; torch ops {wall(nce_stock):.0f} us")
with KernelTimer(64) as kt:
    nce_fused(); torch.cuda.synchronize()
for name, us in kt.records: print(f"    {name:14s} {us:8.1f} us")
