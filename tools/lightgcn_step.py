#!/usr/bin/env python3
"""One LightGCN training step at the Yelp2018 shape (C5: U=31 668, I=38 048, nnz(A)=2.25 M, D=64, L=3, 2048 BPR
triples, weight_decay on): model(adj) -> fused BPR over the propagated tables + reg loss -> backward, as the
reference's _train_step does (src/trainer/lightgcn.py:380-421, optimizer excluded), launched eagerly and replayed
as ONE hipGraph; and the same step in stock torch ops on the same GPU (torch CSR matmul) for scale."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kbench2 import yelp_graph  # noqa: E402

import recsys_benchmark_amd as pkg  # noqa: E402
from recsys_benchmark_amd.lightgcn import LightGCN  # noqa: E402
from recsys_benchmark_amd.losses import bpr_loss_rows  # noqa: E402

dev = torch.device("cuda")
U, I, D, L, B = 31668, 38048, 64, 3, 2048


def wall(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6


def main():
    adj = yelp_graph().to(dev)
    torch.manual_seed(0)
    model = LightGCN(U, I, num_layers=L, hidden_size=D).to(dev).train()
    g = torch.Generator().manual_seed(1)
    users = torch.randint(0, U, (B,), generator=g).to(dev)
    pos, neg = torch.randint(0, I, (B,), generator=g).to(dev), torch.randint(0, I, (B,), generator=g).to(dev)
    one = torch.ones((), device=dev)

    def step():
        au, ai = model(adj)
        loss = bpr_loss_rows(au, ai, users, pos, neg) + 1e-4 * model.get_reg_loss(users, pos, neg)
        loss.backward(one)
        return loss

    model.zero_grad(set_to_none=True)
    t_eager = wall(lambda: (model.zero_grad(set_to_none=True), step()))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            model.zero_grad(set_to_none=True)
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    model.zero_grad(set_to_none=True)
    with torch.cuda.graph(graph):
        loss = step()
    t_graph = wall(graph.replay)
    gn = float(sum(p.grad.abs().sum() for p in model.parameters() if p.grad is not None))
    print(f"[product] eager {t_eager:.0f} us/step, hipGraph replay {t_graph:.0f} us/step ({B / t_graph:.2f} M triples/s); "
          f"loss {float(loss):.4f}, |grad|_1 {gn:.3e}")

    # the whole _train_step (zero_grad + Adam included) as trainer.GraphedCFTrainStep replays it
    from recsys_benchmark_amd import optim, trainer
    for nce in (0.0, 0.1):
        m2 = LightGCN(U, I, num_layers=L, hidden_size=D).to(dev).train()
        tstep = trainer.GraphedCFTrainStep(m2, adj, optim.Adam(m2.parameters(), lr=1e-3), weight_decay=1e-4, info_nce_weight=nce)
        t_train = wall(lambda: tstep(users, pos, neg))
        print(f"[product, whole step with Adam, info_nce_weight={nce}] {t_train:.0f} us/step = {B / t_train:.2f} M triples/s "
              f"({'one hipGraph' if tstep._graph is not None else 'eager: torch.unique has a data-dependent shape'})")

    # the same step in stock torch ops (reference op sequence on the GPU)
    Eu = model.user_emb_table.get_weight().detach().clone().requires_grad_(True)
    Ei = model.item_emb_table.get_weight().detach().clone().requires_grad_(True)

    def stock():
        Eu.grad = Ei.grad = None
        e = torch.cat([Eu, Ei])
        res = e
        for _ in range(L):
            e = adj @ e
            res = res + e
        res = res / (L + 1)
        au, ai = torch.split(res, [U, I])
        u, p, n = au.index_select(0, users), ai.index_select(0, pos), ai.index_select(0, neg)
        rec = -torch.nn.functional.logsigmoid((u * p).sum(1) - (u * n).sum(1)).mean()
        reg = (Eu[users].norm(2).pow(2) + Ei[pos].norm(2).pow(2) + Ei[neg].norm(2).pow(2)) / (2 * B)
        (rec + 1e-4 * reg).backward()

    print(f"[stock torch ops on the same GPU] {wall(stock, 10):.0f} us/step")
    pkg.check_index_errors()


if __name__ == "__main__":
    main()
