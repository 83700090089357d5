#!/usr/bin/env python3
"""Minimal counterpart of the reference's CTR training loop (scripts/deepfm/train_deepfm.py:103-182 +
src/trainer/deepfm.py:17-91) on synthetic Criteo-shaped data, using only drop-in pieces:

    model = get_ctr_model(field_dims, cfg["model"])            # recsys_benchmark_amd
    optimizers = get_optimizers(model, cfg)                    # fused row-sparse Adam + dense Adam
    for x, y in batches: loss = BCEWithLogits(model(x), y); zero_grad; backward; step

The synthetic labels follow a planted logistic model over a few fields so the loss visibly falls.
GPU only.   python examples/train_deepfm_synthetic.py [--steps 300] [--batch 4096] [--model deepfm|dcn_mix|dcn]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recsys_benchmark_amd as pkg  # noqa: E402
from recsys_benchmark_amd.optim import get_optimizers  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--model", default="deepfm", choices=["deepfm", "dcn_mix", "dcn"])
    ap.add_argument("--embedding", default="vanilla", choices=["vanilla", "qr", "cerp"])
    a = ap.parse_args()
    dev = torch.device("cuda")
    torch.manual_seed(2023)
    field_dims = [1460, 583, 100000, 22026, 305, 24, 12517, 633, 3, 93145, 5683, 83515, 3194, 27, 14992, 54613, 10, 5652,
                  2173, 4, 70465, 18, 15, 28618, 105, 14257]
    emb_cfg = {"vanilla": {"name": "vanilla", "sparse": True}, "qr": {"name": "qr", "divider": 4},
               "cerp": {"name": "cerp", "bucket_size": 8000}}[a.embedding]
    mcfg = {"name": a.model, "num_factor": 16, "hidden_sizes": [400, 400, 400], "p_dropout": 0.2,
            "embedding_config": emb_cfg}
    if a.model == "deepfm":
        mcfg["use_batchnorm"] = True
    model = pkg.get_ctr_model(field_dims, mcfg).to(dev)
    sparse = a.embedding == "vanilla"
    cfg = {"sparse": sparse, "optimizer": "adam", "learning_rate": 1e-3, "weight_decay": 1e-6}
    optimizers = get_optimizers(model, cfg) if a.model == "deepfm" else [torch.optim.Adam(model.parameters(), lr=1e-3)]
    if a.model != "deepfm" and sparse:
        raise SystemExit("use --embedding qr/cerp (dense grads) with the DCN models in this example")
    # planted model: the label depends on hashed ids of 6 fields
    gen = torch.Generator().manual_seed(7)
    planted = [torch.randn(d, generator=gen).to(dev) for d in field_dims[:6]]
    lossf = torch.nn.BCEWithLogitsLoss()

    def batch():
        x = torch.stack([torch.randint(0, d, (a.batch,), device=dev) for d in field_dims], 1)
        logit = sum(p[x[:, i]] for i, p in enumerate(planted)) * 0.8 - 1.0
        return x, torch.bernoulli(torch.sigmoid(logit))

    model.train()
    t0, hist = None, []
    for step in range(a.steps):
        if step == 20:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        x, y = batch()
        loss = lossf(model(x), y)
        for o in optimizers:
            o.zero_grad()
        loss.backward()
        for o in optimizers:
            o.step()
        if step % 50 == 0 or step == a.steps - 1:
            hist.append(float(loss))
            print(f"step {step:4d}  loss {hist[-1]:.4f}", flush=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (a.steps - 20)
    pkg.check_index_errors()
    print(f"{a.model}/{a.embedding}: {dt*1e3:.2f} ms/step eager (data gen + fwd + bwd + optimizer) = {a.batch/dt/1e6:.2f} M samples/s; "
          f"loss {hist[0]:.4f} -> {hist[-1]:.4f}")
    assert hist[-1] < hist[0], "the loss should fall on the planted data"


if __name__ == "__main__":
    main()
