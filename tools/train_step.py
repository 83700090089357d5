"""Whole DeepFM training step (zero_grad, forward, BCE-with-logits, backward, optimizer steps) at the headline shape —
what the reference's train_epoch does per batch (src/trainer/deepfm.py:40-60) — under the optimizer configs the
reference ships:

    dense    configs/deepfm/base_config.yaml: one Adam over everything, dense table gradient
    sparse   configs/deepfm/base_config_sparse.yaml: SparseAdam on the table, Adam (weight decay) on the rest
    sparse+  the same with the first-order table's gradient in row form too (DeepFM(fc_sparse=True); extension)

    python tools/train_step.py [--batch 4096] [--iters 30]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import recsys_benchmark_amd as rb  # noqa: E402
from recsys_benchmark_amd import losses, optim, trainer  # noqa: E402
from bench import CRITEO_KAGGLE_26, synth_batch  # noqa: E402


def run(name, model, opts, x, y, iters):
    lossf = losses.BCEWithLogitsLoss()

    def step():
        for o in opts:
            o.zero_grad(set_to_none=True)
        lossf(model(x), y).backward()
        for o in opts:
            o.step()

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    parts = {}
    for label, fn in (("fwd+bwd", lambda: lossf(model(x), y).backward()), ("optimizers", lambda: [o.step() for o in opts])):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(iters):
            if label == "fwd+bwd":
                for o in opts:
                    o.zero_grad(set_to_none=True)
            fn()
        torch.cuda.synchronize()
        parts[label] = (time.perf_counter() - t) / iters * 1e3
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / iters * 1e3
    gstep = trainer.GraphedTrainStep(model, opts, lossf)
    for _ in range(5):
        gstep(x, y)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters):
        gstep(x, y)
    torch.cuda.synchronize()
    gms = (time.perf_counter() - t) / iters * 1e3
    print(f"{name:8s} eager {ms:7.3f} ms/step = {x.shape[0] / ms / 1e3:5.2f} M samples/s (fwd+bwd {parts['fwd+bwd']:.3f}, "
          f"optimizer steps {parts['optimizers']:.3f});  one hipGraph {gms:7.3f} ms/step = {x.shape[0] / gms / 1e3:5.2f} M samples/s"
          f"{'' if gstep._graph is not None else '  [NOT captured]'}")


def eval_forward(x, iters):
    """validate_epoch's forward (eval mode: running statistics, no dropout), eager vs replayed"""
    dims = list(CRITEO_KAGGLE_26)
    torch.manual_seed(0)
    model = rb.DeepFM(dims, 16, [400, 400, 400], p_dropout=0.5, use_batchnorm=True).to(x.device).eval()
    out = {}
    for name, fwd in (("eager", trainer.GraphedForward(model, use_graph=False)), ("hipGraph", trainer.GraphedForward(model))):
        for _ in range(5):
            fwd(x)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(iters):
            fwd(x)
        torch.cuda.synchronize()
        out[name] = (time.perf_counter() - t) / iters * 1e3
    print(f"eval forward: eager {out['eager']:.3f} ms = {x.shape[0] / out['eager'] / 1e3:.1f} M samples/s;  one hipGraph "
          f"{out['hipGraph']:.3f} ms = {x.shape[0] / out['hipGraph'] / 1e3:.1f} M samples/s")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    dims = list(CRITEO_KAGGLE_26)
    x, y = synth_batch(dims, a.batch, 7, dev)
    base = {"optimizer": "adam", "learning_rate": 1e-3, "weight_decay": 1e-6}
    for name, sparse, fc_sparse in (("dense", False, False), ("sparse", True, False), ("sparse+", True, True)):
        if a.only and a.only != name:
            continue
        torch.manual_seed(0)
        model = rb.DeepFM(dims, 16, [400, 400, 400], p_dropout=0.5, use_batchnorm=True,
                          embedding_config={"name": "vanilla", "sparse": sparse}, fc_sparse=fc_sparse).to(dev)
        opts = optim.get_optimizers(model, dict(base, sparse=sparse))
        run(name, model, opts, x, y, a.iters)
        del model, opts
        torch.cuda.empty_cache()
    if not a.only:
        eval_forward(x, a.iters)


if __name__ == "__main__":
    main()
