#!/usr/bin/env python3
"""One DCN-Mix / DCNv2 training step at the C3 shape (F=22 Avazu-shaped fields, N=2.02 M rows, D=16 -> d=352, QR
`divider: 2` embedding, E=4, r=64, L=3, MLP 400x3 + BN, B=4096): model(x) -> BCE-with-logits -> backward, launched
eagerly and replayed as ONE hipGraph."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recsys_benchmark_amd as pkg  # noqa: E402
from recsys_benchmark_amd import mlp as _mlp  # noqa: E402
from recsys_benchmark_amd.dcn import DCN_Mix, DCNv2  # noqa: E402
from recsys_benchmark_amd.losses import BCEWithLogitsLoss  # noqa: E402

AVAZU_22 = [241, 8, 8, 3697, 4614, 25, 5481, 329, 31, 381763, 1611748, 6793, 6, 5, 2509, 9, 10, 432, 5, 68, 169, 61]
dev = torch.device("cuda")
B = 4096


def wall(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6


def run(name, model):
    model = model.to(dev).train()
    g = torch.Generator().manual_seed(1)
    x = torch.stack([torch.randint(0, d, (B,), generator=g) for d in AVAZU_22], 1).to(dev)
    y = (torch.rand(B, generator=g) < 0.2).float().to(dev)
    lossf, one = BCEWithLogitsLoss(), torch.ones((), device=dev)

    def step():
        loss = lossf(model(x), y)
        loss.backward(one)
        return loss

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
    print(f"[{name}] eager {t_eager:.0f} us/step, hipGraph replay {t_graph:.0f} us/step = {B / t_graph:.2f} M samples/s "
          f"(loss {float(loss):.4f})")
    # the whole training step (+ zero_grad and the optimizers of get_optimizers) as trainer.GraphedTrainStep replays it
    from recsys_benchmark_amd import optim, trainer
    sparse = bool(getattr(model.embedding, "sparse_grad", False))
    opts = optim.get_optimizers(model, {"optimizer": "adam", "learning_rate": 1e-3, "weight_decay": 1e-6, "sparse": sparse})
    tstep = trainer.GraphedTrainStep(model, opts, lossf)
    t_train = wall(lambda: tstep(x, y))
    print(f"    + optimizers ({'SparseAdam + Adam' if sparse else 'Adam'}), one hipGraph: {t_train:.0f} us/step = "
          f"{B / t_train:.2f} M samples/s{'' if tstep._graph is not None else '  [NOT captured]'}")
    pkg.check_index_errors()


def main():
    _mlp.TUNE_BACKWARD_GEMMS = True
    torch.manual_seed(0)
    emb = {"name": "qr", "divider": 2, "operation": "mult"}
    run("DCN-Mix  QR div2", DCN_Mix(AVAZU_22, 16, [400, 400, 400], num_layers=3, num_experts=4, rank=64,
                                    embedding_config=emb, p_dropout=0.5))
    run("DCN-Mix  vanilla", DCN_Mix(AVAZU_22, 16, [400, 400, 400], num_layers=3, num_experts=4, rank=64,
                                    embedding_config={"name": "vanilla"}, p_dropout=0.5))
    run("DCN-Mix  vanilla, row-form grads", DCN_Mix(AVAZU_22, 16, [400, 400, 400], num_layers=3, num_experts=4, rank=64,
                                                    embedding_config={"name": "vanilla", "sparse": True}, p_dropout=0.5))
    run("DCNv2 stacked   ", DCNv2(AVAZU_22, 16, [400, 400, 400], num_layers=3, embedding_config={"name": "vanilla"}))


if __name__ == "__main__":
    main()
