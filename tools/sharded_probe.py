"""Where does the host time of one sharded step go?  World = 1 RCCL group on one GPU; CPU wall time of
each phase without device syncs (the step is host-bound when these add up to the step time)."""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from recsys_benchmark_amd.losses import BCEWithLogitsLoss  # noqa: E402
from recsys_benchmark_amd.sharded import ShardedDeepFM, _Exchange  # noqa: E402


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29561")
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    dims = bench.CRITEO_KAGGLE_26
    B = 4096
    model = ShardedDeepFM(dims, 16, [400, 400, 400], p_dropout=0.5, use_batchnorm=True, device=dev)
    model.train()
    if "--graphed-step" not in sys.argv:
        model.enable_graphs(B)
    x, y = bench.synth_batch(dims, B, 1, dev)
    lossf = BCEWithLogitsLoss()
    acc = {}

    def tick(name, t0):
        t1 = time.perf_counter()
        acc[name] = acc.get(name, 0.0) + (t1 - t0)
        return t1

    def step(record):
        t = time.perf_counter()
        model.zero_grad(set_to_none=True)
        t = tick("zero_grad", t) if record else t
        recv, slot = _Exchange.apply(x, model.embedding_shard, model.fc_shard, model)
        t = tick("exchange_fwd", t) if record else t
        out = model._graphed_local(recv, slot)
        t = tick("graph_fwd", t) if record else t
        loss = lossf(out, y)
        t = tick("loss_fwd", t) if record else t
        loss.backward()
        t = tick("backward", t) if record else t
        model.allreduce_dense_grads()
        t = tick("allreduce", t) if record else t

    if "--graphed-step" in sys.argv:
        gstep = model.make_graphed_step(lossf, B)

        def step(record):  # noqa: F811
            gstep(x, y)

    for _ in range(20):
        step(False)
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
        step(True)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"host {t_host / n * 1e6:.0f} us/step, with final sync {t_all / n * 1e6:.0f} us/step")
    for k, v in acc.items():
        print(f"  {k:14s} {v / n * 1e6:8.1f} us")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
