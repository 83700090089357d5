"""Child of tests/test_sharded_gpu.py: TWO ranks sharing one GPU (gloo carries the collectives, because RCCL
refuses two ranks on one device) run the product's HIP routing / packing / slot kernels with world = 2 —
eager forward/backward and the one-graph step — against the unsharded product DeepFM on the concatenated batch."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import recsys_benchmark_amd as pkg  # noqa: E402
from recsys_benchmark_amd.losses import BCEWithLogitsLoss  # noqa: E402
from recsys_benchmark_amd.sharded import ShardedDeepFM, local_num_rows, shard_rows  # noqa: E402


def main():
    rank, world, port = int(sys.argv[1]), 2, sys.argv[2]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        probe = torch.arange(4, device=dev)
        got = torch.empty_like(probe)
        dist.all_to_all_single(got, probe)
    except Exception as e:  # noqa: BLE001
        print(f"GLOO_NO_CUDA_ALLTOALL {type(e).__name__}: {e}", flush=True)
        dist.destroy_process_group()
        return
    torch.manual_seed(3)                      # same reference model and data on both ranks
    dims, D, B, hidden = [50, 7, 1000, 3, 211], 16, 64, [32, 16]
    ref = pkg.DeepFM(dims, D, hidden, p_dropout=0.0, use_batchnorm=False,
                     embedding_config={"name": "vanilla", "sparse": True}, fc_sparse=True).to(dev)
    x_all = torch.stack([torch.randint(0, d, (B * world,)) for d in dims], 1).to(dev)
    y_all = (torch.rand(B * world) < 0.3).float().to(dev)
    x, y = x_all[rank * B:(rank + 1) * B], y_all[rank * B:(rank + 1) * B]
    N = sum(dims)
    lossf = BCEWithLogitsLoss()
    ref_loss = lossf(ref(x_all), y_all)
    ref_loss.backward()
    gW, g1 = ref.embedding.get_weight().grad.to_dense(), ref.fc.weight.grad.to_dense()

    def check(model, logits, what):
        torch.testing.assert_close(logits, ref(x_all)[rank * B:(rank + 1) * B].detach(), rtol=1e-5, atol=1e-5,
                                   msg=lambda m: f"{what} logits: {m}")
        n = local_num_rows(N, rank, world)
        torch.testing.assert_close(model.embedding_shard.grad.to_dense()[:n], shard_rows(gW, rank, world), rtol=1e-4,
                                   atol=1e-6, msg=lambda m: f"{what} table grad: {m}")
        torch.testing.assert_close(model.fc_shard.grad.to_dense()[:n], shard_rows(g1, rank, world), rtol=1e-4,
                                   atol=1e-6, msg=lambda m: f"{what} first-order grad: {m}")
        for (k, p), (_, q) in zip(model._deep_branch.named_parameters(), ref._deep_branch.named_parameters()):
            torch.testing.assert_close(p.grad, q.grad, rtol=1e-4, atol=1e-6, msg=lambda m: f"{what} {k}: {m}")
        torch.testing.assert_close(model._bias.grad, ref._bias.grad, rtol=1e-4, atol=1e-6)

    def fresh():
        m = ShardedDeepFM(dims, D, hidden, p_dropout=0.0, use_batchnorm=False, device=dev)
        m.load_full_tables(ref.embedding.get_weight().data, ref.fc.weight.data)
        m._deep_branch.load_state_dict(ref._deep_branch.state_dict())
        with torch.no_grad():
            m._bias.copy_(ref._bias)
        return m

    eager = fresh()
    out = eager(x)
    lossf(out, y).backward()
    eager.allreduce_dense_grads()
    check(eager, out.detach(), "eager")
    eager.check_overflow()

    stepper = fresh()
    step = stepper.make_graphed_step(lossf, B)
    for _ in range(2):
        loss = step(x, y)
    # each rank's loss is the mean over ITS half; the reference loss is the mean over both
    both = loss.detach().clone().reshape(1)
    dist.all_reduce(both)
    torch.testing.assert_close(both / world, ref_loss.detach().reshape(1), rtol=1e-5, atol=1e-6)
    check(stepper, stepper(x).detach(), "graphed step")
    stepper.check_overflow()
    # training: four optimizer steps of the sharded model (one-graph step + its optimizers) track the unsharded model
    # stepping on the concatenated batches
    from recsys_benchmark_amd.optim import get_optimizers

    cfg = {"sparse": True, "optimizer": "adam", "learning_rate": 1e-2, "weight_decay": 1e-6}
    trainee = fresh()
    tstep = trainee.make_graphed_step(lossf, B)
    ropts, sopts = get_optimizers(ref, cfg), trainee.get_optimizers(cfg)
    gen = torch.Generator().manual_seed(11)
    for _ in range(4):
        xa = torch.stack([torch.randint(0, d, (B * world,), generator=gen) for d in dims], 1).to(dev)
        ya = (torch.rand(B * world, generator=gen) < 0.3).float().to(dev)
        for o in ropts:
            o.zero_grad()
        lossf(ref(xa), ya).backward()
        for o in ropts:
            o.step()
        tstep(xa[rank * B:(rank + 1) * B], ya[rank * B:(rank + 1) * B])
        for o in sopts:
            o.step()
    n = local_num_rows(N, rank, world)
    torch.testing.assert_close(trainee.embedding_shard[:n], shard_rows(ref.embedding.get_weight().detach(), rank, world),
                               rtol=5e-3, atol=1e-4, msg=lambda m: f"table after training: {m}")
    torch.testing.assert_close(trainee.fc_shard[:n], shard_rows(ref.fc.weight.detach(), rank, world), rtol=5e-3, atol=1e-4,
                               msg=lambda m: f"first-order table after training: {m}")
    for (k, p_), (_, q_) in zip(trainee._deep_branch.named_parameters(), ref._deep_branch.named_parameters()):
        torch.testing.assert_close(p_, q_, rtol=5e-3, atol=1e-4, msg=lambda m: f"{k} after training: {m}")
    trainee.check_overflow()
    pkg.check_index_errors()
    torch.cuda.synchronize()
    print("TWO_RANK_OK", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
