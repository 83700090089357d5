"""Child process of tests/test_sharded_gpu.py: graphed local compute of the sharded DeepFM must
equal its eager execution (logits and every gradient).  Run isolated because hipGraph capture in
a process that has already run other autograd work segfaulted in capture_end on this stack."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recsys_benchmark_amd.sharded import ShardedDeepFM  # noqa: E402


def main():
    os.environ["MASTER_ADDR"] = "127.0.0.1"    # never inherit the parent test process's rendezvous
    os.environ["MASTER_PORT"] = sys.argv[1] if len(sys.argv) > 1 else "29547"
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    torch.manual_seed(5)
    dims, D, B = [50, 7, 1000, 3], 16, 64
    eager = ShardedDeepFM(dims, D, [32, 16], p_dropout=0.0, use_batchnorm=True, device=dev)
    graphed = ShardedDeepFM(dims, D, [32, 16], p_dropout=0.0, use_batchnorm=True, device=dev)
    graphed.load_state_dict(eager.state_dict())
    graphed.enable_graphs(B)
    # enable_graphs ran warm-up passes that moved the BatchNorm running stats: re-align them
    graphed.load_state_dict(eager.state_dict())
    x = torch.stack([torch.randint(0, d, (B,)) for d in dims], 1).to(dev)
    y = (torch.rand(B) < 0.3).float().to(dev)
    lossf = torch.nn.BCEWithLogitsLoss()
    a, b = eager(x), graphed(x)
    torch.testing.assert_close(b, a, rtol=1e-5, atol=1e-6)
    lossf(a, y).backward()
    lossf(b, y).backward()
    for (k, p), (_, q) in zip(graphed.named_parameters(), eager.named_parameters()):
        g = p.grad.to_dense() if p.grad.is_sparse else p.grad
        r = q.grad.to_dense() if q.grad.is_sparse else q.grad
        torch.testing.assert_close(g, r, rtol=1e-4, atol=1e-6, msg=lambda m: f"{k}: {m}")
    for (k, p), (_, q) in zip(graphed.named_buffers(), eager.named_buffers()):
        torch.testing.assert_close(p, q, rtol=1e-5, atol=1e-6, msg=lambda m: f"buffer {k}: {m}")
    graphed.check_overflow()

    # the one-graph training step: same loss, same p.grad on every parameter, same buffers, twice in a row
    from recsys_benchmark_amd.losses import BCEWithLogitsLoss

    stepper = ShardedDeepFM(dims, D, [32, 16], p_dropout=0.0, use_batchnorm=True, device=dev)
    stepper.load_state_dict(eager.state_dict())
    step = stepper.make_graphed_step(BCEWithLogitsLoss(), B)
    for it in range(2):
        eager.load_state_dict(stepper.state_dict())        # warm-ups / earlier steps moved the BN running stats
        x = torch.stack([torch.randint(0, d, (B,)) for d in dims], 1).to(dev)
        y = (torch.rand(B) < 0.3).float().to(dev)
        eager.zero_grad(set_to_none=True)
        ref_loss = lossf(eager(x), y)
        ref_loss.backward()
        eager.allreduce_dense_grads()
        loss = step(x, y)
        torch.testing.assert_close(loss.reshape(()), ref_loss.detach().reshape(()), rtol=1e-5, atol=1e-6)
        for (k, p), (_, q) in zip(stepper.named_parameters(), eager.named_parameters()):
            g = p.grad.to_dense() if p.grad.is_sparse else p.grad
            r = q.grad.to_dense() if q.grad.is_sparse else q.grad
            torch.testing.assert_close(g, r, rtol=1e-4, atol=1e-6, msg=lambda m: f"step {it} {k}: {m}")
        for (k, p), (_, q) in zip(stepper.named_buffers(), eager.named_buffers()):
            torch.testing.assert_close(p, q, rtol=1e-5, atol=1e-6, msg=lambda m: f"step {it} buffer {k}: {m}")
    stepper.check_overflow()
    torch.cuda.synchronize()
    print("GRAPHED_SHARDED_OK", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
