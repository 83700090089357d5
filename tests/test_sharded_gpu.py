"""GPU: the row-sharded DeepFM through its default HIP lookup / FM kernels on a 1-rank RCCL
group must reproduce the unsharded product DeepFM (same weights, dropout off)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

from conftest import assert_close

import recsys_benchmark_amd as pkg
from recsys_benchmark_amd.sharded import ShardedDeepFM

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def one_rank_group():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield
    dist.destroy_process_group()


def test_sharded_world1_equals_unsharded(one_rank_group):
    torch.manual_seed(3)
    dims, D, B = [50, 7, 1000, 3], 16, 64
    dev = torch.device("cuda", 0)
    ref = pkg.DeepFM(dims, D, [32, 16], p_dropout=0.0, use_batchnorm=True,
                     embedding_config={"name": "vanilla", "sparse": True}, fc_sparse=True).to(dev)
    sh = ShardedDeepFM(dims, D, [32, 16], p_dropout=0.0, use_batchnorm=True, device=dev)
    sh.load_full_tables(ref.embedding.get_weight().data, ref.fc.weight.data)
    sh._deep_branch.load_state_dict(ref._deep_branch.state_dict())
    with torch.no_grad():
        ref._bias.fill_(0.25)
        sh._bias.fill_(0.25)
    x = torch.stack([torch.randint(0, d, (B,)) for d in dims], 1).to(dev)
    y = (torch.rand(B) < 0.3).float().to(dev)
    lossf = torch.nn.BCEWithLogitsLoss()
    a, b = ref(x), sh(x)
    assert_close(b, a, 1e-5, 1e-6, "logits")
    lossf(a, y).backward()
    lossf(b, y).backward()
    sh.allreduce_dense_grads()
    N = sum(dims)
    gW, g1 = sh.embedding_shard.grad.to_dense(), sh.fc_shard.grad.to_dense()
    assert_close(gW[:N], ref.embedding.get_weight().grad.to_dense(), 1e-5, 1e-7, "table grad")
    assert_close(g1[:N], ref.fc.weight.grad.to_dense(), 1e-5, 1e-7, "first-order grad")
    assert not gW[N].any() and not g1[N].any()                 # the sink row
    assert_close(sh._bias.grad, ref._bias.grad, 1e-5, 1e-7)
    for (k, p), (_, q) in zip(sh._deep_branch.named_parameters(), ref._deep_branch.named_parameters()):
        assert_close(p.grad, q.grad, 1e-4, 1e-6, k)
    pkg.check_index_errors()


@pytest.mark.parametrize("graphed,criterion", [(False, "torch"), (True, "torch"), (True, "package")])
def test_sharded_training_steps_track_the_unsharded_model(one_rank_group, graphed, criterion):
    """Same batches, same optimizers: after several steps the shards hold the unsharded model's tables.  criterion "package":
    recsys_benchmark_amd.BCEWithLogitsLoss — the graphed step then hands labels and its 1 / world seed to the local compute and
    the criterion is evaluated inside the tail's head launch."""
    from recsys_benchmark_amd.optim import get_optimizers

    torch.manual_seed(4)
    dims, D, B = [50, 7, 1000, 3], 16, 64
    dev = torch.device("cuda", 0)
    ref = pkg.DeepFM(dims, D, [32, 16], p_dropout=0.0, use_batchnorm=True,
                     embedding_config={"name": "vanilla", "sparse": True}, fc_sparse=True).to(dev)
    sh = ShardedDeepFM(dims, D, [32, 16], p_dropout=0.0, use_batchnorm=True, device=dev)
    sh.load_full_tables(ref.embedding.get_weight().data, ref.fc.weight.data)
    sh._deep_branch.load_state_dict(ref._deep_branch.state_dict())
    cfg = {"sparse": True, "optimizer": "adam", "learning_rate": 1e-2, "weight_decay": 1e-6}
    ropts, sopts = get_optimizers(ref, cfg), sh.get_optimizers(cfg)
    lossf = torch.nn.BCEWithLogitsLoss() if criterion == "torch" else pkg.BCEWithLogitsLoss()
    gstep = sh.make_graphed_step(lossf, B) if graphed else None
    gen = torch.Generator().manual_seed(5)
    for _ in range(6):
        x = torch.stack([torch.randint(0, d, (B,), generator=gen) for d in dims], 1).to(dev)
        y = (torch.rand(B, generator=gen) < 0.3).float().to(dev)
        for o in ropts:
            o.zero_grad()
        lossf(ref(x), y).backward()
        for o in ropts:
            o.step()
        if graphed:
            gstep(x, y)
        else:
            for o in sopts:
                o.zero_grad()
            lossf(sh(x), y).backward()
            sh.allreduce_dense_grads()
        for o in sopts:
            o.step()
    N = sum(dims)
    # six Adam steps apart: the usual amplification of last-bit differences where a gradient is almost zero
    assert_close(sh.embedding_shard[:N], ref.embedding.get_weight(), 5e-3, 1e-4, "table")
    assert_close(sh.fc_shard[:N], ref.fc.weight, 5e-3, 1e-4, "first-order table")
    assert not sh.embedding_shard[N].any() and not sh.fc_shard[N].any(), "the sink row moved"
    for (k, p), (_, q) in zip(sh._deep_branch.named_parameters(), ref._deep_branch.named_parameters()):
        assert_close(p, q, 5e-3, 1e-4, k)
    pkg.check_index_errors()


def test_direct_communicator_equals_the_process_group_collectives(one_rank_group, monkeypatch):
    """The library's own RCCL communicator (collectives on the compute stream, csrc/comm.hip) against torch.distributed's:
    byte-identical all-to-all and all-reduce results, and the sharded model uses it by default on an RCCL group."""
    from recsys_benchmark_amd.sharded import DirectComm

    dev = torch.device("cuda", 0)
    comm = DirectComm.create(None, dev)
    assert comm is not None, "an RCCL group must get the direct communicator"
    src = torch.randn(1000, 20, device=dev)
    out, ref = torch.empty_like(src), torch.empty_like(src)
    comm.all_to_all(out, src)
    dist.all_to_all_single(ref, src)
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    ids = torch.randint(0, 1 << 40, (4096,), device=dev)
    oi, ri = torch.empty_like(ids), torch.empty_like(ids)
    comm.all_to_all(oi, ids)
    dist.all_to_all_single(ri, ids)
    assert torch.equal(oi, ri)                                   # int64 ids travel as bytes
    a = torch.randn(12345, device=dev)
    b = a.clone()
    comm.all_reduce_sum(a)
    dist.all_reduce(b)
    assert torch.equal(a, b)
    comm.close()
    dims = [50, 7, 1000, 3]
    sh = ShardedDeepFM(dims, 16, [32, 16], p_dropout=0.0, use_batchnorm=True, device=dev)
    assert sh._comm is not None
    monkeypatch.setenv("MI_DIRECT_RCCL", "0")
    sh2 = ShardedDeepFM(dims, 16, [32, 16], p_dropout=0.0, use_batchnorm=True, device=dev)
    assert sh2._comm is None                                     # the switch keeps torch.distributed's collectives
    sh2.load_state_dict(sh.state_dict())
    x = torch.stack([torch.randint(0, d, (64,)) for d in dims], 1).to(dev)
    assert torch.equal(sh(x), sh2(x))


def test_graphed_local_compute_matches_eager():
    """Isolated child process (see tests/_graphed_sharded_check.py for why)."""
    import subprocess
    import sys

    from conftest import ROOT

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_graphed_sharded_check.py"), str(port)],
                       capture_output=True, text=True, timeout=240)
    assert r.returncode == 0 and "GRAPHED_SHARDED_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_two_ranks_sharing_one_gpu_match_the_unsharded_model():
    """world = 2 through the HIP routing kernels end to end (see tests/_two_rank_gpu_check.py)."""
    import subprocess
    import sys

    from conftest import ROOT

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = str(s.getsockname()[1])
    script = os.path.join(ROOT, "tests", "_two_rank_gpu_check.py")
    procs = [subprocess.Popen([sys.executable, script, str(r), port], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(2)]
    outs = [p.communicate(timeout=300) for p in procs]
    if any("GLOO_NO_CUDA_ALLTOALL" in o for o, _ in outs):
        pytest.skip("this gloo build has no all_to_all on device tensors: " + outs[0][0].strip()[:200])
    for r, (p, (o, e)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and "TWO_RANK_OK" in o, f"rank {r}:\n{o[-1500:]}\n{e[-3000:]}"
