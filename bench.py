#!/usr/bin/env python3
"""bench.py — samples/sec of DeepFM fwd+bwd on the Criteo-26-field shape, B=4096 per GPU.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1], SURVEY.md §8d "C2"): F=26 Criteo-Kaggle categorical
cardinalities (N=33,762,577 rows, D=16 fp32 = 2.16 GB table), MLP 400x3 + BatchNorm +
dropout 0.5 (configs/deepfm/base_config.yaml of the reference), B=4096, seeded synthetic ids
resident in HBM before the timed region.  A step = model(x) -> BCEWithLogits -> backward
(the metric is fwd+bwd; no optimizer step), gradients of the two tables in row (COO) form.

One JSON line on rank 0: the contract fields + `roofline` (the slower of the two gather+FM
kernels, HIP-event timed inside the timed region, algorithmic bytes per SURVEY.md §8d) +
`cpu_baseline` (the oracle's restatement of the reference op sequence on the host cores).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CRITEO_KAGGLE_26 = [1460, 583, 10131227, 2202608, 305, 24, 12517, 633, 3, 93145, 5683, 8351593, 3194,
                    27, 14992, 5461306, 10, 5652, 2173, 4, 7046547, 18, 15, 286181, 105, 142572]
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def alg_bytes_per_sample(F, D):
    """fwd = 12F + 8FD + 4, bwd = 12F + 12FD + 4 (fp32 rows, int64 ids)."""
    return 12 * F + 8 * F * D + 4, 12 * F + 12 * F * D + 4


def synth_batch(dims, B, seed, device):
    gen = torch.Generator().manual_seed(seed)
    x = torch.stack([torch.randint(0, d, (B,), generator=gen) for d in dims], 1)
    y = (torch.rand(B, generator=gen) < 0.25).float()
    return x.to(device), y.to(device)


def measured_copy_ceiling(dev, nbytes=1 << 30, reps=10):
    """What a plain streaming copy reaches on THIS box (SURVEY.md §8d asks for it next to the 8 TB/s spec peak):
    GB/s of a 1 GiB device-to-device copy counting the bytes read plus the bytes written, and of a read-only pass."""
    a = torch.empty(nbytes // 4, dtype=torch.float32, device=dev).normal_()
    b = torch.empty_like(a)
    out = {}
    for name, fn, moved in (("copy_read_plus_write_GBps", lambda: b.copy_(a), 2 * nbytes),
                            ("read_only_GBps", lambda: a.sum(), nbytes)):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out[name] = round(moved * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
    del a, b
    torch.cuda.empty_cache()
    return out


def cpu_baseline(dims, D, hidden, B, p_dropout, budget_s=20.0):
    """Oracle (= reference op sequence in stock PyTorch CPU ops, dense grads like the
    reference's default nn.Embedding) timed on the host cores; bounded sample."""
    from oracle import reference_ops as ro

    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    gen = torch.Generator().manual_seed(2023)
    N = sum(dims)
    bound = (6.0 / (N + D)) ** 0.5
    p = {
        "offsets": ro.field_offsets(dims),
        "embedding._emb_module.weight": (torch.rand(N, D, generator=gen) * 2 - 1) * bound,
        "fc.weight": torch.randn(N, 1, generator=gen),
        "_bias": torch.zeros(1),
    }
    inp = len(dims) * D
    i = 0
    for h in hidden:
        lin = torch.nn.Linear(inp, h)
        p[f"_deep_branch.{i}.weight"], p[f"_deep_branch.{i}.bias"] = lin.weight.detach(), lin.bias.detach()
        p[f"_deep_branch.{i+1}.weight"], p[f"_deep_branch.{i+1}.bias"] = torch.ones(h), torch.zeros(h)
        p[f"_deep_branch.{i+1}.running_mean"], p[f"_deep_branch.{i+1}.running_var"] = torch.zeros(h), torch.ones(h)
        inp = h
        i += 4
    lin = torch.nn.Linear(inp, 1)
    p[f"_deep_branch.{i}.weight"], p[f"_deep_branch.{i}.bias"] = lin.weight.detach(), lin.bias.detach()
    for k, v in p.items():
        if v.is_floating_point() and "running_" not in k:
            v.requires_grad_(True)
    x, y = synth_batch(dims, B, 2023, "cpu")
    lossf = torch.nn.BCEWithLogitsLoss()

    def step():
        for v in p.values():
            v.grad = None
        lossf(ro.deepfm_forward(x, p, len(hidden), True, True, p_dropout=p_dropout), y).backward()

    step()  # warm-up
    times = []
    t_end = time.perf_counter() + budget_s
    while len(times) < 10 and (time.perf_counter() < t_end or len(times) < 2):
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    return {"value": B / med, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"{len(times)} fwd+bwd steps of the same B={B} workload (median {med*1e3:.1f} ms/step), "
                      "oracle/reference_ops.py deepfm_forward on torch CPU, dense weight.grad as the reference"}


def self_launch(n):
    """Parent of an N-rank run: start `torch.distributed.run` as a CHILD process (never exec: see the GPU-box rules), one
    rank per device, rendezvous on 127.0.0.1; forward the ranks' stderr, print exactly rank 0's JSON line on stdout and
    return the launcher's exit code (non-zero when any rank failed or no line was produced)."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for out in proc.stdout:
        if out.startswith('{"metric"'):
            line = out.strip()
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if rc == 0 and line is None:
        print("bench: the ranks exited cleanly but rank 0 printed no JSON line", file=sys.stderr)
        rc = 1
    if line is not None and rc == 0:
        print(line, flush=True)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=4096, help="per-GPU batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dense-grads", action="store_true", help="reference-style dense weight.grad (atomic scatter)")
    ap.add_argument("--ring", type=int, default=16, help="distinct pre-generated id batches rotated through, one per step")
    ap.add_argument("--sharded", action="store_true", help="use the row-sharded model even on 1 GPU (exercises the N>1 path)")
    ap.add_argument("--no-graph", action="store_true", help="launch every step eagerly instead of replaying a hipGraph")
    ap.add_argument("--c4", action="store_true", help="BASELINE config 4: the same 26 fields with the largest one scaled so that "
                    "the table has 1e9 rows (64 GB fp32); row-sharded model (implies --sharded)")
    ap.add_argument("--no-gemm-tuning", action="store_true", help="leave the MLP's backward GEMMs on PyTorch's default hipBLASLt heuristic")
    ap.add_argument("--fused-tail", action="store_true", help="run the MLP tail on the fused MFMA kernels of csrc/tail.hip "
                    "(deterministic; slower than the library GEMMs at this shape, see DESIGN.md)")
    ap.add_argument("--dry-launch", action="store_true", help="launch-contract check without a GPU: the ranks join a gloo "
                    "all-reduce and rank 0 prints a JSON line with no measurement in it (tests/test_bench_launch.py)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` with no launcher around it: this process has made no GPU call yet, so it
        # only starts N fresh rank processes (one per device), relays rank 0's JSON line and exits with their code
        raise SystemExit(self_launch(args.gpus))

    # RCCL prints a version banner on stdout at communicator creation; the contract is ONE JSON line
    # on stdout, so everything before the final print goes to stderr.
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.dry_launch:
        import torch.distributed as dist

        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.ones(1)
        dist.all_reduce(t)
        dist.destroy_process_group()
        if os.environ.get("MI_BENCH_DRY_FAIL_RANK") == str(rank):
            raise SystemExit(3)
        if rank == 0:
            os.dup2(real_stdout, 1)
            print(json.dumps({"metric": "dry-launch (no measurement)", "value": None, "n_gpus": world,
                              "joined_ranks": int(t.item())}), flush=True)
        return
    import torch.distributed as dist

    # Rehearsal knobs (NOT a measurement mode): MI_BENCH_REHEARSE=1 runs N ranks on ONE GPU with gloo carrying the
    # collectives (RCCL refuses two ranks on a device), to exercise the N>1 launch contract on a 1-GPU box.
    rehearse = os.environ.get("MI_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    sharded = world > 1 or args.sharded or args.c4
    if sharded:
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        # how many ranks the collective backend really joins: every rank adds 1 through an all-reduce
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        torch.cuda.synchronize()
        collective_ranks = int(ones.item())
        if collective_ranks != world:
            raise SystemExit(f"all-reduce over {dist.get_backend()} joined {collective_ranks} ranks, expected {world}")

    import recsys_benchmark_amd as pkg
    from recsys_benchmark_amd import mlp as _mlp
    from recsys_benchmark_amd.profiling import KernelTimer

    # the MLP's two backward GEMMs per layer: let PyTorch pick the fastest rocBLAS/hipBLASLt solution per shape
    # (searched once, during the warm-up steps)
    _mlp.TUNE_BACKWARD_GEMMS = not args.no_gemm_tuning
    _mlp.FUSED_TAIL = bool(args.fused_tail)

    dims, D, hidden, p_drop = list(CRITEO_KAGGLE_26), 16, [400, 400, 400], 0.5
    if args.c4:
        big = max(range(len(dims)), key=lambda i: dims[i])
        dims[big] += 1_000_000_000 - sum(dims)
    F, B = len(dims), args.batch
    torch.manual_seed(2023)
    sparse = not args.dense_grads
    emb_cfg = {"name": "vanilla", "sparse": True} if sparse else {"name": "vanilla"}
    if sharded:
        from recsys_benchmark_amd.sharded import ShardedDeepFM

        model = ShardedDeepFM(dims, D, hidden, p_dropout=p_drop, use_batchnorm=True, device=dev)
        parallelism = f"table row-sharded x{world} (RCCL all-to-all) + dp{world} MLP" + (" [REHEARSAL: one GPU, gloo]" if rehearse else "")
    else:
        model = pkg.DeepFM(dims, D, hidden, p_dropout=p_drop, use_batchnorm=True,
                           embedding_config=emb_cfg, fc_sparse=sparse).to(dev)
        parallelism = "single"
    model.train()
    # A ring of distinct id batches, all resident in HBM before the timed region; every step copies
    # the next one into the static input (868 KB device-to-device, part of the timed step) so the
    # gathers see fresh rows like a real epoch instead of re-reading Infinity-Cache-resident ones.
    # ids and labels of a batch travel as ONE blob (int64 ids, then fp32 labels): one device copy per step
    nx = B * F * 8

    def blob_of(xb, yb):
        b = torch.empty(nx + B * 4, dtype=torch.uint8, device=dev)
        b[:nx].view(torch.int64).view(B, F).copy_(xb)
        b[nx:].view(torch.float32).copy_(yb)
        return b

    ring = [blob_of(*synth_batch(dims, B, 2023 + 7919 * rank + 104729 * i, dev)) for i in range(max(1, args.ring))]
    cur = ring[0].clone()
    x, y = cur[:nx].view(torch.int64).view(B, F), cur[nx:].view(torch.float32)
    from recsys_benchmark_amd.losses import BCEWithLogitsLoss

    lossf = BCEWithLogitsLoss()      # same criterion as the reference trainer, one launch each way
    state = {"i": 0}

    def next_batch():
        cur.copy_(ring[state["i"] % len(ring)])
        state["i"] += 1

    def eager_step():
        next_batch()
        model.zero_grad(set_to_none=True)
        loss = lossf(model(x), y)
        loss.backward()
        if sharded:
            model.allreduce_dense_grads()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # The step is ~60 short launches: replay it as ONE hipGraph (the captured work is the
    # identical kernel sequence; gradients land in the graph's static buffers each replay).
    # The sharded step is NOT captured: recording RCCL collectives into a hipGraph hung on this
    # stack (even with one rank), so N>1 launches eagerly — without any per-step host sync
    # (fixed-capacity all-to-all buckets).
    use_graph = not args.no_graph and not sharded
    step = eager_step
    graphed_local = False
    if sharded and not args.no_graph:
        # everything between the collectives (gather+FM+MLP forward, criterion, whole backward) as ONE hipGraph
        try:
            graphed_step = model.make_graphed_step(lossf, B, static_labels=y)   # y is refreshed by the batch copy
            graphed_local = True

            def step():
                next_batch()
                graphed_step(x)
        except Exception as e:  # noqa: BLE001 - keep the run alive: the eager step is always valid
            print(f"[bench] rank {rank}: capturing the local compute failed ({type(e).__name__}: {e}); eager",
                  file=sys.stderr, flush=True)
    if use_graph:
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(3):
                eager_step()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        model.zero_grad(set_to_none=True)
        from recsys_benchmark_amd.losses import unit_scalar

        one = unit_scalar(dev)      # d(loss)/d(loss) = 1 from a resident scalar: no fill per step, and the fused criterion
        #                             hands back the gradient its forward already wrote
        with torch.cuda.graph(graph):
            lossf(model(x), y).backward(one)

        def step():
            next_batch()
            graph.replay()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    pkg.check_index_errors()
    if sharded:
        model.check_overflow()
    # the timed steps really produced gradients (a replayed graph reading a freed seed tensor would give zeros)
    # (weights only: a Linear bias in front of a training-mode BatchNorm has an exactly zero gradient)
    probe = [model._bias.grad] + [p.grad for p in model._deep_branch.parameters() if p.grad is not None and p.dim() == 2][:2]
    for gprobe in probe:
        if gprobe is None or not bool(torch.isfinite(gprobe).all()) or float(gprobe.abs().sum()) == 0.0:
            raise SystemExit("bench: a gradient of the timed steps is missing, zero or non-finite")

    # roofline leg: the same step launched eagerly, every library kernel timed by its own dispatch
    # begin/end events (a graph replay cannot carry per-kernel events: probed, they are not stamped).
    # Eager launches read ~15-20 % slower than the same kernels inside the replayed graph
    # (rocprofv3, profiles/): the reported fraction is conservative.
    n_prof = min(args.steps, 100)
    with KernelTimer(capacity=32 * n_prof + 64) as kt:
        for _ in range(n_prof):
            eager_step()
        torch.cuda.synchronize()
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        ks = kt.summary()
        fb, bb = alg_bytes_per_sample(F, D)
        alg = {"gather_fm_fwd": fb * B, "gather_fm_bwd_rows": bb * B, "gather_fm_bwd_dense": bb * B}
        if sharded:  # same kernels over the exchanged packed rows, addressed by slot: same bytes per sample
            alg = {"slot_fm_fwd": fb * B, "slot_fm_bwd": bb * B}
        kernels = {}
        for k, s in ks.items():
            e = {"avg_us": round(s["avg_us"], 3), "min_us": round(s["min_us"], 3), "launches": s["count"]}
            if k in alg:
                e["alg_bytes"] = alg[k]
                e["GBps"] = round(alg[k] / (s["avg_us"] * 1e-6) / 1e9, 1)
            kernels[k] = e
        cand = [k for k in kernels if k in alg]
        dom = max(cand, key=lambda k: kernels[k]["avg_us"]) if cand else None
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if dom and os.path.exists(tpath):
            traffic = json.load(open(tpath)).get(dom)
        roofline = None
        if dom:
            ach = kernels[dom]["GBps"]
            pair_us = sum(kernels[k]["avg_us"] for k in cand)
            roofline = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                        "avg_us": kernels[dom]["avg_us"],
                        "fwd_bwd_pair": {"us": round(pair_us, 3),
                                         "GBps": round((fb + bb) * B / (pair_us * 1e-6) / 1e9, 1),
                                         "frac": round((fb + bb) * B / (pair_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}}
        if roofline is not None and not args.c4:
            roofline["measured_stream_ceiling"] = measured_copy_ceiling(dev)
        out = {
            "metric": "samples/sec fwd+bwd, Criteo-26field DeepFM b=4096; HBM GB/s vs roofline",
            "value": round(B * world * args.steps / elapsed, 1),
            "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "rccl_ranks": (collective_ranks if (sharded and not rehearse) else None),
            "collective_backend": (dist.get_backend() if sharded else None),
            "launch": "hipGraph replay" if use_graph else ("eager RCCL collectives + one hipGraph for the local compute"
                                                           if (sharded and graphed_local) else "eager"),
            "config": {"workload": f"{'C4' if args.c4 else 'C2'} DeepFM Criteo-26field full embedding: F={F}, D={D}, N={sum(dims)} rows, "
                                   f"MLP 400x3+BN+dropout0.5, B={B}/GPU, fwd+bwd, {len(ring)} distinct uniform-id batches "
                                   f"rotated (fresh ids every step), "
                                   f"{'row-form (COO)' if sparse else 'dense'} table grads",
                       "global_batch": B * world, "parallelism": parallelism},
            "roofline": roofline,
            "roofline_method": "dispatch begin/end HIP events (hipExtLaunchKernelGGL) on every library launch of an eager pass",
            "kernels": kernels,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(dims, D, hidden, B, p_drop)
        elif world == 1:
            out["cpu_baseline"] = None
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
    if sharded:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
