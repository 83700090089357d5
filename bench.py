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


def synth_batch(dims, B, seed, device, ids="uniform"):
    """ids per field (i) uniform or (ii) Zipf(alpha = 1.05) over the field's vocabulary, raw (pre-offset)
    int64 [B, F]; labels Bernoulli(0.25); torch.Generator().manual_seed(seed)."""
    gen = torch.Generator().manual_seed(seed)
    if ids == "zipf":
        x = torch.stack([zipf_ids(d, B, gen) for d in dims], 1)
    else:
        x = torch.stack([torch.randint(0, d, (B,), generator=gen) for d in dims], 1)
    y = (torch.rand(B, generator=gen) < 0.25).float()
    return x.to(device), y.to(device)


ZIPF_ALPHA = 1.05
_zipf_cdf = {}


def zipf_ids(d, B, gen):
    """B draws of P(rank k) ~ 1 / (k + 1)^1.05 over k in [0, d) by inverting the CDF (float64), the rank mapped to an id by
    a fixed multiplicative scramble so that a field's hot rows are not one contiguous run of the table (ranks that
    collide under it simply share an id: the skew is what matters)."""
    cdf = _zipf_cdf.get(d)
    if cdf is None:
        w = torch.arange(1, d + 1, dtype=torch.float64).pow_(-ZIPF_ALPHA)
        cdf = _zipf_cdf[d] = torch.cumsum(w, 0).div_(float(w.sum()))
    rank = torch.searchsorted(cdf, torch.rand(B, generator=gen, dtype=torch.float64)).clamp_(max=d - 1)
    return (rank * 1000003 + 12345) % d


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


def physical_cores():
    """(physical cores, logical cpus) of the host; the CPU legs run on the physical count (SMT siblings only add
    scheduling noise to a memory-bound embedding step)."""
    logical = os.cpu_count() or 1
    try:
        import psutil

        phys = psutil.cpu_count(logical=False) or logical
    except Exception:  # noqa: BLE001
        phys = logical
    return int(phys), int(logical)


def timed_cpu(step, warm=3, timed=10, budget_s=30.0):
    """SURVEY.md §8d protocol: `warm` untimed + `timed` timed iterations, median; stops early (never below 2 timed
    iterations) when the budget is spent and says how many it got."""
    t_end = time.perf_counter() + budget_s
    for _ in range(warm):
        step()
        if time.perf_counter() > t_end:
            break
    times = []
    while len(times) < timed and (time.perf_counter() < t_end or len(times) < 2):
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
    times.sort()
    return times[len(times) // 2], len(times)


def deepfm_cpu_params(dims, D, hidden, gen):
    from oracle import reference_ops as ro

    N = sum(dims)
    bound = (6.0 / (N + D)) ** 0.5
    p = {
        "offsets": ro.field_offsets(dims),
        "embedding._emb_module.weight": (torch.rand(N, D, generator=gen) * 2 - 1) * bound,
        "fc.weight": torch.randn(N, 1, generator=gen),
        "_bias": torch.zeros(1),
    }
    inp, i = len(dims) * D, 0
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
    return p


def cpu_baseline(dims, D, hidden, B, p_dropout, budget_s=25.0):
    """The oracle (= the reference's op sequence in stock PyTorch CPU ops, oracle/reference_ops.py) timed on this box's
    host cores, 3 warm-up + 10 timed iterations, median, threads = physical cores.

    Headline (`value`): the SAME workload as the GPU line, like for like — fwd+bwd with row-form (sparse=True) table
    gradients, the reference's configs/deepfm/base_config_sparse.yaml:8-10 path — so the two numbers compare the same
    work.  `lines` adds what the reference's default training script really executes (src/models/deepfm.py:155-219): dense
    weight.grad, and dense Adam(weight_decay=1e-6) over every parameter; both at the survey's F=39 / N~1M shape where 13
    iterations fit the time budget, and the dense-gradient step at the full C2 table with as many iterations as the
    budget allows (its 2.16 GB gradient makes one step take seconds)."""
    from oracle import reference_ops as ro

    phys, logical = physical_cores()
    torch.set_num_threads(phys)
    lossf = torch.nn.BCEWithLogitsLoss()

    def make_step(dims_, sparse, adam):
        gen = torch.Generator().manual_seed(2023)
        p = deepfm_cpu_params(dims_, D, hidden, gen)
        x, y = synth_batch(dims_, B, 2023, "cpu")
        opt = torch.optim.Adam([v for v in p.values() if v.requires_grad], lr=1e-3, weight_decay=1e-6) if adam else None

        def fwd():      # sparse: nn.Embedding(sparse=True) / EmbeddingBag(sparse=True) gradients, same op sequence
            return ro.deepfm_forward(x, p, len(hidden), True, True, p_dropout=p_dropout, sparse=sparse, fc_sparse=sparse)

        def step():
            for v in p.values():
                v.grad = None
            lossf(fwd(), y).backward()
            if opt is not None:
                opt.step()

        return step

    lines = []

    def line(name, dims_, sparse, adam, budget):
        med, n = timed_cpu(make_step(dims_, sparse, adam), budget_s=budget)
        lines.append({"what": name, "samples_per_s": round(B / med, 1), "ms_per_step": round(med * 1e3, 2), "timed_iters": n,
                      "shape": f"F={len(dims_)}, N={sum(dims_)}, D={D}, B={B}"})
        return med, n

    med, n = line("fwd+bwd, row-form (sparse=True) table grads — like for like with the GPU line", dims, True, False, budget_s * 0.35)
    survey39 = [50] * 13 + [max(2, d * 1_000_000 // sum(dims)) for d in dims]          # F=39, N~1M (SURVEY.md §6 shape)
    line("fwd+bwd, dense weight.grad (reference default), F=39 N~1M", survey39, False, False, budget_s * 0.2)
    line("fwd+bwd + dense Adam(weight_decay=1e-6) (what train_deepfm.py runs), F=39 N~1M", survey39, False, True, budget_s * 0.2)
    line("fwd+bwd, dense weight.grad (reference default), full C2 table", dims, False, False, budget_s * 0.25)
    return {"value": round(B / med, 1), "unit": "samples/s", "cores": phys, "logical_cpus": logical, "kind": "port",
            "sample": f"3 warm-up + {n} timed fwd+bwd steps (median {med*1e3:.1f} ms/step) of the same B={B} C2 workload with "
                      "row-form table gradients, oracle/reference_ops.py op sequence on torch CPU, "
                      f"torch.set_num_threads({phys}) = physical cores",
            "lines": lines}


class GatherBench:
    """The two gather+FM kernels of a step launched straight through the C-ABI on the model's own tables (whatever their
    layout: the row strides are read off the parameters), so that they can be replayed alone inside hipGraphs.

    The activations rotate through `sets` buffer sets (emb, g_emb, row gradients) totalling more than the 256 MiB Infinity
    Cache, and callers rotate through more id batches than it can hold the rows of: every launch then finds its inputs
    where a training step finds them — in HBM — instead of in whatever the previous launch of the same graph left in the
    caches (with one buffer set and 16 id batches the forward reads 20 % and the backward 35 % faster than inside a step)."""

    def __init__(self, model, B, F, D, dev, sets=None):
        from recsys_benchmark_amd import _lib as L

        self.L, self.lib = L, L.load()
        self.W, self.w1 = model.embedding.get_weight().detach(), model.fc.weight.detach()
        self.ldw = int(self.W.stride(0)) if self.W.shape[0] > 1 else D
        self.ldw1 = int(self.w1.stride(0)) if self.w1.shape[0] > 1 else 1
        self.bias, self.off = model._bias.detach(), model.offsets.reshape(-1).contiguous()
        self.N, self.B, self.F, self.D, self.dev = self.W.shape[0], B, F, D, dev
        set_bytes = 3 * B * F * D * 4
        self.sets = sets or max(2, min(16, -(-(320 << 20) // set_bytes)))
        self.emb = [torch.empty(B, F, D, device=dev) for _ in range(self.sets)]
        self.gemb = [torch.randn(B, F, D, device=dev) for _ in range(self.sets)]
        self.gvals = [torch.empty(B, F, D, device=dev) for _ in range(self.sets)]
        self.yfm, self.gy = torch.empty(B, device=dev), torch.randn(B, device=dev)
        self.rows = torch.empty(B, F, dtype=torch.int64, device=dev)
        self.g1, self.gb = torch.empty(B, F, device=dev), torch.empty(1, device=dev)
        self.err = L.err_word(dev)

    def fwd(self, x, i=0):
        L, B, F, D = self.L, self.B, self.F, self.D
        L.check(self.lib.mi_gather_fm_fwd_ld(x.data_ptr(), self.off.data_ptr(), self.W.data_ptr(), self.ldw, self.w1.data_ptr(),
                                             self.ldw1, self.bias.data_ptr(), self.emb[i % self.sets].data_ptr(), self.yfm.data_ptr(),
                                             self.rows.data_ptr(), B, F, D, self.N, self.err.data_ptr(), L.stream_ptr(self.dev)), "fwd")

    def bwd(self, i=0):
        L, B, F, D = self.L, self.B, self.F, self.D
        k = i % self.sets
        L.check(self.lib.mi_gather_fm_bwd_rows(self.emb[k].data_ptr(), self.gy.data_ptr(), self.gemb[k].data_ptr(),
                                               self.gvals[k].data_ptr(), self.g1.data_ptr(), self.gb.data_ptr(), B, F, D,
                                               L.stream_ptr(self.dev)), "bwd")


class DgradBench:
    """The tail's FIRST input-gradient product (dz[B, H] . W1[H, F*D]) launched through the C-ABI in its two forms: plain
    (OUT = da, what round 3 ran in front of mi_gather_fm_bwd_rows) and with the lookup's backward in its epilogue
    (mi_tail_dgrad_gemm_fm: reads the saved rows + their per-sample sums, writes the table's row-form gradient).  The
    difference of their in-graph wall times is what the fused epilogue costs a step.  Saved rows and outputs rotate through
    `sets` buffer sets larger than the Infinity Cache (the epilogue then reads the rows from HBM: a step's own rows are
    ~250 us old and may still be cache-resident, so this is the pessimistic side)."""

    def __init__(self, B, F, D, H, dev, sets=None):
        from recsys_benchmark_amd import _lib as L

        self.L, self.lib, self.B, self.F, self.D, self.H, self.dev = L, L.load(), B, F, D, H, dev
        K = F * D
        set_bytes = 2 * B * K * 4
        self.sets = sets or max(2, min(24, -(-(320 << 20) // set_bytes)))
        g = torch.Generator().manual_seed(11)
        self.emb = [torch.randn(B, K, generator=g).to(dev) for _ in range(self.sets)]
        self.out = [torch.empty(B, K, device=dev) for _ in range(self.sets)]
        self.esum = torch.randn(B, D, generator=g).to(dev)
        self.gy, self.g1 = torch.randn(B, generator=g).to(dev), torch.empty(B, F, device=dev)
        self.DY = [torch.randn(B, H, generator=g).to(dev) for _ in range(2)]
        self.Z = [torch.randn(B, H, generator=g).to(dev) for _ in range(2)]
        self.W = (torch.randn(H, K, generator=g) * 0.05).to(dev)
        self.c = [torch.randn(H, generator=g).to(dev) * 0.1 for _ in range(4)]      # mu, al, bz, de

    def run(self, i, fm):
        L, B, H, K = self.L, self.B, self.H, self.F * self.D
        k, j = i % self.sets, i % 2
        mu, al, bz, de = self.c
        if fm:
            L.check(self.lib.mi_tail_dgrad_gemm_fm(self.DY[j].data_ptr(), self.Z[j].data_ptr(), H, mu.data_ptr(), al.data_ptr(),
                                                   bz.data_ptr(), de.data_ptr(), self.W.data_ptr(), K, self.out[k].data_ptr(), None,
                                                   B, H, K, None, self.emb[k].data_ptr(), self.esum.data_ptr(), self.gy.data_ptr(),
                                                   self.g1.data_ptr(), self.D, L.stream_ptr(self.dev)), "dgrad_fm")
        else:
            L.check(self.lib.mi_tail_dgrad_gemm_s(self.DY[j].data_ptr(), self.Z[j].data_ptr(), H, mu.data_ptr(), al.data_ptr(),
                                                  bz.data_ptr(), de.data_ptr(), self.W.data_ptr(), K, None, K, None, None, None,
                                                  0.0, None, self.out[k].data_ptr(), K, None, 0, None, B, H, K, None,
                                                  L.stream_ptr(self.dev)), "dgrad")


def dgrad_epilogue_us(B, F, D, H, dev, copies=32, reps=20):
    """(plain, with the lookup backward in the epilogue) in-graph wall microseconds of the first input-gradient product."""
    db = DgradBench(B, F, D, H, dev)
    plain = graph_wall_us(lambda i: db.run(i, False), copies, reps, dev)
    fm = graph_wall_us(lambda i: db.run(i, True), copies, reps, dev)
    # interleaved once more: the two forms alternate inside ONE graph (same clocks, same cache state), halves attributed
    both = graph_wall_us(lambda i: (db.run(2 * i, False), db.run(2 * i + 1, True)), copies // 2, reps, dev)
    del db
    torch.cuda.empty_cache()
    return {"plain": plain, "fm": fm, "alternating_pair": both}


def graph_wall_us(enqueue, copies, reps, dev):
    """WALL microseconds per enqueue(i) inside a replayed hipGraph holding `copies` of them back to back (HIP events on the
    stream the graph is launched on, around `reps` replays)."""
    from recsys_benchmark_amd import sharded as _sh

    enqueue(0)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    _sh.note_capture(+1)          # (a sharded run's communicator watchdog skips its event polls while a capture is open)
    try:
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            for i in range(copies):
                enqueue(i)
    finally:
        _sh.note_capture(-1)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    # the MEDIAN of five such measurements: differences of two of these numbers (the epilogue's cost, the prefetched gather)
    # are a microsecond or two, and a single measurement is now and then a microsecond off
    samples = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        samples.append(e0.elapsed_time(e1) * 1e3 / (copies * reps))
    samples.sort()
    return samples[len(samples) // 2]


def gather_in_graph_us(model, xs, B, F, D, dev, copies=32, reps=20):
    """In-graph WALL time of the gather+FM forward, of the backward and of the pair: three graphs of `copies` launches
    each, every launch on ids, rows and activations that are NOT cache-resident (GatherBench).  `warm`: the same with one
    buffer set and the first 16 id batches only — the back-to-back figure rounds 1-2 quoted, cache-assisted."""
    gb = GatherBench(model, B, F, D, dev)
    for k in range(gb.sets):          # the backward's saved activations exist before the backward-only graph reads them
        gb.fwd(xs[k % len(xs)], k)
    fwd = graph_wall_us(lambda i: gb.fwd(xs[i % len(xs)], i), copies, reps, dev)
    bwd = graph_wall_us(lambda i: gb.bwd(i), copies, reps, dev)
    half = gb.sets // 2

    def pair(i):                      # the backward of the sample batch whose forward ran sets/2 pairs earlier
        gb.fwd(xs[i % len(xs)], i)
        gb.bwd(i + half)
    out = {"fwd": fwd, "bwd": bwd, "pair": graph_wall_us(pair, copies, reps, dev)}
    del gb
    warm = GatherBench(model, B, F, D, dev, sets=1)
    xw = xs[:16]
    out["warm"] = {"fwd": graph_wall_us(lambda i: warm.fwd(xw[i % len(xw)]), copies, reps, dev),
                   "bwd": graph_wall_us(lambda i: warm.bwd(), copies, reps, dev)}
    return out


def prefetched_gather_us(model, xs, B, F, D, hidden, dev, copies=32, reps=20):
    """The gather + FM forward in the regime a step with DeepFM.prefetch_next runs it in: the PREVIOUS launch was the tail's
    weight-gradient launch (three products dz^T a, MFMA-bound) carrying, in extra workgroups, the touch of exactly the table rows
    this gather reads.  In-graph wall per launch of
        P = [weight gradients + riders for batch i]                     Q = [the same, then gather(batch i)]
        R = [weight gradients alone]
    gather after a prefetch = Q - P, what the riders cost the launch that carries them = P - R."""
    from recsys_benchmark_amd import _kernels as K_

    gb = GatherBench(model, B, F, D, dev)
    g = torch.Generator().manual_seed(17)
    widths = [F * D] + list(hidden)
    dz = [torch.randn(B, n, generator=g).to(dev) for n in widths[1:]]
    act = [torch.randn(B, k, generator=g).to(dev) for k in widths[:-1]]
    dW = [torch.zeros(n, k, device=dev) for k, n in zip(widths[:-1], widths[1:])]
    probs = [dict(A=dz[l], B=act[l], C=dW[l], M=widths[l + 1], N=widths[l], K=B, lda=widths[l + 1], ldb=widths[l], ldc=widths[l])
             for l in range(len(dW))]

    def job(i):
        x = xs[i % len(xs)]
        return K_.PrefetchRowsJob(x.data_ptr(), gb.off.data_ptr(), gb.W.data_ptr(), gb.w1.data_ptr(), gb.ldw, gb.ldw1, B, gb.N, F)

    jobs = [job(i) for i in range(len(xs))]

    def wgrad(i, ride):
        K_.gemm_multi(probs, transA=True, ride=jobs[i % len(jobs)] if ride else None)

    R = graph_wall_us(lambda i: wgrad(i, False), copies, reps, dev)
    P = graph_wall_us(lambda i: wgrad(i, True), copies, reps, dev)
    Q = graph_wall_us(lambda i: (wgrad(i, True), gb.fwd(xs[i % len(xs)], i)), copies, reps, dev)
    C = graph_wall_us(lambda i: (wgrad(i, False), gb.fwd(xs[i % len(xs)], i)), copies, reps, dev)
    del gb
    torch.cuda.empty_cache()
    return {"wgrad": R, "wgrad_with_riders": P, "riders_cost": max(P - R, 0.0), "gather_after_prefetch": Q - P,
            "gather_cold_behind_wgrad": C - R}


def batch_sweep(model, dims, F, D, dev, ids, sizes=(4096, 16384, 65536, 262144)):
    """SURVEY.md §7 'also report a bandwidth-saturating batch': the same two kernels at growing B (fresh ids per launch),
    in-graph wall per kernel, algorithmic GB/s and fraction of the 8 TB/s peak."""
    out = []
    for B in sizes:
        nb = max(2, min(64, -(-(320 << 20) // (B * F * 128))))        # id batches whose rows exceed the Infinity Cache
        xs = [synth_batch(dims, B, 99 + 13 * i + B, dev, ids)[0] for i in range(nb)]
        copies = max(nb, min(32, (1 << 20) // B))
        g = gather_in_graph_us(model, xs, B, F, D, dev, copies=copies, reps=8)
        fwd, bwd, pair = g["fwd"], g["bwd"], g["pair"]
        fb, bb = alg_bytes_per_sample(F, D)
        out.append({"B": B, "fwd_us": round(fwd, 2), "bwd_us": round(bwd, 2), "pair_us": round(pair, 2),
                    "fwd_GBps": round(fb * B / fwd / 1e3, 1), "bwd_GBps": round(bb * B / bwd / 1e3, 1),
                    "fwd_frac": round(fb * B / fwd / 1e3 / HBM_PEAK_GBS, 4), "bwd_frac": round(bb * B / bwd / 1e3 / HBM_PEAK_GBS, 4),
                    "pair_frac": round((fb + bb) * B / pair / 1e3 / HBM_PEAK_GBS, 4)})
        del xs
    torch.cuda.empty_cache()
    return out


def train_step_lines(dims, D, hidden, p_drop, B, dev, ids, layout, iters=20):
    """What train_deepfm.py really runs per batch (src/trainer/deepfm.py:40-60: forward, loss, zero_grad, backward,
    optimizer steps; optimizers as src/models/deepfm.py:155-219 builds them), as ONE replayed hipGraph, under the reference's
    two optimizer configs + the row-form extension — the GPU counterpart of cpu_baseline.lines[2]."""
    import recsys_benchmark_amd as pkg
    from recsys_benchmark_amd import optim, trainer
    from recsys_benchmark_amd.losses import BCEWithLogitsLoss

    x, y = synth_batch(dims, B, 7, dev, ids)
    base = {"optimizer": "adam", "learning_rate": 1e-3, "weight_decay": 1e-6}
    out = []
    for name, sparse, fc_sparse, what in (
            ("configs/deepfm/base_config_sparse.yaml", True, False, "SparseAdam on the embedding table, Adam(weight_decay=1e-6) on the rest incl. the dense first-order table"),
            ("base_config_sparse.yaml + fc_sparse (extension)", True, True, "both tables row-form -> SparseAdam, Adam on the MLP"),
            ("configs/deepfm/base_config.yaml", False, False, "one dense Adam(weight_decay=1e-6) over all parameters (dense [N,D] weight.grad)")):
        torch.manual_seed(0)
        with torch.device(dev):
            model = pkg.DeepFM(dims, D, hidden, p_dropout=p_drop, use_batchnorm=True,
                               embedding_config={"name": "vanilla", "sparse": sparse}, fc_sparse=fc_sparse)
        packed = layout == "packed128" and sparse and fc_sparse      # the dense optimizers step contiguous tensors
        if packed:
            model.pack_tables()
        model.train()
        opts = optim.get_optimizers(model, dict(base, sparse=sparse))
        gstep = trainer.GraphedTrainStep(model, opts, BCEWithLogitsLoss())
        for _ in range(4):
            gstep(x, y)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            gstep(x, y)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / iters * 1e3
        out.append({"config": name, "optimizers": what, "ms_per_step": round(ms, 4), "samples_per_s": round(B / ms * 1e3, 1),
                    "one_hipGraph": gstep._graph is not None, "tables": "packed128" if packed else "two tensors"})
        del model, opts, gstep
        torch.cuda.empty_cache()
    return out


AVAZU_22 = [241, 8, 8, 3697, 4614, 25, 5481, 329, 31, 381763, 1611748, 6793, 6, 5, 2509, 9, 10, 432, 5, 68, 169, 61]
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense f32-input MFMA (= the fp32 vector rate)


def init_ranks(args, dev_index):
    """Replica runs (C3 / C5 do not shard: SURVEY.md §8e): every rank runs the whole workload on its own GPU; the only
    collective is the timing fence (barrier + MAX of the elapsed time)."""
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    return rank, world, dev


def time_graphed(step, args, world, dev):
    """Capture `step` (forward + loss + backward into static gradients) as ONE hipGraph — or, given a LIST of steps (one per
    resident batch, each reading its inputs in place), one graph per step sharing a memory pool, replayed in turn — W warm-up
    replays, K timed replays between fences; returns seconds (max over ranks)."""
    steps = list(step) if isinstance(step, (list, tuple)) else [step]
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for _ in range(3):
            steps[0]()
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize()
    if args.no_graph:            # eager launches (counter collection passes)
        class _Eager:
            def __init__(self, fn):
                self.replay = fn
        graphs = [_Eager(fn) for fn in steps]
    else:
        graphs, pool = [], None
        for fn in steps:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool):
                fn()
            pool = g.pool()
            graphs.append(g)
        # the FIRST replay of a captured graph uploads it to the device: part of building the step, like the capture — with
        # 16 per-batch graphs and W = 5 warm-up steps eleven of those uploads used to fall inside the timed region
        for g in graphs:
            g.replay()
        torch.cuda.synchronize()

    class _Ring:
        i = 0

        def replay(self):
            graphs[self.i % len(graphs)].replay()
            self.i += 1
    graph = _Ring()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist

            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(0, args.settle)):      # (clock ramp after the captures: see bench_c2)
        graph.replay()
    for _ in range(args.warmup):
        graph.replay()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        graph.replay()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist

        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def kernel_table(kt):
    return {k: {"avg_us": round(v["avg_us"], 3), "min_us": round(v["min_us"], 3), "launches": v["count"]} for k, v in kt.summary().items()}


def emit(out, real_stdout):
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    print(json.dumps(out), flush=True)


def bench_c3(args, real_stdout):
    """BASELINE config 3 (SURVEY.md §8d "C3"): DCN-Mix on the Avazu-shaped 22 fields (N = 2.02 M rows, D = 16 -> d = 352) with
    the QR `divider: 2` embedding (configs/avazu/qr_2.yaml:9-10), E = 4 experts, rank 64, L = 3 cross layers
    (src/models/dcn.py:19-21), MLP 400x3 + BatchNorm + dropout 0.5, B = 4096.  Step = model(x) -> BCE-with-logits ->
    backward.  Roofline: the CrossNet contractions are MFMA-bound (the library's `gemm_f32` launches): algorithmic flops per
    sample and layer E(2dr + 2r^2 + 2rd) + 2Ed forward, x3 layers, x3 for forward + backward."""
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    rank, world, dev = init_ranks(args, local_rank)
    import recsys_benchmark_amd as pkg
    from recsys_benchmark_amd import _lib as L
    from recsys_benchmark_amd import mlp as _mlp
    from recsys_benchmark_amd.dcn import DCN_Mix
    from recsys_benchmark_amd.losses import BCEWithLogitsLoss, unit_scalar
    from recsys_benchmark_amd.profiling import KernelTimer

    _mlp.TUNE_BACKWARD_GEMMS = not args.no_gemm_tuning
    _mlp.FUSED_TAIL = (bool(args.fused_tail) or _mlp.FUSED_TAIL) and not args.library_tail     # (default on; MI_FUSED_TAIL=0)
    dims, D, hidden, B, E, r, nl = list(AVAZU_22), 16, [400, 400, 400], args.batch, 4, 64, 3
    F, d = len(dims), len(dims) * D
    torch.manual_seed(2023)
    # the quotient table's gradient in row (COO) form, like C2's tables (an extension: the reference's QR has no sparse
    # option; --dense-grads gives its dense weight.grad: a 64 MB zero-fill + scattered float atomics per step)
    emb_cfg = {"name": "qr", "divider": 2} if args.dense_grads else {"name": "qr", "divider": 2, "sparse": True}
    model = DCN_Mix(dims, D, hidden, num_layers=nl, num_experts=E, rank=r, embedding_config=emb_cfg, p_dropout=0.5).to(dev).train()
    # resident batches with distinct ids, one hipGraph each (fresh rows every step, no input copy) — like the C2 leg
    batches = [synth_batch(dims, B, 2023 + 7919 * rank + 104729 * i, dev) for i in range(max(1, min(args.ring, 8)))]
    x, y = batches[0]
    lossf, one = BCEWithLogitsLoss(), unit_scalar(dev)

    def step_on(xb, yb):
        def fn():
            model.zero_grad(set_to_none=True)
            # (labels handed to the forward, as the package's trainer does: head + criterion + head backward in one launch)
            lossf(model(xb) if args.no_head_loss else model(xb, labels=yb), yb).backward(one)
        return fn

    step = step_on(x, y)
    elapsed = time_graphed([step_on(xb, yb) for xb, yb in batches], args, world, dev)
    pkg.check_index_errors()
    n_prof = min(args.steps, 50)
    with KernelTimer(capacity=96 * n_prof + 64) as kt:
        for _ in range(n_prof):
            step()
            L.load().mi_prof_empty_launch(256, 256, L.stream_ptr(dev))
        torch.cuda.synchronize()
    if rank != 0:
        return
    kernels = kernel_table(kt)
    flops_step = 3.0 * nl * (E * (2 * d * r + 2 * r * r + 2 * r * d) + 2 * E * d) * B
    tail_wgrad = 0.0
    if _mlp.FUSED_TAIL:      # the own MLP tail runs its weight gradients as a multi-problem launch too: same kernel name, so its
        widths = [d] + hidden      # flops join the count (its other products are tail_* kernels and stay out)
        tail_wgrad = 2.0 * B * sum(a * b for a, b in zip(widths[:-1], widths[1:]))
        flops_step += tail_wgrad
    # every launch that carries CrossNet products: 64x64-tile GEMMs, the head's weight gradients (one multi-problem launch
    # per backward), the layer products on 64-row panels and the per-expert kernels with the r x r product in their epilogue
    prods = [kernels[k] for k in ("gemm_f32", "gemm_f32_multi", "gemm_tn_multi", "gemm_f32_panel", "mix_expert_fwd", "mix_expert_bwd")
             if k in kernels]
    roofline = None
    if prods:
        launches = sum(o["launches"] for o in prods)
        per_step_us = sum(o["avg_us"] * o["launches"] for o in prods) / n_prof
        ach = flops_step / (per_step_us * 1e-6) / 1e12
        roofline = {"bound": "mfma", "kernel": "all CrossNet products of a step (gemm_f32 / _panel / _multi | gemm_tn_multi, mix_expert_fwd / _bwd)" +
                    (" + the MLP tail's weight gradients (the same multi-problem kernel)" if tail_wgrad else ""),
                    "achieved": round(ach, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / MFMA_F32_PEAK_TFLOPS, 4), "traffic": None,
                    "launches_per_step": launches / n_prof, "us_per_step": round(per_step_us, 2),
                    "alg_flops_per_step": flops_step, "floor_us": kernels.get("empty", {}).get("avg_us")}
    out = {"metric": "samples/sec fwd+bwd, Avazu-22field DCN-Mix (QR divider 2) b=4096; MFMA TFLOP/s vs roofline",
           "value": round(B * world * args.steps / elapsed, 1), "unit": "samples/s", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "launch": "hipGraph replay",
           "config": {"workload": f"C3 DCN-Mix Avazu-22field: F={F}, D={D}, d={d}, N={sum(dims)} rows, QR divider 2 (mult), E={E}, "
                                  f"rank={r}, L={nl}, MLP 400x3+BN+dropout0.5, B={B}/GPU, fwd+bwd, quotient-table grad {'dense' if args.dense_grads else 'row-form (COO)'}, {len(batches)} distinct uniform-id batches rotated "
                                  f"(one hipGraph per resident batch)",
                      "global_batch": B * world, "parallelism": "single" if world == 1 else f"{world} independent replicas"},
           "roofline": roofline, "kernels": kernels}
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline_c3(model, dims, D, nl, len(hidden), B)
    emit(out, real_stdout)


def cpu_baseline_c3(model, dims, D, nl, n_hidden, B):
    from oracle import reference_ops as ro

    phys, logical = physical_cores()
    torch.set_num_threads(phys)
    p = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    for k, v in p.items():
        if v.is_floating_point() and "running_" not in k:
            v.requires_grad_(True)
    x, y = synth_batch(dims, B, 2023, "cpu")
    lossf = torch.nn.BCEWithLogitsLoss()
    divider = 2

    def step():
        for v in p.values():
            v.grad = None
        rows = x + p["offsets"]
        emb = ro.qr_forward(rows, p["embedding.emb1.weight"], p["embedding.emb2.weight"], divider, "mult")
        lossf(ro.dcn_mix_forward(x, p, emb, nl, n_hidden, True), y).backward()

    med, n = timed_cpu(step, budget_s=20.0)
    return {"value": round(B / med, 1), "unit": "samples/s", "cores": phys, "logical_cpus": logical, "kind": "port",
            "sample": f"3 warm-up + {n} timed fwd+bwd steps (median {med*1e3:.1f} ms/step) of the same C3 workload, "
                      "oracle/reference_ops.py qr_forward + dcn_mix_forward on torch CPU (dropout off: the oracle is functional)"}


def yelp_graph(U=31668, I=38048, nnz=1128375, seed=2023):
    """edges user-uniform x item power-law (item = floor(I u^2)), symmetric normalised adjacency of the
    bipartite graph as the reference builds it (src/graph_utils.py:47-98), CSR fp32."""
    gen = torch.Generator().manual_seed(seed)
    u = torch.randint(0, U, (nnz,), generator=gen)
    i = (I * torch.rand(nnz, generator=gen).pow(2)).long().clamp_(max=I - 1)
    n = U + I
    idx = torch.stack([torch.cat([u, i + U]), torch.cat([i + U, u])])
    adj = torch.sparse_coo_tensor(idx, torch.ones(2 * nnz), size=(n, n)).coalesce()
    deg = torch.sparse.sum(adj, dim=1).to_dense().clamp_(min=1).pow(-0.5)
    ii = adj.indices()
    return torch.sparse_coo_tensor(ii, adj.values() * deg[ii[0]] * deg[ii[1]], size=(n, n)).coalesce().to_sparse_csr()


def bench_c5(args, real_stdout):
    """BASELINE config 5 (SURVEY.md §8d "C5"): LightGCN, Yelp2018-shaped (U = 31 668, I = 38 048, 1 128 375 interactions ->
    nnz(A) ~ 2.25 M, D = 64, L = 3, configs/yelp2018/lightgcn_config.yaml:1-7), 2048 BPR triples per step.  Step = what the
    reference's _train_step computes before the optimizer (src/trainer/lightgcn.py:380-421): full propagation, BPR over the
    batch rows, L2 regulariser, backward.  Roofline: the CSR SpMM launches are HBM-bound; algorithmic (compulsory) bytes per
    layer 8 nnz + 4 (N + 1) + 8 N D."""
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    rank, world, dev = init_ranks(args, local_rank)
    import recsys_benchmark_amd as pkg
    from recsys_benchmark_amd import _lib as L
    from recsys_benchmark_amd.lightgcn import LightGCN
    from recsys_benchmark_amd.losses import bpr_loss_rows, unit_scalar
    from recsys_benchmark_amd.profiling import KernelTimer
    from recsys_benchmark_amd._kernels import spmm as _kernels_spmm

    U, I, D, nl, B = 31668, 38048, 64, 3, 2048
    adj_cpu = yelp_graph(U, I)
    adj = adj_cpu.to(dev)
    N, nnz = U + I, int(adj_cpu.values().numel())
    torch.manual_seed(2023)
    model = LightGCN(U, I, num_layers=nl, hidden_size=D).to(dev).train()
    gen = torch.Generator().manual_seed(2023 + rank)
    users = torch.randint(0, U, (B,), generator=gen).to(dev)
    pos, neg = torch.randint(0, I, (B,), generator=gen).to(dev), torch.randint(0, I, (B,), generator=gen).to(dev)
    one = unit_scalar(dev)

    def step():
        model.zero_grad(set_to_none=True)
        if args.separate_reg:      # the reference's two calls: model(adj), then model.get_reg_loss(...)
            au, ai = model(adj)
            reg = model.get_reg_loss(users, pos, neg)
        else:                      # what trainer.GraphedCFTrainStep runs: both as one autograd node, the last layer at the
            au, ai, reg = model.forward_with_reg_loss(adj, users, pos, neg,          # rows the losses read
                                                      batch_rows_only=not args.full_last_layer)
        if args.separate_reg:
            (bpr_loss_rows(au, ai, users, pos, neg) + 1e-4 * reg).backward(one)
        else:                      # ... and loss = bpr + weight_decay * reg out of the BPR launch itself
            bpr_loss_rows(au, ai, users, pos, neg, plus=reg, plus_weight=1e-4).backward(one)

    elapsed = time_graphed(step, args, world, dev)
    pkg.check_index_errors()
    n_prof = min(args.steps, 50)
    with KernelTimer(capacity=64 * n_prof + 64) as kt:
        for _ in range(n_prof):
            step()
            L.load().mi_prof_empty_launch(256, 256, L.stream_ptr(dev))
        torch.cuda.synchronize()
    # roofline leg: FULL layers only (in the step one backward layer skips the zero rows of its operand and — unless
    # --full-last-layer — the last forward layer computes the batch's rows only: both move fewer bytes than a layer's
    # algorithmic count), each the product of the whole adjacency with a dense [N, D] operand
    Xfull = torch.randn(N, D, device=dev)
    with torch.no_grad(), KernelTimer(capacity=4 * n_prof + 16) as kt_full:
        for _ in range(n_prof):
            Xfull = _kernels_spmm(adj, Xfull)
        torch.cuda.synchronize()
    if rank != 0:
        return
    kernels = kernel_table(kt)
    full = kernel_table(kt_full)
    alg = 8 * nnz + 4 * (N + 1) + 8 * N * D
    kname = next((n for n in ("spmm_sliced", "spmm_tiled", "spmm_csr") if n in full), "spmm_csr")
    k = full.get(kname)
    roofline = None
    if k:
        ach = alg / (k["avg_us"] * 1e-6) / 1e9
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        traffic = json.load(open(tpath)).get(kname) if os.path.exists(tpath) else None
        roofline = {"bound": "hbm", "kernel": kname, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "avg_us": k["avg_us"], "alg_bytes": alg,
                    "what": "full layers (A x dense [N, D]) launched back to back outside the step; the step's own six launches "
                            "are in `kernels` (one skips zero operand rows, one computes the batch's rows only)",
                    "launches_per_step": kernels.get(kname, {}).get("launches", 0) / n_prof, "nnz": nnz,
                    "gather_bytes_without_reuse": nnz * (8 + 4 * D), "G_nnz_per_s": round(nnz / (k["avg_us"] * 1e-6) / 1e9, 2),
                    "floor_us": kernels.get("empty", {}).get("avg_us")}
    out = {"metric": "BPR triples/sec fwd+bwd, LightGCN Yelp2018-shaped 3-layer (propagation over the whole graph every step); HBM GB/s vs roofline",
           "value": round(B * world * args.steps / elapsed, 1), "unit": "triples/s", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "launch": "hipGraph replay",
           "config": {"workload": f"C5 LightGCN Yelp2018-shaped: U={U}, I={I}, N={N}, nnz(A)={nnz}, D={D}, L={nl}, {B} BPR triples/step "
                                  "+ L2 reg, fwd+bwd (no optimizer); " +
                                  ("the reference's two calls model(adj) + get_reg_loss()" if args.separate_reg else
                                   "propagation + regulariser as one autograd node" +
                                   ("" if args.full_last_layer else ", last layer at the rows the losses read (the batch's)")),
                      "global_batch": B * world, "parallelism": "single" if world == 1 else f"{world} independent replicas (the path does not shard: SURVEY.md §8e)"},
           "roofline": roofline, "kernels": kernels}
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline_c5(model, adj_cpu, users.cpu(), pos.cpu(), neg.cpu(), U, nl, B)
    emit(out, real_stdout)


def cpu_baseline_c5(model, adj_cpu, users, pos, neg, U, nl, B):
    from oracle import reference_ops as ro

    phys, logical = physical_cores()
    torch.set_num_threads(phys)
    Eu = model.user_emb_table.get_weight().detach().cpu().clone().requires_grad_(True)
    Ei = model.item_emb_table.get_weight().detach().cpu().clone().requires_grad_(True)

    def step():
        Eu.grad = Ei.grad = None
        res = ro.lightgcn_propagate(adj_cpu, torch.cat([Eu, Ei]), nl)
        au, ai = res[:U], res[U:]
        loss = ro.bpr_loss(au[users], ai[pos], ai[neg]) + 1e-4 * ro.l2_reg_loss(Eu[users], Ei[pos], Ei[neg])
        loss.backward()

    med, n = timed_cpu(step, budget_s=25.0)
    return {"value": round(B / med, 1), "unit": "triples/s", "cores": phys, "logical_cpus": logical, "kind": "port",
            "sample": f"3 warm-up + {n} timed fwd+bwd steps (median {med*1e3:.1f} ms/step) of the same C5 workload, "
                      "oracle/reference_ops.py lightgcn_propagate (torch CSR matmul, as src/models/lightgcn.py:79-87) + bpr_loss + l2_reg_loss on torch CPU"}


def bench_infer(args, real_stdout):
    """The reference's ONLY timing harnesses are inference: scripts/deepfm/infer_deepfm.py:36,318-352 — one
    eval-mode forward under torch.no_grad() on a resident batch (default batch 64), wall clock after one warm-up, or
    timeit.repeat(number=20) — and scripts/lightgcn/infer_lightgcn.py:56-114,459-467 — forward / matching / filter / top-k of
    one request (user_id = [0], k = 20), each phase fenced by cuda.synchronize(), mean over 20 runs.  This leg runs both
    protocols on the C2 / C5 shapes (eval mode: BatchNorm on running statistics, no dropout — every product on the library's
    own kernels, csrc/tail.hip + mi_tail_affine_consts) and the CPU oracle by the survey's 3 + 10 protocol beside them."""
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    import statistics

    import recsys_benchmark_amd as pkg
    from recsys_benchmark_amd import _lib as L
    from recsys_benchmark_amd.lightgcn import LightGCN, score_topk, train_items_csr
    from recsys_benchmark_amd.profiling import KernelTimer

    out = {"metric": "inference latency by the reference's own timing harnesses (eval-mode forward; LightGCN request phases)",
           "unit": "us", "n_gpus": 1, "dtype": "f32", "data": "synthetic", "higher_is_better": False, "infer": {}}

    # ---------------------------------------------------------------- DeepFM (C2 shape), eval + no_grad
    dims, D, hidden = list(CRITEO_KAGGLE_26), 16, [400, 400, 400]
    F = len(dims)
    torch.manual_seed(2023)
    model = pkg.DeepFM(dims, D, hidden, p_dropout=0.5, use_batchnorm=True).to(dev)
    model.pack_tables()
    model.eval()
    deep = []
    for B in (64, 4096):
        xs = [synth_batch(dims, B, 555 + i, dev)[0] for i in range(16)]
        with torch.no_grad():
            model(xs[0])                                   # the reference's single warm-up
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            model(xs[1]).cpu()                             # "--- Inference ---": one forward + the copy of its output
            single = time.perf_counter() - t0
            reps = []
            for r in range(5):                             # timeit.repeat(number=20): 5 repeats of 20 forwards
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(20):
                    model(xs[(r * 20 + i) % len(xs)])
                torch.cuda.synchronize()
                reps.append((time.perf_counter() - t0) / 20)
            # the same forward replayed as a hipGraph per resident batch (what a serving loop would keep)
            graphs = []
            pool = None
            for xb in xs[:8]:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=pool):
                    yb = model(xb)
                pool = g.pool()
                graphs.append((g, yb))
            for g, _ in graphs:
                g.replay()
            torch.cuda.synchronize()
            lat = []
            for i in range(200):
                t0 = time.perf_counter()
                graphs[i % len(graphs)][0].replay()
                torch.cuda.synchronize()
                lat.append(time.perf_counter() - t0)
            t0 = time.perf_counter()
            for i in range(200):
                graphs[i % len(graphs)][0].replay()
            torch.cuda.synchronize()
            thr = (time.perf_counter() - t0) / 200
            with KernelTimer(capacity=256) as kt:
                model(xs[2])
                torch.cuda.synchronize()
        lat.sort()
        deep.append({"B": B, "single_forward_plus_output_copy_us": round(single * 1e6, 1),
                     "eager_forward_us_timeit20": {"p50": round(statistics.median(reps) * 1e6, 1), "min": round(min(reps) * 1e6, 1)},
                     "graph_replay_latency_us": {"p50": round(lat[len(lat) // 2] * 1e6, 1), "p99": round(lat[int(len(lat) * 0.99)] * 1e6, 1)},
                     "graph_replay_back_to_back_us": round(thr * 1e6, 2),
                     "p50_us": round(lat[len(lat) // 2] * 1e6, 1),
                     "samples_per_s": round(B / thr, 1),
                     "kernels_per_forward": {k: v["count"] for k, v in kt.summary().items()}})
        del graphs
    pkg.check_index_errors()
    out["infer"]["deepfm"] = {"shape": f"C2: F={F}, D={D}, N={sum(dims)} rows (packed [N,32] table), MLP 400x3 + BatchNorm (eval) + Linear(400,1)",
                              "lines": deep,
                              "note": "every kernel of the forward is the library's own (gather_fm_fwd, tail_affine_consts, tail_fwd_gemm x3, "
                                      "tail_head_fwd): kernels_per_forward lists what one eager forward launched"}
    if not args.no_cpu_baseline:
        from oracle import reference_ops as ro

        phys, logical = physical_cores()
        torch.set_num_threads(phys)
        p = {k: v.detach().cpu().clone().contiguous() for k, v in model.state_dict().items()}
        cpu_lines = []
        for B in (64, 4096):
            x = synth_batch(dims, B, 555, "cpu")[0]
            with torch.no_grad():
                med, n = timed_cpu(lambda: ro.deepfm_forward(x, p, len(hidden), True, False), budget_s=8.0)
            cpu_lines.append({"B": B, "p50_us": round(med * 1e6, 1), "samples_per_s": round(B / med, 1), "timed_iters": n})
        out["infer"]["deepfm"]["cpu_baseline"] = {"kind": "port", "cores": phys, "lines": cpu_lines,
                                                  "sample": "3 warm-up + 10 timed eval forwards (median), oracle/reference_ops.deepfm_forward under no_grad"}
    del model
    torch.cuda.empty_cache()

    # ---------------------------------------------------------------- LightGCN (C5 shape): request phases
    U, I, Dg, nl, k = 31668, 38048, 64, 3, 20
    adj_cpu = yelp_graph(U, I)
    adj = adj_cpu.to(dev)
    torch.manual_seed(2023)
    gm = LightGCN(U, I, num_layers=nl, hidden_size=Dg).to(dev).eval()
    # the users' train items (what the filter masks): the bipartite block of the adjacency's rows
    crow = adj_cpu.crow_indices()
    colv = adj_cpu.col_indices()
    graph = {u: (colv[int(crow[u]):int(crow[u + 1])] - U).tolist() for u in range(2048)}
    train_csr = train_items_csr(graph, 2048, dev)      # (crow, col) of the users a request may name
    from recsys_benchmark_amd import _kernels as K_
    lg = []
    for users_n in (1, 2048):
        users = torch.arange(users_n, device=dev)
        acc = {"forward": 0.0, "matching": 0.0, "filter_topk": 0.0}
        n_runs = 20
        lib = L.load()
        with torch.no_grad():
            gm(adj)
            score_topk(*gm(adj), users, k, train_csr)
            torch.cuda.synchronize()
            for _ in range(n_runs):
                t0 = time.perf_counter()
                ue, ie = gm(adj)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                # matching as the product runs it (lightgcn.score_topk): row gather + fp32 MFMA product
                rows = K_.gather_rows(users, ue)
                scores = torch.empty((users_n, I), dtype=torch.float32, device=dev)
                K_.gemm(rows, ie.contiguous(), scores, users_n, I, Dg, Dg, Dg, I, transB=True)
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                # filter + top-k: ONE launch, a workgroup per user masks the user's train items and selects the k best
                topk = torch.empty((users_n, k), dtype=torch.int64, device=dev)
                L.check(lib.mi_mask_topk_rows(scores.data_ptr(), scores.stride(0), users_n, I, users.data_ptr(), train_csr[0].data_ptr(),
                                              train_csr[1].data_ptr(), k, topk.data_ptr(), None, L.stream_ptr(dev)), "mask_topk")
                torch.cuda.synchronize()
                t3 = time.perf_counter()
                acc["forward"] += t1 - t0
                acc["matching"] += t2 - t1
                acc["filter_topk"] += t3 - t2
        lg.append({"users": users_n, "k": k, "forward_us": round(acc["forward"] / n_runs * 1e6, 1),
                   "matching_us": round(acc["matching"] / n_runs * 1e6, 1),
                   "filter_plus_topk_us": round(acc["filter_topk"] / n_runs * 1e6, 1),
                   "note": "filter and top-k are one launch here (a workgroup per user masks the user's train items and selects the "
                           "k best): no host loop over users — the reference builds the filter's index lists in Python per request"})
    out["infer"]["lightgcn"] = {"shape": f"C5: U={U}, I={I}, nnz(A)={int(adj_cpu.values().numel())}, D={Dg}, L={nl}", "lines": lg,
                                "protocol": "scripts/lightgcn/infer_lightgcn.py: phases fenced by synchronize, mean of 20 runs; users = [0] "
                                            "(the script's request) and the first 2048 users (a validation batch)"}
    if not args.no_cpu_baseline:
        from oracle import reference_ops as ro

        Eu = gm.user_emb_table.get_weight().detach().cpu()
        Ei = gm.item_emb_table.get_weight().detach().cpu()
        cpu_lg = []
        for users_n in (1, 2048):
            uid = list(range(users_n))
            ph = {"forward": [], "matching": [], "filter": [], "topk": []}
            for it in range(3 + 5):
                t0 = time.perf_counter()
                with torch.no_grad():
                    res = ro.lightgcn_propagate(adj_cpu, torch.cat([Eu, Ei]), nl)
                    t1 = time.perf_counter()
                    scores = res[:U][uid] @ res[U:].T
                    t2 = time.perf_counter()
                    ind0 = torch.tensor([], dtype=torch.long)
                    ind1 = torch.tensor([], dtype=torch.long)
                    for idx, user in enumerate(uid):                                  # the reference's own loop (:93-99)
                        ind0 = torch.cat((ind0, torch.tensor([idx] * len(graph[user]), dtype=torch.long)))
                        ind1 = torch.cat((ind1, torch.tensor(graph[user], dtype=torch.long)))
                    scores[ind0, ind1] = float("-inf")
                    t3 = time.perf_counter()
                    torch.topk(scores, k, dim=1)
                    t4 = time.perf_counter()
                if it >= 3:
                    for name, v in zip(ph, (t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
                        ph[name].append(v)
            cpu_lg.append({"users": users_n, **{f"{n}_us": round(statistics.median(v) * 1e6, 1) for n, v in ph.items()}})
        phys, _ = physical_cores()
        out["infer"]["lightgcn"]["cpu_baseline"] = {"kind": "port", "cores": phys, "lines": cpu_lg,
                                                    "sample": "3 warm-up + 5 timed requests (median per phase), the reference's op sequence incl. its Python filter loop"}
    d64 = deep[0]
    out["value"] = d64["p50_us"]
    out["config"] = {"workload": "inference: DeepFM C2 eval forward at B=64 (value = p50 graph-replay latency) and B=4096; LightGCN C5 request phases"}
    emit(out, real_stdout)


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
    env.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")      # a failed / timed-out collective takes the rank down
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
    ap.add_argument("--config", choices=["c2", "c3", "c5"], default="c2", help="c2 (default): the headline DeepFM workload; c3: DCN-Mix "
                    "Avazu-shaped with the QR embedding; c5: LightGCN Yelp2018-shaped (SURVEY.md §8d); --c4 is c2's 1e9-row variant")
    ap.add_argument("--fused-tail", action="store_true", help="(the default now) the MLP tail on the own fused MFMA kernels of "
                    "csrc/tail.hip")
    ap.add_argument("--copy-batch", action="store_true", help="one graph with a static input and a device copy of the next resident "
                    "batch into it every step (the round-1 form) instead of one graph per resident batch")
    ap.add_argument("--library-tail", action="store_true", help="the MLP tail's contractions on hipBLASLt / rocBLAS through PyTorch "
                    "(+ the fused BatchNorm passes of csrc/mlp.hip) instead of the own fused kernels; same speed (DESIGN.md §5b)")
    ap.add_argument("--layout", choices=["split", "packed128"], default="packed128", help="storage of the two lookup tables: the "
                    "reference's two tensors (split) or one [N,32] buffer holding row + first-order weight per 128-B line "
                    "(DeepFM.pack_tables(); the default: same arithmetic, one random line per lookup instead of two sectors)")
    ap.add_argument("--ids", choices=["uniform", "zipf"], default="uniform", help="SURVEY.md §8d id distributions: uniform, or "
                    "Zipf(alpha=1.05) per field")
    ap.add_argument("--fields", type=int, choices=[26, 39], default=26, help="26: the headline Criteo categorical fields; 39: the "
                    "reference-faithful variant with 13 fields of 50 buckets prepended (src/dataset/criteo/utils.py:8-9)")
    ap.add_argument("--no-eager-leg", action="store_true", help="skip the eager pass that times every kernel by dispatch events "
                    "(so that a profiler run sees in-graph launches only)")
    ap.add_argument("--full-last-layer", action="store_true", help="c5: compute every row of the last propagation layer (the "
                    "default computes the rows the losses read: the batch's)")
    ap.add_argument("--separate-reg", action="store_true", help="c5: call model(adj) and model.get_reg_loss() separately (the "
                    "reference's call shape) instead of LightGCN.forward_with_reg_loss (what the mirrored trainer runs)")
    ap.add_argument("--no-gather-leg", action="store_true", help="skip the roofline leg (graphs holding only gather+FM kernels), so "
                    "that a profiler run sees those kernels in the step's replays only")
    ap.add_argument("--probe-empties", action="store_true", help="diagnostic: empty kernels of grid 1, 2 (in front of the forward) "
                    "and 3 (behind the backward) inside every captured step, to read what a profiler charges a launch that does "
                    "nothing at those places (invalidates the measurement as a benchmark line)")
    ap.add_argument("--no-sweep", action="store_true", help="skip roofline.batch_sweep")
    ap.add_argument("--no-train-step", action="store_true", help="skip the train_step object (fwd+bwd+optimizers as one graph)")
    ap.add_argument("--settle", type=int, default=200, help="replays of the captured step(s) between building them and the W warm-up "
                    "steps (the GPU's clocks ramp up over the first milliseconds of load after the host-bound capture phase); 0: none")
    ap.add_argument("--no-head-loss", action="store_true", help="model(x) without the step's labels: head, criterion and head backward as "
                    "three launches (default: model(x, labels=y), one launch)")
    ap.add_argument("--no-prefetch", action="store_true", help="default: every step touches the NEXT batch's table rows in extra "
                    "workgroups of its weight-gradient launch (DeepFM.prefetch_next), the next forward gathers from the Infinity "
                    "Cache; with this flag the steps know nothing of the next batch")
    ap.add_argument("--infer", action="store_true", help="the reference's own timing harnesses (scripts/deepfm/infer_deepfm.py, "
                    "scripts/lightgcn/infer_lightgcn.py): eval-mode forward latency at B=64 / 4096, LightGCN request phases, CPU oracle beside")
    ap.add_argument("--windows", type=int, default=10, help="extra windows of --steps replays after the timed region, for "
                    "ms_per_step_windows {min, median, max}")
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
    if args.infer and not args.dry_launch:
        return bench_infer(args, real_stdout)
    if args.config == "c3" and not args.dry_launch:
        return bench_c3(args, real_stdout)
    if args.config == "c5" and not args.dry_launch:
        return bench_c5(args, real_stdout)
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
            import datetime

            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=datetime.timedelta(seconds=300))
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
    _mlp.FUSED_TAIL = (bool(args.fused_tail) or _mlp.FUSED_TAIL) and not args.library_tail     # (default on; MI_FUSED_TAIL=0)

    dims, D, hidden, p_drop = list(CRITEO_KAGGLE_26), 16, [400, 400, 400], 0.5
    if args.fields == 39:
        dims = [50] * 13 + dims
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
        if args.layout == "packed128":
            model.pack_tables()
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

    ring = [blob_of(*synth_batch(dims, B, 2023 + 7919 * rank + 104729 * i, dev, args.ids)) for i in range(max(1, args.ring))]
    cur = ring[0].clone()
    x, y = cur[:nx].view(torch.int64).view(B, F), cur[nx:].view(torch.float32)
    from recsys_benchmark_amd.losses import BCEWithLogitsLoss

    lossf = BCEWithLogitsLoss()      # same criterion as the reference trainer, one launch each way
    state = {"i": 0}
    # the labels handed to the forward (DeepFM.forward(x, labels=y), what the package's trainer does): the head launch also
    # evaluates the criterion and the head's backward sums, lossf() picks them up (--no-head-loss: the reference's two calls
    # as they stand — head, criterion and head backward are three launches)
    labels_in_forward = not args.no_head_loss and not sharded
    # one graph per resident batch, replayed in turn: graph i is followed by graph i + 1, whose ids it can touch ahead
    prefetching = (not args.no_prefetch and not args.copy_batch and not sharded and hasattr(model, "prefetch_next")
                   and not args.no_graph)

    def fwd(xb, yb):
        return model(xb, labels=yb) if labels_in_forward else model(xb)

    def next_batch():
        cur.copy_(ring[state["i"] % len(ring)])
        state["i"] += 1

    def eager_step():
        next_batch()
        model.zero_grad(set_to_none=True)
        loss = lossf(fwd(x, y), y)
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

            def timed_phases(iters=20):
                """Per-phase device time of the sharded step: HIP events between the phases of `iters` steps enqueued back
                to back like the timed region's (no host sync between steps, so the host stays ahead of the device as it
                does there and an interval is the phase's own device time, not the host's launch latency); the collectives
                and the graph replay are the same calls the timed region makes.  `synced`: the same with a host sync after
                every step (each phase then starts from an idle device: what a step costs when nothing is queued behind it)."""
                def run(sync_each):
                    names, marks = [], []
                    for it in range(iters + 2):
                        evs = [torch.cuda.Event(enable_timing=True)]
                        evs[0].record()
                        labels = []

                        def mark(name):
                            e = torch.cuda.Event(enable_timing=True)
                            e.record()
                            evs.append(e)
                            labels.append(name)
                        next_batch()
                        graphed_step(x, mark=mark)
                        if sync_each:
                            torch.cuda.synchronize()
                        marks.append(evs)
                        names = labels
                    torch.cuda.synchronize()
                    acc = {}
                    for evs in marks[2:]:
                        for k, name in enumerate(names):
                            acc[name] = acc.get(name, 0.0) + evs[k].elapsed_time(evs[k + 1]) * 1e3
                    return [{"phase": n, "us": round(acc[n] / iters, 2)} for n in names]
                return {"pipelined": run(False), "synced": run(True)}
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
        model.zero_grad(set_to_none=True)
        from recsys_benchmark_amd.losses import unit_scalar

        one = unit_scalar(dev)      # d(loss)/d(loss) = 1 from a resident scalar: no fill per step, and the fused criterion
        #                             hands back the gradient its forward already wrote
        # One graph PER resident batch, each reading its ids and labels in place (all graphs share one memory pool: their
        # replays never overlap): the step consumes the next resident batch without first copying it into a static input —
        # inputs are in HBM when the timed region starts, and nothing but the step itself is timed.  (--copy-batch keeps
        # the one-graph form with the 868 KB copy per step.)
        graphs = []
        slots = [cur] if args.copy_batch else ring
        pool = None
        for gi, blob in enumerate(slots):
            xb, yb = blob[:nx].view(torch.int64).view(B, F), blob[nx:].view(torch.float32)
            if prefetching:
                # graph i is followed by graph i + 1 in the timed ring: its step touches that batch's table rows
                nb = slots[(gi + 1) % len(slots)]
                model.prefetch_next(nb[:nx].view(torch.int64).view(B, F))
            g = torch.cuda.CUDAGraph()
            model.zero_grad(set_to_none=True)          # every capture builds its own gradient buffers (no accumulation)
            with torch.cuda.graph(g, pool=pool):
                if args.probe_empties:
                    from recsys_benchmark_amd import _lib as _pl
                    _pl.load().mi_prof_empty_launch(1, 256, _pl.stream_ptr(dev))
                    _pl.load().mi_prof_empty_launch(2, 256, _pl.stream_ptr(dev))
                lossf(fwd(xb, yb), yb).backward(one)
                if args.probe_empties:
                    _pl.load().mi_prof_empty_launch(3, 256, _pl.stream_ptr(dev))
            pool = g.pool()
            graphs.append(g)
        # the FIRST replay of a captured graph uploads it to the device: part of building the step, like the capture — with
        # 16 per-batch graphs and W = 5 warm-up steps eleven of those uploads used to fall inside the timed region (0.276 ms
        # per step at --steps 20 against 0.258 in every later window)
        for g in graphs:
            g.replay()
        torch.cuda.synchronize()

        def step():
            if args.copy_batch:
                next_batch()
                graphs[0].replay()
            else:
                graphs[state["i"] % len(graphs)].replay()
                state["i"] += 1

    # part of BUILDING the step, like the captures and the graphs' first replays above: the clocks of a GPU that has just sat
    # through seconds of host-side capture work take some milliseconds of load to come up (measured: the first 20 steps
    # after the captures run 3.5 % slower than every later window, 0.2345 vs 0.2262 ms) — the ring is replayed --settle times
    # first; then the contract's W warm-up steps and K timed steps
    for _ in range(max(0, args.settle)):
        step()
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    # the timed region above is the contract's; the same K replays a few more times give it an error bar
    window_ms = []
    for _ in range(max(0, args.windows)):
        fence()
        tw = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        window_ms.append((time.perf_counter() - tw) / args.steps * 1e3)
    if sharded:
        model.check_index_errors()       # collective: every rank raises together
        model.check_overflow()
    else:
        pkg.check_index_errors()
    if use_graph:
        graphs[-1].replay()         # p.grad are the buffers of the graph captured last: fill them for the check below
        torch.cuda.synchronize()
    # the timed steps really produced gradients (a replayed graph reading a freed seed tensor would give zeros)
    # (weights only: a Linear bias in front of a training-mode BatchNorm has an exactly zero gradient)
    probe = [model._bias.grad] + [p.grad for p in model._deep_branch.parameters() if p.grad is not None and p.dim() == 2][:2]
    for gprobe in probe:
        if gprobe is None or not bool(torch.isfinite(gprobe).all()) or float(gprobe.abs().sum()) == 0.0:
            raise SystemExit("bench: a gradient of the timed steps is missing, zero or non-finite")

    # Per-kernel table (diagnostic): the same step launched eagerly, every library kernel timed by its own dispatch
    # begin/end events (a graph replay cannot carry per-kernel events: probed, they are not stamped).  Eager launches read
    # slower than the same kernels inside the replayed graph, so the ROOFLINE numbers below do not come from this pass.
    n_prof = min(args.steps, 100)
    from recsys_benchmark_amd import _lib as _mlib

    ks = {}
    if not args.no_eager_leg:
        with KernelTimer(capacity=48 * n_prof + 64) as kt:
            for _ in range(n_prof):
                eager_step()
                _mlib.load().mi_prof_empty_launch(max(1, B // 4), 256, _mlib.stream_ptr(dev))
            torch.cuda.synchronize()
        ks = kt.summary()
    # Roofline leg: the two gather+FM kernels inside replayed hipGraphs (how the timed region runs them), wall per kernel.
    ingraph = sweep = None
    if not sharded and not args.no_gather_leg:
        xs_ring = [blob[:nx].view(torch.int64).view(B, F) for blob in ring]
        xs_ring += [synth_batch(dims, B, 31337 + i, dev, args.ids)[0] for i in range(max(0, 64 - len(xs_ring)))]
        ingraph = gather_in_graph_us(model, xs_ring, B, F, D, dev, copies=64)
        ingraph["dgrad"] = dgrad_epilogue_us(B, F, D, hidden[0], dev)
        ingraph["prefetched"] = prefetched_gather_us(model, xs_ring, B, F, D, hidden, dev) if prefetching else None
        if not args.no_sweep and rank == 0:
            sweep = batch_sweep(model, dims, F, D, dev, args.ids)
    sharded_dgrad = dgrad_epilogue_us(B, F, D, hidden[0], dev) if (sharded and rank == 0 and not args.no_gather_leg) else None
    sharded_phases = None
    if sharded and graphed_local:
        sharded_phases = timed_phases()          # every rank runs them (collectives inside); rank 0 reports its own
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
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
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        traffic_all = json.load(open(tpath)) if os.path.exists(tpath) else {}

        def rl(us, nbytes):
            return {"us": round(us, 3), "GBps": round(nbytes / us / 1e3, 1), "frac": round(nbytes / us / 1e3 / HBM_PEAK_GBS, 4)}

        roofline = None
        if ingraph is not None:
            fwd_us, bwd_us, pair_us = ingraph["fwd"], ingraph["bwd"], ingraph["pair"]
            dg = ingraph["dgrad"]
            epi_us = max(dg["fm"] - dg["plain"], 0.0)
            from recsys_benchmark_amd import tail as _tailmod
            fused = bool(_tailmod.FM_EPILOGUE and _mlp.FUSED_TAIL and sparse)
            pf = ingraph.get("prefetched") if fused else None
            fwd_step_us = pf["gather_after_prefetch"] if pf else fwd_us      # the forward as the step finds its rows
            step_pair_us = fwd_step_us + epi_us if fused else pair_us
            pair_bytes = (fb + bb) * B
            roofline = {"bound": "hbm",
                        "kernel": ("gather+FM fwd+bwd PAIR as the step runs it: k_gather_fm_fwd" +
                                   (" behind the previous step's weight-gradient launch, whose extra workgroups touched its table "
                                    "rows (DeepFM.prefetch_next)" if pf else "") +
                                   " + the lookup backward in the epilogue of "
                                   "the tail's first input-gradient product (k_tail_dgrad<..., FM>: its time with the epilogue minus "
                                   "its time without)" if fused else "gather+FM fwd+bwd pair: k_gather_fm_fwd + k_gather_fm_bwd_rows"),
                        "achieved": round(pair_bytes / step_pair_us / 1e3, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(pair_bytes / step_pair_us / 1e3 / HBM_PEAK_GBS, 4),
                        "traffic": traffic_all.get("gather_fm_pair_fused" if fused else "gather_fm_fwd"), "avg_us": round(step_pair_us, 3),
                        "alg_bytes": pair_bytes,
                        "cold": ({"what": "the same pair with the forward on rows nobody touched ahead (--no-prefetch; rounds 1-4's figure)",
                                  "avg_us": round(fwd_us + epi_us, 3),
                                  "frac": round(pair_bytes / (fwd_us + epi_us) / 1e3 / HBM_PEAK_GBS, 4)} if pf else None),
                        "prefetch": ({"gather_fm_fwd_after_prefetch_us": round(pf["gather_after_prefetch"], 3),
                                      "gather_fm_fwd_cold_behind_the_same_launch_us": round(pf["gather_cold_behind_wgrad"], 3),
                                      "weight_gradient_launch_us": round(pf["wgrad"], 3),
                                      "weight_gradient_launch_with_riders_us": round(pf["wgrad_with_riders"], 3),
                                      "riders_cost_us": round(pf["riders_cost"], 3),
                                      "note": "in-graph wall per launch, medians of 5: P = [weight gradients + riders for batch i], "
                                              "Q = [P, gather(batch i)], R = [weight gradients alone]; gather after a prefetch = "
                                              "Q - P, riders' cost = P - R (it is inside ms_per_step: the timed steps carry the riders)"}
                                     if pf else None),
                        "attribution": {"gather_fm_fwd_us": round(fwd_step_us, 3), "gather_fm_fwd_cold_us": round(fwd_us, 3),
                                        "tail_dgrad_gemm_plain_us": round(dg["plain"], 3), "tail_dgrad_gemm_fm_us": round(dg["fm"], 3),
                                        "epilogue_us": round(epi_us, 3),
                                        "alternating_plain_then_fm_us_per_pair": round(dg["alternating_pair"], 3),
                                        "note": "algorithmic bytes are SURVEY.md §8d's 24F + 20FD + 8 per sample for the PAIR (the fused "
                                                "form moves fewer: dL/demb is neither written nor read back); time = forward kernel + "
                                                "(first input-gradient product with the lookup backward in its epilogue - the same "
                                                "product without), each an in-graph wall per launch over cold buffers"},
                        "clock": "WALL per kernel inside a replayed hipGraph of 32-64 launches of that kernel, HIP events on the launching "
                                 "stream around 20 replays; ids, table rows and activations rotate through more data than the "
                                 "256 MiB Infinity Cache holds, so every launch reads from HBM as inside a training step",
                        "cache_assisted_back_to_back": {"note": "the same graphs over ONE activation buffer set and 16 id batches (their "
                                                                "rows stay in the Infinity Cache): optimistic, not what a step sees",
                                                        "gather_fm_fwd": rl(ingraph["warm"]["fwd"], fb * B),
                                                        "gather_fm_bwd_rows": rl(ingraph["warm"]["bwd"], bb * B)},
                        "alg_bytes_per_sample": {"fwd": fb, "bwd": bb},
                        "gather_fm_fwd": rl(fwd_us, fb * B), "gather_fm_bwd_rows": rl(bwd_us, bb * B),
                        "two_kernel_pair": dict(rl(pair_us, (fb + bb) * B), what="the round-3 form (MI_FUSED_FM_EPILOGUE=0): wall per fwd + bwd_rows pair in a graph of 64 pairs",
                                                sum_of_the_two_kernels_us=round(fwd_us + bwd_us, 3)),
                        "target_us_for_half_of_peak": round((fb + bb) * B / (0.5 * HBM_PEAK_GBS) / 1e3, 2),
                        "table_layout": args.layout if not sharded else "sharded packed rows"}
            if kernels:
                roofline["eager_dispatch_clock"] = {
                    "note": "diagnostic: the same kernels in an EAGER pass by their dispatch begin/end events (reads higher than in-graph)",
                    **{k: kernels[k]["avg_us"] for k in ("gather_fm_fwd", "gather_fm_bwd_rows", "gather_fm_bwd_dense", "empty") if k in kernels}}
            if sweep is not None:
                roofline["batch_sweep"] = sweep
        elif sharded and "gather_fm_fwd_ride" in kernels and sharded_dgrad is not None:
            # the sharded step's local compute is the one-node form too: the slot lookup (same kernel, rows addressed by slot in
            # the receive buffer, the tail's mask riders in its launch) + the lookup backward in the epilogue of the tail's first
            # input-gradient product.  The lookup's time here is its EAGER dispatch-event time (reads ~1-2 us above the in-graph
            # wall the unsharded line quotes, and includes the riders); the epilogue's is the in-graph difference as above.
            fwd_us = kernels["gather_fm_fwd_ride"]["avg_us"]
            epi_us = max(sharded_dgrad["fm"] - sharded_dgrad["plain"], 0.0)
            pair_bytes = (fb + bb) * B
            roofline = {"bound": "hbm",
                        "kernel": "gather+FM fwd+bwd PAIR of the sharded step's local compute: the slot lookup (k_gather_fm_fwd_ride) + "
                                  "the lookup backward in the epilogue of the tail's first input-gradient product",
                        "achieved": round(pair_bytes / (fwd_us + epi_us) / 1e3, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(pair_bytes / (fwd_us + epi_us) / 1e3 / HBM_PEAK_GBS, 4), "traffic": None,
                        "avg_us": round(fwd_us + epi_us, 3), "alg_bytes": pair_bytes,
                        "attribution": {"slot_lookup_eager_dispatch_us": round(fwd_us, 3),
                                        "tail_dgrad_gemm_plain_us": round(sharded_dgrad["plain"], 3),
                                        "tail_dgrad_gemm_fm_us": round(sharded_dgrad["fm"], 3), "epilogue_us": round(epi_us, 3)},
                        "clock": "lookup: dispatch begin/end events of an eager pass (incl. the mask riders of its launch); epilogue: "
                                 "in-graph wall per launch, product with the epilogue minus without"}
        else:
            cand = [k for k in kernels if k in alg]
            dom = max(cand, key=lambda k: kernels[k]["avg_us"]) if cand else None
            if dom:
                pair_us = sum(kernels[k]["avg_us"] for k in cand)
                roofline = {"bound": "hbm", "kernel": dom, "achieved": kernels[dom]["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(kernels[dom]["GBps"] / HBM_PEAK_GBS, 4), "traffic": traffic_all.get(dom),
                            "avg_us": kernels[dom]["avg_us"], "clock": "dispatch begin/end events of an eager pass",
                            "fwd_bwd_pair": rl(pair_us, (fb + bb) * B)}
        if roofline is not None and not args.c4:
            roofline["measured_stream_ceiling"] = measured_copy_ceiling(dev)
        window_ms.sort()
        out = {
            "metric": "samples/sec fwd+bwd, Criteo-26field DeepFM b=4096; HBM GB/s vs roofline",
            "value": round(B * world * args.steps / elapsed, 1),
            "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "settle_replays_before_warmup": max(0, args.settle),
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "rccl_ranks": (collective_ranks if (sharded and not rehearse) else None),
            "collective_backend": (dist.get_backend() if sharded else None),
            "launch": (("hipGraph replay, one static input refreshed by a device copy per step" if args.copy_batch else
                        f"hipGraph replay, one graph per resident batch ({len(ring)}; ids and labels read in place)")
                       if use_graph else "eager RCCL collectives + one hipGraph for the local compute"
                                                           if (sharded and graphed_local) else "eager"),
            "config": {"workload": f"{'C4' if args.c4 else 'C2'} DeepFM Criteo-{F}field full embedding: F={F}, D={D}, N={sum(dims)} rows, "
                                   f"MLP 400x3+BN+dropout0.5, B={B}/GPU, fwd+bwd, {len(ring)} distinct {args.ids}-id batches "
                                   f"rotated (fresh ids every step), "
                                   f"{'row-form (COO)' if sparse else 'dense'} table grads, "
                                   f"{'criterion evaluated in the head launch (model(x, labels=y)), ' if labels_in_forward else ''}"
                                   f"{'next batch known one step ahead (its table rows touched by riders of the weight-gradient launch), ' if prefetching else ''}tables stored as "
                                   f"{'one packed [N,32] buffer (row + first-order weight per 128-B line)' if (args.layout == 'packed128' and not sharded) else 'row-sharded packed rows' if sharded else 'the two reference tensors'}",
                       "global_batch": B * world, "parallelism": parallelism},
            "ms_per_step_windows": ({"min": round(window_ms[0], 4), "median": round(window_ms[len(window_ms) // 2], 4),
                                     "max": round(window_ms[-1], 4), "windows": len(window_ms), "steps_per_window": args.steps}
                                    if window_ms else None),
            "roofline": roofline,
            "kernels_eager_dispatch_clock": kernels,
        }
        if sharded:
            rows0 = ring[0][:nx].view(torch.int64).view(B, F) + model.offsets
            fill = torch.bincount((rows0 % world).reshape(-1), minlength=world)
            out["sharded"] = {
                "direct_rccl": model.__dict__.get("_comm") is not None,
                "bucket_capacity_per_peer": int(model.capacity(B)), "observed_max_bucket_fill": int(fill.max()),
                "bucket_slack": model.bucket_slack, "shard_rows": int(model.num_local_rows),
                "phases_us": sharded_phases["pipelined"] if sharded_phases else None,
                "phases_us_synced": sharded_phases["synced"] if sharded_phases else None,
                "before_the_local_graph_us": (round(sum(p["us"] for p in sharded_phases["pipelined"][:4]), 2)
                                              if sharded_phases else None),
                "phases_note": "device time between HIP events placed after each phase of 20 steps enqueued back to back as in "
                               "the timed region (the host stays ahead of the device); phases_us_synced: the same with a host "
                               "sync after every step (every phase then includes the host's launch latency: rounds 1-4's "
                               "figure); at world=1 the all-to-alls are one-rank copies"}
        if not args.no_train_step and world == 1 and not sharded:
            out["train_step"] = train_step_lines(dims, D, hidden, p_drop, B, dev, args.ids, args.layout)
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
    try:
        main()
    except SystemExit:
        raise
    except BaseException:          # noqa: BLE001 - a rank that fails must say why and must not linger
        import traceback

        traceback.print_exc()
        sys.stderr.flush()
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:
            # peers may be blocked in a collective this rank will never join: leave at once with a failure status (the
            # launcher then tears the other ranks down) instead of waiting in the process group's destructors
            os._exit(1)
        raise
