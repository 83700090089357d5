"""GPU: the fused MLP tail (recsys-benchmark_amd/tail.py over csrc/tail.hip) against the reference's op sequence —
(Linear, BatchNorm1d, ReLU, Dropout) x k + Linear(., 1), src/models/deepfm.py:53-66,100-102 — evaluated in float64 on the
CPU.  The claim tested is the one the tolerances of a float32 pipeline rest on: BOTH the stock float32 modules and the
fused kernels lie within a few float32 ulps (scaled by the size of the summed terms) of the float64 value; the fused
path is held to the same multiple of the stock path's own error.  Dropout: the oracle is given the very mask the kernels
use (tests/tail_helpers.py restates the bit generator).  Integer / bookkeeping state (num_batches_tracked, masks,
bit-identical reruns) is exact."""
import copy
import os

import pytest
import torch
from torch import nn

from conftest import assert_close
from tail_helpers import tail_keep_scale

from recsys_benchmark_amd import mlp as _mlp
from recsys_benchmark_amd.mlp import run_tail
from recsys_benchmark_amd.tail import SALT, fused_tail_plan

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(autouse=True, params=["sums", "joins", "finalize"],
                ids=["atomic-sums-derived-in-prologue", "joins-in-prologue", "finalize-launches"])
def _fused_tail_on(request, monkeypatch):
    """Every test runs in the three forms of the BatchNorm statistics: shifted sums added with float atomics by the
    producing product and derived into constants by the consuming product's prologue (round 4, the default), the tile
    statistics joined in the consumer's prologue (MI_TAIL_MERGE_JOINS=1: round 3, measured slower), and tile statistics
    with finalize launches (MI_TAIL_STAT_SUMS=0; what deterministic mode runs)."""
    from recsys_benchmark_amd import tail as _tail_mod

    monkeypatch.setattr(_mlp, "FUSED_TAIL", True)
    monkeypatch.setattr(_tail_mod, "STAT_SUMS", request.param == "sums")
    monkeypatch.setattr(_tail_mod, "MERGE_JOINS", request.param == "joins")


def _seq(inp, hidden, p, bn=True):
    layers = []
    for h in hidden:
        layers += [nn.Linear(inp, h)] + ([nn.BatchNorm1d(h)] if bn else []) + [nn.ReLU(), nn.Dropout(p)]
        inp = h
    layers.append(nn.Linear(inp, 1))
    seq = nn.Sequential(*layers)
    for m in seq:
        if isinstance(m, nn.BatchNorm1d):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.3)
            m.running_mean.normal_(0, 0.3)
            m.running_var.uniform_(0.5, 1.5)
    return seq


KINK = 3e-7     # |float64 pre-activation| below which a float32 evaluation may land on either side of the ReLU kink


def _reference(seq, x, add, masks, dtype, flips=None, near_out=None):
    """The reference op sequence with explicit dropout masks, in `dtype`; returns (modules, x, add, out).
    A pre-activation within rounding of 0 lands on either side of the ReLU kink in ANY float32 evaluation (the fused products
    sum in another order than torch's) and one flip changes that unit's whole gradient contribution (one row of dx): the
    float64 pass lists the (at most 4) pre-activations nearest the kink below KINK in `near_out`, and `flips` (a set of
    positions in that list) evaluates the reference with those ReLU decisions inverted."""
    seq = copy.deepcopy(seq).to(dtype)
    x = x.detach().clone().to(dtype).requires_grad_(True)
    add = add.detach().clone().to(dtype).requires_grad_(True)
    h, li, layer = x, 0, 0
    cands = []
    for m in seq:
        if isinstance(m, nn.Dropout):
            h = h * masks[li].to(dtype)
            li += 1
        elif isinstance(m, nn.ReLU):
            pre = h.detach()
            on = pre > 0
            if near_out is not None:
                a = pre.abs().flatten()
                vals, pos = torch.topk(a, min(4, a.numel()), largest=False)
                cands += [(float(v), layer, int(q)) for v, q in zip(vals, pos) if float(v) < KINK]
            if flips:
                for j, (_, lay, q) in enumerate(flips["list"]):
                    if lay == layer and j in flips["which"]:
                        on.view(-1)[q] = ~on.view(-1)[q]
            h = h * on.to(dtype)
            layer += 1
        else:
            h = m(h)
    if near_out is not None:
        near_out.extend(sorted(cands)[:4])
    out = h + add.view(-1, 1)
    return seq, x, add, out


def _run_fused(seq, x, add, seed_value):
    dev = torch.device(DEV, 0)
    seed = _mlp._seed_word(dev)
    seed.fill_(seed_value)
    seq = copy.deepcopy(seq).to(DEV)
    xd = x.detach().clone().to(DEV).requires_grad_(True)
    ad = add.detach().clone().to(DEV).requires_grad_(True)
    assert fused_tail_plan(seq, xd, _mlp._groups(seq)) is not None, "the fused node must take this pattern"
    out = run_tail(seq, xd, last_add=ad)
    return seq, xd, ad, out


CASES = [
    (4096, 416, [400, 400, 400], 0.5),     # the headline tail
    (4096, 352, [400, 400, 400], 0.0),     # DCN-Mix width, no dropout
    (200, 48, [40, 72], 0.5),              # ragged: partial 64-row tiles, narrow column tiles
    (67, 16, [8], 0.25),                   # one hidden layer, 8 columns
    (1000, 128, [120, 264], 0.1),          # 264 columns = 3 column tiles
]


# the four ways a layer normalises: batch statistics (training-mode BatchNorm1d), running statistics (model.eval(): the
# reference's validation / inference path, scripts/deepfm/infer_deepfm.py), and none at all (use_batchnorm=False, DeepFM's
# constructor default src/models/deepfm.py:20) in training (with dropout) and in eval mode
MODES = ["bn-train", "bn-eval", "nobn-train", "nobn-eval"]


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("M,K,hidden,p", CASES)
def test_fused_tail_brackets_float64_like_the_stock_modules(M, K, hidden, p, mode, monkeypatch):
    if mode != "bn-train" and (M, K) == (4096, 352):
        pytest.skip("one full-size case per mode is enough")
    # the general path (library products through PyTorch) must not be what computes any of this
    monkeypatch.setattr(_mlp, "_LinearFn", None)
    torch.manual_seed(M + K + len(hidden))
    training = mode.endswith("train")
    seq = _seq(K, hidden, p, bn=mode.startswith("bn")).train(training)
    x = torch.randn(M, K) * 0.7 + 0.2
    add = torch.randn(M)
    G = torch.randn(M, 1)
    seed_value = 4242 + M
    if not training:
        p = 0.0
    masks = [tail_keep_scale(seed_value, SALT * (i + 1), M, h, p) for i, h in enumerate(hidden)]

    near = []
    ref64 = _reference(seq, x, add, masks, torch.float64, near_out=near)
    ref32 = _reference(seq, x, add, masks, torch.float32)
    for r in (ref64, ref32):
        (r[3] * G.to(r[3].dtype)).sum().backward()
    fs, fx, fa, fout = _run_fused(seq, x, add, seed_value)
    (fout * G.to(DEV)).sum().backward()
    lin_in_front_of_bn = {f"{i}.bias" for i, m in enumerate(seq) if isinstance(m, nn.Linear) and i + 1 < len(seq)
                          and isinstance(seq[i + 1], nn.BatchNorm1d) and training}

    def failures(ref64, k=8.0, floor=1e-6):
        """Checks |got - r64| <= k * max(|r32 - r64|, floor) (relative to max |r64|) for the output, every gradient and the
        running statistics: the fused result is as close to the float64 value as the stock float32 modules are, up to the
        factor k (different summation order / fused multiply-adds).  k = 8: over every tensor of every case of this test the
        measured ratio is at most 4.1 (MI_TEST_REPORT=1 prints them; round 2 allowed 32).  Returns the failed checks."""
        bad = []

        def check(name, got, r64, r32):
            got, r64, r32 = got.detach().double().cpu(), r64.detach().double(), r32.detach().double()
            scale = r64.abs().max().clamp_min(1e-30)
            err_f = (got - r64).abs().max() / scale
            err_s = (r32 - r64).abs().max() / scale
            if os.environ.get("MI_TEST_REPORT"):
                print(f"RATIO {name} M={M} K={K} p={p}: fused {float(err_f):.3e} stock {float(err_s):.3e} ratio {float(err_f / err_s.clamp_min(1e-30)):.2f}")
            if not err_f <= max(k * err_s, floor):
                bad.append(f"{name}: fused {err_f:.3e} vs stock {err_s:.3e} (relative to max |ref|)")

        check("out", fout, ref64[3], ref32[3])
        check("dx", fx.grad, ref64[1].grad, ref32[1].grad)
        check("dadd", fa.grad, ref64[2].grad, ref32[2].grad)
        p64, p32, pf = dict(ref64[0].named_parameters()), dict(ref32[0].named_parameters()), dict(fs.named_parameters())
        for name in p64:
            if name in lin_in_front_of_bn:
                # analytically zero (the batch mean is removed); the stock modules return rounding noise, the fused node 0
                assert float(pf[name].grad.abs().max()) == 0.0
                continue
            check(name, pf[name].grad, p64[name].grad, p32[name].grad)
        b64, bf = dict(ref64[0].named_buffers()), dict(fs.named_buffers())
        b32 = dict(ref32[0].named_buffers())
        for name in b64:
            if name.endswith("num_batches_tracked"):
                assert int(bf[name]) == int(b64[name])
            else:
                check(name, bf[name], b64[name], b32[name])
        return bad

    bad = failures(ref64)
    if bad and near:
        # pre-activations on the kink: the fused result has to match the float64 evaluation under ONE assignment of their
        # ReLU decisions (tools/fuzz.py holds its tail cases to the same rule)
        for bits in range(1, 1 << len(near)):
            alt = _reference(seq, x, add, masks, torch.float64, flips={"list": near, "which": {j for j in range(len(near)) if bits >> j & 1}})
            (alt[3] * G.double()).sum().backward()
            if not failures(alt):
                bad = []
                break
    assert not bad, f"{bad} (pre-activations within {KINK:g} of the kink: {near})"
    # the seed word advanced exactly once per step when there is dropout
    assert int(_mlp._seed_word(torch.device(DEV, 0))) == seed_value + (1 if p > 0 else 0)


def test_fused_tail_is_bit_reproducible_and_graph_capturable(monkeypatch):
    """Deterministic mode: every reduction of the fused tail is joined in a fixed order (its own slab-summed weight-gradient
    kernel instead of the split-K multi-problem launch), so two runs of a step agree bit for bit."""
    from recsys_benchmark_amd import _kernels

    monkeypatch.setattr(_kernels, "DETERMINISTIC", True)
    torch.manual_seed(3)
    M, K, hidden, p = 4096, 416, [400, 400, 400], 0.5
    seq = _seq(K, hidden, p).train().to(DEV)
    x = torch.randn(M, K, device=DEV)
    add = torch.randn(M, device=DEV)
    dev = torch.device(DEV, 0)
    state0 = copy.deepcopy(seq.state_dict())

    def step():
        seq.zero_grad(set_to_none=True)
        xd = x.clone().requires_grad_(True)
        out = run_tail(seq, xd, last_add=add)
        out.square().mean().backward()
        return out.detach().clone(), xd.grad.clone(), [q.grad.clone() for q in seq.parameters()]

    _mlp._seed_word(dev).fill_(99)
    a = step()
    seq.load_state_dict(state0)
    _mlp._seed_word(dev).fill_(99)
    b = step()
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for u, v in zip(a[2], b[2]):
        assert torch.equal(u, v)                      # no atomics anywhere: bitwise identical
    # captured into a hipGraph: the replay gives what the eager step gives from the same state and seed
    seq.load_state_dict(state0)
    _mlp._seed_word(dev).fill_(99)
    xs = x.clone().requires_grad_(True)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            run_tail(seq, xs, last_add=add).square().mean().backward()
    torch.cuda.current_stream().wait_stream(side)
    seq.zero_grad(set_to_none=True)
    xs.grad = None
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = run_tail(seq, xs, last_add=add)
        out.square().mean().backward()
    seq.load_state_dict(state0)
    _mlp._seed_word(dev).fill_(99)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, a[0]) and torch.equal(xs.grad, a[1])


@pytest.mark.parametrize("seed_value", [None, 0.25])
@pytest.mark.parametrize("M,K,hidden,p", [(4096, 416, [400, 400, 400], 0.5), (300, 64, [96], 0.0), (64, 32, [512, 32], 0.3)])
def test_head_launch_with_the_criterion_and_a_named_upstream_seed(M, K, hidden, p, seed_value):
    """run_tail(..., labels=y, loss_seed=s): the head launch evaluates BCE-with-logits and the head's backward for an
    upstream gradient s (a device scalar; None = the resident 1) — the table-sharded step seeds 1 / world.  Against the same
    step with the labels withheld, the criterion as its own launch and the same seed, on the same dropout seed."""
    from recsys_benchmark_amd.losses import BCEWithLogitsLoss, unit_scalar

    torch.manual_seed(M + K)
    seq0 = _seq(K, hidden, p).train()
    x = torch.randn(M, K)
    add = torch.randn(M)
    y = (torch.rand(M) < 0.4).float()
    dev = torch.device(DEV, 0)
    up = unit_scalar(dev) if seed_value is None else torch.full((), seed_value, device=dev)
    res = {}
    for fused in (True, False):
        seq = copy.deepcopy(seq0).to(DEV)
        xd = x.to(DEV).requires_grad_(True)
        yd, ad = y.to(DEV), add.to(DEV).requires_grad_(True)
        _mlp._seed_word(dev).fill_(1234)
        out = run_tail(seq, xd, last_add=ad, labels=yd if fused else None, loss_seed=up if (fused and seed_value is not None) else None)
        loss = BCEWithLogitsLoss()(out.squeeze(-1), yd)
        # (the head launch carries the criterion only in the default form of the statistics: its backward sums go by atomics)
        from recsys_benchmark_amd import tail as _tail_mod
        assert (type(loss.grad_fn).__name__ == "_HeadBCEFnBackward") == (fused and _tail_mod.STAT_SUMS and not _tail_mod.MERGE_JOINS)
        loss.backward(up)
        res[fused] = (out.detach(), loss.detach(), xd.grad, ad.grad, [q.grad for q in seq.parameters()])
    a, b = res[True], res[False]
    assert_close(a[0], b[0], 1e-5, 1e-6, "logits")
    assert_close(a[1], b[1], 1e-5, 1e-6, "loss")
    assert_close(a[2], b[2], 2e-4, 1e-7, "dx")
    assert_close(a[3], b[3], 1e-5, 1e-9, "d last_add")
    for u, v, (name, _) in zip(a[4], b[4], copy.deepcopy(seq0).named_parameters()):
        assert_close(u, v, 2e-4, 1e-6 + 2e-8 * M, name)


def test_fused_tail_default_and_deterministic_weight_gradients_agree():
    """Default mode keeps a(z) and dz as the operand loads computed them and runs the three weight gradients as ONE
    multi-problem launch (split-K atomics); deterministic mode recomputes them inside its own kernel.  Same numbers up to
    summation order; everything that does not pass through the split (outputs, input gradient, BatchNorm gradients) is
    bit-identical."""
    from recsys_benchmark_amd import _kernels

    torch.manual_seed(5)
    M, K, hidden, p = 4096, 416, [400, 400, 400], 0.5
    seq0 = _seq(K, hidden, p).train().to(DEV)
    x = torch.randn(M, K, device=DEV)
    add = torch.randn(M, device=DEV)
    res = {}
    for det in (False, True):
        _kernels.DETERMINISTIC = det
        try:
            seq = copy.deepcopy(seq0)
            _mlp._seed_word(torch.device(DEV, 0)).fill_(77)
            xd = x.clone().requires_grad_(True)
            out = run_tail(seq, xd, last_add=add)
            out.square().mean().backward()
            res[det] = (out.detach(), xd.grad, {n: q.grad for n, q in seq.named_parameters()})
        finally:
            _kernels.DETERMINISTIC = False
    from recsys_benchmark_amd import tail as _tail_mod

    a, b = res[False], res[True]

    def same(u, v, what):
        if _tail_mod.STAT_SUMS:
            # the default accumulates the BatchNorm statistics with float atomics (order varies in the last bits);
            # deterministic mode joins tile statistics in a fixed order: equal within float32 rounding, not bitwise
            assert float((u - v).abs().max()) <= 2e-5 * float(v.abs().max().clamp_min(1e-30)), what
        else:
            assert torch.equal(u, v), what

    same(a[0], b[0], "out")
    same(a[1], b[1], "dx")
    for name in a[2]:
        ga, gb = a[2][name], b[2][name]
        if name.endswith("weight") and ga.dim() == 2 and ga.shape[0] > 1:          # a hidden Linear's weight: through the split
            scale = float(gb.abs().max())
            assert float((ga - gb).abs().max()) <= 2e-5 * scale, name
        else:
            same(ga, gb, name)


def test_inference_without_grad_runs_the_own_kernels_and_keeps_nothing(monkeypatch):
    """torch.no_grad() + eval(): what the reference's inference scripts time (scripts/deepfm/infer_deepfm.py:318-352, batch
    64 by default; one sample must work too).  Same values as the stock modules, no general-path product, no state change."""
    monkeypatch.setattr(_mlp, "_LinearFn", None)
    for bn in (True, False):
        torch.manual_seed(3)
        seq = _seq(416, [400, 400, 400], 0.5, bn=bn).to(DEV).eval()
        before = copy.deepcopy(seq.state_dict())
        for M in (1, 64, 4096):
            x, add = torch.randn(M, 416, device=DEV), torch.randn(M, device=DEV)
            with torch.no_grad():
                out = run_tail(seq, x, last_add=add)
                ref = seq.double()(x.double()) + add.double().view(-1, 1)
                seq.float()
            assert out.shape == (M, 1) and not out.requires_grad
            assert float((out.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max().clamp_min(1.0))
        for k, v in seq.state_dict().items():
            assert torch.equal(v, before[k]), k


def test_patterns_outside_the_fused_node_keep_the_general_path():
    seq = _seq(16, [8], 0.0).to(DEV)
    x = torch.randn(32, 16, device=DEV)
    plan = fused_tail_plan(seq.eval(), x, _mlp._groups(seq))                    # eval-mode BatchNorm: fixed statistics
    assert plan is not None and all(L.fixed for L in plan) and plan.grad
    seq.train()
    plan = fused_tail_plan(seq, x, _mlp._groups(seq))
    assert plan is not None and not any(L.fixed for L in plan)
    with torch.no_grad():
        plan = fused_tail_plan(seq, x, _mlp._groups(seq))                      # no autograd: nothing is kept for a backward
        assert plan is not None and not plan.grad
    assert fused_tail_plan(seq, x[:1], _mlp._groups(seq)) is None               # batch statistics need two samples
    odd = _seq(16, [12], 0.0).to(DEV).train()                                   # 12 columns: not a multiple of 8
    assert fused_tail_plan(odd, x, _mlp._groups(odd)) is None
    out = run_tail(odd, x)                                                      # still runs (general path)
    assert out.shape == (32, 1)


def test_dropout_bits_match_the_restated_generator_and_rate():
    import ctypes

    from recsys_benchmark_amd import _lib

    lib = _lib.load()
    M, ld, p = 333, 40, 0.3
    seed = torch.tensor([777], dtype=torch.int64, device=DEV)
    bits = torch.empty(M * ld // 8, dtype=torch.uint8, device=DEV)
    salts, ps, lds = (ctypes.c_int64 * 1)(5), (ctypes.c_float * 1)(p), (ctypes.c_int32 * 1)(ld)
    ptrs = (ctypes.c_void_p * 1)(bits.data_ptr())
    _lib.check(lib.mi_tail_dropout_masks(seed.data_ptr(), 1, ctypes.addressof(salts), ctypes.addressof(ps), ctypes.addressof(lds),
                                         ctypes.addressof(ptrs), M, _lib.stream_ptr(torch.device(DEV, 0))), "masks")
    want = tail_keep_scale(777, 5, M, ld, p) > 0
    got = ((bits.cpu().view(-1, 1) >> torch.arange(8, dtype=torch.uint8)) & 1).bool().view(M, ld)
    assert torch.equal(got, want)                                               # integer work: bit-exact
    big = tail_keep_scale(1, 2, 4096, 400, 0.5) > 0
    assert abs(float(big.float().mean()) - 0.5) < 5e-3
