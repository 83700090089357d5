"""The MLP tail of DeepFM / DCN: (Linear, [BatchNorm1d], ReLU, Dropout) x k +
Linear — reference: src/models/deepfm.py:53-66,100-102 and src/models/dcn.py:56-66.

The modules stay ordinary nn.Linear / nn.BatchNorm1d / nn.ReLU / nn.Dropout inside the same
nn.Sequential (state_dict keys unchanged); `run_tail` walks the Sequential.

A tail of (Linear, [BatchNorm1d], ReLU, [Dropout]) groups with a 1-output last Linear — the
reference's configs — runs on the library's own fused MFMA kernels (tail.py / csrc/tail.hip, see
FUSED_TAIL below) in training, eval() and without BatchNorm alike (round 4).  The patterns those
kernels do not take (widths that are not multiples of 8, headless stacks) use the GENERAL path of this file: each
(Linear, [BatchNorm1d], ReLU, [Dropout]) group is the contraction on hipBLASLt/rocBLAS through
PyTorch, then ONE fused BN+ReLU+Dropout HIP pass each way (mi_bn_relu_dropout_*).
Launch count matters at B=4096 (every kernel is a few microseconds), so: all reduction targets
of a pass live in ONE zero-filled workspace, the dropout-seed and num_batches_tracked bumps ride
inside the library's kernels, and the bias gradient of a Linear that feeds a training-mode
BatchNorm is returned as the exact zero it is (sum_m dz = 0 when the batch mean is removed)
instead of being reduced.  Anything that does not match a group pattern runs as the plain module.
"""
import os
from typing import Dict, List, Optional

import torch
from torch import nn

from . import _kernels, _lib

_seeds: Dict[int, torch.Tensor] = {}


def _seed_word(dev: torch.device) -> torch.Tensor:
    i = dev.index if dev.index is not None else torch.cuda.current_device()
    w = _seeds.get(i)
    if w is None:
        w = torch.tensor([torch.initial_seed() & 0x7FFFFFFFFFFF], dtype=torch.int64, device=torch.device("cuda", i))
        _seeds[i] = w
    return w


# PyTorch's TunableOp searches the rocBLAS / hipBLASLt solutions for each GEMM shape at its first call (~1 s per
# shape) and then always launches the fastest.  Measured on the tail's shapes: input-gradient
# 20 -> 16.5 us, weight-gradient 27 -> 25 us, but the bias-fused forward gets slower — so it is applied to the two
# backward products only, and only when a training loop opts in (it makes the first step of every new shape slow).
TUNE_BACKWARD_GEMMS = False

# FUSED_TAIL (default on; MI_FUSED_TAIL=0 or mlp.FUSED_TAIL = False turns it off): a training-mode
# (Linear, BatchNorm1d, ReLU, Dropout) x k + Linear(., 1) tail runs as ONE autograd node over the fused MFMA kernels of
# csrc/tail.hip (tail.py) — the contractions are the library's OWN kernels, BatchNorm / ReLU / Dropout sit inside their
# operand loads and epilogues, the weight gradients are one multi-problem launch; with use_deterministic_algorithms(True)
# there are no atomics at all (bit-reproducible steps).  Measured on MI355X, same box, A/B, round 2: DeepFM headline 0.2906
# vs 0.2886 ms per step for the general path below (hipBLASLt / rocBLAS products + the fused passes of csrc/mlp.hip) —
# level, without TunableOp's per-shape search at start-up; round 4: 0.226 ms (one-node step, statistics by atomics, criterion
# in the head launch: DESIGN.md §4) against 0.289 for the general path.  Eval-mode and no-BatchNorm stacks run on the same
# kernels with fixed per-column constants; tails that do not fit (widths not multiples of 8, ...) take the general path.
FUSED_TAIL = os.environ.get("MI_FUSED_TAIL", "1") != "0"


class _tuned_gemms:
    _named = False

    def __enter__(self):
        self.on = bool(TUNE_BACKWARD_GEMMS)
        if self.on:
            self.was = torch.cuda.tunable.is_enabled()
            if not _tuned_gemms._named:      # keep its results file out of the working directory
                import os
                import tempfile

                torch.cuda.tunable.set_filename(os.path.join(tempfile.gettempdir(), "mi355x_recsys_tunableop.csv"), True)
                torch.cuda.tunable.set_max_tuning_duration(10)       # ms per candidate: ~20 us GEMMs need no more
                torch.cuda.tunable.set_max_tuning_iterations(50)
                _tuned_gemms._named = True
            torch.cuda.tunable.enable(True)
        return self

    def __exit__(self, *exc):
        if self.on:
            torch.cuda.tunable.enable(self.was)
        return False


def _pad4(n: int) -> int:
    return (n + 3) // 4 * 4


class _LinearFn(torch.autograd.Function):
    """z = x W^T + b on rocBLAS/hipBLASLt.  db: the library's column sum into `db_buf` (a zeroed
    workspace slice), or — zero_db — that zeroed slice itself when z feeds a training BatchNorm."""

    @staticmethod
    def forward(ctx, x, W, b, db_buf, zero_db: bool, skip_bias: bool = False):
        """skip_bias: z feeds a training-mode BatchNorm, where the bias cancels (it shifts z and its batch mean
        alike): the contraction runs without it — a plain GEMM, which TunableOp serves better than the
        bias-fused one — and the BatchNorm pass adds it to the running mean (mean_offset)."""
        ctx.save_for_backward(x, W, db_buf)
        ctx.has_bias = b is not None
        ctx.zero_db = zero_db
        if b is None or skip_bias:
            with _tuned_gemms():
                return x @ W.t()
        return torch.addmm(b, x, W.t())

    @staticmethod
    def backward(ctx, g):
        x, W, db_buf = ctx.saved_tensors
        g = g.contiguous()
        with _tuned_gemms():
            dx = g @ W if ctx.needs_input_grad[0] else None
            dW = g.t() @ x if ctx.needs_input_grad[1] else None
        db = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            M, N = g.shape
            db = db_buf if db_buf is not None else torch.zeros((N,), dtype=torch.float32, device=g.device)
            if not ctx.zero_db:
                _lib.check(_lib.load().mi_colsum(g.data_ptr(), N, None, 0, db.data_ptr(), None, M, N,
                                                 _lib.stream_ptr(g.device)), "mi_colsum")
        return dx, dW, db, None, None, None


class _Linear1Fn(torch.autograd.Function):
    """A Linear with ONE output (the tail's last layer): a rank-1 layer is three memory-bound passes
    (row-dot; scaled column sum for dW; outer product for dx), not three GEMM launches —
    hipBLASLt spent ~40 us per step on it at [4096, 400] (rocprof r01)."""

    @staticmethod
    def forward(ctx, x, W, b, addend, red_buf=None):
        """addend (optional, [M]): added to the output row-wise inside the same kernel (DeepFM's y_fm).
        red_buf (optional): N+1 zeroed floats of the pass workspace for the backward's two column sums."""
        dev = _lib.require_gpu(x, W)
        x = _kernels._f32c(x)
        M, N = x.shape
        out = torch.empty((M, 1), dtype=torch.float32, device=dev)
        add = None if addend is None else _kernels._f32c(addend)
        _lib.check(_lib.load().mi_rowdot(x.data_ptr(), N, W.data_ptr(), _lib.ptr(b), _lib.ptr(add), out.data_ptr(), M, N,
                                         _lib.stream_ptr(dev)), "mi_rowdot")
        ctx.save_for_backward(x, W, red_buf)
        ctx.has_bias = b is not None
        ctx.add_shape = None if addend is None else tuple(addend.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        x, W, red_buf = ctx.saved_tensors
        dev = x.device
        M, N = x.shape
        g = _kernels._f32c(g).view(M)
        lib, s = _lib.load(), _lib.stream_ptr(dev)
        dx = dW = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((M, N), dtype=torch.float32, device=dev)
            _lib.check(lib.mi_outer(g.data_ptr(), W.data_ptr(), dx.data_ptr(), M, N, s), "mi_outer")
        red = red_buf if red_buf is not None else torch.zeros((N + 1,), dtype=torch.float32, device=dev)
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            # dW = g^T x and db = sum g in ONE launch (db is the sum of the very row scales dW is weighted with)
            _lib.check(lib.mi_colsum(x.data_ptr(), N, g.data_ptr(), 1, red.data_ptr(),
                                     red[N:].data_ptr() if want_db else None, M, N, s), "mi_colsum")
            dW = red[:N].view(1, N)
        elif want_db:
            _lib.check(lib.mi_colsum(g.data_ptr(), 1, None, 0, red[N:].data_ptr(), None, M, 1, s), "mi_colsum")
        if want_db:
            db = red[N:]
        dadd = g.view(ctx.add_shape) if (ctx.add_shape is not None and ctx.needs_input_grad[3]) else None
        return dx, dW, db, dadd, None


class _BNReLUDropFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, gamma, beta, running_mean, running_var, nbt, has_bn, training, momentum, eps, p, seed, salt,
                bump_seed, stats, dgb, mean_offset=None, lin1_W=None, lin1_b=None, lin1_add=None, lin1_red=None):
        """lin1_* (optional): the tail's final 1-output Linear (weight [1, N], bias, row-wise addend, N+1 zeroed floats)
        rides in the same autograd node: the node returns that layer's [M, 1] output, and its backward feeds the
        rank-1 gradient g[m] * W[n] to the BatchNorm backward kernels directly instead of writing it out."""
        dev = _lib.require_gpu(z)
        z = _kernels._f32c(z)
        M, N = z.shape
        bn_train = bool(has_bn and training)
        drop = bool(training and p > 0.0)
        y = torch.empty_like(z)
        keep = torch.empty((M, N), dtype=torch.uint8, device=dev) if drop else None
        if bn_train and stats is None:
            stats = torch.zeros((2 * N,), dtype=torch.float32, device=dev)
        save = torch.empty((2, N), dtype=torch.float32, device=dev) if has_bn else None
        _lib.check(
            _lib.load().mi_bn_relu_dropout_fwd(
                z.data_ptr(), N, M, N, int(has_bn), int(training), _lib.ptr(gamma), _lib.ptr(beta),
                _lib.ptr(running_mean), _lib.ptr(running_var), float(momentum), float(eps), float(p if drop else 0.0),
                _lib.ptr(seed), int(salt), int(bool(bump_seed and bn_train)), _lib.ptr(nbt) if bn_train else None,
                _lib.ptr(stats) if bn_train else None, _lib.ptr(mean_offset) if bn_train else None, y.data_ptr(),
                _lib.ptr(keep),
                save[0].data_ptr() if has_bn else None, save[1].data_ptr() if has_bn else None, _lib.stream_ptr(dev)),
            "mi_bn_relu_dropout_fwd",
        )
        ctx.meta = (M, N, bool(has_bn), bool(training), float(p if drop else 0.0))
        ctx.lin1 = lin1_W is not None
        if not ctx.lin1:
            ctx.save_for_backward(z, gamma, beta, keep, save, dgb)
            return y
        out = torch.empty((M, 1), dtype=torch.float32, device=dev)
        add = None if lin1_add is None else _kernels._f32c(lin1_add)
        _lib.check(_lib.load().mi_rowdot(y.data_ptr(), N, lin1_W.data_ptr(), _lib.ptr(lin1_b), _lib.ptr(add),
                                         out.data_ptr(), M, N, _lib.stream_ptr(dev)), "mi_rowdot")
        ctx.save_for_backward(z, gamma, beta, keep, save, dgb, y, lin1_W, lin1_red)
        ctx.lin1_meta = (lin1_b is not None, None if lin1_add is None else tuple(lin1_add.shape))
        return out

    @staticmethod
    def backward(ctx, dy):
        M, N, has_bn, training, p = ctx.meta
        lib = _lib.load()
        dW1 = db1 = dadd = gvec = W1 = None
        if ctx.lin1:
            z, gamma, beta, keep, save, dgb, y, W1, red = ctx.saved_tensors
            dev = z.device
            s = _lib.stream_ptr(dev)
            has_b1, add_shape = ctx.lin1_meta
            gvec = _kernels._f32c(dy).view(M)
            if red is None:
                red = torch.zeros((N + 1,), dtype=torch.float32, device=dev)
            want_db1 = has_b1 and ctx.needs_input_grad[18]
            if ctx.needs_input_grad[17]:
                _lib.check(lib.mi_colsum(y.data_ptr(), N, gvec.data_ptr(), 1, red.data_ptr(),
                                         red[N:].data_ptr() if want_db1 else None, M, N, s), "mi_colsum")
                dW1 = red[:N].view(1, N)
            elif want_db1:
                _lib.check(lib.mi_colsum(gvec.data_ptr(), 1, None, 0, red[N:].data_ptr(), None, M, 1, s), "mi_colsum")
            if want_db1:
                db1 = red[N:]
            if add_shape is not None and ctx.needs_input_grad[19]:
                dadd = gvec.view(add_shape)
            dy_ptr = None
        else:
            z, gamma, beta, keep, save, dgb = ctx.saved_tensors
            dev = z.device
            dy = _kernels._f32c(dy)
            dy_ptr = dy.data_ptr()
        dz = torch.empty_like(z)
        if has_bn and dgb is None:
            dgb = torch.zeros((2 * N,), dtype=torch.float32, device=dev)
        _lib.check(
            lib.mi_bn_relu_dropout_bwd(
                dy_ptr, z.data_ptr(), N, M, N, int(has_bn), int(training), _lib.ptr(keep), p, _lib.ptr(gamma),
                _lib.ptr(beta), save[0].data_ptr() if has_bn else None, save[1].data_ptr() if has_bn else None,
                _lib.ptr(dgb), dz.data_ptr(), _lib.ptr(gvec), _lib.ptr(W1), _lib.stream_ptr(dev)),
            "mi_bn_relu_dropout_bwd",
        )
        dgamma = dgb[:N] if (has_bn and gamma is not None and ctx.needs_input_grad[1]) else None
        dbeta = dgb[N:2 * N] if (has_bn and beta is not None and ctx.needs_input_grad[2]) else None
        return (dz, dgamma, dbeta) + (None,) * 14 + (dW1, db1, dadd, None)


def hidden_stack(width: int, hidden_sizes, p_dropout: float, use_batchnorm: bool = True):
    """[Linear, (BatchNorm1d), ReLU, Dropout] per hidden size (src/models/deepfm.py:53-64, src/models/dcn.py:56-64);
    returns (modules, output width)."""
    mods: List[nn.Module] = []
    for h in hidden_sizes:
        mods.append(nn.Linear(width, h))
        if use_batchnorm:
            mods.append(nn.BatchNorm1d(h))
        mods.extend((nn.ReLU(), nn.Dropout(p_dropout)))
        width = h
    return mods, width


def field_offsets(field_dims) -> torch.Tensor:
    """int64[1, F]: the first row of every field in the concatenated table (src/models/deepfm.py:71-76)."""
    starts = torch.tensor([0] + list(field_dims[:-1]), dtype=torch.long).cumsum(0)
    return starts.unsqueeze(0)


def _groups(seq: nn.Sequential) -> List[List]:
    """Split the Sequential into fusable (Linear, [BN], ReLU, [Dropout]) groups and single modules."""
    mods = list(seq)
    out, i = [], 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, nn.Linear):
            j = i + 1
            bn = mods[j] if j < len(mods) and isinstance(mods[j], nn.BatchNorm1d) else None
            if bn is not None:
                j += 1
            if j < len(mods) and isinstance(mods[j], nn.ReLU):
                j += 1
                dp = mods[j] if j < len(mods) and isinstance(mods[j], nn.Dropout) else None
                if dp is not None:
                    j += 1
                if bn is None or (bn.momentum is not None and bn.track_running_stats):
                    out.append(["fused", m, bn, dp])
                    i = j
                    continue
        out.append(["plain", m])
        i += 1
    return out


def run_tail(seq: nn.Sequential, x: torch.Tensor, last_add: Optional[torch.Tensor] = None,
             labels: Optional[torch.Tensor] = None, loss_seed: Optional[torch.Tensor] = None) -> torch.Tensor:
    """last_add ([B], optional): added to the tail's [B,1] output inside its last kernel when that layer is a
    1-output Linear (DeepFM: scores = y_fm + deep(emb)); otherwise added with a plain op.
    labels ([B], optional): the step's targets when the caller goes on to BCEWithLogitsLoss(out.squeeze(-1), labels) — the
    fused tail's head launch then evaluates that criterion and the head's backward as well (tail.mi_tail_head_bce); ignored
    everywhere else.  loss_seed (device scalar, optional): the gradient the caller will seed that criterion's backward with
    (default: losses.unit_scalar — `loss.backward(unit_scalar(dev))`)."""
    dev = x.device
    groups = _groups(seq)
    if FUSED_TAIL and x.is_cuda:
        from . import tail as _tail

        plan = _tail.fused_tail_plan(seq, x, groups)
        if plan is not None:       # training-mode BatchNorm tail ending in Linear(., 1): one node over csrc/tail.hip
            return _tail.run_fused_tail(plan, groups[-1][1], _seed_word(dev), x, last_add,
                                        labels if seq.training else None, loss_seed)
    seed = _seed_word(dev)
    # one zero-filled workspace for every reduction target of this pass (forward statistics,
    # backward dgamma/dbeta, bias gradients): a single fill launch instead of one per buffer
    need = 0
    for grp in groups:
        lin = grp[1]
        if isinstance(lin, nn.Linear):
            need += _pad4(lin.out_features) * (5 if grp[0] == "fused" else 1)
            if grp[0] == "plain" and lin.out_features == 1:
                need += _pad4(lin.in_features + 1)       # the rank-1 layer's backward column sums
    ws = torch.zeros((need,), dtype=torch.float32, device=dev) if (need and torch.is_grad_enabled()) else None
    if ws is None and need:
        ws = torch.zeros((need,), dtype=torch.float32, device=dev)
    off = 0

    def take(n: int) -> Optional[torch.Tensor]:
        nonlocal off
        t = ws[off:off + n]
        off += _pad4(n)
        return t

    need_bump = True
    skip_next = False
    for k, grp in enumerate(groups):
        if skip_next:                       # the final 1-output Linear already ran inside the previous group's node
            skip_next = False
            continue
        if grp[0] == "plain":
            m = grp[1]
            if isinstance(m, nn.Linear) and m.out_features == 1 and x.dim() == 2:
                red = take(m.in_features + 1)
                fuse = last_add is not None and k == len(groups) - 1
                x = _Linear1Fn.apply(x, m.weight, m.bias, last_add if fuse else None, red)
                if fuse:
                    last_add = None
            elif isinstance(m, nn.Linear):
                x = _LinearFn.apply(x, m.weight, m.bias, take(m.out_features), False)
            else:
                x = m(x)
            continue
        _, lin, bn, dp = grp
        N = lin.out_features
        training_bn = bn is not None and bn.training
        p = dp.p if (dp is not None and dp.training) else 0.0
        skip_bias = bool(training_bn and lin.bias is not None)
        z = _LinearFn.apply(x, lin.weight, lin.bias, take(N), bool(training_bn), skip_bias)
        stats, dgb = take(2 * N), take(2 * N)
        bump_in_kernel = False
        if p > 0.0 and need_bump:
            if training_bn:
                bump_in_kernel = True      # the statistics launch does seed += 1
            else:
                seed.add_(1)               # no statistics launch to ride on
            need_bump = False
        # the tail's final Linear(., 1) directly behind this group: same autograd node (its rank-1 input gradient is
        # consumed by the BatchNorm backward kernels without ever being written out)
        lin1 = None
        if k + 2 == len(groups) and groups[k + 1][0] == "plain" and torch.is_grad_enabled():
            nxt = groups[k + 1][1]
            if isinstance(nxt, nn.Linear) and nxt.out_features == 1 and nxt.in_features == N and z.dim() == 2:
                lin1 = nxt
        fn_args = (
            z, bn.weight if bn is not None else None, bn.bias if bn is not None else None,
            bn.running_mean if bn is not None else None, bn.running_var if bn is not None else None,
            bn.num_batches_tracked if bn is not None else None,
            bn is not None, bool(p > 0.0) if bn is None else bool(bn.training),
            bn.momentum if bn is not None else 0.0, bn.eps if bn is not None else 0.0, p, seed, 7919 * (k + 1),
            bump_in_kernel, stats, dgb, lin.bias if skip_bias else None)
        if lin1 is not None:
            x = _BNReLUDropFn.apply(*fn_args, lin1.weight, lin1.bias, last_add, take(N + 1))
            last_add = None
            skip_next = True
        else:
            x = _BNReLUDropFn.apply(*fn_args)
    if last_add is not None:
        x = x + last_add.view(-1, *([1] * (x.dim() - 1)))
    return x
