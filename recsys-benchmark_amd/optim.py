"""Row-sparse optimizers for the row-form (COO) table gradients + the reference's optimizer factory.

SURVEY.md §8f rank 1.  `SparseAdam` has torch.optim.SparseAdam's constructor, state
(`step`, `exp_avg`, `exp_avg_sq`) and update rule, but consumes the uncoalesced COO gradient the
lookup kernels emit directly: one torch.sort of the row ids (no host sync) + ONE fused HIP kernel
(mi_sparse_adam_sorted) instead of coalesce + ~10 sparse tensor ops.  `SparseSGD` is
`p[rows] -= lr * g` as float-atomic adds (linear, so duplicates need no coalescing).
`get_optimizers(model, config)` mirrors src/models/deepfm.py:155-219.
"""
import ctypes
import math
from typing import Dict, List

import torch

from . import _kernels, _lib


def _coo_parts(grad: torch.Tensor):
    if not grad.is_sparse:
        raise RuntimeError("this optimizer consumes row-form (sparse COO) gradients; build the embedding with "
                           "sparse=True or use a dense optimizer")
    rows = grad._indices()[0].contiguous()
    vals = grad._values().contiguous()
    return rows, vals.view(vals.shape[0], -1)


def sort_rows(rows: torch.Tensor, N: int):
    """(sorted row ids, permutation) of a row-form gradient's ids.  Ids produced by a multi-field lookup are sorted
    field by field in LDS (one launch, ≈4x faster than a generic sort at the headline shape); anything else by
    torch.sort (stable is not needed for correctness, only the grouping is)."""
    _lib.require_gpu(rows)
    found = _kernels.sort_field_rows(rows, N)
    return found if found is not None else tuple(torch.sort(rows))


class SparseAdam(torch.optim.Optimizer):
    """torch.optim.SparseAdam's arguments plus `capturable`: the step count and the bias-corrected step size live on the
    device (state["step"] becomes a 0-dim float tensor, as in torch's capturable Adam) and are advanced by a one-thread
    kernel, so step() has no host-side state and can be captured into a hipGraph (trainer.GraphedTrainStep)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, capturable=False):
        if not 0.0 < lr:
            raise ValueError(f"Invalid learning rate: {lr}")
        if not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0:
            raise ValueError(f"Invalid betas: {betas}")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, capturable=bool(capturable)))
        self._workspace = {}

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        lib = _lib.load()
        sorted_ids = {}      # both DeepFM tables receive gradients over the same ids: sort them once per step
        for group in self.param_groups:
            beta1, beta2 = group["betas"]
            capturable = group.get("capturable", False)
            todo = []
            for p in group["params"]:
                if p.grad is None:
                    continue
                dev = _lib.require_gpu(p)
                rows, vals = _coo_parts(p.grad)
                state = self.state[p]
                if not state:
                    state["step"] = torch.zeros((), dtype=torch.float32, device=dev) if capturable else 0
                    # contiguous [N, D] also for a parameter that is a view of a packed table (DeepFM.pack_tables())
                    state["exp_avg"] = torch.zeros(p.shape, dtype=p.dtype, device=dev)
                    state["exp_avg_sq"] = torch.zeros(p.shape, dtype=p.dtype, device=dev)
                todo.append((p, dev, rows, vals, state))
            if not todo:
                continue
            if capturable:          # one launch advances every table's step count and bias-corrected step size
                for p, dev, *_ in todo:
                    if (p, "step_size") not in self._workspace:
                        self._workspace[(p, "step_size")] = torch.zeros(1, dtype=torch.float32, device=dev)
                n = len(todo)
                arr = ctypes.c_void_p * n
                _lib.check(lib.mi_adam_tick_multi(arr(*[st["step"].data_ptr() for *_, st in todo]),
                                                  arr(*[self._workspace[(p, "step_size")].data_ptr() for p, *_ in todo]), n,
                                                  group["lr"], beta1, beta2, _lib.stream_ptr(todo[0][1])), "mi_adam_tick_multi")
            for p, dev, rows, vals, state in todo:
                stream = _lib.stream_ptr(dev)
                if capturable:
                    step_size, step_size_dev = 0.0, self._workspace[(p, "step_size")].data_ptr()
                else:
                    state["step"] += 1
                    t = state["step"]
                    step_size, step_size_dev = group["lr"] * math.sqrt(1 - beta2 ** t) / (1 - beta1 ** t), None
                N = p.shape[0]
                D = p.numel() // N
                key = (rows.data_ptr(), rows.numel(), N)
                if key not in sorted_ids:
                    sorted_ids[key] = sort_rows(rows, N) if rows.numel() else (rows, rows)
                rows_sorted, perm = sorted_ids[key]
                acc = self._workspace.get(p)           # scratch, not optimizer state
                if acc is None or acc.numel() < vals.numel():
                    acc = self._workspace[p] = torch.empty(vals.numel(), dtype=torch.float32, device=dev)
                ldw = D if p.is_contiguous() else _kernels._row_strided(p.view(N, D), align=4 if D >= 4 else 1)[1]
                if not p.is_contiguous() and (p.dim() != 2 or p.stride(0) != ldw):
                    raise RuntimeError("SparseAdam: a non-contiguous parameter must be a row-strided 2-D view (packed table)")
                _lib.check(
                    lib.mi_sparse_adam_sorted_ld(rows_sorted.data_ptr(), perm.data_ptr(), vals.data_ptr(), p.data_ptr(), ldw,
                                                 state["exp_avg"].data_ptr(), state["exp_avg_sq"].data_ptr(),
                                                 acc.data_ptr(), rows.numel(), D, N, step_size, step_size_dev, beta1, beta2,
                                                 group["eps"], stream),
                    "mi_sparse_adam_sorted_ld",
                )
        return loss


class SparseSGD(torch.optim.Optimizer):
    """p[rows] -= lr * g for row-form gradients (weight_decay is 0 for the embedding group in the
    reference's sparse SGD branch, src/models/deepfm.py:206-214)."""

    def __init__(self, params, lr=1e-2):
        super().__init__(params, dict(lr=lr))

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        lib = _lib.load()
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                dev = _lib.require_gpu(p)
                rows, vals = _coo_parts(p.grad)
                N = p.shape[0]
                if not p.is_contiguous():       # a view of a packed table (DeepFM.pack_tables()): torch's strided scatter-add
                    p.index_add_(0, rows, vals.view((rows.numel(),) + tuple(p.shape[1:])), alpha=-group["lr"])
                    continue
                _lib.check(lib.mi_scatter_axpy_rows(rows.data_ptr(), vals.data_ptr(), -group["lr"], p.data_ptr(),
                                                    rows.numel(), p.numel() // N, N, _lib.stream_ptr(dev)),
                           "mi_scatter_axpy_rows")
        return loss


class Adam(torch.optim.Adam):
    """torch.optim.Adam (same constructor, param_groups and state_dict: it IS a torch.optim.Adam with `capturable=True`,
    i.e. device-side step counts) whose step over fp32 GPU parameters is mi_adam_dense_multi: one launch per 24 tensors,
    one read and one write of p, m, v.  At the headline model: the 574 M-parameter dense config 12.5 ms (torch's default
    multi-pass implementation) / 3.8 ms (torch `fused=True`) -> see DESIGN.md §5; the MLP's 0.5 M parameters 42 us
    (torch fused, 8 workgroups) -> one 5 us launch.  Options the kernel does not implement (amsgrad, maximize, decoupled
    weight decay, gradient scaling, non-fp32 or CPU parameters) take torch's own step."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, **kwargs):
        params = list(params)
        tensors = [t for g in params for t in (g["params"] if isinstance(g, dict) else [g[1] if isinstance(g, tuple) else g])]
        # device-side step counts only where they are supported (torch rejects capturable=True for CPU parameters)
        kwargs.setdefault("capturable", bool(tensors) and all(t.is_cuda for t in tensors))
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, **kwargs)

    def _tickets(self, dev, n_tensors: int) -> torch.Tensor:
        """Completion tickets of the launches (zero between steps): the last workgroup of a launch advances the counts."""
        need = -(-n_tensors // 24)
        t = self.__dict__.get("_ticket_buf")
        if t is None or t.numel() < need or t.device != dev:
            t = self.__dict__["_ticket_buf"] = torch.zeros(max(need, 4), dtype=torch.int32, device=dev)
        return t

    def _kernel_ok(self, group) -> bool:
        if group["amsgrad"] or group["maximize"] or group.get("decoupled_weight_decay", False) or group["differentiable"]:
            return False
        if not group["capturable"] or getattr(self, "grad_scale", None) is not None:
            return False
        if isinstance(group["lr"], torch.Tensor) or any(isinstance(b, torch.Tensor) for b in group["betas"]):
            return False
        return all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in group["params"])

    @torch.no_grad()
    def step(self, closure=None):
        if not all(self._kernel_ok(g) for g in self.param_groups):
            return super().step(closure)
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        for group in self.param_groups:
            params, grads, exp_avgs, exp_avg_sqs, max_sqs, steps = [], [], [], [], [], []
            self._init_group(group, params, grads, exp_avgs, exp_avg_sqs, max_sqs, steps)   # torch's own state layout
            if not params:
                continue
            dev = _lib.require_gpu(*params)
            grads = [g if g.is_contiguous() else g.contiguous() for g in grads]
            n = len(params)
            arr = ctypes.c_void_p * n
            beta1, beta2 = group["betas"]
            _lib.check(lib.mi_adam_dense_multi(arr(*[t.data_ptr() for t in params]), arr(*[t.data_ptr() for t in grads]),
                                               arr(*[t.data_ptr() for t in exp_avgs]),
                                               arr(*[t.data_ptr() for t in exp_avg_sqs]),
                                               arr(*[t.data_ptr() for t in steps]),
                                               (ctypes.c_int64 * n)(*[t.numel() for t in params]), n, group["lr"], beta1,
                                               beta2, group["eps"], group["weight_decay"], self._tickets(dev, n).data_ptr(),
                                               _lib.stream_ptr(dev)),
                       "mi_adam_dense_multi")
        return loss


def _dense_adam(params, lr, weight_decay):
    """The reference's `torch.optim.Adam(params, lr=..., weight_decay=...)`: on the GPU the subclass above (device-side
    step counts, so that a whole training step can be replayed as a hipGraph), otherwise torch's own."""
    if len(params) > 0 and all(p.is_cuda and p.dtype == torch.float32 for p in params):
        return Adam(params, lr=lr, weight_decay=weight_decay)
    return torch.optim.Adam(params, lr=lr, weight_decay=weight_decay)


def get_optimizers(model, config: Dict) -> List[torch.optim.Optimizer]:
    """src/models/deepfm.py:155-219 with the sparse branches on the fused row-sparse steps."""
    sparse: bool = config.get("sparse", False)
    optimizer_name: str = config.get("optimizer", "adam")
    lr_emb = config.get("learning_rate_emb", config["learning_rate"])
    if sparse:
        # extension (off unless DeepFM(fc_sparse=True)): a first-order table that emits row-form gradients joins the
        # row-sparse group, since torch's dense Adam/SGD reject sparse gradients
        fc = getattr(model, "fc", None)
        fc_rows = list(fc.parameters()) if getattr(fc, "sparse", False) else []
        decay_param = [p for name, p in model.named_parameters()
                       if "embedding." not in name and not any(p is q for q in fc_rows)]
        no_decay_param = list(model.embedding.parameters()) + fc_rows
    if sparse and optimizer_name == "adam":
        return [SparseAdam(no_decay_param, lr=lr_emb, capturable=all(p.is_cuda for p in no_decay_param)),
                _dense_adam(decay_param, lr=config["learning_rate"], weight_decay=config["weight_decay"])]
    if optimizer_name == "adam":
        return [_dense_adam(list(model.parameters()), lr=config["learning_rate"], weight_decay=config["weight_decay"])]
    elif optimizer_name == "sgd":
        if not sparse:
            return [torch.optim.SGD(model.parameters(), lr=config["learning_rate"], weight_decay=config["weight_decay"])]
        return [SparseSGD(no_decay_param, lr=lr_emb),
                torch.optim.SGD(decay_param, lr=config["learning_rate"], weight_decay=config["weight_decay"])]
    raise ValueError(f"{optimizer_name=} is not recognized")
