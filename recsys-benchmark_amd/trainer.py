"""The training / validation loops around the hot path — CTR (reference: src/trainer/deepfm.py:17-139) and LightGCN
(src/trainer/lightgcn.py:14-165, 378-421) — with the per-batch work — forward, loss, backward, every optimizer step —
replayed as ONE hipGraph.

At the headline shape the eager step is bound by the host (≈0.85 ms of Python / autograd / launch work around ≈0.45 ms of
kernels); a captured step has no host work beyond two input copies and one graph launch.  What makes the whole step
capturable: the lookup kernels emit row-form gradients without host syncs, `optim.SparseAdam(capturable=True)` keeps its
step count on the device, `optim.Adam` steps from device-side counts too (`optim.get_optimizers` builds both on a GPU),
dropout seeds and BatchNorm counters advance inside kernels.

`train_epoch` / `validate_epoch` keep the reference's signatures and return values; the loss is accumulated on the device
and read back at the logging steps only (the reference calls `.item()` every batch).
"""
import datetime
import logging
import inspect
import warnings
from typing import Dict, List, Optional, Sequence, Tuple, Union

import torch

from . import _lib, losses
from .lightgcn import score_topk, train_items_csr

logger = logging.getLogger("recsys_benchmark_amd.trainer")
now = datetime.datetime.now


def _check_errors(model) -> None:
    """Deferred index / overflow checks at a logging step.  A row-sharded model checks COLLECTIVELY (the ranks agree on the
    flags, then all raise together: a rank raising alone would leave its peers waiting in the next all-to-all)."""
    if hasattr(model, "check_overflow"):
        model.check_index_errors()
        model.check_overflow()       # a peer bucket that overflowed dropped lookups
    else:
        _lib.check_index_errors()


class _capture(torch.cuda.graph):
    """torch.cuda.graph that also tells a direct RCCL communicator's watchdog thread (sharded.DirectComm) that a
    global-mode capture is under way, so it leaves its event queries alone until the capture has ended."""

    def __enter__(self):
        from . import sharded

        sharded.note_capture(+1)
        try:
            return super().__enter__()
        except BaseException:
            sharded.note_capture(-1)
            raise

    def __exit__(self, *exc):
        from . import sharded

        try:
            return super().__exit__(*exc)
        finally:
            sharded.note_capture(-1)


def _capturable(optimizers) -> bool:
    """False when an optimizer keeps its step count on the host (torch's Adam family without `capturable=True`): its
    step() refuses to run under capture, and a capture abandoned half-way is not something to recover from — so this
    is checked up front.  `optim.get_optimizers` / `optim.Adam` / `optim.SparseAdam(capturable=True)` all qualify."""
    return all(group.get("capturable", True) for opt in optimizers for group in opt.param_groups)


class GraphedTrainStep:
    """step(inputs, labels): one optimisation step of `model` on the batch, in the reference's order (forward, loss,
    zero_grad, backward, optimizer steps).  The first `warmup` calls run eagerly (they are ordinary training steps and
    let every lazily created buffer come into being), the next call of the same batch shape is captured and from then on
    replayed; batches of any other shape (the ragged last batch of an epoch) run eagerly.

    `loss_sum` (0-dim device tensor) accumulates the batch losses, `steps` counts them; `last_loss` is the latest one.
    """

    def __init__(self, model: torch.nn.Module, optimizers, criterion: Optional[torch.nn.Module] = None, warmup: int = 2,
                 use_graph: bool = True, clip_grad: float = 0, extra_loss=None, extra_weight: float = 1.0):
        """extra_loss: optional callable returning a scalar tensor; `extra_weight` times it is added to the criterion
        before the backward (the CERP trainer's `prune_loss_weight * model.embedding.get_prune_loss()`); `extra_sum`
        accumulates the unweighted values."""
        self.model = model
        self.extra_loss, self.extra_weight = extra_loss, extra_weight
        self.extra_sum: Optional[torch.Tensor] = None
        self.optimizers: List[torch.optim.Optimizer] = optimizers if isinstance(optimizers, list) else [optimizers]
        self.criterion = criterion if criterion is not None else losses.BCEWithLogitsLoss()
        self._labels_in_forward = (isinstance(self.criterion, losses.BCEWithLogitsLoss)
                                   and "labels" in inspect.signature(model.forward).parameters)
        self._prefetches = callable(getattr(model, "prefetch_next", None))
        self._static_next = None
        self.warmup = warmup
        self.clip_grad = clip_grad
        # clip_grad_norm_ reads the norm back on some paths and row-form gradients have no dense norm: eager only
        self.use_graph = use_graph and not clip_grad
        if self.use_graph and not _capturable(self.optimizers):
            warnings.warn("an optimizer keeps its step count on the host (capturable=False): the training step runs "
                          "eagerly; build the optimizers with recsys_benchmark_amd.optim.get_optimizers / optim.Adam")
            self.use_graph = False
        self.steps = 0
        self.loss_sum: Optional[torch.Tensor] = None
        self.last_loss: Optional[torch.Tensor] = None
        self._graph: Optional[torch.cuda.CUDAGraph] = None
        self._static = None
        self._shape = None
        self._seen = 0

    def _body(self, inputs, labels):
        labels = labels.float()
        # (a model that takes the step's labels — DeepFM's fused step — evaluates the criterion inside its head launch; the
        #  criterion call below then only picks the result up)
        outputs = self.model(inputs, labels=labels) if self._labels_in_forward else self.model(inputs)
        loss = self.criterion(outputs, labels)
        total = loss
        if self.extra_loss is not None:
            extra = self.extra_loss()
            total = loss + self.extra_weight * extra
            self.extra_sum += extra.detach()
        for opt in self.optimizers:
            opt.zero_grad(set_to_none=True)
        total.backward(self._one)          # d(loss)/d(loss) from a resident scalar: no fill launch per step
        if self.clip_grad:
            torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.clip_grad)
        for opt in self.optimizers:
            opt.step()
        loss = loss.detach()
        self.loss_sum += loss
        return loss

    def _capture(self, inputs, labels):
        static_in, static_lab = inputs.clone(), labels.clone()
        # a model that can use the NEXT batch's ids (DeepFM.prefetch_next: the step's weight-gradient launch touches that
        # batch's table rows in extra workgroups) reads them from a static buffer the caller refreshes per step
        static_next = inputs.clone() if self._prefetches else None
        for opt in self.optimizers:
            opt.zero_grad(set_to_none=True)        # the captured backward allocates the gradients in the graph's pool
        graph = torch.cuda.CUDAGraph()
        if static_next is not None:
            self.model.prefetch_next(static_next)
        with _capture(graph):
            static_loss = self._body(static_in, static_lab)
        self._graph, self._static, self._static_next = graph, (static_in, static_lab, static_loss), static_next

    def __call__(self, inputs: torch.Tensor, labels: torch.Tensor, next_inputs: Optional[torch.Tensor] = None) -> torch.Tensor:
        """next_inputs (optional): the batch of the FOLLOWING step, already on the device — a loop that is one batch ahead of
        the step (train_epoch is) lets the step pull that batch's table rows into the Infinity Cache."""
        if not inputs.is_cuda:
            raise RuntimeError("GraphedTrainStep runs on the GPU: move the batch to the model's device first")
        if self.loss_sum is None:
            self.loss_sum = torch.zeros((), dtype=torch.float32, device=inputs.device)
            self.extra_sum = torch.zeros((), dtype=torch.float32, device=inputs.device)
            self._one = losses.unit_scalar(inputs.device)
        self.steps += 1
        shape = (tuple(inputs.shape), inputs.dtype, tuple(labels.shape), labels.dtype)
        if self.use_graph and self._graph is None and self._seen >= self.warmup and self._shape == shape:
            try:
                self._capture(inputs, labels)
            except Exception as exc:               # leave training running on the eager HIP path
                warnings.warn(f"hipGraph capture of the training step failed ({exc!r}); continuing with eager steps")
                torch.cuda.synchronize()
                self.use_graph, self._graph = False, None
        if self._graph is not None and self._shape == shape:
            static_in, static_lab, static_loss = self._static
            static_in.copy_(inputs, non_blocking=True)
            static_lab.copy_(labels, non_blocking=True)
            if self._static_next is not None and next_inputs is not None and next_inputs.shape == self._static_next.shape:
                self._static_next.copy_(next_inputs, non_blocking=True)
            self._graph.replay()
            self.last_loss = static_loss
            return static_loss
        if self._graph is None:
            if self._shape != shape:
                self._shape, self._seen = shape, 0
            self._seen += 1
        if self._prefetches and next_inputs is not None and next_inputs.shape == inputs.shape:
            self.model.prefetch_next(next_inputs)
        self.last_loss = self._body(inputs, labels)
        return self.last_loss


def train_epoch(dataloader, model, optimizers: Union[List[torch.optim.Optimizer], torch.optim.Optimizer], device="cuda",
                log_step=10, profiler=None, clip_grad=0, step: Optional[GraphedTrainStep] = None) -> Dict[str, float]:
    """src/trainer/deepfm.py:17-93.  Pass the same `step` object to successive epochs to keep one captured graph."""
    if not isinstance(optimizers, list):
        optimizers = [optimizers]
    model.train()
    model.to(device)
    if step is None:
        step = GraphedTrainStep(model, optimizers, clip_grad=clip_grad)
    first_steps = step.steps
    first_sum = float(step.loss_sum) if step.loss_sum is not None else 0.0
    load_data_time, train_time = datetime.timedelta(), datetime.timedelta()
    first_start = start = now()
    idx = -1
    # one batch ahead of the step (the DataLoader's workers are anyway): the step is told the next batch's ids
    batches = iter(dataloader)
    ahead = next(batches, None)
    if ahead is not None:
        ahead = (ahead[0].to(device, non_blocking=True), ahead[1].to(device, non_blocking=True))
    while ahead is not None:
        idx += 1
        inputs, labels = ahead
        ahead = next(batches, None)
        if ahead is not None:
            ahead = (ahead[0].to(device, non_blocking=True), ahead[1].to(device, non_blocking=True))
        load_data_time += now() - start
        start_train = now()
        step(inputs, labels, next_inputs=ahead[0] if ahead is not None else None)
        if log_step and idx % log_step == 0:
            logger.info("Idx: %d - loss: %.4g", idx, (float(step.loss_sum) - first_sum) / (idx + 1))
            _check_errors(model)             # the reference's nn.Embedding raises on the offending batch; here at the next sync
        if profiler:
            profiler.step()
        end_train = start = now()
        train_time += end_train - start_train
    n = step.steps - first_steps
    loss_dict = {"loss": (float(step.loss_sum) - first_sum) / n if n else 0.0}
    _check_errors(model)
    logger.info("train_time: %s", train_time)
    logger.info("load_data_time: %s", load_data_time)
    logger.info("total_time: %s", now() - first_start)
    return loss_dict


def train_epoch_cerp(dataloader, model, optimizer, device="cuda", log_step=10, profiler=None, clip_grad=0,
                     target_sparsity=0.8, prune_loss_weight=0, step: Optional[GraphedTrainStep] = None) -> Dict[str, float]:
    """src/trainer/deepfm.py:142-248: `train_epoch` plus `prune_loss_weight * model.embedding.get_prune_loss()` in the
    loss, the table's sparsity checked at the logging steps, and an early return (running sums, as in the reference) once
    it reaches `target_sparsity`.  Returns {"loss", "prune_loss", "log_loss", "sparsity", "num_params"}."""
    model.train()
    model.to(device)
    if step is None:
        step = GraphedTrainStep(model, optimizer, clip_grad=clip_grad, extra_loss=lambda: model.embedding.get_prune_loss(),
                                extra_weight=prune_loss_weight)
    first_steps = step.steps
    first = (float(step.loss_sum), float(step.extra_sum)) if step.loss_sum is not None else (0.0, 0.0)

    def sums():
        log_loss, prune = float(step.loss_sum) - first[0], float(step.extra_sum) - first[1]
        return {"loss": log_loss + prune_loss_weight * prune, "prune_loss": prune, "log_loss": log_loss}

    idx = -1
    for idx, (inputs, labels) in enumerate(dataloader):
        step(inputs.to(device, non_blocking=True), labels.to(device, non_blocking=True))
        if log_step and idx % log_step == 0:
            sparsity, num_params = model.embedding.get_sparsity(get_n_params=True)
            running = sums()
            _lib.check_index_errors()
            logger.info("Idx: %d - loss: %.4g - sparsity: %.4g - num_params: %d", idx, running["loss"] / (idx + 1), sparsity,
                        num_params)
            if sparsity >= target_sparsity:
                return dict(running, sparsity=sparsity, num_params=num_params)
        if profiler:
            profiler.step()
    n = max(step.steps - first_steps, 1)
    sparsity, num_params = model.embedding.get_sparsity(get_n_params=True)
    _lib.check_index_errors()
    return dict({k: v / n for k, v in sums().items()}, sparsity=sparsity, num_params=num_params)


def binary_auc(y_true: torch.Tensor, y_score: torch.Tensor) -> float:
    """Area under the ROC curve as sklearn.metrics.roc_auc_score computes it for binary labels (ties share their average
    rank: the Mann-Whitney statistic), on the device in float64."""
    y_true = y_true.reshape(-1).to(torch.float64)
    uniq, inverse, counts = torch.unique(y_score.reshape(-1), sorted=True, return_inverse=True, return_counts=True)
    ends = torch.cumsum(counts, 0).to(torch.float64)
    avg_rank = ends - (counts.to(torch.float64) - 1.0) / 2.0          # 1-based average rank of each distinct score
    n_pos = y_true.sum()
    n_neg = y_true.numel() - n_pos
    if float(n_pos) == 0.0 or float(n_neg) == 0.0:
        raise ValueError("Only one class present in y_true. ROC AUC score is not defined in that case.")
    rank_sum = (avg_rank[inverse] * y_true).sum()
    return float((rank_sum - n_pos * (n_pos + 1.0) / 2.0) / (n_pos * n_neg))


class GraphedForward:
    """model(x) under no_grad with the forward of each input shape seen twice replayed as a hipGraph (the first call of a
    shape runs eagerly).  The returned tensor is the graph's static output: consume it before the next call."""

    def __init__(self, model: torch.nn.Module, use_graph: bool = True):
        self.model, self.use_graph = model, use_graph
        self._graphs: Dict[tuple, object] = {}

    @torch.no_grad()
    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        key = (tuple(x.shape), x.dtype)
        entry = self._graphs.get(key)
        if not self.use_graph or not x.is_cuda:
            return self.model(x)
        if entry is None:
            self._graphs[key] = "seen"
            return self.model(x)
        if entry == "seen":
            try:
                static_in = x.clone()
                graph = torch.cuda.CUDAGraph()
                with _capture(graph):
                    static_out = self.model(static_in)
                entry = self._graphs[key] = (graph, static_in, static_out)
            except Exception as exc:
                warnings.warn(f"hipGraph capture of the forward failed ({exc!r}); continuing with eager launches")
                torch.cuda.synchronize()
                self.use_graph = False
                return self.model(x)
        graph, static_in, static_out = entry
        static_in.copy_(x, non_blocking=True)
        graph.replay()
        return static_out


@torch.no_grad()
def validate_epoch(val_loader, model, device="cuda") -> Dict[str, float]:
    """src/trainer/deepfm.py:96-139: {"auc", "log_loss"}; labels and predictions stay on the device, the forward of the
    full-size batches is replayed as a hipGraph."""
    model.eval()
    model = model.to(device)
    forward = GraphedForward(model)
    criterion = torch.nn.BCEWithLogitsLoss(reduction="sum")
    log_loss = torch.zeros((), dtype=torch.float64, device=device)
    y_true, y_pred = [], []
    for inputs, labels in val_loader:
        inputs, labels = inputs.to(device), labels.to(device)
        outputs = forward(inputs)
        log_loss += criterion(outputs, labels.float())
        y_true.append(labels.reshape(-1))
        y_pred.append(torch.sigmoid(outputs).reshape(-1))
    y_true, y_pred = torch.cat(y_true), torch.cat(y_pred)
    _lib.check_index_errors()
    return {"auc": binary_auc(y_true, y_pred), "log_loss": float(log_loss) / y_pred.numel()}


# --------------------------------------------------------------------------------------------------------------------
# collaborative filtering (LightGCN): reference src/trainer/lightgcn.py:14-165, 378-421
class GraphedCFTrainStep:
    """step(users, pos_items, neg_items): one LightGCN optimisation step as the reference's `_train_step` does it —
    propagate, BPR over the batch rows, `weight_decay * get_reg_loss`, optional InfoNCE on the batch's distinct rows,
    zero_grad, backward, optimizer step — replayed as one hipGraph for the batch size seen first (other sizes run eagerly
    on the same kernels).  The reference's `torch.unique` (a data-dependent shape) is replaced by a first-occurrence mask.

    `sums` (device tensor [4]) accumulates loss, rec_loss, reg_loss, cl_loss; `steps` counts the calls."""

    def __init__(self, model, adj, optimizer, weight_decay: float = 0, info_nce_weight: float = 0, warmup: int = 2,
                 use_graph: bool = True):
        self.model, self.adj, self.optimizer = model, adj, optimizer
        self.weight_decay, self.info_nce_weight = weight_decay, info_nce_weight
        self.warmup = warmup
        self.use_graph = use_graph
        if self.use_graph and not _capturable([optimizer]):
            warnings.warn("the optimizer keeps its step count on the host (capturable=False): the LightGCN step runs "
                          "eagerly; use recsys_benchmark_amd.optim.Adam")
            self.use_graph = False
        self.steps = 0
        self.sums: Optional[torch.Tensor] = None
        self._graph, self._static, self._shape, self._seen = None, None, None, 0

    def _body(self, users, pos_items, neg_items):
        fused = self.weight_decay > 0 and hasattr(self.model, "forward_with_reg_loss")
        if fused:      # propagation + regulariser as one node: the regulariser's gradient rows join the propagation's; every
            # term below reads the propagated tables at the batch's rows only, so the last layer computes only those
            all_user_emb, all_item_emb, reg_loss = self.model.forward_with_reg_loss(self.adj, users, pos_items, neg_items,
                                                                                    batch_rows_only=True)
        else:
            all_user_emb, all_item_emb = self.model(self.adj)
        if fused:      # rec_loss + weight_decay * reg_loss out of the BPR launch itself (no scale / add launches)
            rec_plus_reg, rec_loss = losses.bpr_loss_rows(all_user_emb, all_item_emb, users, pos_items, neg_items,
                                                          plus=reg_loss, plus_weight=self.weight_decay, return_parts=True)
        else:
            rec_loss = losses.bpr_loss_rows(all_user_emb, all_item_emb, users, pos_items, neg_items)
        zero = torch.zeros((), device=rec_loss.device)
        if not fused:
            reg_loss = self.model.get_reg_loss(users, pos_items, neg_items) if self.weight_decay > 0 else zero
        cl_loss = zero
        if self.info_nce_weight > 0:           # SGL without augmentation (src/trainer/lightgcn.py:405-417)
            # view1 = rows of the batch's DISTINCT users and positives; here: all batch rows, repeats masked out (the loss is
            # a mean over rows of a softmax over columns: the order of the rows is immaterial), so no shape depends on data
            view = torch.cat([torch.index_select(all_user_emb, 0, users), torch.index_select(all_item_emb, 0, pos_items)], 0)
            valid = torch.cat([losses.first_occurrence(users, all_user_emb.shape[0]),
                               losses.first_occurrence(pos_items, all_item_emb.shape[0])])
            cl_loss = losses.info_nce(view, view, 0.2, valid=valid) * self.info_nce_weight
        if fused:
            loss = rec_plus_reg + cl_loss if self.info_nce_weight > 0 else rec_plus_reg
        else:
            loss = rec_loss + self.weight_decay * reg_loss + cl_loss
        self.optimizer.zero_grad(set_to_none=True)
        loss.backward(self._one)
        self.optimizer.step()
        parts = torch.stack([loss.detach(), rec_loss.detach(), reg_loss.detach(), cl_loss.detach()])
        self.sums += parts
        return parts

    def __call__(self, users, pos_items, neg_items) -> torch.Tensor:
        if self.sums is None:
            self.sums = torch.zeros(4, dtype=torch.float32, device=users.device)
            self._one = losses.unit_scalar(users.device)
        self.steps += 1
        shape = (tuple(users.shape), users.dtype)
        if self.use_graph and self._graph is None and self._seen >= self.warmup and self._shape == shape:
            try:
                static = (users.clone(), pos_items.clone(), neg_items.clone())
                self.optimizer.zero_grad(set_to_none=True)
                graph = torch.cuda.CUDAGraph()
                with _capture(graph):
                    parts = self._body(*static)
                self._graph, self._static = graph, static + (parts,)
            except Exception as exc:
                warnings.warn(f"hipGraph capture of the LightGCN step failed ({exc!r}); continuing with eager steps")
                torch.cuda.synchronize()
                self.use_graph, self._graph = False, None
        if self._graph is not None and self._shape == shape:
            for dst, src in zip(self._static[:3], (users, pos_items, neg_items)):
                dst.copy_(src, non_blocking=True)
            self._graph.replay()
            return self._static[3]
        if self._graph is None:
            if self._shape != shape:
                self._shape, self._seen = shape, 0
            self._seen += 1
        return self._body(users, pos_items, neg_items)


def train_epoch_cf(dataloader, model, optimizer, device="cuda", log_step=10, weight_decay=0, profiler=None,
                   info_nce_weight=0, step: Optional[GraphedCFTrainStep] = None) -> Dict[str, float]:
    """src/trainer/lightgcn.py:14-77 (`train_epoch`): {"loss", "reg_loss", "rec_loss", "cl_loss"} averaged over the
    batches.  `dataloader.dataset.get_norm_adj()` supplies the normalised adjacency, as in the reference."""
    model.train()
    model.to(device)
    if step is None:
        adj = dataloader.dataset.get_norm_adj().to(device)
        step = GraphedCFTrainStep(model, adj, optimizer, weight_decay, info_nce_weight)
    first_steps = step.steps
    first = step.sums.clone() if step.sums is not None else None
    idx = -1
    for idx, (users, pos_items, neg_items) in enumerate(dataloader):
        step(users.to(device, non_blocking=True), pos_items.to(device, non_blocking=True),
             neg_items.to(device, non_blocking=True))
        if log_step and idx % log_step == 0:
            done = (step.sums - first if first is not None else step.sums) / (idx + 1)
            logger.info("Idx: %d - loss: %.2g - rec_loss: %.2g", idx, float(done[0]), float(done[1]))
            _lib.check_index_errors()
        if profiler:
            profiler.step()
    n = step.steps - first_steps
    total = (step.sums - first if first is not None else step.sums) if n else torch.zeros(4)
    avg = (total / max(n, 1)).tolist()
    _lib.check_index_errors()
    return {"loss": avg[0], "rec_loss": avg[1], "reg_loss": avg[2], "cl_loss": avg[3]}


def train_epoch_pep(dataloader, model, optimizer, device="cuda", log_step=10, weight_decay=0, profiler=None,
                    info_nce_weight=0, target_sparsity=0, step: Optional[GraphedCFTrainStep] = None) -> Dict[str, float]:
    """src/trainer/lightgcn.py:294-375: `train_epoch` for a LightGCN on PEP tables — the tables' sparsity is read at the
    logging steps (`get_sparsity_and_param`) and the epoch ends early once it exceeds `target_sparsity`.  Returns the
    four averaged losses plus "sparsity" / "num_params" of the last check."""
    from .lightgcn import get_sparsity_and_param

    model.train()
    model.to(device)
    if step is None:
        step = GraphedCFTrainStep(model, dataloader.dataset.get_norm_adj().to(device), optimizer, weight_decay, info_nce_weight)
    first_steps = step.steps
    first = step.sums.clone() if step.sums is not None else None
    extra = {}
    for idx, (users, pos_items, neg_items) in enumerate(dataloader):
        step(users.to(device, non_blocking=True), pos_items.to(device, non_blocking=True),
             neg_items.to(device, non_blocking=True))
        if log_step and idx % log_step == 0:
            sparsity, num_params = get_sparsity_and_param(model)
            extra = {"sparsity": sparsity, "num_params": num_params}
            logger.info("Idx: %d - sparsity: %.2f - num_params: %d", idx, sparsity, num_params)
            _lib.check_index_errors()
            if sparsity > target_sparsity:
                logger.info("Found target sparsity")
                break
        if profiler:
            profiler.step()
    n = max(step.steps - first_steps, 1)
    avg = ((step.sums - first if first is not None else step.sums) / n).tolist() if step.sums is not None else [0.0] * 4
    _lib.check_index_errors()
    return dict({"loss": avg[0], "rec_loss": avg[1], "reg_loss": avg[2], "cl_loss": avg[3]}, **extra)


def ndcg_recall_at_k(y_pred: torch.Tensor, y_true: Sequence[Union[Sequence[int], set]], k: int = 20) -> Tuple[float, float]:
    """src/metrics.py:9-43, 70-108 (`get_ndcg`, `get_ndcg_recall`) for a [users, >=k] tensor of recommended item ids:
    the relevance test is one broadcast comparison against the padded true-item lists on y_pred's device, float64."""
    dev = y_pred.device
    n = len(y_true)
    if n == 0 or y_pred.shape[0] != n:
        raise ValueError("y_pred must hold one row of recommendations per entry of y_true")
    lens = torch.tensor([len(t) for t in y_true], dtype=torch.int64)
    width = max(int(lens.max()), 1)
    padded = torch.full((n, width), -1, dtype=torch.int64)
    for i, t in enumerate(y_true):
        if len(t):
            padded[i, :len(t)] = torch.as_tensor(sorted(t) if isinstance(t, (set, frozenset)) else list(t), dtype=torch.int64)
    padded, lens = padded.to(dev), lens.to(dev)
    pred = y_pred[:, :k]
    relevant = (pred.unsqueeze(2) == padded.unsqueeze(1)).any(2).to(torch.float64)          # [n, k]
    weight = 1.0 / torch.log2(torch.arange(2, pred.shape[1] + 2, dtype=torch.float64, device=dev))
    dcg = (relevant * weight).sum(1)
    length = torch.clamp(lens, max=k)
    ideal = torch.cumsum(1.0 / torch.log2(torch.arange(2, k + 2, dtype=torch.float64, device=dev)), 0)
    idcg = ideal[(length - 1).clamp(min=0)]
    ndcg = (dcg / idcg).mean()
    recall = (relevant.sum(1) / length.to(torch.float64)).mean()
    return float(ndcg), float(recall)


@torch.no_grad()
def validate_epoch_cf(train_dataset, val_loader, model, device="cuda", k=20, filter_item_on_train=True, profiler=None,
                      metrics: Optional[List[str]] = None) -> Dict[str, float]:
    """src/trainer/lightgcn.py:80-165 (`validate_epoch`): {"ndcg"} or {"ndcg", "recall"}.  Scores, the train-item mask
    (a CSR of `train_dataset.get_graph()` built once, not a Python loop per batch) and the top-k run in `score_topk`;
    the recommendations stay on the device until the metric."""
    adj = train_dataset.get_norm_adj().to(device)
    graph = train_dataset.get_graph()
    model.eval()
    model = model.to(device)
    user_embs, item_embs = model(adj)
    csr = train_items_csr(graph, user_embs.shape[0], device) if filter_item_on_train else None
    preds, truths = [], []
    for users, pos_items in val_loader:
        preds.append(score_topk(user_embs, item_embs, torch.as_tensor(users).to(device), k, csr))
        truths.extend(pos_items)
        if profiler:
            profiler.step()
    ndcg, recall = ndcg_recall_at_k(torch.cat(preds), truths, k)
    _lib.check_index_errors()
    if metrics is not None and "ndcg" in metrics and "recall" in metrics:
        return {"ndcg": ndcg, "recall": recall}
    return {"ndcg": ndcg}
