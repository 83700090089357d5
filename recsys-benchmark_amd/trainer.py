"""The CTR training / validation loops around the hot path (reference: src/trainer/deepfm.py:17-139), with the per-batch
work — forward, BCE-with-logits, backward, every optimizer step — replayed as ONE hipGraph.

At the headline shape the eager step is bound by the host (≈0.85 ms of Python / autograd / launch work around ≈0.45 ms of
kernels); a captured step has no host work beyond two input copies and one graph launch.  What makes the whole step
capturable: the lookup kernels emit row-form gradients without host syncs, `optim.SparseAdam(capturable=True)` keeps its
step count on the device, torch's Adam runs fused + capturable (`optim.get_optimizers` builds both that way on a GPU),
dropout seeds and BatchNorm counters advance inside kernels.

`train_epoch` / `validate_epoch` keep the reference's signatures and return values; the loss is accumulated on the device
and read back at the logging steps only (the reference calls `.item()` every batch).
"""
import datetime
import logging
import warnings
from typing import Dict, List, Optional, Union

import torch

from . import losses

logger = logging.getLogger("recsys_benchmark_amd.trainer")
now = datetime.datetime.now


class GraphedTrainStep:
    """step(inputs, labels): one optimisation step of `model` on the batch, in the reference's order (forward, loss,
    zero_grad, backward, optimizer steps).  The first `warmup` calls run eagerly (they are ordinary training steps and
    let every lazily created buffer come into being), the next call of the same batch shape is captured and from then on
    replayed; batches of any other shape (the ragged last batch of an epoch) run eagerly.

    `loss_sum` (0-dim device tensor) accumulates the batch losses, `steps` counts them; `last_loss` is the latest one.
    """

    def __init__(self, model: torch.nn.Module, optimizers, criterion: Optional[torch.nn.Module] = None, warmup: int = 2,
                 use_graph: bool = True, clip_grad: float = 0):
        self.model = model
        self.optimizers: List[torch.optim.Optimizer] = optimizers if isinstance(optimizers, list) else [optimizers]
        self.criterion = criterion if criterion is not None else losses.BCEWithLogitsLoss()
        self.warmup = warmup
        self.clip_grad = clip_grad
        # clip_grad_norm_ reads the norm back on some paths and row-form gradients have no dense norm: eager only
        self.use_graph = use_graph and not clip_grad
        self.steps = 0
        self.loss_sum: Optional[torch.Tensor] = None
        self.last_loss: Optional[torch.Tensor] = None
        self._graph: Optional[torch.cuda.CUDAGraph] = None
        self._static = None
        self._shape = None
        self._seen = 0

    def _body(self, inputs, labels):
        outputs = self.model(inputs)
        loss = self.criterion(outputs, labels.float())
        for opt in self.optimizers:
            opt.zero_grad(set_to_none=True)
        loss.backward()
        if self.clip_grad:
            torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.clip_grad)
        for opt in self.optimizers:
            opt.step()
        loss = loss.detach()
        self.loss_sum += loss
        return loss

    def _capture(self, inputs, labels):
        static_in, static_lab = inputs.clone(), labels.clone()
        for opt in self.optimizers:
            opt.zero_grad(set_to_none=True)        # the captured backward allocates the gradients in the graph's pool
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            static_loss = self._body(static_in, static_lab)
        self._graph, self._static = graph, (static_in, static_lab, static_loss)

    def __call__(self, inputs: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
        if not inputs.is_cuda:
            raise RuntimeError("GraphedTrainStep runs on the GPU: move the batch to the model's device first")
        if self.loss_sum is None:
            self.loss_sum = torch.zeros((), dtype=torch.float32, device=inputs.device)
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
            self._graph.replay()
            self.last_loss = static_loss
            return static_loss
        if self._graph is None:
            if self._shape != shape:
                self._shape, self._seen = shape, 0
            self._seen += 1
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
    for idx, (inputs, labels) in enumerate(dataloader):
        load_data_time += now() - start
        start_train = now()
        step(inputs.to(device, non_blocking=True), labels.to(device, non_blocking=True))
        if log_step and idx % log_step == 0:
            logger.info("Idx: %d - loss: %.4g", idx, (float(step.loss_sum) - first_sum) / (idx + 1))
        if profiler:
            profiler.step()
        end_train = start = now()
        train_time += end_train - start_train
    n = step.steps - first_steps
    loss_dict = {"loss": (float(step.loss_sum) - first_sum) / n if n else 0.0}
    logger.info("train_time: %s", train_time)
    logger.info("load_data_time: %s", load_data_time)
    logger.info("total_time: %s", now() - first_start)
    return loss_dict


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


@torch.no_grad()
def validate_epoch(val_loader, model, device="cuda") -> Dict[str, float]:
    """src/trainer/deepfm.py:96-139: {"auc", "log_loss"}; labels and predictions stay on the device."""
    model.eval()
    model = model.to(device)
    criterion = torch.nn.BCEWithLogitsLoss(reduction="sum")
    log_loss = torch.zeros((), dtype=torch.float64, device=device)
    y_true, y_pred = [], []
    for inputs, labels in val_loader:
        inputs, labels = inputs.to(device), labels.to(device)
        outputs = model(inputs)
        log_loss += criterion(outputs, labels.float())
        y_true.append(labels.reshape(-1))
        y_pred.append(torch.sigmoid(outputs).reshape(-1))
    y_true, y_pred = torch.cat(y_true), torch.cat(y_pred)
    return {"auc": binary_auc(y_true, y_pred), "log_loss": float(log_loss) / y_pred.numel()}
