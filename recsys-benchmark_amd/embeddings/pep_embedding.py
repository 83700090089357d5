"""PEP soft-threshold pruned embedding and its retrain variant — reference:
src/models/embeddings/pep_embedding.py:12-229.

Same constructors, parameters (`emb.weight`, `s`; `mask` for the retrain variant), threshold
types, checkpoint side files and `train_callback`.  The reference soft-thresholds the WHOLE table
on every forward and then gathers; here the threshold (broadcast per type) is applied on the fly
inside the gather (mi_xform_gather_fwd/bwd), so traffic is proportional to the looked-up rows.
"""
import os
from typing import List, Optional, Union

import torch
from torch import nn

from .. import _kernels
from .base import IEmbedding


_THRESHOLD_SHAPES = {          # one threshold logit per ...
    "global": lambda n, d: (1,),
    "dimension": lambda n, d: (d,),
    "feature": lambda n, d: (n, 1),
    "feature_dim": lambda n, d: (n, d),
}


def _num_rows(field_dims: Union[List[int], int]) -> int:
    return field_dims if isinstance(field_dims, int) else sum(field_dims)


def _soft(v: torch.Tensor, s: torch.Tensor) -> torch.Tensor:
    """sign(v) * relu(|v| - sigmoid(s)): the PEP re-parametrisation of a pruned weight."""
    return torch.sign(v) * torch.relu(torch.abs(v) - torch.sigmoid(s))


def _initial_weights_file(directory: str, field_name: str) -> str:
    return os.path.join(directory, field_name + ".pth")


def _milestone_file(directory: str, sparsity) -> str:
    return os.path.join(directory, f"{sparsity}.pth")


class PepEmbeeding(IEmbedding):   # (sic: the reference's class name)
    def __init__(
        self,
        field_dims: Union[List[int], int],
        hidden_size: int,
        mode: Optional[str] = None,
        ori_weight_dir: str = "",
        checkpoint_weight_dir: str = "checkpoints",
        field_name: str = "",
        init_threshold: float = -150,
        threshold_type: str = "feature_dim",
        sparsity: Optional[List[float]] = None,
    ):
        super().__init__()
        rows = _num_rows(field_dims)
        milestones = [0.8, 0.9, 0.99] if sparsity is None else sparsity
        if not (isinstance(milestones, list) and isinstance(milestones[0], float)):
            raise AssertionError("sparsity must be a list of floats")
        self.sparsity = sorted(milestones)
        self._cur_min_spar_idx = 0            # milestones below this index have been written out
        self._mode = mode
        self.field_name = field_name
        self.threshold_type = threshold_type

        self.emb = nn.Embedding(rows, hidden_size)
        nn.init.xavier_uniform_(self.emb.weight)
        self.s = self.init_threshold(init_threshold, rows, hidden_size)

        # side files: the untouched initial table (for the retrain stage) and one state_dict per sparsity milestone
        if ori_weight_dir:
            os.makedirs(ori_weight_dir, exist_ok=True)
            torch.save({"state_dict": self.emb.state_dict()}, _initial_weights_file(ori_weight_dir, field_name))
        self.checkpoint_weight_dir = os.path.join(checkpoint_weight_dir, field_name) if field_name else checkpoint_weight_dir
        os.makedirs(self.checkpoint_weight_dir, exist_ok=True)

    def init_threshold(self, init, num_item, hidden_size) -> nn.Parameter:
        kind = self.threshold_type
        if kind in ("field", "field_dim"):
            raise NotImplementedError()
        if kind not in _THRESHOLD_SHAPES:
            raise ValueError("Invalid threshold_type: {}".format(kind))
        return nn.Parameter(torch.full(_THRESHOLD_SHAPES[kind](num_item, hidden_size), float(init)))

    def soft_threshold(self, v, s):
        return _soft(v, s)

    def _all_ids(self):
        return torch.arange(self.emb.num_embeddings, device=self.emb.weight.device)

    def forward(self, x):
        # the reference thresholds the whole table, then gathers; the kernel thresholds the gathered rows
        return _kernels.bag_reduce(_kernels.soft_threshold_gather(x, self.emb.weight, self.s), self._mode)

    def get_weight(self):
        return _kernels.soft_threshold_gather(self._all_ids(), self.emb.weight, self.s)

    def get_num_params(self) -> int:
        return torch.count_nonzero(_soft(self.emb.weight, self.s)).item()

    def get_sparsity(self, get_n_params=False):
        kept = self.get_num_params()
        ratio = 1 - kept / self.emb.weight.numel()
        return (ratio, kept) if get_n_params else ratio

    def train_callback(self):
        """Write `state_dict()` to {checkpoint_weight_dir}/{milestone}.pth for every milestone the table's sparsity
        has passed since the last call."""
        with torch.no_grad():
            reached = self.get_sparsity()
        pending = self.sparsity[self._cur_min_spar_idx:]
        passed = [m for m in pending if m < reached]          # the list is sorted: a prefix of `pending`
        for milestone in passed:
            torch.save(self.state_dict(), _milestone_file(self.checkpoint_weight_dir, milestone))
        self._cur_min_spar_idx += len(passed)


class RetrainPepEmbedding(IEmbedding):
    def __init__(
        self,
        field_dims: Union[List[int], int],
        hidden_size,
        mode: Optional[str],
        checkpoint_weight_dir,
        sparsity: Union[float, str] = 0.8,
        ori_weight_dir: Optional[str] = None,
        field_name: str = "",
        sparse=False,
    ):
        super().__init__()
        self._sparse, self._mode = sparse, mode
        self.emb = nn.Embedding(_num_rows(field_dims), hidden_size)
        if ori_weight_dir:                      # lottery-ticket style: restart from the saved initial table
            saved = torch.load(_initial_weights_file(ori_weight_dir, field_name), map_location="cpu")
            self.emb.load_state_dict(saved["state_dict"])
        found = torch.load(_milestone_file(os.path.join(checkpoint_weight_dir, field_name), sparsity), map_location="cpu")
        keep = (torch.abs(found["emb.weight"]) - torch.sigmoid(found["s"])) > 0      # where the pruned table is non-zero
        self.mask = nn.Parameter(keep, requires_grad=False)
        self._nnz = keep.sum()
        self.sparsity = 1 - (self._nnz / torch.prod(torch.tensor(keep.size()))).item()

    def forward(self, x):
        gather = _kernels.masked_gather_row_grad if self._sparse else _kernels.masked_gather
        return _kernels.bag_reduce(gather(x, self.emb.weight, self.mask), self._mode)

    def get_weight(self):
        ids = torch.arange(self.emb.num_embeddings, device=self.emb.weight.device)
        return _kernels.masked_gather(ids, self.emb.weight, self.mask)

    def get_num_params(self):
        return self._nnz

    def get_sparsity(self, get_n_params=False):
        return (self.sparsity, self._nnz) if get_n_params else self.sparsity
