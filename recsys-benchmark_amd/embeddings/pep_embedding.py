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
        if isinstance(field_dims, int):
            field_dims = [field_dims]
        num_item = sum(field_dims)
        if sparsity is None:
            sparsity = [0.8, 0.9, 0.99]
        assert isinstance(sparsity, list) and isinstance(sparsity[0], float)
        self.sparsity = list(sorted(sparsity))
        self._cur_min_spar_idx = 0

        self.emb = nn.Embedding(num_item, hidden_size)
        nn.init.xavier_uniform_(self.emb.weight)
        if ori_weight_dir:
            os.makedirs(ori_weight_dir, exist_ok=True)
            torch.save({"state_dict": self.emb.state_dict()}, os.path.join(ori_weight_dir, field_name + ".pth"))

        self.threshold_type = threshold_type
        self.s = self.init_threshold(init_threshold, num_item, hidden_size)
        self.field_name = field_name
        if field_name:
            checkpoint_weight_dir = os.path.join(checkpoint_weight_dir, field_name)
        os.makedirs(checkpoint_weight_dir, exist_ok=True)
        self.checkpoint_weight_dir = checkpoint_weight_dir
        self._mode = mode

    def soft_threshold(self, v, s):
        return torch.sign(v) * torch.relu(torch.abs(v) - torch.sigmoid(s))

    def get_weight(self):
        arr = torch.arange(self.emb.num_embeddings, device=self.emb.weight.device)
        return _kernels.soft_threshold_gather(arr, self.emb.weight, self.s)

    def forward(self, x):
        rows = _kernels.soft_threshold_gather(x, self.emb.weight, self.s)
        return _kernels.bag_reduce(rows, self._mode)

    def init_threshold(self, init, num_item, hidden_size) -> nn.Parameter:
        """One threshold logit per (global | dimension | feature | feature x dimension)."""
        shapes = {"global": (1,), "dimension": (hidden_size,), "feature": (num_item, 1),
                  "feature_dim": (num_item, hidden_size)}
        if self.threshold_type in ("field", "field_dim"):
            raise NotImplementedError()
        if self.threshold_type not in shapes:
            raise ValueError("Invalid threshold_type: {}".format(self.threshold_type))
        return nn.Parameter(torch.full(shapes[self.threshold_type], float(init)))

    def get_sparsity(self, get_n_params=False):
        total_params = self.emb.weight.numel()
        n_params = self.get_num_params()
        if get_n_params:
            return (1 - n_params / total_params), n_params
        return 1 - n_params / total_params

    def get_num_params(self) -> int:
        return torch.count_nonzero(self.soft_threshold(self.emb.weight, self.s)).item()

    def train_callback(self):
        """Save the state to {checkpoint_weight_dir}/{sparsity}.pth when a target sparsity is passed."""
        with torch.no_grad():
            cur_sparsity = self.get_sparsity()
        while self._cur_min_spar_idx < len(self.sparsity) and self.sparsity[self._cur_min_spar_idx] < cur_sparsity:
            sparsity = self.sparsity[self._cur_min_spar_idx]
            torch.save(self.state_dict(), os.path.join(self.checkpoint_weight_dir, f"{sparsity}.pth"))
            self._cur_min_spar_idx += 1


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
        if isinstance(field_dims, int):
            field_dims = [field_dims]
        num_item = sum(field_dims)
        self.emb = nn.Embedding(num_item, hidden_size)
        if ori_weight_dir:
            ori = torch.load(os.path.join(ori_weight_dir, field_name + ".pth"), map_location="cpu")["state_dict"]
            self.emb.load_state_dict(ori)
        finish = torch.load(os.path.join(checkpoint_weight_dir, field_name, f"{sparsity}.pth"), map_location="cpu")
        weight, s = finish["emb.weight"], finish["s"]
        self.mask = nn.Parameter((torch.abs(weight) - torch.sigmoid(s)) > 0, False)
        nnz = self.mask.sum()
        self._nnz = nnz
        self.sparsity = 1 - (nnz / torch.prod(torch.tensor(self.mask.size()))).item()
        self._mode = mode
        if sparse:
            raise NotImplementedError("RetrainPepEmbedding(sparse=True): row-form grads are not built for the "
                                      "masked table; use the dense form")
        self._sparse = sparse

    def get_weight(self):
        arr = torch.arange(self.emb.num_embeddings, device=self.emb.weight.device)
        return _kernels.masked_gather(arr, self.emb.weight, self.mask)

    def forward(self, x):
        rows = _kernels.masked_gather(x, self.emb.weight, self.mask)
        return _kernels.bag_reduce(rows, self._mode)

    def get_sparsity(self, get_n_params=False):
        if get_n_params:
            return self.sparsity, self._nnz
        return self.sparsity

    def get_num_params(self):
        return self._nnz
