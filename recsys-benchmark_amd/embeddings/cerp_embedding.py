"""CERP compositional + soft-threshold-pruned embedding.

Reference: src/models/embeddings/cerp_embedding.py:14-207 (CerpEmbedding) and :209-378
(RetrainCerpEmbedding).  Same constructors, parameter names (`p_weight`, `q_weight`,
`p_threshold`, `q_threshold`; masks `p_mask`, `q_mask`), initialisers and helper methods.
The reference materialises the pruned copies of both tables on every forward and then does two
F.embedding calls; here index math (trunc-div / mod), soft-threshold (or mask) and the add are
fused into one HIP gather over the raw tables (mi_dual_gather_fwd, xform 1 / 2).
"""
import os
from typing import List, Literal, Optional, Tuple, Union

import numpy as np
import torch
from torch import nn

from .. import _kernels
from .base import IEmbedding


class CerpEmbedding(IEmbedding):
    def __init__(
        self,
        field_dims: Union[List[int], int],
        hidden_size: int,
        mode: Optional[str] = None,
        bucket_size: int = 8000,
        threshold_init: float = -100.0,
        threshold_init_method="all-ones",
        field_name: str = "",
    ):
        super().__init__()
        dims = [field_dims] if isinstance(field_dims, int) else list(field_dims)
        assert mode in [None, "sum", "mean", "max"]
        self._field_dims = torch.tensor(dims)
        self._mode, self.field_name = mode, field_name
        self._num_item, self._hidden_size, self._bucket_size = sum(dims), hidden_size, bucket_size
        # a Q row is shared by ceil(N / bucket) consecutive ids; a P row by every bucket-th id
        self.q_entity_per_row = int(np.ceil(self._num_item / bucket_size))

        def table():
            return nn.Parameter(nn.init.xavier_uniform_(torch.zeros(bucket_size, hidden_size)))

        def threshold():
            return self.init_threshold("element-wise", threshold_init, row_size=bucket_size, col_size=hidden_size,
                                       threshold_init_method=threshold_init_method)

        self.p_weight, self.q_weight = table(), table()
        self.q_threshold, self.p_threshold = threshold(), threshold()

    @staticmethod
    def init_threshold(
        threshold_type: Literal["global", "element-wise"],
        init: float,
        row_size: int,
        col_size: int,
        threshold_init_method: str = "all_ones",
    ) -> nn.Parameter:
        """`init` times a [0,1] pattern: all ones by default (any unknown method name, like the
        reference's own default string "all-ones"); "uniform" / "normal" / "xavier_uniform" draw a
        pattern that is squeezed into [0,1] (sigmoid for the global form, per-row min-max for the
        element-wise form; a plain U(0,1) draw is already in range)."""
        if threshold_type not in ("global", "element-wise"):
            raise ValueError("Invalid threshold_type: {}".format(threshold_type))
        if threshold_type == "global":
            if threshold_init_method == "xavier_uniform":
                raise NotImplementedError
            draws = {"uniform": lambda: torch.rand(1), "normal": lambda: torch.normal(0.0, 1.0, size=(1,))}
            pattern = torch.ones(1)
            if threshold_init_method in draws:
                pattern = pattern * draws[threshold_init_method]()
                if threshold_init_method == "normal":
                    pattern = torch.sigmoid(pattern)
            return nn.Parameter(pattern * init)

        shape = (row_size, col_size)
        draws = {
            "uniform": lambda: torch.nn.init.uniform_(torch.zeros(shape)),
            "normal": lambda: torch.normal(0.0, 1.0, size=shape),
            "xavier_uniform": lambda: nn.init.xavier_uniform_(torch.zeros(shape)),
        }
        pattern = torch.ones(shape)
        if threshold_init_method in draws:
            pattern = pattern * draws[threshold_init_method]()
            lo = pattern.min(dim=1, keepdim=True)[0]
            hi = pattern.max(dim=1, keepdim=True)[0]
            pattern = (pattern - lo) / (hi - lo)
        assert bool(((pattern >= 0) & (pattern <= 1)).all())
        return nn.Parameter(init * pattern)

    def _pruned_tables(self):
        soft = lambda w, t: torch.sign(w) * torch.relu(torch.abs(w) - torch.sigmoid(t))   # noqa: E731
        return soft(self.q_weight, self.q_threshold), soft(self.p_weight, self.p_threshold)

    def apply_pruning(self):
        """Materialised pruned tables, for the bookkeeping helpers only (the lookup fuses this)."""
        self.sparse_q_weight, self.sparse_p_weight = self._pruned_tables()

    def forward(self, x):
        emb = _kernels.dual_gather(
            x, self.p_weight, self.q_weight, mod1=self._bucket_size, div2=self.q_entity_per_row,
            op="add", S1=self.p_threshold, S2=self.q_threshold,
        )
        # sum/mean of (Q rows + P rows) == sum/mean(Q rows) + sum/mean(P rows); max is taken
        # over the summed rows in the reference as well (cerp_embedding.py:160-175)
        return _kernels.bag_reduce(emb, self._mode)

    def get_sparsity(self, get_n_params=False):
        total_params = self._num_item * self._hidden_size
        n_params = self.get_num_params()
        if get_n_params:
            return (1 - n_params / total_params), n_params
        return 1 - n_params / total_params

    def get_weight(self):
        all_idxes = torch.arange(self._num_item, device=self.p_weight.data.device)
        return self(all_idxes)

    def get_num_params(self):
        self.apply_pruning()
        n_params = 0
        for w in [self.sparse_p_weight, self.sparse_q_weight]:
            n_params += torch.count_nonzero(w).item()
        return n_params

    def get_prune_loss(self, K=100):
        # same value as the reference's (cerp_embedding.py get_prune_loss); the pruned tables are NOT stashed on the module
        # here: a tensor with autograd history that outlives a hipGraph capture on a module attribute crashes torch's
        # capture_end, and this loss is part of the captured training step (trainer.train_epoch_cerp)
        q, p = self._pruned_tables()
        return -torch.tanh((p + q) * K).norm(2) ** 2


class RetrainCerpEmbedding(IEmbedding):
    """Fixed-mask retraining variant; weights/masks come from
    {checkpoint_weight_dir}/{field_name}/{initial,<weight_name>}.pth exactly as in the reference."""

    def __init__(
        self,
        field_dims: Union[List[int], int],
        hidden_size: int,
        mode: Optional[str],
        checkpoint_weight_dir: str,
        field_name: str = "",
        weight_name: str = "target",
        bucket_size: int = 8000,
        sparse: bool = False,
    ):
        super().__init__()
        if isinstance(field_dims, int):
            field_dims = [field_dims]

        mask_weight_path = os.path.join(checkpoint_weight_dir, field_name, f"{weight_name}.pth")
        init_weight_path = os.path.join(checkpoint_weight_dir, field_name, "initial.pth")
        assert os.path.exists(mask_weight_path), f"Weight not found at {mask_weight_path} to re-init mask"
        assert os.path.exists(
            init_weight_path
        ), f"Weight not found at {init_weight_path} to re-init original weight"

        num_item = sum(field_dims)
        self._field_dims = torch.tensor(field_dims)
        self._mode = mode
        self.field_name: str = field_name
        self._bucket_size = bucket_size
        self._hidden_size = hidden_size

        self.p_weight = nn.Parameter(torch.zeros(bucket_size, hidden_size))
        self.q_weight = nn.Parameter(torch.zeros(bucket_size, hidden_size))
        init_weight = torch.load(init_weight_path, map_location="cpu")

        self.q_mask, self.p_mask = None, None
        assert init_weight["q_weight"].shape == (bucket_size, hidden_size), (
            f"{init_weight['q_weight'].shape} != {(bucket_size, hidden_size)}")
        assert init_weight["p_weight"].shape == (bucket_size, hidden_size)
        self.q_weight.data = init_weight["q_weight"]
        self.p_weight.data = init_weight["p_weight"]
        self.q_mask, self.p_mask = self.load_mask(mask_weight_path)

        self._num_item = num_item
        self.q_entity_per_row = int(np.ceil(self._num_item / self._bucket_size))
        self._sparse = sparse

    def load_mask(self, weight_path: str) -> List[nn.Parameter]:
        checkpoint = torch.load(weight_path, map_location="cpu")
        names: Tuple[Tuple[str, str], Tuple[str, str]] = (
            ("q_weight", "q_threshold"),
            ("p_weight", "p_threshold"),
        )
        masks = []
        for weight_name, threshold_name in names:
            weight = checkpoint[weight_name]
            threshold = checkpoint[threshold_name]
            mask = (weight.abs() - torch.sigmoid(threshold)) > 0
            assert mask.shape == (self._bucket_size, self._hidden_size)
            masks.append(nn.Parameter(mask, False))
        return masks

    def get_weight(self):
        all_idxes = torch.arange(self._num_item, device=self.p_weight.data.device)
        return self(all_idxes)

    def forward(self, x):
        if self._sparse:       # row-form gradients of the two bucket tables (torch.optim.SparseAdam's input)
            emb = _kernels.dual_masked_gather_row_grad(x, self.p_weight, self.q_weight, self.p_mask, self.q_mask,
                                                       self._bucket_size, self.q_entity_per_row)
        else:
            emb = _kernels.dual_gather(
                x, self.p_weight, self.q_weight, mod1=self._bucket_size, div2=self.q_entity_per_row,
                op="add", M1=self.p_mask, M2=self.q_mask,
            )
        return _kernels.bag_reduce(emb, self._mode)

    def get_num_params(self):
        return (torch.count_nonzero(self.q_mask) + torch.count_nonzero(self.p_mask)).item()
