"""IEmbedding plug-in surface + the full (vanilla) table.

Mirrors src/models/embeddings/base.py:8-75 of the reference: same class names,
constructor arguments, `get_weight()` / `get_num_params()` contract and
`state_dict` keys (`_emb_module.weight`), so reference checkpoints load and the
modules drop in under src/models/{deepfm,dcn,lightgcn}.py.  The lookup itself is
the HIP row gather of libmi355x_recsys.so, not nn.Embedding's.
"""
from abc import abstractmethod
from typing import Iterable, Optional, Union

import torch
from torch import nn

from .. import _kernels


class IEmbedding(nn.Module):
    """forward(idx[B] | idx[B,F]) -> [B,D] | [B,F,D]; get_weight() -> [N,D] (differentiable)."""

    @abstractmethod
    def get_weight(self) -> torch.Tensor:
        ...

    def get_num_params(self) -> int:
        return sum([p.numel() for p in self.parameters()])


class VanillaEmbedding(IEmbedding):
    """One concatenated table of sum(field_dims) rows (reference base.py:23-75).

    `_emb_module` is kept as the parameter holder (nn.Embedding / nn.EmbeddingBag,
    xavier-uniform or N(0, 0.1) initialised exactly like the reference) so that
    `state_dict()` keys, `.sparse`, `.weight` and checkpoint loading are unchanged.
    """

    _emb_module: Union[nn.Embedding, nn.EmbeddingBag]

    def __init__(
        self,
        field_dims: Union[Iterable[int], int],
        hidden_size: int,
        mode: Optional[str] = None,
        initializer="xavier",
        **kwargs,
    ):
        super().__init__()
        assert mode in [None, "sum", "mean", "max"]
        rows = field_dims if isinstance(field_dims, int) else sum(field_dims)
        self._mode = mode
        # the holder module only owns the parameter (and its `sparse` flag): lookups never call it
        self._emb_module = (nn.Embedding(rows, hidden_size, **kwargs) if mode is None
                            else nn.EmbeddingBag(rows, hidden_size, mode=mode, **kwargs))
        if initializer == "xavier":
            nn.init.xavier_uniform_(self._emb_module.weight)
        else:
            nn.init.normal_(self._emb_module.weight, std=0.1)

    @property
    def sparse_grad(self) -> bool:
        return bool(getattr(self._emb_module, "sparse", False))

    def get_weight(self):
        return self._emb_module.weight

    def forward(self, x):
        rows = _kernels.gather_rows(x, self._emb_module.weight, self.sparse_grad)
        return _kernels.bag_reduce(rows, self._mode)
