"""LightGCN / SingleLightGCN with the propagation as fused CSR SpMM HIP kernels.

Drop-in for src/models/lightgcn.py:11-173 and src/models/base.py:8-35: same constructors,
`forward(matrix) -> (user_emb, item_emb)`, `get_reg_loss`, `get_embs`, attribute names
(`user_emb_table`, `item_emb_table`, `emb_table`) and state_dict keys.  The embedding tables
come from the same plug-in registry through get_weight() (differentiable).
"""
from abc import abstractmethod
from dataclasses import dataclass
from typing import List, Tuple, Union

import torch
from torch import nn

from . import _kernels
from .embeddings import IEmbedding, get_embedding
from .layers import SparseDropout


class IGraphBaseCore(nn.Module):
    @abstractmethod
    def get_emb_table(self, matrix) -> Tuple[torch.Tensor, torch.Tensor]:
        ...

    @abstractmethod
    def get_reg_loss(self, users, pos_items, neg_items) -> torch.Tensor:
        ...

    def forward(self, matrix):
        return self.get_emb_table(matrix)

    @abstractmethod
    def get_embs(self) -> List[Tuple[str, nn.Module]]:
        ...


class LightGCN(IGraphBaseCore):
    item_emb_table: IEmbedding
    user_emb_table: IEmbedding

    def __init__(self, num_user, num_item, num_layers=2, hidden_size=64, p_dropout=0, embedding_config=None):
        super().__init__()
        if embedding_config is None:
            embedding_config = {"name": "vanilla"}
        self.embedding_config = embedding_config
        self._init_embedding(num_user, num_item, hidden_size)
        self.num_layers = num_layers
        self._num_user = num_user
        self._num_item = num_item
        self._hidden_size = hidden_size
        if p_dropout > 0:
            self.sparse_dropout = SparseDropout(p_dropout)
        else:
            self.sparse_dropout = torch.nn.Identity(p_dropout)

    def _init_embedding(self, num_user, num_item, hidden_size):
        self.user_emb_table = get_embedding(self.embedding_config, num_user, hidden_size, field_name="user")
        self.item_emb_table = get_embedding(self.embedding_config, num_item, hidden_size, field_name="item")

    def get_emb_table(self, matrix):
        """matrix: sparse (num_user+num_item)^2 normalised adjacency -> (user_emb, item_emb)."""
        matrix = self.sparse_dropout(matrix)
        # E^0 = [user table; item table] is read as two row segments: no torch.cat copy
        res = _kernels.lightgcn_propagate(
            matrix, self.user_emb_table.get_weight(), self.item_emb_table.get_weight(), self.num_layers)
        return torch.split(res, (self._num_user, self._num_item))

    def get_reg_loss(self, users, pos_items, neg_items) -> torch.Tensor:
        user_emb = self.user_emb_table(users)
        pos_item_emb = self.item_emb_table(pos_items)
        neg_item_emb = self.item_emb_table(neg_items)
        reg_loss = (
            user_emb.norm(2).pow(2) + pos_item_emb.norm(2).pow(2) + neg_item_emb.norm(2).pow(2)
        ) / (2 * len(users))
        return reg_loss

    def get_embs(self):
        return [("user", self.user_emb_table), ("item", self.item_emb_table)]


class SingleLightGCN(IGraphBaseCore):
    emb_table: IEmbedding

    def __init__(self, num_user, num_item, num_layers=2, hidden_size=64, p_dropout=0, embedding_config=None):
        super().__init__()
        if embedding_config is None:
            embedding_config = {"name": "vanilla"}
        self.embedding_config = embedding_config
        self._init_embedding(num_user, num_item, hidden_size)
        self.num_layers = num_layers
        self._num_user = num_user
        self._num_item = num_item
        self._hidden_size = hidden_size
        if p_dropout > 0:
            self.sparse_dropout = SparseDropout(p_dropout)
        else:
            self.sparse_dropout = torch.nn.Identity(p_dropout)

    def _init_embedding(self, num_user, num_item, hidden_size):
        self.emb_table = get_embedding(self.embedding_config, [num_user, num_item], hidden_size,
                                       field_name="user-item")

    def get_emb_table(self, matrix):
        matrix = self.sparse_dropout(matrix)
        res = _kernels.lightgcn_propagate(matrix, self.emb_table.get_weight(), None, self.num_layers)
        return torch.split(res, (self._num_user, self._num_item))

    def get_reg_loss(self, users, pos_items, neg_items) -> torch.Tensor:
        indices = torch.cat([users, pos_items + self._num_user, neg_items + self._num_user])
        emb = self.emb_table(indices)
        return emb.norm(2).pow(2) / (2 * len(users))

    def get_embs(self):
        return [("user-item", self.emb_table)]


@dataclass
class LightGCNConfig:
    num_user: int
    num_item: int
    hidden_size: int = 64
    num_layers: int = 2


def get_sparsity_and_param(model: Union[LightGCN, SingleLightGCN]):
    if isinstance(model, LightGCN):
        embs = [model.user_emb_table, model.item_emb_table]
    elif isinstance(model, SingleLightGCN):
        embs = [model.emb_table]
    else:
        raise ValueError()
    max_params = (model._num_user + model._num_item) * model._hidden_size
    num_params = sum(emb.get_num_params() for emb in embs)
    return 1 - num_params / max_params, num_params
