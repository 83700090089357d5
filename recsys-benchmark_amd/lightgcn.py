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
from .embeddings import IEmbedding, VanillaEmbedding, get_embedding
from .layers import SparseDropout


class IGraphBaseCore(nn.Module):
    """The interface of src/models/base.py:8-35: forward(adjacency) -> (user embeddings, item embeddings)."""

    @abstractmethod
    def get_emb_table(self, matrix) -> Tuple[torch.Tensor, torch.Tensor]:
        ...

    @abstractmethod
    def get_reg_loss(self, users, pos_items, neg_items) -> torch.Tensor:
        ...

    @abstractmethod
    def get_embs(self) -> List[Tuple[str, nn.Module]]:
        ...

    def forward(self, matrix):
        return self.get_emb_table(matrix)


def _squared_l2(t: torch.Tensor) -> torch.Tensor:
    return t.norm(2).pow(2)          # as the reference forms it (norm, then square)


class _Propagating(IGraphBaseCore):
    """Constructor bookkeeping shared by the two-table and the one-table model (same arguments, same attributes)."""

    def __init__(self, num_user, num_item, num_layers=2, hidden_size=64, p_dropout=0, embedding_config=None):
        super().__init__()
        self.embedding_config = {"name": "vanilla"} if embedding_config is None else embedding_config
        self._num_user, self._num_item, self._hidden_size = num_user, num_item, hidden_size
        self.num_layers = num_layers
        self._init_embedding(num_user, num_item, hidden_size)
        # dropout on the adjacency's stored values; an Identity keeps the attribute (and module tree) when it is off
        self.sparse_dropout = SparseDropout(p_dropout) if p_dropout > 0 else torch.nn.Identity(p_dropout)

    def _init_embedding(self, num_user, num_item, hidden_size):
        raise NotImplementedError


class LightGCN(_Propagating):
    item_emb_table: IEmbedding
    user_emb_table: IEmbedding

    def _init_embedding(self, num_user, num_item, hidden_size):
        self.user_emb_table = get_embedding(self.embedding_config, num_user, hidden_size, field_name="user")
        self.item_emb_table = get_embedding(self.embedding_config, num_item, hidden_size, field_name="item")

    def get_emb_table(self, matrix):
        """matrix: sparse (num_user+num_item)^2 normalised adjacency -> (user_emb, item_emb).  E^0 = [user table; item
        table] is read as two row segments (no concatenated copy) and the result comes back as the two tables."""
        return _kernels.lightgcn_propagate(self.sparse_dropout(matrix), self.user_emb_table.get_weight(),
                                           self.item_emb_table.get_weight(), self.num_layers)

    def _plain_tables(self) -> bool:
        return all(type(t) is VanillaEmbedding and t._mode is None and not t.sparse_grad
                   for t in (self.user_emb_table, self.item_emb_table))

    def get_reg_loss(self, users, pos_items, neg_items) -> torch.Tensor:
        """(|e_u|^2 + |e_i+|^2 + |e_i-|^2) / (2 * batch) over the batch's rows of the INPUT tables."""
        users_t, items_t = self.user_emb_table, self.item_emb_table
        if self._plain_tables() and users.dim() == 1:
            # gathers, squares and the sum in one launch each way (~30 small launches otherwise)
            from .losses import reg_loss_rows

            return reg_loss_rows(users_t.get_weight(), items_t.get_weight(), users, pos_items, neg_items)
        total = _squared_l2(users_t(users)) + _squared_l2(items_t(pos_items)) + _squared_l2(items_t(neg_items))
        return total / (2 * len(users))

    def forward_with_reg_loss(self, matrix, users, pos_items, neg_items, batch_rows_only: bool = False):
        """(user_emb, item_emb, reg_loss) = (*self(matrix), self.get_reg_loss(users, pos_items, neg_items)) — the two calls
        of the reference's `_train_step` (src/trainer/lightgcn.py:389-402) as one autograd node when the tables are plain,
        so that the regulariser's 3 B gradient rows are added into the propagation's gradient instead of travelling as two
        zero-filled dense tensors (an extension: the two separate calls stay valid and give the same numbers).

        batch_rows_only=True: the caller promises to read user_emb only at `users` and item_emb only at `pos_items` /
        `neg_items` (the BPR and InfoNCE terms of the training step do exactly that): the last propagation layer then
        computes only those rows — every other row of the two returned tables is UNDEFINED.  Same values at the rows
        that are read, same gradients."""
        if self._plain_tables() and users.dim() == 1 and self.num_layers > 0:
            return _kernels.lightgcn_propagate_reg(self.sparse_dropout(matrix), self.user_emb_table.get_weight(),
                                                   self.item_emb_table.get_weight(), self.num_layers, users, pos_items, neg_items,
                                                   batch_rows_only=batch_rows_only)
        user_emb, item_emb = self(matrix)
        return user_emb, item_emb, self.get_reg_loss(users, pos_items, neg_items)

    def get_embs(self):
        return [("user", self.user_emb_table), ("item", self.item_emb_table)]


class SingleLightGCN(_Propagating):
    emb_table: IEmbedding

    def _init_embedding(self, num_user, num_item, hidden_size):
        self.emb_table = get_embedding(self.embedding_config, [num_user, num_item], hidden_size, field_name="user-item")

    def get_emb_table(self, matrix):
        both = _kernels.lightgcn_propagate(self.sparse_dropout(matrix), self.emb_table.get_weight(), None, self.num_layers)
        return torch.split(both, (self._num_user, self._num_item))

    def get_reg_loss(self, users, pos_items, neg_items) -> torch.Tensor:
        item_base = self._num_user                         # items follow the users in the one table
        batch_rows = torch.cat([users, pos_items + item_base, neg_items + item_base])
        return _squared_l2(self.emb_table(batch_rows)) / (2 * len(users))

    def forward_with_reg_loss(self, matrix, users, pos_items, neg_items, batch_rows_only: bool = False):
        """LightGCN.forward_with_reg_loss for the one-table model: (user_emb, item_emb, reg_loss) with propagation and
        regulariser as one autograd node when the table is plain."""
        t = self.emb_table
        if type(t) is VanillaEmbedding and t._mode is None and not t.sparse_grad and users.dim() == 1 and self.num_layers > 0:
            both, reg = _kernels.lightgcn_propagate_reg(self.sparse_dropout(matrix), t.get_weight(), None, self.num_layers,
                                                        users, pos_items, neg_items, batch_rows_only=batch_rows_only,
                                                        item_base=self._num_user)
            user_emb, item_emb = torch.split(both, (self._num_user, self._num_item))
            return user_emb, item_emb, reg
        user_emb, item_emb = self(matrix)
        return user_emb, item_emb, self.get_reg_loss(users, pos_items, neg_items)

    def get_embs(self):
        return [("user-item", self.emb_table)]


@dataclass
class LightGCNConfig:
    num_user: int
    num_item: int
    hidden_size: int = 64
    num_layers: int = 2


def get_sparsity_and_param(model: Union[LightGCN, SingleLightGCN]):
    """(1 - stored parameters / dense parameters, stored parameters) over the model's tables."""
    if not isinstance(model, (LightGCN, SingleLightGCN)):
        raise ValueError()
    stored = sum(table.get_num_params() for _, table in model.get_embs())
    dense = (model._num_user + model._num_item) * model._hidden_size
    return 1 - stored / dense, stored


def train_items_csr(graph, num_users: int, device=None):
    """CSR (crow, col) of the train interactions, from the reference dataset's `graph`
    (user -> iterable of item ids, CFGraphDataset.get_graph()).  Built once per dataset on the host."""
    lens = [len(graph[u]) if u in graph else 0 for u in range(num_users)] if isinstance(graph, dict) else \
        [len(graph[u]) for u in range(num_users)]
    crow = torch.zeros(num_users + 1, dtype=torch.int64)
    crow[1:] = torch.cumsum(torch.tensor(lens, dtype=torch.int64), 0)
    flat = [i for u in range(num_users) for i in (graph[u] if (not isinstance(graph, dict) or u in graph) else ())]
    col = torch.tensor(flat, dtype=torch.int64)
    return crow.to(device), col.to(device)


def score_topk(user_embs: torch.Tensor, item_embs: torch.Tensor, users: torch.Tensor, k: int, train_csr=None):
    """The scoring tail of validation (src/trainer/lightgcn.py:122-138): `user_embs[users] @ item_embs.T`,
    -inf on the items each user has in train (`train_csr` = train_items_csr(graph, ...); None = no filter) and
    torch.topk(scores, k) indices — gather, fp32 MFMA GEMM, and one mask + top-k workgroup per user, no Python
    loop over users.  Returns int64 [len(users), k] (score descending, ties by ascending item id)."""
    from . import _kernels, _lib

    dev = _lib.require_gpu(user_embs, item_embs, users)
    users = _kernels._i64c(users).view(-1)
    rows = _kernels.gather_rows(users, user_embs.detach())
    items = _kernels._f32c(item_embs.detach())
    B, D, I = rows.shape[0], rows.shape[1], items.shape[0]
    scores = torch.empty((B, I), dtype=torch.float32, device=dev)
    _kernels.gemm(rows, items, scores, B, I, D, D, D, I, transB=True)               # rows . items^T
    out = torch.empty((B, k), dtype=torch.int64, device=dev)
    crow, col = (None, None) if train_csr is None else train_csr
    _lib.check(_lib.load().mi_mask_topk_rows(scores.data_ptr(), scores.stride(0), B, I, users.data_ptr(),
                                             _lib.ptr(crow), _lib.ptr(col), k, out.data_ptr(), None,
                                             _lib.stream_ptr(dev)), "mi_mask_topk_rows")
    return out
