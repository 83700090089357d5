"""CSR-stored pruned embedding table (inference only).

Reference: src/models/embeddings/pruned_embedding.py:11-204 — a numba `@cuda.jit` kernel
(32-thread blocks, one thread per id, serial over the row) plus a numba CPU kernel.  Here the
CSR triple lives in torch buffers and the row densify is a HIP kernel (mi_csr_rows_fwd).
"""
from typing import List, Optional, Union

import torch

from .. import _kernels
from .base import IEmbedding


class PrunedEmbedding(IEmbedding):
    def __init__(self, field_dims: Union[int, List[int]], hidden_size: int, mode: Optional[str] = None):
        super().__init__()
        if isinstance(field_dims, int):
            field_dims = [field_dims]
        self._hidden_size = hidden_size
        self._num_item = sum(field_dims)
        self._mode = mode
        self.is_cuda = False
        self.register_buffer("values", torch.zeros(0, dtype=torch.float32))
        self.register_buffer("crow_indices", torch.zeros(self._num_item + 1, dtype=torch.int64))
        self.register_buffer("col_indices", torch.zeros(0, dtype=torch.int64))

    @classmethod
    @torch.no_grad()
    def from_other_emb(cls, emb: IEmbedding, mode=None) -> "PrunedEmbedding":
        return cls.from_weight(emb.get_weight(), mode)

    @classmethod
    def from_weight(cls, weight: torch.Tensor, mode=None) -> "PrunedEmbedding":
        num_item, hidden_size = weight.shape
        result = cls(num_item, hidden_size, mode)
        weight = weight.detach()
        if weight.layout != torch.sparse_csr:
            weight = weight.to_sparse_csr()
        result.values = weight.values().to(torch.float32).contiguous()
        result.crow_indices = weight.crow_indices().to(torch.int64).contiguous()
        result.col_indices = weight.col_indices().to(torch.int64).contiguous()
        result.is_cuda = result.values.is_cuda
        return result

    def to_cuda(self):
        """Move the CSR triple to the GPU (reference API, pruned_embedding.py:51-65)."""
        self.to("cuda")
        self.is_cuda = True

    def _apply(self, fn, *a, **kw):
        out = super()._apply(fn, *a, **kw)
        self.is_cuda = self.crow_indices.is_cuda
        return out

    def get_weight(self):
        sparse_csr = torch.sparse_csr_tensor(
            self.crow_indices, self.col_indices, self.values,
            size=(self._num_item, self._hidden_size), dtype=torch.float32,
        )
        return sparse_csr.to_dense()

    def forward(self, x):
        out = _kernels.csr_rows(self.values, self.crow_indices, self.col_indices, x,
                                self._hidden_size, self._num_item)
        return _kernels.bag_reduce(out, self._mode)
