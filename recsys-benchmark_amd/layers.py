"""SparseDropout — reference: src/models/layers.py:5-37 (dropout on the stored values of a sparse
matrix, pattern unchanged).  Identity at the BASELINE config (p_dropout = 0)."""
import torch
from torch import nn


class SparseDropout(nn.Module):
    def __init__(self, p=0.5, inplace=False):
        super().__init__()
        self._dropout = nn.Dropout(p, inplace)

    def forward(self, matrix: torch.Tensor) -> torch.Tensor:
        if matrix.is_sparse_csr:
            values = self._dropout(matrix.values())
            return torch.sparse_csr_tensor(matrix.crow_indices(), matrix.col_indices(), values, matrix.size())
        elif matrix.layout == torch.sparse_coo:
            matrix = matrix.coalesce()
            values = self._dropout(matrix.values())
            return torch.sparse_coo_tensor(matrix.indices(), values, matrix.size())
        raise ValueError(f"Not supported matrix layout: {matrix.layout}")
