"""SparseDropout — reference: src/models/layers.py:5-37: dropout applied to the STORED values of a sparse matrix (the
sparsity pattern is kept; a COO matrix is coalesced first).  Identity at the BASELINE config (p_dropout = 0)."""
import torch
from torch import nn


class SparseDropout(nn.Module):
    def __init__(self, p=0.5, inplace=False):
        super().__init__()
        self._dropout = nn.Dropout(p, inplace)          # a submodule, so .train() / .eval() reach it

    def forward(self, matrix: torch.Tensor) -> torch.Tensor:
        layout, shape = matrix.layout, matrix.size()
        if layout == torch.sparse_csr:
            pattern = (matrix.crow_indices(), matrix.col_indices())
            rebuild = torch.sparse_csr_tensor
        elif layout == torch.sparse_coo:
            matrix = matrix.coalesce()
            pattern = (matrix.indices(),)
            rebuild = torch.sparse_coo_tensor
        else:
            raise ValueError(f"Not supported matrix layout: {layout}")
        return rebuild(*pattern, self._dropout(matrix.values()), shape)
