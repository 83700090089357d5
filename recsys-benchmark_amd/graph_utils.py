"""Symmetric-normalised adjacency builders — reference: src/graph_utils.py:6-98.

Host-side, once per dataset: this defines the SpMM operand, so the reference's
conventions are kept exactly — a node's degree counts STORED entries (a duplicate (user, item) edge
counts twice), every stored entry gets d_i^-1/2 d_j^-1/2, duplicates are summed when the matrix is
coalesced (the CSR conversion), and an omitted `num_user` defaults to the largest user id.
"""
from typing import Dict, List, Literal, Optional

import torch


def _edge_arrays(graph: Dict[int, List[int]]):
    """(user of every interaction, item of every interaction) in the dict's iteration order."""
    users, items = [], []
    for user, its in graph.items():
        users += [user] * len(its)
        items += list(its)
    return torch.tensor(users, dtype=torch.int64), torch.tensor(items, dtype=torch.int64)


def _inv_sqrt_degree(index: torch.Tensor, size: int) -> torch.Tensor:
    # number of stored entries per index, as float32 like the reference's sum over a matrix of ones
    return torch.bincount(index, minlength=size).to(torch.float32).pow(-0.5)


def get_adj(graph: Dict[int, List[int]], num_item: int, num_user: Optional[int] = None,
            normalize=False) -> torch.Tensor:
    """Rectangular user x item interaction matrix (sparse COO); with `normalize`, D_u^-1/2 R D_i^-1/2
    coalesced (reference graph_utils.py:6-44)."""
    num_user = num_user or max(graph.keys())
    users, items = _edge_arrays(graph)
    where = torch.stack([users, items])
    if not normalize:
        return torch.sparse_coo_tensor(where, torch.ones(users.numel()), size=(num_user, num_item))
    weight = _inv_sqrt_degree(users, num_user)[users] * _inv_sqrt_degree(items, num_item)[items]
    return torch.sparse_coo_tensor(where, weight, size=(num_user, num_item)).coalesce()


def calculate_sparse_graph_adj_norm(graph: Dict[int, List[int]], num_item: int, num_user: Optional[int] = None,
                                    layout: Literal["coo", "csr"] = "csr") -> torch.Tensor:
    """A_hat = D^-1/2 [[0, R], [R^T, 0]] D^-1/2 over the (users ++ items) node set (reference
    graph_utils.py:47-98).  "coo" keeps the reference's entry order — per user its R entries, then
    the mirrored R^T entries — and stays uncoalesced; "csr" is the coalesced form."""
    num_user = num_user or max(graph.keys())
    n = num_user + num_item
    rows: List[int] = []
    cols: List[int] = []
    for user, its in graph.items():
        nodes = [it + num_user for it in its]           # item nodes follow the user nodes
        rows += [user] * len(its) + nodes
        cols += nodes + [user] * len(its)
    rows_t, cols_t = torch.tensor(rows, dtype=torch.int64), torch.tensor(cols, dtype=torch.int64)
    scale = _inv_sqrt_degree(cols_t, n)                  # the matrix is symmetric: column counts are the degrees
    matrix = torch.sparse_coo_tensor(torch.stack([rows_t, cols_t]), scale[rows_t] * scale[cols_t], size=(n, n))
    return matrix.to_sparse_csr() if layout == "csr" else matrix
