"""QR (quotient-remainder) hashed embedding — reference: src/models/embeddings/qr_embedding.py:10-113.

Drop-in: same constructor arguments, parameter holders `emb1` / `emb2` (state_dict keys
`emb1.weight`, `emb2.weight`), initial distributions and quirks —
  * `emb1` has `divider` rows and is addressed by the REMAINDER, `emb2` has (N-1)//divider+1 rows
    and is addressed by the QUOTIENT (names inverted w.r.t. the paper: `divider: 2` = a 2-row emb1);
  * operation "cat" concatenates along dim=1, so [B,F] ids give [B,2F,D/2].
Index math, both row gathers and the combine are one HIP kernel (mi_dual_gather_fwd);
get_weight() is that kernel over arange(N) and stays differentiable.
"""
import functools
import math
from typing import List, Literal, Optional, Union

import torch
from torch import nn

from .. import _kernels
from .base import IEmbedding

_COMBINE = ("cat", "add", "mult")


def table_rows(num_item: int, divider: Optional[int]):
    """(divider, rows of emb1, rows of emb2); divider defaults to floor(sqrt(N))."""
    d = int(math.sqrt(num_item)) if divider is None else divider
    return d, d, (num_item - 1) // d + 1


class QRHashingEmbedding(IEmbedding):
    def __init__(
        self,
        field_dims: Union[int, List[int]],
        hidden_size: int,
        mode: Optional[str] = None,
        divider: Optional[int] = None,
        operation: Literal["cat", "add", "mult"] = "mult",
        initializer="uniform",
        sparse: bool = False,
    ):
        """Arguments as the reference's (src/models/embeddings/qr_embedding.py:14-22).  Extra, optional:

        sparse: hand the QUOTIENT table's gradient (emb2: ~N / divider rows, the big one) out in row (COO) form, like
            nn.Embedding(sparse=True) — for `optim.SparseAdam`; the remainder table (divider rows) stays dense.  Off by
            default: the reference's optimizers see dense gradients."""
        super().__init__()
        self._sparse2 = bool(sparse) and mode is None
        assert operation in _COMBINE
        assert operation != "cat" or hidden_size % 2 == 0
        self._operation, self._mode, self._hidden_size = operation, mode, hidden_size
        self._num_item = field_dims if isinstance(field_dims, int) else sum(field_dims)
        self._field_dims = None if isinstance(field_dims, int) else [int(v) for v in field_dims]
        self._hint = None                  # (device, small-field hint) for the backward, built on first use
        self._divider, rows1, rows2 = table_rows(self._num_item, divider)

        width = hidden_size // 2 if operation == "cat" else hidden_size
        make = nn.Embedding if mode is None else functools.partial(nn.EmbeddingBag, mode=mode)
        self.emb1, self.emb2 = make(rows1, width), make(rows2, width)

        init = {"normal": self._init_normal_weight, "uniform": self._init_uniform_weight}.get(initializer)
        if init is not None:
            init()

    def _tables(self):
        return self.emb1.weight, self.emb2.weight

    def _init_uniform_weight(self):
        # U(sqrt(1/N), 1) on both tables (the DLRM QR trick's range: a product is never ~0*0)
        low = math.sqrt(1 / self._num_item)
        for w in self._tables():
            nn.init.uniform_(w, low, 1)

    def _init_normal_weight(self):
        std = {"add": 0.1 / 2, "mult": math.sqrt(0.1)}.get(self._operation, 0.1)
        for w in self._tables():
            nn.init.normal_(w, std=std)

    def forward(self, tensor: torch.Tensor, offsets: Optional[torch.Tensor] = None):
        """offsets (optional extension, [F]): `tensor` then holds the model's raw per-field ids [B, F] and the per-field
        row offsets are added inside the lookup; returns (embeddings, row ids) — the reference's callers pass x + offsets
        (src/models/dcn.py:204), an elementwise launch of its own on the GPU."""
        w1, w2 = self._tables()
        d = self._divider
        if offsets is not None:
            if not self.takes_offsets(tensor):
                rows = tensor + offsets
                return self.forward(rows), rows
        if self._mode is None:
            fields = None
            if self._field_dims is not None and tensor.dim() == 2 and tensor.shape[1] == len(self._field_dims):
                # ids arrive as per-field ids + cumulative offsets (the CTR models' shared table): tell the backward which
                # fields have a handful of values
                if self._hint is None or self._hint[0] != tensor.device:
                    self._hint = (tensor.device, _kernels.small_field_hint(self._field_dims, d, tensor.device))
                fields = self._hint[1]
            return _kernels.dual_gather(tensor, w1, w2, mod1=d, div2=d, op=self._operation, fields=fields,
                                        sparse2=self._sparse2, offsets=offsets)
        # EmbeddingBag modes reduce each table's rows BEFORE the combine (two bags in the reference)
        bag1 = _kernels.bag_reduce(_kernels.gather_rows(tensor % d, w1), self._mode)
        bag2 = _kernels.bag_reduce(_kernels.gather_rows(tensor // d, w2), self._mode)
        if self._operation == "mult":
            return bag1 * bag2
        return bag1 + bag2 if self._operation == "add" else torch.cat([bag1, bag2], dim=1)

    def takes_offsets(self, tensor: torch.Tensor) -> bool:
        """Whether forward(tensor, offsets=...) folds the addition into the lookup kernel (float4 rows, [B, F] ids on the GPU)."""
        De = self.emb1.weight.shape[1]
        return bool(self._mode is None and tensor.is_cuda and tensor.dim() == 2 and 4 <= De <= 256 and De % 4 == 0
                    and (De // 4) & (De // 4 - 1) == 0)

    def get_weight(self):
        return self(torch.arange(self._num_item, device=self.emb1.weight.device))
