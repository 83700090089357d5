"""QR (quotient-remainder) hashed embedding — reference: src/models/embeddings/qr_embedding.py:10-113.

Same constructor, `emb1` / `emb2` parameter holders (state_dict keys `emb1.weight`,
`emb2.weight`), initialisers and quirks:
  * `emb1` has `divider` rows and is indexed by the REMAINDER, `emb2` has
    (N-1)//divider+1 rows and is indexed by the QUOTIENT (the naming is inverted
    relative to the paper; `divider: 2` means a 2-row emb1);
  * `operation="cat"` concatenates along dim=1, so a [B,F] input gives [B,2F,D/2].
The lookup, the `%` / `//` index math and the combine run in one HIP kernel
(mi_dual_gather_fwd); get_weight() is that kernel over arange(N), differentiable.
"""
import math
from typing import List, Literal, Optional, Union

import torch
from torch import nn

from .. import _kernels
from .base import IEmbedding


class QRHashingEmbedding(IEmbedding):
    def __init__(
        self,
        field_dims: Union[int, List[int]],
        hidden_size: int,
        mode: Optional[str] = None,
        divider: Optional[int] = None,
        operation: Literal["cat", "add", "mult"] = "mult",
        initializer="uniform",
    ):
        super().__init__()
        assert operation in ["cat", "add", "mult"]
        if operation == "cat":
            assert hidden_size % 2 == 0

        if isinstance(field_dims, int):
            field_dims = [field_dims]

        num_item = sum(field_dims)
        if divider is None:
            divider = int(math.sqrt(num_item))

        emb_size = hidden_size
        if operation == "cat":
            emb_size = hidden_size // 2

        self._operation = operation
        size = (num_item - 1) // divider + 1

        if mode is None:
            self.emb1 = nn.Embedding(divider, emb_size)
            self.emb2 = nn.Embedding(size, emb_size)
        else:
            self.emb1 = nn.EmbeddingBag(divider, emb_size, mode=mode)
            self.emb2 = nn.EmbeddingBag(size, emb_size, mode=mode)
        self._mode = mode

        self._hidden_size = hidden_size
        self._divider = divider
        self._num_item = num_item

        if initializer == "normal":
            self._init_normal_weight()
        elif initializer == "uniform":
            self._init_uniform_weight()

    def _init_uniform_weight(self):
        alpha = math.sqrt(1 / self._num_item)
        nn.init.uniform_(self.emb1.weight, alpha, 1)
        nn.init.uniform_(self.emb2.weight, alpha, 1)

    def _init_normal_weight(self):
        std = 0.1
        if self._operation == "add":
            std = std / 2
        elif self._operation == "mult":
            std = math.sqrt(std)
        nn.init.normal_(self.emb1.weight, std=std)
        nn.init.normal_(self.emb2.weight, std=std)

    def forward(self, tensor: torch.Tensor):
        if self._mode is None:
            return _kernels.dual_gather(tensor, self.emb1.weight, self.emb2.weight,
                                        mod1=self._divider, div2=self._divider, op=self._operation)
        # bag modes reduce each table's rows BEFORE the combine (two EmbeddingBags in the
        # reference, qr_embedding.py:60-63,98-99): two row gathers, then the small combine.
        inp1 = tensor % self._divider
        inp2 = tensor // self._divider
        emb1 = _kernels.bag_reduce(_kernels.gather_rows(inp1, self.emb1.weight), self._mode)
        emb2 = _kernels.bag_reduce(_kernels.gather_rows(inp2, self.emb2.weight), self._mode)
        if self._operation == "cat":
            return torch.cat([emb1, emb2], dim=1)
        elif self._operation == "add":
            return emb1 + emb2
        return emb1 * emb2

    def get_weight(self):
        arr = torch.arange(self._num_item, device=self.emb1.weight.device)
        return self(arr)
