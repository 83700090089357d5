"""DCN-Mix and DCNv2 — reference: src/models/dcn.py:11-222.  Same constructors, state_dict keys
(`offsets, embedding.*, cross_head.*, _dnn.*` / `linear_model.weight, _last_fc.*`), forward
signature and `load`.  Embedding lookup and cross network run as HIP kernels; the `_dnn` MLP
(Linear/BatchNorm1d/ReLU/Dropout) stays on rocBLAS through PyTorch like DeepFM's tail."""
from typing import Any, Dict, List, Optional, Union, cast

import torch
from torch import nn

from . import _kernels
from .embeddings import IEmbedding, get_embedding
from .layer_dcn import DCN_MixHead, DCNHead
from .mlp import run_tail


def _offsets(field_dims):
    t = torch.cat([torch.tensor([0], dtype=torch.long), torch.tensor(field_dims)])
    return torch.cumsum(t[:-1], 0).unsqueeze(0)


def _mlp(inp_size, hidden_sizes, p_dropout):
    layers: List[nn.Module] = []
    for size in hidden_sizes:
        layers += [nn.Linear(inp_size, size), nn.BatchNorm1d(size), nn.ReLU(), nn.Dropout(p_dropout)]
        inp_size = size
    return layers, inp_size


class DCN_Mix(nn.Module):
    embedding: IEmbedding

    def __init__(self, field_dims: List[int], num_factor: int, hidden_sizes: List[int], num_layers: int = 3,
                 num_experts: int = 4, rank: int = 64, activation: Optional[str] = None,
                 embedding_config: Optional[Dict] = None, p_dropout=0.5, empty_embedding=False):
        super().__init__()
        if not embedding_config:
            embedding_config = {"name": "vanilla"}
        if not empty_embedding:
            self.embedding = get_embedding(embedding_config, field_dims, num_factor, mode=None, field_name="dcn")
        inp_size = num_factor * len(field_dims)
        self.cross_head = DCN_MixHead(num_experts, num_layers, rank, inp_size, activation)
        layers, inp_size = _mlp(inp_size, hidden_sizes, p_dropout)
        layers.append(nn.Linear(inp_size, 1))
        self._dnn = nn.Sequential(*layers)
        self.register_buffer("offsets", _offsets(field_dims))
        self._num_rows = int(sum(field_dims))

    def forward(self, x):
        """x: int [B, F] -> logits [B]."""
        x = x + self.offsets
        _kernels.note_field_layout(x, self.offsets, self._num_rows)     # lets the sparse optimizer sort field by field
        emb = self.embedding(x)
        bs = x.shape[0]
        cross_logit = self.cross_head(emb.reshape(bs, -1))
        return run_tail(self._dnn, cross_logit).squeeze(-1)

    @classmethod
    def load(cls, checkpoint: Union[str, Dict[str, Any]], strict=True, *, empty_embedding=False):
        if isinstance(checkpoint, str):
            checkpoint = torch.load(checkpoint, map_location="cpu")
        checkpoint = cast(Dict[str, Any], checkpoint)
        model_config = dict(checkpoint["model_config"])
        model_config.pop("compile_model", None)   # nothing to compile: the cross net is the kernel path
        model_config.pop("name", None)
        model = cls(checkpoint["field_dims"], **model_config, empty_embedding=empty_embedding)
        state = {k.replace("_orig_mod.", "", 1): v for k, v in checkpoint["state_dict"].items()}
        model.load_state_dict(state, strict=strict)
        return model


class DCNv2(nn.Module):
    def __init__(self, field_dims: List[int], num_factor: int, hidden_sizes: List[int], num_layers: int = 3,
                 embedding_config: Optional[Dict] = None, p_dropout: float = 0.5, empty_embedding: bool = False,
                 structure: str = "Stacked"):
        super().__init__()
        if not embedding_config:
            embedding_config = {"name": "vanilla"}
        if not empty_embedding:
            self.embedding = get_embedding(embedding_config, field_dims, num_factor, mode=None, field_name="dcn")
        inp_size = num_factor * len(field_dims)
        self.linear_model = nn.EmbeddingBag(sum(field_dims), 1, mode="sum")
        self.cross_head = DCNHead(num_layers, inp_size)
        self.structure = structure
        layers, dnn_out = _mlp(inp_size, hidden_sizes, p_dropout)
        if structure == "Stacked":
            self._last_fc = nn.Linear(dnn_out, 1)
        else:
            self._last_fc = nn.Linear(inp_size + dnn_out, 1)
        self._dnn = nn.Sequential(*layers)
        self.register_buffer("offsets", _offsets(field_dims))
        self._num_rows = int(sum(field_dims))

    def forward(self, x):
        x = x + self.offsets
        _kernels.note_field_layout(x, self.offsets, self._num_rows)     # lets the sparse optimizer sort field by field
        emb = self.embedding(x)
        bs = x.shape[0]
        emb = emb.reshape(bs, -1)
        cross_logit = self.cross_head(emb)
        if self.structure == "Stacked":
            logit = run_tail(self._dnn, cross_logit)
        else:
            logit = torch.concat([cross_logit, run_tail(self._dnn, emb)], dim=1)
        # first-order term: EmbeddingBag(N,1,"sum") over the row ids = a D=1 row gather + bag sum
        linear = _kernels.gather_rows(x, self.linear_model.weight, bool(self.linear_model.sparse)).sum(1)
        return (self._last_fc(logit) + linear).squeeze(-1)
