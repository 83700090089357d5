"""Post-training-quantised embedding tables (inference) — reference: src/models/embeddings/ptq_emb.py:7-96.
Same constructors (the table is quantised from `ori_checkpoint_path`'s
`state_dict["embedding._emb_module.weight"]`), buffers (`weight`, `scale`, `bias`) and rounding; the
lookup dequantises inside the HIP gather (mi_gather_rows_quant): 32 / 16-byte rows at D=16 instead of 64."""
import torch

from .. import _kernels
from .base import IEmbedding


class PTQEmb_Fp16(IEmbedding):
    def __init__(self, field_dims, num_factor, mode, ori_checkpoint_path):
        super().__init__()
        checkpoint = torch.load(ori_checkpoint_path, map_location="cpu")
        emb = checkpoint["state_dict"]["embedding._emb_module.weight"]
        self.register_buffer("weight", emb.to(torch.float16))

    def forward(self, x):
        return _kernels.gather_rows_quant(x, self.weight)

    def get_weight(self) -> torch.Tensor:
        return self.weight

    def get_num_params(self) -> int:
        return self.weight.shape[0] * self.weight.shape[1]


class PTQEmb_Int(IEmbedding):
    def __init__(self, field_dims, num_factor, mode, ori_checkpoint_path, n_bits=8):
        super().__init__()
        checkpoint = torch.load(ori_checkpoint_path, map_location="cpu")
        emb: torch.Tensor = checkpoint["state_dict"]["embedding._emb_module.weight"]
        assert n_bits in [4, 8, 16]
        self.n_bits = n_bits
        q_min = (-1) * (1 << (self.n_bits - 1))
        q_max = (1 << (self.n_bits - 1)) - 1
        r_min = emb.min().cpu()
        scale = (emb.max().item() - r_min) / (q_max - q_min)
        dtype = torch.int16 if self.n_bits == 16 else torch.int8
        bias = (q_min - r_min / scale).to(dtype)
        self.register_buffer("scale", scale)
        self.register_buffer("bias", bias)
        weight = emb / scale + bias
        torch.round_(weight)
        if n_bits < 8:
            torch.clamp_(weight, q_min, q_max)
        self.register_buffer("weight", weight.to(dtype))

    def forward(self, x):
        return _kernels.gather_rows_quant(x, self.weight, self.scale.reshape(1), self.bias.reshape(1))

    def get_weight(self) -> torch.Tensor:
        return (self.weight - self.bias) * self.scale

    def get_num_params(self) -> int:
        return self.weight.shape[0] * self.weight.shape[1]
