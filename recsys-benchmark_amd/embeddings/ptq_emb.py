"""Post-training-quantised embedding tables (inference only) — reference: src/models/embeddings/ptq_emb.py:7-96.

Constructor arguments, buffer names (`weight`, `scale`, `bias`) and the quantisation arithmetic are the reference's, so a
table quantised here is bit-identical to one quantised there (pinned by tests/golden/ptq.npz); the lookup dequantises
inside the HIP gather (mi_gather_rows_quant): 32- or 16-byte rows at D=16 instead of 64-byte fp32 rows.
"""
import torch

from .. import _kernels
from .base import IEmbedding

_TABLE_KEY = "embedding._emb_module.weight"


def _trained_table(checkpoint_path) -> torch.Tensor:
    """The fp32 table of a DeepFM / DCN checkpoint as the reference's trainers save it."""
    return torch.load(checkpoint_path, map_location="cpu")["state_dict"][_TABLE_KEY]


def affine_codes(table: torch.Tensor, n_bits: int):
    """(codes, scale, zero_point) with table ~ (codes - zero_point) * scale over the signed n_bits range: the scale spans
    [min, max] of the whole table, the zero point is truncated to the storage type, codes are rounded half-to-even
    (and clamped below 8 bits, where the storage type is wider than the range)."""
    lowest, highest = -(1 << (n_bits - 1)), (1 << (n_bits - 1)) - 1
    storage = torch.int16 if n_bits == 16 else torch.int8
    floor = table.min().cpu()                               # 0-dim fp32: the arithmetic below stays in fp32 tensors
    scale = (table.max().item() - floor) / (highest - lowest)
    zero_point = (lowest - floor / scale).to(storage)
    codes = torch.round(table / scale + zero_point)
    if n_bits < 8:
        codes = codes.clamp(lowest, highest)
    return codes.to(storage), scale, zero_point


class _FrozenTable(IEmbedding):
    """field_dims / num_factor / mode are accepted for the registry's calling convention only."""

    weight: torch.Tensor

    def get_num_params(self) -> int:
        rows, width = self.weight.shape
        return rows * width


class PTQEmb_Fp16(_FrozenTable):
    def __init__(self, field_dims, num_factor, mode, ori_checkpoint_path):
        super().__init__()
        self.register_buffer("weight", _trained_table(ori_checkpoint_path).to(torch.float16))

    def forward(self, x):
        return _kernels.gather_rows_quant(x, self.weight)                     # fp16 rows -> fp32

    def get_weight(self) -> torch.Tensor:
        return self.weight


class PTQEmb_Int(_FrozenTable):
    def __init__(self, field_dims, num_factor, mode, ori_checkpoint_path, n_bits=8):
        super().__init__()
        if n_bits not in (4, 8, 16):
            raise AssertionError("n_bits must be 4, 8 or 16")
        self.n_bits = n_bits
        codes, scale, zero_point = affine_codes(_trained_table(ori_checkpoint_path), n_bits)
        self.register_buffer("scale", scale)
        self.register_buffer("bias", zero_point)
        self.register_buffer("weight", codes)

    def forward(self, x):
        return _kernels.gather_rows_quant(x, self.weight, self.scale.reshape(1), self.bias.reshape(1))

    def get_weight(self) -> torch.Tensor:
        return (self.weight - self.bias) * self.scale
