"""Test-side restatement of the fused MLP tail's dropout mask (recsys-benchmark_amd/csrc/tail_gemm.hpp: `Drop`), so that
the oracle can be given the very mask the kernels recompute: 16 bits of splitmix64(seed + salt', (m*ld + c) / 4) per
feature, keep when >= round(p * 65536).  Integer arithmetic only (torch int64 wraps like uint64)."""
import torch

_MASK64 = (1 << 64) - 1


def _s64(x: int) -> int:
    x &= _MASK64
    return x - (1 << 64) if x >= (1 << 63) else x


def _lsr(z: torch.Tensor, s: int) -> torch.Tensor:
    return (z >> s) & ((1 << (64 - s)) - 1)


def tail_keep_scale(seed: int, salt: int, M: int, ld: int, p: float, device="cpu") -> torch.Tensor:
    """[M, ld] multipliers (0 or 1/(1-p)) of the activation whose row pitch is `ld` (ld % 4 == 0)."""
    if p <= 0.0:
        return torch.ones(M, ld, device=device)
    assert ld % 4 == 0
    sd = _s64(seed + 0xD1B54A32D192ED03 * salt)
    idx = torch.arange(M * ld // 4, dtype=torch.int64, device=device)
    z = sd + _s64(0x9E3779B97F4A7C15) * (idx + 1)
    z = (z ^ _lsr(z, 30)) * _s64(0xBF58476D1CE4E5B9)
    z = (z ^ _lsr(z, 27)) * _s64(0x94D049BB133111EB)
    h = z ^ _lsr(z, 31)
    thr = int(p * 65536.0 + 0.5)
    parts = [(_lsr(h, 16 * j) & 0xFFFF) if j else (h & 0xFFFF) for j in range(4)]
    u = torch.stack(parts, 1).reshape(M, ld)
    return (u >= thr).float() / (1.0 - p)
