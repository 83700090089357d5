"""DHE deep-hash embedding — reference: src/models/embeddings/dh_embedding.py:16-362.

k universal hashes ((a*(id+prefix+1)+b) mod p) mod m -> uniform [-1,1] -> MLP(Linear/BN/Mish).
The reference precomputes a [N, k] fp32 cache (4 KB per row at k=1024) and gathers from it; here
the hash features are generated on the fly by a HIP kernel (mi_dhe_hash, int64 floor-mod,
bit-exact with torch), so no cache is stored: `cached` only selects the same control flow as the
reference.  The MLP is a dense GEMM chain and stays on rocBLAS through PyTorch.
The slopes / biases / primes are drawn exactly as the reference draws them
(torch.Generator().manual_seed(0)); its prime asset (all 74,518 primes in (1e6, 2,059,181]) is
regenerated here with a sieve.

use_universal_hash=False (dh_embedding.py:155-196): every item gets its OWN k hash functions, drawn from torch's global
CPU generator re-seeded with item + prefix.  That stream only exists on the host, so — exactly like the reference —
the [N, k] feature table is built once on the host at construction (cached=True) or per unique id (cached=False, eval),
moved to the MLP's device, and the hot path is the HIP row gather from it followed by the MLP.
"""
from typing import Final, List, Optional, Union

import numpy as np
import torch
from torch import nn

from .. import _kernels
from .base import IEmbedding

LARGE_INT: Final[int] = int(1e9)
NEGATIVE_LARGE_INT: Final[int] = -LARGE_INT
_PRIME_LO, _PRIME_HI, _PRIME_COUNT = 1_000_000, 2_059_181, 74518
_primes_cache: Optional[torch.Tensor] = None


def large_primes() -> torch.Tensor:
    """All primes p with 1e6 < p <= 2,059,181 (== src/assets/large_prime_74518.json)."""
    global _primes_cache
    if _primes_cache is None:
        sieve = np.ones(_PRIME_HI + 1, dtype=bool)
        sieve[:2] = False
        for i in range(2, int(_PRIME_HI ** 0.5) + 1):
            if sieve[i]:
                sieve[i * i:: i] = False
        p = np.nonzero(sieve)[0]
        p = p[p > _PRIME_LO]
        assert len(p) == _PRIME_COUNT
        _primes_cache = torch.from_numpy(p.astype(np.int64))
    return _primes_cache


import os as _os

OWN_MLP = _os.environ.get("MI_DHE_OWN_MLP", "1") == "1"      # 0: the MLP as nn.Sequential on the library GEMM (round-3 form)


class DHEmbedding(IEmbedding):
    COUNTER = 0

    def __init__(
        self,
        field_dims: Union[int, List[int]],
        out_size: int,
        mode: Optional[str] = None,
        inp_size: int = 1024,
        hidden_sizes: Optional[List[int]] = None,
        use_bn: Union[bool, int] = 2,
        cached: bool = True,
        prime_file: Optional[str] = None,
        cache_path: str = "",
        compute_v2=False,
        use_universal_hash=True,
    ):
        super().__init__()
        if isinstance(field_dims, int):
            field_dims = [field_dims]
        if isinstance(use_bn, bool):
            use_bn = int(use_bn)

        num_item = sum(field_dims)
        self.m = int(1e6)
        self._prefix = DHEmbedding.COUNTER
        DHEmbedding.COUNTER += num_item

        if prime_file is None:
            primes = large_primes()
        else:
            import json

            with open(prime_file) as fin:
                primes = torch.tensor(json.load(fin))
        self._primes = primes
        self._inp_size = inp_size
        self._num_item = num_item
        self._use_universal_hash = bool(use_universal_hash)

        if self._use_universal_hash:     # (the per-item form registers no buffers: same state_dict keys as the reference)
            rng = torch.Generator()
            rng.manual_seed(0)
            self.register_buffer("_slopes", self._random_nonzero_int(inp_size, rng))
            self.register_buffer("_bias", self._random_nonzero_int(inp_size, rng))
            p_idx = torch.randint(0, len(primes), (inp_size,), generator=rng)
            self.register_buffer("_primes_choices", self._primes[p_idx])

        layers: List[nn.Module] = []
        if hidden_sizes is None:
            hidden_sizes = []
        hidden_sizes.append(out_size)  # (mutates the caller's list, as the reference does)
        for size in hidden_sizes:
            layers.append(nn.Linear(inp_size, size))
            if use_bn == 1:
                layers.append(nn.Mish())
                layers.append(nn.BatchNorm1d(size))
            elif use_bn == 2:
                layers.append(nn.BatchNorm1d(size))
                layers.append(nn.Mish())
            else:
                layers.append(nn.Mish())
            inp_size = size
        self._seq = nn.Sequential(*layers)

        self._use_cache = cached
        self._use_bn = use_bn
        self._out_size = out_size
        self.compute_v2 = compute_v2
        self._mode = mode
        self._emb = None
        # per-item hashes: the [N, k] feature table, host-built (one reseed of the global generator per item)
        self._cache: Optional[torch.Tensor] = None
        if not self._use_universal_hash and cached:
            if cache_path:
                import os

                if os.path.exists(cache_path):
                    self._cache = torch.load(cache_path)
            if self._cache is None:
                self._cache = self._seeded_hash_rows(range(self._num_item))
                if cache_path:
                    torch.save(self._cache, cache_path)

    def _seeded_hash_rows(self, items) -> torch.Tensor:
        """[len(items), k] features of the per-item hash family (dh_embedding.py:155-196).  Per item: reseed the GLOBAL
        torch generator with item + prefix (as the reference does — the global stream is left where it leaves it), draw k
        slopes, k non-zero offsets, k prime picks; feature = ((a*(item+1)+b) mod p mod m) / (m-1) * 2 - 1 with torch's
        floor-mod on int64 and float32 arithmetic."""
        k, m, primes = self._inp_size, self.m, self._primes
        rows = []
        for item in items:
            item = int(item)
            torch.manual_seed(item + self._prefix)
            a = torch.randint(NEGATIVE_LARGE_INT, LARGE_INT, (k,))
            b = self._random_nonzero_int(k)
            p = primes[torch.randint(0, len(primes), (k,))]
            h = (a * (item + 1) + b) % p % m
            rows.append((h / (m - 1)) * 2 - 1)
        if not rows:
            return torch.empty((0, k), dtype=torch.float32)
        return torch.stack(rows)

    def _cache_on(self, device) -> torch.Tensor:
        if self._cache.device != device:
            self._cache = self._cache.to(device)       # moved once, kept (the reference re-copies every forward)
        return self._cache

    def _random_nonzero_int(self, num_element, rng=None):
        b = torch.randint(NEGATIVE_LARGE_INT, LARGE_INT, (num_element,), generator=rng)
        mask = b == 0
        while mask.sum() > 0:
            num_values = mask.sum().item()
            b[mask] = torch.randint(NEGATIVE_LARGE_INT, LARGE_INT, (num_values,), generator=rng)
            mask = b == 0
        return b

    def _get_universal_hash_batch(self, item: torch.Tensor) -> torch.Tensor:
        """[n] ids -> [n, k] features (dh_embedding.py:213-236), computed by the HIP kernel."""
        return _kernels.dhe_hash(item, self._slopes, self._bias, self._primes_choices, self._prefix, self.m)

    def get_weight(self):
        arr = torch.arange(self._num_item, device=self._seq[0].weight.device)
        return self(arr)

    def forward(self, inp: torch.Tensor):
        mode = self._mode
        if not self.training and self._emb is not None:
            return _kernels.gather_rows(inp, self._emb)

        if not self._use_universal_hash:
            device = self._seq[0].weight.device
            if self._use_cache:
                cache = self._cache_on(device)
                if self.compute_v2:
                    uniques, inverse_idx = inp.unique(return_inverse=True)
                    x = self._forward_mlp(_kernels.gather_rows(uniques, cache))
                    return _kernels.gather_rows(inverse_idx, x)
                return self._forward_mlp(_kernels.bag_reduce(_kernels.gather_rows(inp, cache), mode))
            if self.training:
                raise NotImplementedError()  # as the reference (dh_embedding.py:325-326)
            uniques, inverse_idx = inp.unique(return_inverse=True)
            feats = self._seeded_hash_rows(uniques.tolist()).to(device)
            return _kernels.gather_rows(inverse_idx, self._seq(feats))

        if (self.compute_v2 and self._use_cache) or not self._use_cache:
            if not self._use_cache and self.training:
                raise NotImplementedError()  # as the reference (dh_embedding.py:325-326)
            uniques, inverse_idx = inp.unique(return_inverse=True)
            x = self._forward_mlp(self._get_universal_hash_batch(uniques))
            return _kernels.gather_rows(inverse_idx, x)

        feats = self._get_universal_hash_batch(inp)
        feats = _kernels.bag_reduce(feats, mode)
        return self._forward_mlp(feats)

    def _forward_mlp(self, embs):
        is_flatten = False
        if len(embs.shape) == 3:
            is_flatten = True
            batch, num_field, dimension = embs.shape
            embs = embs.reshape(batch * num_field, dimension)
        # the MLP on the library's own kernels (mish_mlp.py: MFMA products + csrc/mish_mlp.hip) when the pattern fits;
        # nn.Sequential (library products through PyTorch) otherwise
        from .. import mish_mlp as _mm

        plan = _mm.mish_mlp_plan(self._seq, self._use_bn, embs) if OWN_MLP else None
        outs = _mm.run_mish_mlp(plan, self._use_bn, embs) if plan is not None else self._seq(embs)
        if is_flatten:
            outs = outs.reshape(batch, num_field, -1)
        return outs

    def set_extra_state(self, state):
        self._prefix = state["_prefix"]

    def get_extra_state(self):
        return {"_prefix": self._prefix}
