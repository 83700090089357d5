"""TT-Rec embedding in its pure-PyTorch semantics — reference:
src/models/embeddings/tensortrain_embeddings.py:153-270 (`TTRecTorch`, registry key `tt_emb_torch`).

Same constructor, `tt_cores` ParameterList ([1, p_i, r_i*q_i*r_{i+1}] each, state_dict keys
`tt_cores.{i}`), shape suggestion and initialisers.  The lookup (mixed-radix index split, slice
gather, chained contraction) is one HIP kernel (mi_tt_fwd / mi_tt_bwd).  `get_weight()` is that
kernel over arange(N): the reference's own test pins forward(idx) == get_weight()[idx]
(tests/test_emb.py:458-478).  `tt_emb` (the FBTT CUDA extension) is not in the reference tree
and is out of scope.
"""
from typing import List, Optional

import numpy as np
import torch
from torch import nn

from .. import _kernels
from .base import IEmbedding


def suggested_tt_shapes(n: int, d: int = 3, allow_round_up: bool = True) -> List[int]:
    """Factor n (optionally rounded up to a rounder number) into d balanced factors: the shape with
    the highest entropy among all multiset partitions of the prime factors — same procedure and the
    same sympy/scipy calls as the reference (tt_embedding_ops.py:387-447), so the same answer."""
    from itertools import cycle, islice

    from scipy.stats import entropy
    from sympy.ntheory import factorint
    from sympy.utilities.iterables import multiset_partitions

    def interleave(*its):
        pending = len(its)
        nexts = cycle(iter(it).__next__ for it in its)
        while pending:
            try:
                for nxt in nexts:
                    yield nxt()
            except StopIteration:
                pending -= 1
                nexts = cycle(islice(nexts, pending))

    def best_shape(m: int) -> List[int]:
        primes: List[int] = []
        for prime, mult in factorint(m).items():
            primes += [prime] * mult
        if len(primes) < d:
            primes = primes + [1] * (d - len(primes))

        def canon(blocks):
            prods = sorted(np.prod(b) for b in blocks)
            half = len(prods) // 2
            return tuple(interleave(prods[:half], prods[half:]))

        shapes = list(set(canon(part) for part in multiset_partitions(primes, d)))
        return list(shapes[int(np.argmax([entropy(s) for s in shapes]))])

    def roundup(m: int, k: int) -> int:
        return int(np.ceil(m / 10 ** k)) * 10 ** k

    if not allow_round_up:
        return best_shape(n)
    scores = [entropy(best_shape(roundup(n, i))) for i in range(len(str(n)))]
    return best_shape(roundup(n, int(np.argmax(scores))))


def get_num_params(tt_p_shapes, tt_q_shapes, tt_ranks) -> int:
    return sum(tt_p_shapes[i] * tt_q_shapes[i] * tt_ranks[i] * tt_ranks[i + 1] for i in range(len(tt_p_shapes)))


def _init_cores(num_embeddings, embedding_dim, tt_ranks, weight_dist, tt_cores, tt_ndim, tt_p_shapes=None, tt_q_shapes=None):
    """Initialisers of the reference (tt_embedding_ops.py:818-986,989-1036).  They draw from numpy's global generator
    (and, for 'approx-uniform', Python's `random`) in the reference's order, so a seeded construction gives the same
    cores."""
    assert weight_dist in ["uniform", "naive-uniform", "normal", "approx-uniform", "approx-normal"]
    if weight_dist == "uniform":
        stddev = np.sqrt(2.0 / (num_embeddings + embedding_dim))
        var = np.prod(np.array(tt_ranks) ** (-1.0 / (2 * tt_ndim)))
        core_stddev = stddev ** (1.0 / tt_ndim) * var
        for c in tt_cores:
            nn.init.uniform_(c, 0.0, core_stddev)
    elif weight_dist == "naive-uniform":
        for c in tt_cores:
            nn.init.uniform_(c, 0.0, 1 / np.sqrt(num_embeddings))
    elif weight_dist == "normal":
        for c in tt_cores:
            nn.init.normal_(c, 0.0, 1.0 / np.sqrt(num_embeddings))
            c.data *= 1.0 / tt_ranks[0]
    elif weight_dist == "approx-normal":
        # every element is re-drawn until |w| >= 2 (the reference's loop keeps resampling the
        # |w| < 2 entries), then scaled by (3 N)^(-1/6)
        scale = np.power(1 / np.sqrt(3 * num_embeddings), 1 / 3)
        for c in tt_cores:
            W = np.random.normal(0.0, 1.0, size=tuple(c.shape)).astype(np.float32).flatten()
            redo = np.abs(W) < 2
            while redo.sum() > 0:
                W[redo] = np.random.normal(0.0, 1.0, size=(int(redo.sum()),)).astype(np.float32)
                again = np.zeros(len(W), dtype=np.bool_)
                again[redo] = np.abs(W[redo]) < 2
                redo = again
            c.data = torch.tensor(W.reshape(tuple(c.shape)) * scale, dtype=torch.float32, device=c.data.device)
    else:
        _init_approx_uniform(num_embeddings, tt_ranks, tt_cores, tt_ndim, tt_p_shapes, tt_q_shapes)


def _comb(count: int, teeth: int = 15, width: float = 0.7 / 30.0) -> np.ndarray:
    """`count` draws of the reference's "flat saw tooth" density: a tooth centre j/teeth, j uniform on -(teeth-1)..teeth-1,
    plus uniform(-width/2, width/2) (tt_embedding_ops.py:863-879; one randint call, then one rand call)."""
    centre = np.random.randint(-(teeth - 1), teeth, count)
    return centre * (1.0 / teeth) + (-width / 2.0 + width * np.random.rand(count))


def _init_approx_uniform(num_embeddings, tt_ranks, tt_cores, tt_ndim, tt_p_shapes, tt_q_shapes, sigma: float = 0.01):
    """3-core construction whose product is approximately uniform (tt_embedding_ops.py:861-986): core 0 ~ N(1/sqrt(r1),
    sigma) ; core 1 ~ N(1/sqrt(r1), sigma) except that for every (row, column) pair one EVEN outgoing-rank slice is
    N(0, sigma^2 sqrt(r1)) with a single saw-tooth entry (pre-divided by 1/sqrt(r1)); core 2 ~ N(0, sigma) with one
    saw-tooth entry on an ODD incoming rank per (row, column).  All three scaled by N^(-1/6) and stored [1, p_i, r_i q_i r_i+1]
    (rank-major slices transposed to row-major).  Draw order: numpy block, numpy comb, then per pair a Python-`random`
    slice pick, (core 1) r1 numpy normals, a Python-`random` entry pick — the normals are one [pairs, r1] call, which is the
    same stream."""
    import random

    if tt_ndim != 3:
        raise AssertionError("weight_dist='approx-uniform' needs exactly 3 cores (tt_embedding_ops.py:961)")
    amp = 1.0 / (np.sqrt(num_embeddings) ** (1.0 / 3.0))
    dims = [(tt_ranks[i], tt_p_shapes[i], tt_q_shapes[i], tt_ranks[i + 1]) for i in range(3)]

    def stored(block, i):
        block = (block * amp).astype(np.float32)
        return block.transpose(1, 0, 2, 3).reshape(1, tt_p_shapes[i], -1)

    # core 0: around 1/sqrt(r1) so that head x mid sums to ~1
    r0, p0, q0, r1 = dims[0]
    head = (1.0 / np.sqrt(r1) + np.random.randn(r0 * p0 * q0 * r1) * sigma).reshape(dims[0])

    # core 1
    r1_, p1, q1, r2 = dims[1]
    level = 1.0 / np.sqrt(r1_)
    pairs = p1 * q1
    mid = (level + np.random.randn(r1_ * pairs * r2) * sigma).reshape(r1_, pairs, r2)
    teeth = _comb(pairs) / level
    slices, entries = np.empty(pairs, dtype=np.int64), np.empty(pairs, dtype=np.int64)
    for e in range(pairs):              # Python's generator: slice pick, entry pick, per pair, in this order
        slices[e] = random.randrange(0, r2, 2)
        entries[e] = random.randrange(r1_)
    noise = np.random.randn(pairs, r1_) * (sigma * sigma / level)
    for e in range(pairs):
        mid[:, e, slices[e]] = noise[e]
        mid[entries[e], e, slices[e]] = teeth[e]
    mid = mid.reshape(dims[1])

    # core 2
    r2_, p2, q2, r3 = dims[2]
    tail = (np.random.randn(r2_ * p2 * q2 * r3) * sigma).reshape(r2_, -1)
    cols = tail.shape[1]
    teeth = _comb(cols)
    for e in range(cols):
        tail[random.randrange(1, r2_, 2), e] = teeth[e]
    tail = tail.reshape(dims[2])

    for i, block in enumerate((head, mid, tail)):
        tt_cores[i].data = torch.tensor(stored(block, i), dtype=torch.float32, device=tt_cores[i].data.device)


class TTRecTorch(IEmbedding):
    def __init__(self, field_dims, hidden_size: int, tt_ranks: List[int], mode=None,
                 tt_p_shapes: Optional[List[int]] = None, tt_q_shapes: Optional[List[int]] = None,
                 weight_dist: str = "approx-normal", enforce_embedding_dim: bool = False):
        super().__init__()
        if isinstance(field_dims, int):
            field_dims = [field_dims]
        num_embeddings = sum(field_dims)
        embedding_dim = hidden_size
        self._mode = mode

        self.tt_p_shapes: List[int] = (
            suggested_tt_shapes(num_embeddings, len(tt_ranks) + 1) if tt_p_shapes is None else list(tt_p_shapes))
        self.tt_q_shapes: List[int] = (
            suggested_tt_shapes(embedding_dim, len(tt_ranks) + 1, allow_round_up=(not enforce_embedding_dim))
            if tt_q_shapes is None else list(tt_q_shapes))
        assert len(self.tt_p_shapes) >= 2
        assert len(self.tt_p_shapes) <= 4
        assert len(tt_ranks) + 1 == len(self.tt_p_shapes)
        assert len(self.tt_p_shapes) == len(self.tt_q_shapes)
        assert np.prod(np.array(self.tt_p_shapes)) >= num_embeddings
        assert np.prod(np.array(self.tt_q_shapes)) == embedding_dim
        self.tt_ndim = len(tt_ranks) + 1
        self.num_embeddings = num_embeddings
        self.embedding_dim = embedding_dim
        self.tt_ranks = [1] + list(tt_ranks) + [1]
        self.num_tables = 1

        self.tt_cores = nn.ParameterList()
        for i in range(self.tt_ndim):
            self.tt_cores.append(nn.Parameter(torch.empty(
                [self.num_tables, self.tt_p_shapes[i],
                 self.tt_ranks[i] * self.tt_q_shapes[i] * self.tt_ranks[i + 1]], dtype=torch.float32)))
        _init_cores(self.num_embeddings, self.embedding_dim, self.tt_ranks, weight_dist, self.tt_cores, self.tt_ndim,
                    self.tt_p_shapes, self.tt_q_shapes)

    def get_num_params(self):
        return get_num_params(self.tt_p_shapes, self.tt_q_shapes, self.tt_ranks)

    def _lookup(self, flat_idx):
        return _kernels.tt_lookup(flat_idx, list(self.tt_cores), self.num_embeddings,
                                  [int(v) for v in self.tt_p_shapes], [int(v) for v in self.tt_q_shapes],
                                  [int(v) for v in self.tt_ranks])

    def get_weight(self):
        arr = torch.arange(self.num_embeddings, device=self.tt_cores[0].device)
        return self._lookup(arr)

    def forward(self, x):
        out = self._lookup(x.flatten()).reshape(*x.shape, self.embedding_dim)
        return _kernels.bag_reduce(out, self._mode)
