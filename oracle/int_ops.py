"""ORACLE — ctypes front end of oracle/int_ops.c (test infrastructure, not product code)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle_int.so")
_lib = None
_P = ctypes.c_void_p
_I = ctypes.c_int64


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.check_call(["make", "-C", _HERE])
        _lib = ctypes.CDLL(_SO)
    return _lib


def _i64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int64))


def dhe_hash(ids, slopes, bias, primes, prefix, m=1000000):
    ids, slopes, bias, primes = _i64(ids), _i64(slopes), _i64(bias), _i64(primes)
    n, k = ids.size, slopes.size
    h = np.empty((n, k), dtype=np.int64)
    f = np.empty((n, k), dtype=np.float32)
    lib().oracle_dhe_hash(_P(ids.ctypes.data), _I(n), _P(slopes.ctypes.data), _P(bias.ctypes.data),
                          _P(primes.ctypes.data), _I(k), _I(int(prefix)), _I(int(m)), _P(h.ctypes.data),
                          _P(f.ctypes.data))
    return h, f


def csr_rows(values, crow, col, ids, hidden):
    values = np.ascontiguousarray(np.asarray(values, dtype=np.float32))
    crow, col = _i64(crow), _i64(col)
    shape = np.asarray(ids).shape
    ids = _i64(ids).ravel()
    out = np.empty((ids.size, hidden), dtype=np.float32)
    lib().oracle_csr_rows(_P(values.ctypes.data), _P(crow.ctypes.data), _P(col.ctypes.data), _P(ids.ctypes.data),
                          _P(out.ctypes.data), _I(ids.size), _I(hidden))
    return out.reshape(*shape, hidden)


def qr_split(idx, divider):
    idx = _i64(idx).ravel()
    rem, quo = np.empty_like(idx), np.empty_like(idx)
    lib().oracle_qr_split(_P(idx.ctypes.data), _I(idx.size), _I(divider), _P(rem.ctypes.data), _P(quo.ctypes.data))
    return rem, quo


def cerp_split(idx, q_entity_per_row, bucket):
    idx = _i64(idx).ravel()
    q, p = np.empty_like(idx), np.empty_like(idx)
    lib().oracle_cerp_split(_P(idx.ctypes.data), _I(idx.size), _I(q_entity_per_row), _I(bucket), _P(q.ctypes.data),
                            _P(p.ctypes.data))
    return q, p


def tt_split(idx, p_shapes):
    idx, p_shapes = _i64(idx).ravel(), _i64(p_shapes)
    out = np.empty((p_shapes.size, idx.size), dtype=np.int64)
    lib().oracle_tt_split(_P(idx.ctypes.data), _I(idx.size), _P(p_shapes.ctypes.data), _I(p_shapes.size),
                          _P(out.ctypes.data))
    return out
