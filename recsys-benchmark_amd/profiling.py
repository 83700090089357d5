"""Per-kernel HIP-event timing of the library's launches (mi_prof_* in the C-ABI)."""
import ctypes
from collections import defaultdict
from typing import Dict, List

from . import _lib


class KernelTimer:
    """with KernelTimer(capacity) as kt: ...launch...; kt.summary() -> {kernel: {count, avg_us, ...}}

    Every launcher of libmi355x_recsys.so brackets its kernel with a hipEvent pair on the
    launch stream while the timer is armed.  Reading the records synchronises.
    """

    def __init__(self, capacity: int = 65536):
        self.capacity = capacity
        self.records: List = []

    def __enter__(self):
        _lib.check(_lib.load().mi_prof_enable(self.capacity), "mi_prof_enable")
        return self

    def __exit__(self, *exc):
        lib = _lib.load()
        n = lib.mi_prof_count()
        name = ctypes.create_string_buffer(64)
        ms = ctypes.c_float()
        self.records = []
        for i in range(n):
            _lib.check(lib.mi_prof_read(i, name, ctypes.byref(ms)), "mi_prof_read")
            self.records.append((name.value.decode(), ms.value * 1e3))
        lib.mi_prof_enable(0)
        return False

    def summary(self) -> Dict[str, Dict[str, float]]:
        by = defaultdict(list)
        for k, us in self.records:
            by[k].append(us)
        out = {}
        for k, v in by.items():
            v = sorted(v)
            out[k] = {"count": len(v), "avg_us": sum(v) / len(v), "min_us": v[0], "med_us": v[len(v) // 2]}
        return out
