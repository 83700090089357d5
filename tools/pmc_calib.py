#!/usr/bin/env python3
"""Calibration of FETCH_SIZE for this library's access patterns (run under
`rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv`).  Known byte counts:
  launch 1: gather_rows D=16, n=2^20 UNIQUE random rows of a 2.16 GB table  -> 64 MiB of rows + 8 MiB idx
  launch 2: gather_rows D=16, n=2^20 SEQUENTIAL rows                       -> 64 MiB streamed + 8 MiB idx
  launch 3: gather_rows D=1  (4-B gathers), n=2^20 unique random rows of a 135 MB table -> 4 MiB useful
  launch 4: gather_rows D=64, n=2^18 unique random rows                    -> 64 MiB of 256-B rows + 2 MiB idx
"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recsys_benchmark_amd import _kernels
dev = torch.device("cuda")
N = 33762577
W16 = torch.rand(N, 16, device=dev)
W1 = torch.rand(N, 1, device=dev)
W64 = torch.rand(N // 4, 64, device=dev)
n = 1 << 20
perm = torch.randperm(N, device=dev)[:n]
seq = torch.arange(n, device=dev)
perm64 = torch.randperm(N // 4, device=dev)[: n // 4]
torch.cuda.synchronize()
for _ in range(2):
    _kernels.gather_rows(perm, W16)
    _kernels.gather_rows(seq, W16)
    _kernels.gather_rows(perm, W1)
    _kernels.gather_rows(perm64, W64)
torch.cuda.synchronize()
print("done")
