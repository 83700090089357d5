"""Debug helper: one fused-tail configuration against float64, with the location of the largest errors."""
import copy
import os
import sys

import torch
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from tail_helpers import tail_keep_scale  # noqa: E402
from recsys_benchmark_amd import _kernels, mlp as _mlp  # noqa: E402
from recsys_benchmark_amd.tail import SALT  # noqa: E402

DEV = torch.device("cuda", 0)
M, K, hidden, p, seed = int(sys.argv[1]), int(sys.argv[2]), [int(v) for v in sys.argv[3].split(",")], float(sys.argv[4]), int(sys.argv[5])
det = len(sys.argv) > 6 and sys.argv[6] == "det"
g = torch.Generator().manual_seed(seed)
torch.manual_seed(seed)
layers, inp = [], K
for h in hidden:
    layers += [nn.Linear(inp, h), nn.BatchNorm1d(h), nn.ReLU(), nn.Dropout(p)]
    inp = h
layers.append(nn.Linear(inp, 1))
seq = nn.Sequential(*layers).train()
x, add, G = torch.randn(M, K, generator=g) * 0.7 + 0.2, torch.randn(M, generator=g), torch.randn(M, 1, generator=g)
sv = 4242
masks = [tail_keep_scale(sv, SALT * (i + 1), M, h, p) for i, h in enumerate(hidden)]
r = copy.deepcopy(seq).double()
x64, a64 = x.double().requires_grad_(True), add.double().requires_grad_(True)
h, li = x64, 0
for m in r:
    if isinstance(m, nn.Dropout):
        h = h * masks[li].double(); li += 1
    else:
        h = m(h)
o64 = h + a64.view(-1, 1)
(o64 * G.double()).sum().backward()
_mlp.FUSED_TAIL = True
_kernels.DETERMINISTIC = det
_mlp._seed_word(DEV).fill_(sv)
fs = copy.deepcopy(seq).to(DEV)
xd, ad = x.to(DEV).requires_grad_(True), add.to(DEV).requires_grad_(True)
out = _mlp.run_tail(fs, xd, last_add=ad)
(out * G.to(DEV)).sum().backward()


def report(name, got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double()
    err = (got - ref).abs()
    scale = float(ref.abs().max())
    bad = err > 1e-4 * scale
    print(f"{name:12s} shape {tuple(ref.shape)} max rel {float(err.max()) / scale:.3e}  bad {int(bad.sum())}", end="")
    if bad.any() and ref.dim() == 2:
        rows = bad.any(1).nonzero().view(-1)
        cols = bad.any(0).nonzero().view(-1)
        print(f"  rows {rows[:8].tolist()}..{rows[-3:].tolist()} ({len(rows)})  cols {cols[:8].tolist()}..{cols[-3:].tolist()} ({len(cols)})", end="")
    print()


report("out", out, o64)
report("dx", xd.grad, x64.grad)
p64 = dict(r.named_parameters())
for name, q in fs.named_parameters():
    if q.grad is not None and p64[name].grad is not None:
        report(name, q.grad, p64[name].grad)
