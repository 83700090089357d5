import glob
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")

# a stale shared object (sources edited after the last build) must never be what gets tested
import __graft_entry__  # noqa: E402

__graft_entry__.build()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # a gpu test must never silently pass on a box without a GPU
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no ROCm device")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


class Golden(dict):
    """npz fixture with helpers: .t(key) -> torch tensor, .group(prefix) -> {suffix: tensor}."""

    def t(self, key):
        return torch.from_numpy(np.array(self[key]))

    def group(self, prefix):
        return {k[len(prefix):]: self.t(k) for k in self if k.startswith(prefix)}


def load_golden(name: str) -> Golden:
    """A fixture may name another one in `params_from`: that file's `param/*` arrays are the parameters it ran with (its own
    arrays win) — how the C1 eval fixture avoids a second copy of the 2.3 MB base-config parameters."""
    path = os.path.join(GOLDEN_DIR, name + ".npz")
    with np.load(path) as z:
        out = {k: z[k] for k in z.files}
    base = out.pop("params_from", None)
    if base is not None:
        with np.load(os.path.join(GOLDEN_DIR, str(base) + ".npz")) as z:
            for k in z.files:
                if k.startswith("param/"):
                    out.setdefault(k, z[k])
    return Golden(out)


def golden_names(prefix: str):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, prefix + "*.npz")))


@pytest.fixture
def golden():
    return load_golden


def assert_close(actual, expected, rtol, atol, what=""):
    actual = actual.detach().cpu() if isinstance(actual, torch.Tensor) else torch.as_tensor(actual)
    expected = expected.detach().cpu() if isinstance(expected, torch.Tensor) else torch.as_tensor(expected)
    if actual.is_sparse:
        actual = actual.to_dense()
    assert tuple(actual.shape) == tuple(expected.shape), f"{what}: shape {tuple(actual.shape)} vs {tuple(expected.shape)}"
    torch.testing.assert_close(actual, expected.to(actual.dtype), rtol=rtol, atol=atol, msg=lambda m: f"{what}: {m}")


def assert_mostly_close(actual, expected, rtol, atol, max_bad_frac, what=""):
    """For gradients through ReLU: a pre-activation within rounding of 0 can land on either side of
    the kink on CPU vs GPU, flipping that element's gradient; all but `max_bad_frac` must agree."""
    actual = actual.detach().cpu()
    expected = expected.detach().cpu().to(actual.dtype)
    assert tuple(actual.shape) == tuple(expected.shape), f"{what}: shape"
    bad = (actual - expected).abs() > (atol + rtol * expected.abs())
    frac = bad.float().mean().item() if bad.numel() else 0.0
    assert frac <= max_bad_frac, f"{what}: {frac:.2e} of elements differ (allowed {max_bad_frac:.1e}); " \
                                 f"max abs diff {(actual - expected).abs().max().item():.3e}"


EPS32 = 2.0 ** -24


def assert_within_terms(actual, ref64, sum_abs_terms, k, what="", cpu32=None, max_bad_frac=0.0):
    """The claim behind every tolerance of a float32 reduction: a sum of n float32 terms, added in ANY order (float
    atomics included), lies within ~n_levels * eps32 * sum|terms| of the exact value.  `actual` (the HIP result) — and
    `cpu32`, the stock float32 CPU result, when given — must both satisfy  |x - ref64| <= k * eps32 * sum|terms|
    elementwise, ref64 being the float64 evaluation.  k is the allowed depth factor (a few tens covers tree and serial
    orders of up to ~1e4 terms, since rounding errors add like a random walk)."""
    ref64 = ref64.detach().double().cpu()
    bound = k * EPS32 * sum_abs_terms.detach().double().cpu() + 1e-30
    for name, x in (("HIP fp32", actual), ("CPU fp32", cpu32)):
        if x is None:
            continue
        x = x.detach()
        x = (x.to_dense() if x.is_sparse else x).double().cpu()
        assert tuple(x.shape) == tuple(ref64.shape), f"{what}: shape {tuple(x.shape)} vs {tuple(ref64.shape)}"
        r = (x - ref64).abs() / bound
        ratio = r.max().item() if x.numel() else 0.0
        bad = (r > 1.0).double().mean().item() if x.numel() else 0.0
        # max_bad_frac: elements allowed outside the bound for a DISCRETE reason the bound does not model (a pre-activation
        # within rounding of 0 landing on the other side of a ReLU kink moves its whole term)
        assert bad <= max_bad_frac, (f"{what}: {name} is up to {ratio:.2f}x the bound k*eps*sum|terms| (k={k}) away from the "
                                     f"float64 value on {bad:.2e} of the elements (allowed {max_bad_frac:.1e})")
