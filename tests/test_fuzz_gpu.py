"""GPU: tools/fuzz.py's randomised families for 25 FIXED seeds, so that the driver's `-m gpu` run sees them too (the tool
itself is run for thousands of seeds by hand).  Each family stops at its first mismatch with the seed that reproduces it."""
import importlib.util
import os

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fuzz():
    spec = importlib.util.spec_from_file_location("mi_fuzz_tool", os.path.join(ROOT, "tools", "fuzz.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


SEEDS = list(range(25))


@pytest.mark.parametrize("family", ["one", "gemm_case", "lookup_case", "crossnet_case", "tail_case"])
def test_fuzz_family_for_fixed_seeds(fuzz, family):
    """`one`: field sort + sparse / dense Adam + masked InfoNCE + gather+FM; `gemm_case`: the fp32 MFMA GEMM at random small
    shapes; `lookup_case`: dual (QR / CERP) gathers, SpMM, routing; `crossnet_case`: panel / multi-problem GEMMs, fused
    backward head, per-expert kernels (exact on integer-valued data); `tail_case`: the fused MLP tail against float64."""
    fn = getattr(fuzz, family)
    base = {"one": 1000, "gemm_case": 100000, "lookup_case": 500000, "crossnet_case": 700000, "tail_case": 900000}[family]
    for s in SEEDS:
        try:
            fn(base + s)
        except SystemExit as e:           # the tool reports a mismatch by exiting with a message
            pytest.fail(str(e))
