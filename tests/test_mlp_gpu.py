"""GPU: the fused BatchNorm1d+ReLU+Dropout passes of the MLP tail against stock PyTorch modules on
CPU (fp32; tolerance 2e-5 relative for activations, 1e-4 for gradients — reduction order only)."""
import pytest
import torch
from torch import nn

from conftest import assert_close

from recsys_benchmark_amd.mlp import run_tail

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _seq(inp, hidden, bn, p):
    layers = []
    for h in hidden:
        layers.append(nn.Linear(inp, h))
        if bn:
            layers.append(nn.BatchNorm1d(h))
        layers.append(nn.ReLU())
        layers.append(nn.Dropout(p))
        inp = h
    layers.append(nn.Linear(inp, 1))
    return nn.Sequential(*layers)


@pytest.mark.parametrize("M,inp,hidden", [(4096, 416, [400, 400, 400]), (37, 12, [9, 5]), (2, 8, [4])])
@pytest.mark.parametrize("bn", [True, False])
@pytest.mark.parametrize("training", [True, False])
def test_tail_matches_torch_modules(M, inp, hidden, bn, training):
    torch.manual_seed(M + inp)
    ref = _seq(inp, hidden, bn, 0.0)
    if bn:
        for m in ref:
            if isinstance(m, nn.BatchNorm1d):
                m.running_mean.normal_(0, 0.1)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.data.uniform_(0.5, 1.5)
                m.bias.data.normal_(0, 0.1)
    import copy

    mine = copy.deepcopy(ref).to(DEV)
    ref.train(training)
    mine.train(training)
    X = torch.randn(M, inp) * 2 + 0.5
    G = torch.randn(M, 1)
    x1 = X.clone().requires_grad_(True)
    out_ref = ref(x1)
    (out_ref * G).sum().backward()
    x2 = X.to(DEV).requires_grad_(True)
    out = run_tail(mine, x2)
    assert_close(out, out_ref, 5e-5, 5e-5, "output")
    (out * G.to(DEV)).sum().backward()
    scale = float(x1.grad.abs().max()) + 1e-6
    assert_close(x2.grad, x1.grad, 1e-3, 1e-4 * scale, "grad input")
    for (k, p), (_, q) in zip(mine.named_parameters(), ref.named_parameters()):
        s = float(q.grad.abs().max()) + 1e-6
        assert_close(p.grad, q.grad, 2e-3, 2e-4 * s, f"grad {k}")
    for (k, b), (_, c) in zip(mine.named_buffers(), ref.named_buffers()):
        assert_close(b, c, 1e-5, 1e-6, f"buffer {k}")     # running stats + num_batches_tracked


def test_dropout_statistics_and_mask_consistency():
    torch.manual_seed(0)
    seq = _seq(64, [256], True, 0.5).to(DEV).train()
    x = torch.randn(8192, 64, device=DEV, requires_grad=True)
    y1 = run_tail(nn.Sequential(*list(seq)[:4]), x)        # Linear, BN, ReLU, Dropout
    kept = (y1 != 0).float().mean().item()
    relu_on = 0.5                                            # BN output is ~symmetric
    assert abs(kept - relu_on * 0.5) < 0.02, kept
    y2 = run_tail(nn.Sequential(*list(seq)[:4]), x)
    assert not torch.equal(y1 != 0, y2 != 0), "a new dropout stream per pass"
    y1.sum().backward()                                     # backward uses the saved mask (no error, finite)
    assert torch.isfinite(x.grad).all()
    seq.eval()
    y3 = run_tail(nn.Sequential(*list(seq)[:4]), x)
    y4 = run_tail(nn.Sequential(*list(seq)[:4]), x)
    assert torch.equal(y3, y4), "no dropout in eval"
