import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import reference_ops as ro
from recsys_benchmark_amd.dcn import DCN_Mix
AVAZU = [241, 8, 8, 3697, 4614, 25, 5481, 329, 31, 381763, 1611748, 6793, 6, 5, 2509, 9, 10, 432, 5, 68, 169, 61]
DEV = "cuda"
torch.manual_seed(11)
B = 4096
model = DCN_Mix(AVAZU, 16, [400, 400, 400], num_layers=3, num_experts=4, rank=64, embedding_config={"name": "qr", "divider": 2}, p_dropout=0.0).to(DEV).train()
gen = torch.Generator().manual_seed(2)
x = torch.stack([torch.randint(0, d, (B,), generator=gen) for d in AVAZU], 1).to(DEV)
y = (torch.rand(B, generator=gen) < 0.2).float().to(DEV)
def oracle_run(dtype):
    p = {k: v.detach().clone().to(dtype) if v.is_floating_point() else v.detach().clone() for k, v in model.state_dict().items()}
    for k, v in p.items():
        if v.is_floating_point() and "running_" not in k: v.requires_grad_(True)
    rows_ = x + p["offsets"]
    e = ro.qr_forward(rows_, p["embedding.emb1.weight"], p["embedding.emb2.weight"], 2, "mult")
    r = ro.dcn_mix_forward(x, p, e, 3, 3, True)
    torch.nn.functional.binary_cross_entropy_with_logits(r, y.to(dtype)).backward()
    return p, r
out = model(x)
torch.nn.functional.binary_cross_entropy_with_logits(out, y).backward()
p64, r64 = oracle_run(torch.float64); p32, r32 = oracle_run(torch.float32)
named = dict(model.named_parameters())
print("logits", float((out.double()-r64).abs().max()/r64.abs().max()), float((r32.double()-r64).abs().max()/r64.abs().max()))
for k, v in p64.items():
    if not (v.is_floating_point() and v.requires_grad): continue
    g = named[k].grad; g = g.to_dense() if g.is_sparse else g
    sc = v.grad.abs().max().clamp_min(1e-30)
    e = (g.double() - v.grad).abs() / sc; e32 = (p32[k].grad.double() - v.grad).abs() / sc
    print(f"{k:28s} max|ref| {float(sc):.3e}  mine max {float(e.max()):.2e} med {float(e.median()):.2e}   stock32 max {float(e32.max()):.2e} med {float(e32.median()):.2e}")
