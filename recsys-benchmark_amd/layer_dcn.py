"""CrossNet heads — reference: src/models/layer_dcn.py:8-140.

`DCNHead`      x_{l+1} = x_l + x_0 * (W_l x_l + b_l)                                  (:118-140)
`DCN_MixHead`  x_{l+1} = x_l + sum_e g_e(x_l) * x_0 * (U_e^T tanh(C_e tanh(V_e^T x_l)) + b_l), g_e = x_l . G_e   (:27-115)

Same constructors, parameter containers (`layers.{l}.weight/bias`; `U.{l}`, `C.{l}`, `V.{l}`,
`biases.{l}`, `gates`) and kaiming-normal / zero initialisation, so state_dicts interchange.
Forward and backward are compositions of the fp32 MFMA GEMM with fused epilogues (mi_gemm_f32)
plus a few memory-bound helpers (mi_cross_bwd_pre, mi_colsum, mi_rowdot, mi_mix_gate_bwd); no
[B,E,d] intermediate is ever materialised (the expert sum is one GEMM with K = E*rank).
The reference leans on torch.compile for this block (src/models/__init__.py:76-84).
"""
from typing import Literal, Optional

import torch
from torch import nn

from . import _kernels, _lib
from ._kernels import gemm


def _new(shape, dev):
    return torch.empty(shape, dtype=torch.float32, device=dev)


def _expert_fused(r: int, d: int) -> bool:
    """mi_mix_expert_fwd/bwd cover ranks 16 / 32 / 64 (the reference's configs use 64) and d % 4 == 0."""
    return _kernels.PANEL_GEMM and r in (16, 32, 64) and d % 4 == 0


def _bwd_head(g, x0, lin, gate, E, b, dlin, dx0, accumulate, db, dgs, M, d, s):
    """dlin = g*x0, dx0 (+)= g*lin, db += sum_m dlin*rs(m), dgs[m] = dlin[m,:].b in one launch (mi_cross_bwd_head);
    shapes it does not cover (d % 4, d > 1024) take the three separate passes."""
    lib = _lib.load()
    ok = d % 4 == 0 and d <= 1024
    if ok:
        _lib.check(lib.mi_cross_bwd_head(g.data_ptr(), x0.data_ptr(), lin.data_ptr(), _lib.ptr(gate), E, _lib.ptr(b),
                                         dlin.data_ptr(), dx0.data_ptr(), int(accumulate), db.data_ptr(), _lib.ptr(dgs), M, d, s),
                   "mi_cross_bwd_head")
        return
    _lib.check(lib.mi_cross_bwd_pre(g.data_ptr(), x0.data_ptr(), lin.data_ptr(), dlin.data_ptr(), dx0.data_ptr(), M * d,
                                    int(accumulate), s), "mi_cross_bwd_pre")
    _lib.check(lib.mi_colsum(dlin.data_ptr(), d, _lib.ptr(gate), E if gate is not None else 0, db.data_ptr(), None, M, d, s),
               "mi_colsum")
    if dgs is not None:
        _lib.check(lib.mi_rowdot(dlin.data_ptr(), d, b.data_ptr(), None, None, dgs.data_ptr(), M, d, s), "mi_rowdot")


class _DCNHeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x0, *wb):
        dev = _lib.require_gpu(x0)
        x0 = _kernels._f32c(x0)
        M, d = x0.shape
        L = len(wb) // 2
        Ws = [_kernels._f32c(wb[2 * l]) for l in range(L)]
        bs = [_kernels._f32c(wb[2 * l + 1]) for l in range(L)]
        xs, lins = [x0], []
        for l in range(L):
            out, lin = _new((M, d), dev), _new((M, d), dev)
            if not _kernels.gemm_panel(xs[-1], d, Ws[l], d, 0, out, d, M, d, d, epi="cross", bias=bs[l], R1=xs[-1], R2=x0,
                                       C2=lin):
                gemm(xs[-1], Ws[l], out, M, d, d, d, d, d, transB=True, epi="cross", bias=bs[l],
                     R1=xs[-1], ldr1=d, R2=x0, ldr2=d, C2=lin, ldc2=d)
            xs.append(out)
            lins.append(lin)
        ctx.save_for_backward(*xs[:-1], *lins, *Ws)
        ctx.L = L
        return xs[-1] if L else x0.clone()

    @staticmethod
    def backward(ctx, g):
        L = ctx.L
        saved = ctx.saved_tensors
        xs, lins, Ws = saved[:L], saved[L:2 * L], saved[2 * L:3 * L]
        g = _kernels._f32c(g)
        if L == 0:
            return (g,)
        dev = g.device
        M, d = g.shape
        lib = _lib.load()
        s = _lib.stream_ptr(dev)
        x0 = xs[0]
        dx0 = _new((M, d), dev)
        grads = [None] * (2 * L)
        # every weight / bias gradient of the head in ONE zero-filled buffer: the L weight-gradient products (each only
        # (d/64)^2 tiles with an M-long reduction) wait until the chain is done and go out as one launch (gemm_multi)
        flat = torch.zeros((L * (d * d + d),), dtype=torch.float32, device=dev)
        dWs, dbs = flat[:L * d * d].view(L, d, d), flat[L * d * d:].view(L, d)
        later = []
        for l in range(L - 1, -1, -1):
            dlin = _new((M, d), dev)
            _bwd_head(g, x0, lins[l], None, 0, None, dlin, dx0, l != L - 1, dbs[l], None, M, d, s)
            later.append(dict(A=dlin, B=xs[l], C=dWs[l], M=d, N=d, K=M, lda=d, ldb=d, ldc=d))   # dW[o,i] = sum_m dlin[m,o] x_l[m,i]
            gn = _new((M, d), dev)
            # dx_l = g + dlin W  (+ dx0 at l=0)
            if not _kernels.gemm_panel(dlin, d, Ws[l], d, 1, gn, d, M, d, d, epi="add", R1=g, R2=dx0 if l == 0 else None):
                gemm(dlin, Ws[l], gn, M, d, d, d, d, d, epi="add", R1=g, ldr1=d, R2=dx0 if l == 0 else None, ldr2=d)
            grads[2 * l], grads[2 * l + 1] = dWs[l], dbs[l]
            g = gn
        _kernels.gemm_multi(later, transA=True)
        return (g, *grads)


class DCNHead(nn.Module):
    def __init__(self, num_layers, hidden_size):
        super().__init__()
        self.num_layers = num_layers
        self.hidden_size = hidden_size
        self.layers = nn.ModuleList([nn.Linear(hidden_size, hidden_size) for _ in range(num_layers)])

    def forward(self, x_0):
        """x_0: [B, hidden] -> x_L: [B, hidden]."""
        wb = []
        for layer in self.layers:
            wb += [layer.weight, layer.bias]
        return _DCNHeadFn.apply(x_0, *wb)


class _DCNMixFn(torch.autograd.Function):
    """args: x0, gates[E,d,1], then per layer U[E,r,d], C[E,r,r], V[E,d,r], b[1,d]."""

    @staticmethod
    def forward(ctx, x0, gates, *p):
        dev = _lib.require_gpu(x0)
        c = _kernels._f32c
        x0, G = c(x0), c(gates)
        M, d = x0.shape
        E = G.shape[0]
        L = len(p) // 4
        Us, Cs, Vs, bs = ([c(p[4 * l + k]) for l in range(L)] for k in range(4))
        r = Cs[0].shape[1] if L else 0
        Er = E * r
        xs, saved = [x0], []
        for l in range(L):
            xl = xs[-1]
            gate = _new((M, E), dev)
            if d % 4 == 0 and E <= 8:                                                          # g_e = x_l . G_e
                _lib.check(_lib.load().mi_rowdot_multi(xl.data_ptr(), d, G.data_ptr(), gate.data_ptr(), M, d, E,
                                                       _lib.stream_ptr(dev)), "mi_rowdot_multi")
            else:
                gemm(xl, G, gate, M, E, d, d, d, E, transB=True)
            H1, H2, H2g = _new((M, Er), dev), _new((M, Er), dev), _new((M, Er), dev)
            if _expert_fused(r, d):          # tanh(x_l V_e), tanh(. C_e), * g_e in one launch
                _lib.check(_lib.load().mi_mix_expert_fwd(xl.data_ptr(), Vs[l].data_ptr(), Cs[l].data_ptr(), gate.data_ptr(),
                                                         H1.data_ptr(), H2.data_ptr(), H2g.data_ptr(), M, d, E, r,
                                                         _lib.stream_ptr(dev)), "mi_mix_expert_fwd")
            else:
                if not _kernels.gemm_panel(xl, d, Vs[l], r, 1, H1, Er, M, Er, d, gw=r, gstride=d * r, epi="tanh"):
                    gemm(xl, Vs[l], H1, M, r, d, d, r, Er, batch=E, sB=d * r, sC=r, epi="tanh")    # tanh(x_l V_e)
                gemm(H1, Cs[l], H2, M, r, r, Er, r, Er, batch=E, sA=r, sB=r * r, sC=r, epi="tanh_gate",
                     rowscale=gate, nrs=E, C2=H2g, ldc2=Er, sC2=r)                                 # tanh(. C_e), * g_e
            out, T = _new((M, d), dev), _new((M, d), dev)
            if not _kernels.gemm_panel(H2g, Er, Us[l], d, 1, out, d, M, d, Er, epi="cross", bias=bs[l], rowscale=gate,
                                       nrs=E, R1=xl, R2=x0, C2=T):
                gemm(H2g, Us[l], out, M, d, Er, Er, d, d, epi="cross", bias=bs[l], rowscale=gate, nrs=E,
                     R1=xl, ldr1=d, R2=x0, ldr2=d, C2=T, ldc2=d)                               # sum over experts: K = E*r
            xs.append(out)
            saved += [gate, H1, H2, H2g, T]
        ctx.save_for_backward(G, *xs[:-1], *saved, *Us, *Cs, *Vs, *bs)
        ctx.dims = (L, E, r)
        return xs[-1] if L else x0.clone()

    @staticmethod
    def backward(ctx, g):
        L, E, r = ctx.dims
        g = _kernels._f32c(g)
        if L == 0:
            return g, None
        t = ctx.saved_tensors
        G, xs, sv = t[0], t[1:1 + L], t[1 + L:1 + 6 * L]
        Us, Cs, Vs, bs = (t[1 + 6 * L + k * L: 1 + 6 * L + (k + 1) * L] for k in range(4))
        dev = g.device
        M, d = g.shape
        Er = E * r
        lib = _lib.load()
        s = _lib.stream_ptr(dev)
        x0 = xs[0]
        dx0 = _new((M, d), dev)
        grads = [None] * (4 * L)
        # All parameter gradients of the head live in ONE zero-filled buffer, and the 4L weight-gradient products (dU, dC,
        # dV and the gate's share per layer: a few 64x64 tiles each, reduction length M) are collected while the chain
        # runs and go out as ONE launch at the end (gemm_multi) instead of 4L launches + their zero fills.
        per = Er * d + E * r * r + E * d * r + d + E * d
        flat = torch.zeros((L * per,), dtype=torch.float32, device=dev)

        def part(l, off, shape):
            n = 1
            for k in shape:
                n *= k
            return flat[l * per + off: l * per + off + n].view(shape)

        later = []
        # the layers' shares of dG (one gate matrix serves every layer): all added into layer 0's slot by the split-K atomics
        # of the one launch below — or, in deterministic mode, kept apart and summed in a fixed order afterwards
        shared_dG = not _kernels.DETERMINISTIC
        dGs = [part(0 if shared_dG else l, Er * d + E * r * r + E * d * r + d, (E, d)) for l in range(L)]
        for l in range(L - 1, -1, -1):
            gate, H1, H2, H2g, T = sv[5 * l: 5 * l + 5]
            xl = xs[l]
            dT = _new((M, d), dev)
            dU, dC, dV = part(l, 0, (E, r, d)), part(l, Er * d, (E, r, r)), part(l, Er * d + E * r * r, (E, d, r))
            db = part(l, Er * d + E * r * r + E * d * r, (1, d))
            dgsum = _new((M,), dev)
            _bwd_head(g, x0, T, gate, E, bs[l], dT, dx0, l != L - 1, db, dgsum, M, d, s)
            later.append(dict(A=H2g, B=dT, C=dU, M=Er, N=d, K=M, lda=Er, ldb=d, ldc=d))          # dU = H2g^T dT
            dgate, dZ2, dZ1 = _new((M, E), dev), _new((M, Er), dev), _new((M, Er), dev)
            if _expert_fused(r, d):          # dT U^T, the gate / tanh backward and (dZ2_e C_e^T) * tanh' in one launch
                _lib.check(lib.mi_mix_expert_bwd(dT.data_ptr(), Us[l].data_ptr(), Cs[l].data_ptr(), gate.data_ptr(),
                                                 H1.data_ptr(), H2.data_ptr(), dgsum.data_ptr(), dgate.data_ptr(),
                                                 dZ2.data_ptr(), dZ1.data_ptr(), M, d, E, r, s), "mi_mix_expert_bwd")
            else:
                dH2g = _new((M, Er), dev)
                if not _kernels.gemm_panel(dT, d, Us[l], d, 0, dH2g, Er, M, Er, d):            # dT U^T
                    gemm(dT, Us[l], dH2g, M, Er, d, d, d, Er, transB=True)
                _lib.check(lib.mi_mix_gate_bwd(dH2g.data_ptr(), H2.data_ptr(), gate.data_ptr(), dgsum.data_ptr(),
                                               dgate.data_ptr(), dZ2.data_ptr(), M, E, r, s), "mi_mix_gate_bwd")
                gemm(dZ2, Cs[l], dZ1, M, r, r, Er, r, Er, transB=True, batch=E, sA=r, sB=r * r, sC=r,
                     epi="mul_dtanh", R1=H1, ldr1=Er, sR1=r)                                   # (dZ2_e C_e^T) * tanh'
            later.append(dict(A=H1, B=dZ2, C=dC, M=r, N=r, K=M, lda=Er, ldb=Er, ldc=r, batch=E, sA=r, sB=r,
                              sC=r * r))                                                       # dC_e = H1_e^T dZ2_e
            later.append(dict(A=xl, B=dZ1, C=dV, M=d, N=r, K=M, lda=d, ldb=Er, ldc=r, batch=E, sB=r,
                              sC=d * r))                                                       # dV_e = x_l^T dZ1_e
            gn = _new((M, d), dev)
            # g + sum_e dZ1_e V_e^T (+ dx0 at the first layer) + dgate G (a rank-E term of the epilogue)
            if not _kernels.gemm_panel(dZ1, Er, Vs[l], r, 0, gn, d, M, d, Er, gw=r, gstride=d * r, epi="add", R1=g,
                                       R2=dx0 if l == 0 else None, rowscale=dgate, nrs=E, bias=G):
                gemm(dZ1, Vs[l], gn, M, d, r, Er, r, d, transB=True, kgroups=E, gA=r, gB=d * r,
                     epi="add", R1=g, ldr1=d, R2=dx0 if l == 0 else None, ldr2=d, rowscale=dgate, nrs=E, bias=G)
            later.append(dict(A=dgate, B=xl, C=dGs[l], M=E, N=d, K=M, lda=E, ldb=d, ldc=d,         # layer l's share of dG
                              **({"splitk": 8} if shared_dG else {})))                           # (>= 2 slices: atomic adds)
            grads[4 * l: 4 * l + 4] = [dU, dC, dV, db]
            g = gn
        _kernels.gemm_multi(later, transA=True)
        # the layers' shares, summed in a fixed order (one strided view of the flat buffer: a single reduction launch)
        if shared_dG:
            dG = dGs[0]
        else:
            dG = torch.as_strided(flat, (L, E, d), (per, d, 1), Er * d + E * r * r + E * d * r + d).sum(0)
        return (g, dG.view(E, d, 1), *grads)


class DCN_MixHead(nn.Module):
    """DeepCross Mixture head (low-rank mixture of experts)."""

    def __init__(self, num_experts: int, num_layers: int, rank: int, hidden_size: int,
                 activation: Optional[str] = None, gate_act: Literal["softmax", "identity"] = "identity"):
        super().__init__()
        self.num_experts = num_experts
        self.num_layers = num_layers
        self.rank = rank
        assert gate_act in ["softmax", "identity"]
        if gate_act == "softmax":
            raise NotImplementedError("softmax gating is never selected by the reference's DCN_Mix "
                                      "(src/models/dcn.py:48-54); only the identity gate is built")
        self.U = nn.ParameterList([self._init_parameters((num_experts, rank, hidden_size)) for _ in range(num_layers)])
        self.C = nn.ParameterList([self._init_parameters((num_experts, rank, rank)) for _ in range(num_layers)])
        self.V = nn.ParameterList([self._init_parameters((num_experts, hidden_size, rank)) for _ in range(num_layers)])
        self.biases = nn.ParameterList([self._init_parameters((1, hidden_size), "zeros") for _ in range(num_layers)])
        self.gates = self._init_parameters((num_experts, hidden_size, 1))
        self.gate_act = nn.Identity()
        self.act_name = "tanh"   # the reference hard-codes tanh whatever `activation` says
        self.act = nn.Tanh()

    def _init_parameters(self, shape, dist="he") -> nn.Parameter:
        if dist == "zeros":
            return nn.Parameter(torch.zeros(*shape))
        tensor = torch.empty(*shape)
        nn.init.kaiming_normal_(tensor)
        return nn.Parameter(tensor)

    def forward(self, x_0):
        p = []
        for l in range(self.num_layers):
            p += [self.U[l], self.C[l], self.V[l], self.biases[l]]
        return _DCNMixFn.apply(x_0, self.gates, *p)
