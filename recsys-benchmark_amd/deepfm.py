"""DeepFM with the gather + FM + first-order path as ONE HIP kernel each way.

Drop-in for src/models/deepfm.py:11-134 of the reference: same constructor,
same `state_dict` keys (`_bias, offsets, embedding.*, fc.weight, linear_layer.*,
_deep_branch.*` — `linear_layer` is unused by forward there too), same
`forward(x int[B,F]) -> logits[B]`, `get_ranks`, `load`.

What runs where:
  x + offsets, embedding gather, FM 2nd order, EmbeddingBag(N,1,sum) + bias
      -> mi_gather_fm_fwd(_ride) forward; backward in the epilogue of the tail's first input-gradient product
         (mi_tail_dgrad_gemm_fm) when the whole forward is one autograd node (_fused_step), else
         mi_gather_fm_bwd_{rows,dense}  (vanilla table), or
         IEmbedding.forward + mi_fm_fwd / mi_fm_bwd        (compressed tables)
  the MLP tail (a real GEMM): the library's own fp32-MFMA products with BatchNorm1d / ReLU / Dropout in
      their operand loads and epilogues (tail.py, csrc/tail.hip) in training, eval() and no-BatchNorm stacks alike; the
      general path (mlp.py: contractions on rocBLAS / hipBLASLt through PyTorch + fused BatchNorm passes) only for stacks
      the fused kernels do not take (odd widths) or with MI_FUSED_TAIL=0.
"""
from typing import Any, Dict, List, Optional, Union, cast

import torch
from torch import nn

from . import _kernels
from .embeddings import IEmbedding, VanillaEmbedding, get_embedding
from .mlp import field_offsets, hidden_stack, run_tail


class DeepFM(nn.Module):
    embedding: IEmbedding

    def __init__(
        self,
        field_dims: List[int],
        num_factor: int,
        hidden_sizes: List[int],
        p_dropout: float = 0.1,
        use_batchnorm=False,
        embedding_config: Optional[Dict] = None,
        empty_embedding=False,
        fc_sparse: bool = False,
    ):
        """Arguments as the reference's DeepFM.  Extra, optional:

        fc_sparse: produce the first-order table's gradient in row (COO) form as
            well.  Off by default because the reference's optimizer split
            (src/models/deepfm.py:163-184) feeds `fc.weight` to dense Adam.
        """
        super().__init__()
        # (registration order and the order of the random draws are the reference's — src/models/deepfm.py:26-76 — so that
        #  a seeded construction gives bit-identical initial weights and state_dict() lists the same keys in the same order)
        if not empty_embedding:
            self.embedding = get_embedding(embedding_config or {"name": "vanilla"}, field_dims, num_factor, mode=None,
                                           field_name="deepfm")
        self.fc = nn.EmbeddingBag(sum(field_dims), 1, mode="sum", sparse=fc_sparse)
        self.linear_layer = nn.Linear(1, 1)          # (unused by forward there too; kept for the checkpoint keys)
        self._bias = nn.Parameter(torch.zeros(1))
        stack, top = hidden_stack(num_factor * len(field_dims), hidden_sizes, p_dropout, use_batchnorm)
        self._deep_branch = nn.Sequential(*stack, nn.Linear(top, 1))
        self.register_buffer("offsets", field_offsets(field_dims))

    PACKED_ROW_FLOATS = 32      # one 128-byte line per row: D <= 16 embedding floats, the first-order weight, padding

    def pack_tables(self) -> "DeepFM":
        """Opt-in storage layout for the two lookup tables (an MI355X-side choice; the arithmetic and the reference's
        tensors are unchanged): ONE buffer fp32[N, 32] whose row n holds embedding row n in floats 0..D-1 and
        fc.weight[n] in float D, so a lookup reads one 128-byte line instead of a 64-byte embedding row plus a whole
        64-byte sector for its 4-byte first-order weight (src/models/deepfm.py:47-51 declares them as two tensors).
        `embedding._emb_module.weight` and `fc.weight` stay the SAME Parameter objects (optimizers, `state_dict()` keys,
        shapes [N, D] / [N, 1] and `load_state_dict` are unaffected); their storage becomes a column-slice view of
        the packed buffer.  Call it after the model is on its device (`.to()` copies a view out into a tensor of its
        own).  Needs the plain full table (VanillaEmbedding, mode None) with D in {4, 8, 16}."""
        emb = getattr(self, "embedding", None)
        if type(emb) is not VanillaEmbedding or emb._mode is not None:
            raise NotImplementedError("pack_tables() covers the full (vanilla) embedding table only")
        W, w1 = emb._emb_module.weight, self.fc.weight
        N, D = W.shape
        LD = self.PACKED_ROW_FLOATS
        if D not in (4, 8, 16) or w1.shape != (N, 1) or W.dtype != torch.float32 or W.device != w1.device:
            raise NotImplementedError(f"pack_tables(): unsupported shapes {tuple(W.shape)} / {tuple(w1.shape)}")
        if self.tables_packed:
            return self
        packed = torch.zeros((N, LD), dtype=torch.float32, device=W.device)
        packed[:, :D].copy_(W.detach())
        packed[:, D:D + 1].copy_(w1.detach())
        W.data = packed[:, :D]
        w1.data = packed[:, D:D + 1]
        if not getattr(self, "_packed_hook", False):
            # checkpoints keep the reference's format: contiguous tensors, not views that drag the packed buffer along —
            # whichever module's state_dict() is asked (the model's, `model.embedding`'s as factory.save_ctr_checkpoint
            # does, or `fc`'s)
            def _contiguous_tables(module, state_dict, prefix, local_metadata):
                for key, t in list(state_dict.items()):
                    if key.startswith(prefix) and torch.is_tensor(t) and not t.is_contiguous():
                        state_dict[key] = t.detach().contiguous()
            for mod in (self, emb, emb._emb_module, self.fc):
                mod._register_state_dict_hook(_contiguous_tables)
            self._packed_hook = True
        return self

    @property
    def tables_packed(self) -> bool:
        W, w1 = self.embedding._emb_module.weight, self.fc.weight
        return bool(W.stride(0) == self.PACKED_ROW_FLOATS and w1.stride(0) == self.PACKED_ROW_FLOATS
                    and w1.data_ptr() == W.data_ptr() + 4 * W.shape[1]
                    and W.untyped_storage().data_ptr() == w1.untyped_storage().data_ptr())

    def _fm_and_embedding(self, x):
        emb_mod = self.embedding
        if type(emb_mod) is VanillaEmbedding and emb_mod._mode is None:      # subclasses (QAT) transform the rows
            return _kernels.gather_fm(
                x, self.offsets, emb_mod.get_weight(), self.fc.weight, self._bias,
                sparse_W=emb_mod.sparse_grad, sparse_w1=bool(self.fc.sparse),
            )
        rows = x + self.offsets
        _kernels.note_field_layout(rows, self.offsets, self.fc.weight.shape[0])   # lets the sparse optimizer sort by field
        emb = emb_mod(rows)
        return _kernels.fm_first_order(emb, rows, self.fc.weight, self._bias,
                                       sparse_w1=bool(self.fc.sparse))

    def _fused_step(self, x, labels=None):
        """The whole forward as one autograd node (tail.DeepFMFusedFn) when the pieces fit: the full table with row-form
        gradients (or deterministic mode, which builds its dense gradients from row-form values too) in front of a tail the
        fused kernels take.  The lookup's backward then runs in the epilogue of the tail's first input-gradient product.
        None: the two-node path below."""
        from . import mlp as _mlp, tail as _tail

        emb_mod = getattr(self, "embedding", None)
        if not (_mlp.FUSED_TAIL and _tail.FM_EPILOGUE and x.is_cuda and x.dim() == 2
                and type(emb_mod) is VanillaEmbedding and emb_mod._mode is None):
            return None
        W, w1 = emb_mod.get_weight(), self.fc.weight
        sparse_W, sparse_w1 = bool(emb_mod.sparse_grad), bool(self.fc.sparse)
        if (not _kernels._float4_rows(W.shape[1]) or W.dtype != torch.float32 or W.device != x.device
                or x.shape[1] != self.offsets.numel()):
            return None
        wants = (W.requires_grad or w1.requires_grad) and torch.is_grad_enabled()
        if wants and not ((sparse_W and W.requires_grad) or (sparse_w1 and w1.requires_grad) or _kernels.DETERMINISTIC):
            return None              # dense weight.grad by float atomics: the scatter kernel of the two-node path
        # (no gradient wanted — an inference forward, validation under no_grad: the same node, forward only; the lookup
        #  launch then also computes the constants of the tail's fixed-statistics layers)
        groups = _mlp._groups(self._deep_branch)
        plan = _tail.fused_tail_plan(self._deep_branch, _tail._InputSpec(x.shape[0], x.shape[1] * W.shape[1], W.device), groups)
        if plan is None:
            return None
        return _tail.run_fused_deepfm(plan, groups[-1][1], _mlp._seed_word(W.device), x, self.offsets, W, w1, self._bias,
                                      sparse_W, sparse_w1, labels if self.training else None)

    @staticmethod
    def prefetch_next(x_next) -> None:
        """Tell the next training step (the fused one-node form) which ids the batch AFTER it will look up: that step then
        pulls those table rows into the Infinity Cache on a side stream under its weight-gradient launch.  Optional; a
        DataLoader-driven loop calls it with the batch it already holds for the following iteration."""
        from . import tail as _tail

        _tail.set_next_batch(x_next)

    def forward(self, x, labels=None):
        """x: integer tensor [B, F] of per-field ids -> logits [B] (before sigmoid).
        labels (optional extension, [B]): the step's targets, when the caller is about to evaluate BCEWithLogitsLoss on the
        result (the reference trainer does, src/trainer/deepfm.py:51-52) — the fused training step then evaluates the
        criterion and the head's backward inside its head launch, and recsys_benchmark_amd.BCEWithLogitsLoss(logits, labels)
        finds them.  The logits are the same with or without; anything else ignores the argument."""
        fused = self._fused_step(x, labels)
        if fused is not None:
            return fused.squeeze(-1)
        emb, y_fm = self._fm_and_embedding(x)
        b = emb.shape[0]
        scores = run_tail(self._deep_branch, emb.reshape(b, -1), last_add=y_fm)   # y_fm + deep(emb)
        return scores.squeeze(-1)

    def get_ranks(self, x) -> torch.Tensor:
        scores = self(x)
        return torch.argsort(scores, descending=True)

    @classmethod
    def load(
        cls,
        checkpoint: Union[str, Dict[str, Any]],
        strict=True,
        *,
        empty_embedding=False,
    ) -> "DeepFM":
        if isinstance(checkpoint, str):
            checkpoint = torch.load(checkpoint, map_location="cpu")
        checkpoint = cast(Dict[str, Any], checkpoint)
        model_config = checkpoint["model_config"]
        field_dims = checkpoint["field_dims"]
        model = cls(field_dims, **model_config, empty_embedding=empty_embedding)
        model.load_state_dict(checkpoint["state_dict"], strict=strict)
        return model
