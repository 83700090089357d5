"""recsys_benchmark_amd — MI355X-native embedding-lookup + feature-interaction hot path.

Host-side mirror of the reference's plug-in surface (src/models/**) over the
C-ABI HIP library libmi355x_recsys.so (include/mi355x_recsys.h).  See DESIGN.md.
"""
from . import _lib
from ._lib import MI355XLibraryError, check_index_errors
from .deepfm import DeepFM
from .embeddings import IEmbedding, NAME_TO_CLS, VanillaEmbedding, get_embedding
from .factory import (get_ctr_model, get_graph_model, load_ctr_model, load_graph_model, save_cf_emb_checkpoint,
                      save_ctr_checkpoint)
from .lightgcn import LightGCN, SingleLightGCN
from .losses import BCEWithLogitsLoss



_fused_tail_before = None


def use_deterministic_algorithms(on: bool = True) -> None:
    """Bit-reproducible DeepFM training steps (and LightGCN propagation, whose row-per-wave SpMM has a fixed summation
    order by construction): dense table gradients by sorted, ordered accumulation instead of float atomics, unsplit K in
    the library's GEMM, and the MLP tail on the fused kernels of csrc/tail.hip (no atomics).  Slower than the default
    (see DESIGN.md); the reference's CPU path is deterministic, which is what this mode matches.

    Scope: the CrossNet heads of DCN_Mix / DCNv2 and the QR / CERP table backward still accumulate column sums and
    duplicate rows with float atomics — under this mode those models raise NotImplementedError in training rather than
    hand out gradients that only LOOK reproducible.

    on=True forces the fused tail (the deterministic one) on; on=False restores whatever mlp.FUSED_TAIL was before the
    switch (the MI_FUSED_TAIL environment setting or an explicit assignment), it does not turn the fused tail off."""
    global _fused_tail_before
    from . import _kernels, mlp

    if on:
        if not _kernels.DETERMINISTIC:
            _fused_tail_before = mlp.FUSED_TAIL
        mlp.FUSED_TAIL = True
    elif _kernels.DETERMINISTIC and _fused_tail_before is not None:
        mlp.FUSED_TAIL = _fused_tail_before
        _fused_tail_before = None
    _kernels.DETERMINISTIC = bool(on)


__all__ = [
    "DeepFM", "IEmbedding", "VanillaEmbedding", "NAME_TO_CLS", "get_embedding",
    "LightGCN", "SingleLightGCN", "get_ctr_model", "get_graph_model", "load_ctr_model", "load_graph_model",
    "save_cf_emb_checkpoint", "save_ctr_checkpoint", "MI355XLibraryError", "check_index_errors", "use_deterministic_algorithms",
    "BCEWithLogitsLoss",
]
