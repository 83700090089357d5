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



def use_deterministic_algorithms(on: bool = True) -> None:
    """Bit-reproducible DeepFM / DCN training steps: dense table gradients by sorted, ordered accumulation instead of float
    atomics, unsplit K in the library's GEMM, and the MLP tail on the fused kernels of csrc/tail.hip.  Slower than the
    default; the reference's CPU path is deterministic, which is what this mode matches."""
    from . import _kernels, mlp

    _kernels.DETERMINISTIC = bool(on)
    mlp.FUSED_TAIL = bool(on)


__all__ = [
    "DeepFM", "IEmbedding", "VanillaEmbedding", "NAME_TO_CLS", "get_embedding",
    "LightGCN", "SingleLightGCN", "get_ctr_model", "get_graph_model", "load_ctr_model", "load_graph_model",
    "save_cf_emb_checkpoint", "save_ctr_checkpoint", "MI355XLibraryError", "check_index_errors", "use_deterministic_algorithms",
]
