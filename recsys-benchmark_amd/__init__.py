"""recsys_benchmark_amd — MI355X-native embedding-lookup + feature-interaction hot path.

Host-side mirror of the reference's plug-in surface (src/models/**) over the
C-ABI HIP library libmi355x_recsys.so (include/mi355x_recsys.h).  See DESIGN.md.
"""
from . import _lib
from ._lib import MI355XLibraryError, check_index_errors
from .deepfm import DeepFM
from .embeddings import IEmbedding, NAME_TO_CLS, VanillaEmbedding, get_embedding

__all__ = [
    "DeepFM", "IEmbedding", "VanillaEmbedding", "NAME_TO_CLS", "get_embedding",
    "MI355XLibraryError", "check_index_errors",
]
