"""Embedding plug-in registry (reference: src/models/embeddings/__init__.py:18-73).

Same registry keys, same `get_embedding(embedding_config, field_dims, hidden_size,
mode, field_name)` contract (config deep-copied, "name" popped and restored,
`field_name` forwarded to pep*/cerp* classes).  Keys whose class is outside the
hot-path scope (the OptEmbed search classes, the FBTT CUDA extension) raise
NotImplementedError with the reason instead of silently substituting something.
"""
import copy
from typing import Any, Dict, List, Optional, Tuple, Union

from .base import IEmbedding, VanillaEmbedding
from .cerp_embedding import CerpEmbedding, RetrainCerpEmbedding
from .deepfm_opt_embed import OptEmbed as DeepFMOptEmbed
from .dh_embedding import DHEmbedding
from .pep_embedding import PepEmbeeding, RetrainPepEmbedding
from .pruned_embedding import PrunedEmbedding
from .qat_emb import QAT_EmbInt
from .qr_embedding import QRHashingEmbedding
from .tensortrain_embeddings import TTRecTorch

# (class, extra constructor arguments forced by the key, whether the class takes `field_name`)
_REGISTRY = (
    ("vanilla", VanillaEmbedding, {}, False),
    ("qr", QRHashingEmbedding, {}, False),
    ("dhe", DHEmbedding, {}, False),
    ("pep", PepEmbeeding, {}, True),
    ("pep_retrain", RetrainPepEmbedding, {}, True),
    ("cerp", CerpEmbedding, {}, True),
    ("cerp_retrain", RetrainCerpEmbedding, {}, True),
    ("tt_emb_torch", TTRecTorch, {}, False),
    ("qat", QAT_EmbInt, {}, False),
    ("deepfm_optembed", DeepFMOptEmbed, {}, False),
    ("deepfm_optembed_d", DeepFMOptEmbed, {"t_init": None}, False),   # mask E disabled (reference __init__.py:65-67)
)
NAME_TO_CLS: Dict[str, type] = {key: cls for key, cls, _, _ in _REGISTRY}
_FORCED = {key: forced for key, _, forced, _ in _REGISTRY}
_WANTS_FIELD_NAME = {key for key, _, _, named in _REGISTRY if named}

# registry keys of the reference that this build deliberately does not cover
_SEARCH = "OptEmbed search / retraining classes (SURVEY.md §2.1 #7); the supernet lookup is 'deepfm_optembed'"
OUT_OF_SCOPE = {
    **{key: _SEARCH for key in ("optembed_d", "optembed_d_retrain", "optembed", "optembed_retrain",
                                "deepfm_optembed_retrain")},
    "tt_emb": "FBTT-Embedding CUDA extension, not in the reference tree (SURVEY.md §2.3 K3-K12); use 'tt_emb_torch'",
}


def get_embedding(embedding_config: Dict, field_dims: Union[int, List[int]], hidden_size: int,
                  mode: Optional[str] = None, field_name: str = "") -> IEmbedding:
    """Build the embedding a config names.  The caller's dict is left as it was (the reference pops "name" from a
    deep copy and puts it back)."""
    assert mode in [None, "sum", "mean", "max"], "Unsupported mode"
    kwargs = copy.deepcopy(embedding_config)
    name = kwargs.pop("name")
    if name in OUT_OF_SCOPE:
        raise NotImplementedError(f"embedding '{name}' is outside this build's scope: {OUT_OF_SCOPE[name]}")
    if name not in NAME_TO_CLS:
        raise NotImplementedError(f"{name} not found in mapping from name to class")
    kwargs.update(_FORCED[name])
    if name in _WANTS_FIELD_NAME:
        kwargs["field_name"] = field_name
    return NAME_TO_CLS[name](field_dims, hidden_size, mode=mode, **kwargs)


def detect_special(config: Dict[str, Any]) -> Tuple[Optional[str], bool]:
    """Same answers as the reference's detect_special (src/models/embeddings/__init__.py:76-97)."""
    emb_name = config["model"].get("embedding_config", {"name": "vanilla"})["name"]
    for kw in ["pep", "cerp"]:
        if kw in emb_name:
            return kw, "retrain" in emb_name
    if "optembed_d" in emb_name:
        return "optembed_d", "retrain" in emb_name
    if "optembed" in emb_name:
        return "optembed", "retrain" in emb_name
    return None, False


__all__ = ["IEmbedding", "VanillaEmbedding", "QRHashingEmbedding", "CerpEmbedding", "RetrainCerpEmbedding",
           "DHEmbedding", "PrunedEmbedding", "TTRecTorch", "PepEmbeeding", "RetrainPepEmbedding", "NAME_TO_CLS", "get_embedding", "detect_special"]
