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

NAME_TO_CLS: Dict[str, type] = {
    "vanilla": VanillaEmbedding,
    "qr": QRHashingEmbedding,
    "dhe": DHEmbedding,
    "pep": PepEmbeeding,
    "pep_retrain": RetrainPepEmbedding,
    "cerp": CerpEmbedding,
    "cerp_retrain": RetrainCerpEmbedding,
    "tt_emb_torch": TTRecTorch,
    "qat": QAT_EmbInt,
    "deepfm_optembed": DeepFMOptEmbed,
    "deepfm_optembed_d": DeepFMOptEmbed,      # dimension mask only (t_init forced to None below)
}

# registry keys of the reference that this build deliberately does not cover
OUT_OF_SCOPE = {
    "optembed_d": "LightGCN OptEmbed search classes (SURVEY.md §2.1 #7); the supernet lookup is 'deepfm_optembed'",
    "optembed_d_retrain": "OptEmbed retraining from a searched mask (SURVEY.md §2.1 #7)",
    "optembed": "LightGCN OptEmbed search classes (SURVEY.md §2.1 #7); the supernet lookup is 'deepfm_optembed'",
    "optembed_retrain": "OptEmbed retraining from a searched mask (SURVEY.md §2.1 #7)",
    "deepfm_optembed_retrain": "OptEmbed retraining from a searched mask (SURVEY.md §2.1 #7)",
    "tt_emb": "FBTT-Embedding CUDA extension, not in the reference tree (SURVEY.md §2.3 K3-K12); "
              "use 'tt_emb_torch'",
}


def get_embedding(
    embedding_config: Dict,
    field_dims: Union[int, List[int]],
    hidden_size: int,
    mode: Optional[str] = None,
    field_name: str = "",
) -> IEmbedding:
    assert mode in [None, "sum", "mean", "max"], "Unsupported mode"
    name = embedding_config["name"]
    embedding_config = copy.deepcopy(embedding_config)
    embedding_config.pop("name")

    if name == "vanilla":
        emb = VanillaEmbedding(field_dims, hidden_size, mode=mode, **embedding_config)
    elif name in OUT_OF_SCOPE:
        raise NotImplementedError(f"embedding '{name}' is outside this build's scope: {OUT_OF_SCOPE[name]}")
    elif name not in NAME_TO_CLS:
        raise NotImplementedError(f"{name} not found in mapping from name to class")
    else:
        if name.startswith("pep") or name.startswith("cerp"):
            embedding_config["field_name"] = field_name
        if name == "deepfm_optembed_d":
            embedding_config["t_init"] = None      # mask E disabled (reference __init__.py:65-67)
        cls = NAME_TO_CLS[name]
        emb = cls(field_dims, hidden_size, mode=mode, **embedding_config)

    embedding_config["name"] = name
    return emb


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
