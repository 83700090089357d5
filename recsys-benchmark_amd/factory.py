"""Model factories and checkpoint helpers — reference: src/models/__init__.py:15-131.

Same names, same config-dict handling ("name" popped and restored, `compile_model` popped for
dcn_mix — there is nothing to torch.compile here: the cross network IS the hand-written kernel
path, so the flag is accepted and ignored and state_dict keys never gain `_orig_mod.`).
"""
import os
from typing import Dict, Type, Union

import torch

from .deepfm import DeepFM
from .lightgcn import IGraphBaseCore, LightGCN, SingleLightGCN


def get_graph_model(num_users: int, num_items: int, model_config: Dict) -> IGraphBaseCore:
    name = model_config.pop("name")
    name_to_cls: Dict[str, Type[IGraphBaseCore]] = {
        "lightgcn": LightGCN,
        "single-lightgcn": SingleLightGCN,
    }
    if name == "hccf":
        model_config["name"] = name
        raise NotImplementedError("hccf is outside this build's scope (SURVEY.md §2.1 #14)")
    assert name in name_to_cls
    model = name_to_cls[name](num_users, num_items, **model_config)
    model_config["name"] = name
    return model


def load_graph_model(checkpoint_path: str, strict=True) -> IGraphBaseCore:
    checkpoint = torch.load(checkpoint_path, map_location="cpu")
    model = get_graph_model(checkpoint["num_users"], checkpoint["num_items"], checkpoint["model_config"])
    model.load_state_dict(checkpoint["state_dict"], strict=strict)
    return model


def save_cf_emb_checkpoint(model: Union[LightGCN, SingleLightGCN], checkpoint_dir: str, name: str = "target"):
    """{checkpoint_dir}/{field_name}/{name}.pth per embedding, as the reference."""
    for field_name, emb in model.get_embs():
        field_dir = os.path.join(checkpoint_dir, field_name)
        os.makedirs(field_dir, exist_ok=True)
        torch.save(emb.state_dict(), os.path.join(field_dir, f"{name}.pth"))


def get_ctr_model(field_dims, model_config: dict):
    name = "deepfm"
    if "name" in model_config:
        name = model_config.pop("name")
    if name == "deepfm":
        return DeepFM(field_dims, **model_config)
    elif name == "dcn_mix":
        from .dcn import DCN_Mix

        compile_model = model_config.pop("compile_model", True)
        model = DCN_Mix(field_dims, **model_config)
        model_config["compile_model"] = compile_model
        return model
    elif name == "dcn":
        from .dcn import DCNv2

        return DCNv2(field_dims, **model_config)
    raise NotImplementedError()


def load_ctr_model(model_config, checkpoint, strict=True, *, empty_embedding=False):
    name = "deepfm"
    if "name" in model_config:
        name = model_config.pop("name")
    if name == "deepfm":
        return DeepFM.load(checkpoint, strict, empty_embedding=empty_embedding)
    elif name == "dcn_mix":
        from .dcn import DCN_Mix

        return DCN_Mix.load(checkpoint, strict, empty_embedding=empty_embedding)
    raise NotImplementedError()


def save_ctr_checkpoint(model, checkpoint_dir: str, name: str = "target"):
    """{checkpoint_dir}/{deepfm|dcn}/{name}.pth holding the embedding's state_dict."""
    emb = model.embedding
    if isinstance(model, DeepFM):
        field_name = "deepfm"
    elif type(model).__name__ == "DCN_Mix":
        field_name = "dcn"
    else:
        raise NotImplementedError(f"Not supported for {model.__class__=}")
    field_dir = os.path.join(checkpoint_dir, field_name)
    os.makedirs(field_dir, exist_ok=True)
    torch.save(emb.state_dict(), os.path.join(field_dir, f"{name}.pth"))
