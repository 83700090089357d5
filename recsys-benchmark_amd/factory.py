"""Model factories and checkpoint helpers with the reference's names and config-dict contract
(src/models/__init__.py:15-131), written as one registry + one generic constructor.

The contract that callers rely on and that is kept: the config dict is the keyword set of the model class; its "name" key
selects the class (default "deepfm" for CTR models) and is taken out before construction — put back for graph models,
left out for CTR models, exactly as the reference leaves the caller's dict; `compile_model` belongs to the factory, not to
DCN_Mix (there is nothing to torch.compile here: the cross network IS the hand-written kernel path, so the flag is
accepted, restored and ignored, and state_dict keys never gain `_orig_mod.`).
"""
import importlib
import os
from typing import Callable, Dict, NamedTuple, Optional, Tuple

import torch


class _Entry(NamedTuple):
    module: str                      # module of this package that defines the class (imported on first use)
    cls: str
    checkpoint_field: Optional[str]  # directory name save_ctr_checkpoint uses for the embedding, None: not supported
    factory_keys: Tuple[str, ...]    # config keys that belong to the factory: removed for the constructor, then restored
    restore_name: bool               # whether "name" goes back into the caller's dict after construction


_CTR: Dict[str, _Entry] = {
    "deepfm": _Entry(".deepfm", "DeepFM", "deepfm", (), False),
    "dcn_mix": _Entry(".dcn", "DCN_Mix", "dcn", ("compile_model",), False),
    "dcn": _Entry(".dcn", "DCNv2", None, (), False),
}
_GRAPH: Dict[str, _Entry] = {
    "lightgcn": _Entry(".lightgcn", "LightGCN", None, (), True),
    "single-lightgcn": _Entry(".lightgcn", "SingleLightGCN", None, (), True),
}
_OUT_OF_SCOPE = {"hccf": "hccf is outside this build's scope (SURVEY.md §2.1 #14)"}
_LOADABLE = ("deepfm", "dcn_mix")    # the model kinds load_ctr_model knows (the reference's `load` classmethods)


def _resolve(entry: _Entry):
    return getattr(importlib.import_module(entry.module, __package__), entry.cls)


def _build(registry: Dict[str, _Entry], config: dict, default: Optional[str], make: Callable):
    """Take "name" (and the factory's own keys) out of `config`, call make(cls, config), put back what the contract says."""
    name = config.pop("name", default)
    try:
        if name in _OUT_OF_SCOPE:
            raise NotImplementedError(_OUT_OF_SCOPE[name])
        entry = registry.get(name)
        if entry is None:
            raise NotImplementedError(f"unknown model name {name!r}")
        held = {k: config.pop(k) for k in entry.factory_keys if k in config}
        try:
            return make(_resolve(entry), config)
        finally:
            config.update(held)
    finally:
        entry = registry.get(name)
        if name is not None and (name in _OUT_OF_SCOPE or (entry is not None and entry.restore_name)):
            config["name"] = name


def get_graph_model(num_users: int, num_items: int, model_config: Dict):
    return _build(_GRAPH, model_config, None, lambda cls, cfg: cls(num_users, num_items, **cfg))


def get_ctr_model(field_dims, model_config: dict):
    return _build(_CTR, model_config, "deepfm", lambda cls, cfg: cls(field_dims, **cfg))


def load_ctr_model(model_config, checkpoint, strict=True, *, empty_embedding=False):
    loadable = {k: _CTR[k] for k in _LOADABLE}
    return _build(loadable, model_config, "deepfm",
                  lambda cls, cfg: cls.load(checkpoint, strict, empty_embedding=empty_embedding))


def load_graph_model(checkpoint_path: str, strict=True):
    checkpoint = torch.load(checkpoint_path, map_location="cpu")
    model = get_graph_model(checkpoint["num_users"], checkpoint["num_items"], checkpoint["model_config"])
    model.load_state_dict(checkpoint["state_dict"], strict=strict)
    return model


def _save_embedding(emb, checkpoint_dir: str, field_name: str, name: str) -> None:
    field_dir = os.path.join(checkpoint_dir, field_name)
    os.makedirs(field_dir, exist_ok=True)
    # a table may be a column-slice view of a packed buffer (DeepFM.pack_tables()): torch.save would serialise the whole
    # packed storage behind every view — the file keeps the reference's layout, one contiguous tensor per key
    state = {k: (v.detach().contiguous().clone() if torch.is_tensor(v) and not v.is_contiguous() else v)
             for k, v in emb.state_dict().items()}
    torch.save(state, os.path.join(field_dir, f"{name}.pth"))


def save_cf_emb_checkpoint(model, checkpoint_dir: str, name: str = "target"):
    """{checkpoint_dir}/{field_name}/{name}.pth per embedding, as the reference."""
    for field_name, emb in model.get_embs():
        _save_embedding(emb, checkpoint_dir, field_name, name)


def save_ctr_checkpoint(model, checkpoint_dir: str, name: str = "target"):
    """{checkpoint_dir}/{deepfm|dcn}/{name}.pth holding the embedding's state_dict."""
    fields = {e.cls: e.checkpoint_field for e in _CTR.values() if e.checkpoint_field}
    field_name = next((f for cls in type(model).__mro__ if (f := fields.get(cls.__name__))), None)
    if field_name is None:
        raise NotImplementedError(f"Not supported for {model.__class__=}")
    _save_embedding(model.embedding, checkpoint_dir, field_name, name)
