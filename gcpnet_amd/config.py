"""Config surface of the hot path: the key names under the reference's `configs/model/**` are accepted unchanged.

Hydra / OmegaConf are not required (they are absent from the build image): `AttrDict` gives the attribute access and
the shallow `copy()` the reference relies on (src/models/components/gcpnet.py:867-868,1001-1004), `load_model_config`
composes a `configs/model/*.yaml` file with its `defaults:` list and `${..key}` interpolations, and `instantiate`
resolves `_target_` / `_partial_` entries, mapping the reference's dotted paths onto this package.  When OmegaConf
objects are passed in (a real Hydra run), they are used as they are.
"""
from __future__ import annotations

import functools
import importlib
import os
import re
from typing import Any, Mapping

import yaml


class AttrDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def __copy__(self):
        return AttrDict(self)

    def copy(self):
        return AttrDict(self)


def as_cfg(obj: Any) -> Any:
    """dict -> AttrDict (recursively); OmegaConf / AttrDict objects pass through."""
    if isinstance(obj, AttrDict):
        return obj
    if isinstance(obj, Mapping) and type(obj) is dict:
        return AttrDict({k: as_cfg(v) for k, v in obj.items()})
    if isinstance(obj, list):
        return [as_cfg(v) for v in obj]
    return obj


def to_container(cfg: Any) -> dict:
    """OmegaConf.to_container(cfg, throw_on_missing=True) for either config flavour."""
    if isinstance(cfg, dict):
        return dict(cfg)
    try:
        from omegaconf import OmegaConf  # type: ignore

        return OmegaConf.to_container(cfg, throw_on_missing=True)
    except ImportError:  # pragma: no cover
        return dict(cfg)


# the reference's dotted paths -> this package (drop-in `_target_` mapping)
TARGET_MAP = {
    "src.models.components.gcpnet.GCP2": "gcpnet_amd.gcpnet.GCP2",
    "src.models.components.gcpnet.GCP3": "gcpnet_amd.gcpnet.GCP3",
    "src.models.components.gcpnet.GCPInteractions": "gcpnet_amd.gcpnet.GCPInteractions",
    "src.models.components.gcpnet.GCPInteractions2": "gcpnet_amd.gcpnet.GCPInteractions2",
    "src.models.components.gcpnet.GCPMessagePassing": "gcpnet_amd.gcpnet.GCPMessagePassing",
    "src.models.components.gcpnet.GCPEmbedding": "gcpnet_amd.gcpnet.GCPEmbedding",
    "src.models.gcpnet_nms_module.GCPNetNMSLitModule": "gcpnet_amd.models.GCPNetNMS",
    "src.models.gcpnet_lba_module.GCPNetLBALitModule": "gcpnet_amd.models.GCPNetLBA",
}


def _locate(path: str):
    path = TARGET_MAP.get(path, path)
    mod, _, name = path.rpartition(".")
    return getattr(importlib.import_module(mod), name)


def instantiate(node: Any, **overrides):
    """hydra.utils.instantiate for the subset the model configs use: `_target_`, `_partial_`, nested nodes."""
    if isinstance(node, Mapping) and "_target_" in node:
        kwargs = {k: instantiate(v) for k, v in node.items() if k not in ("_target_", "_partial_")}
        kwargs.update(overrides)
        fn = _locate(node["_target_"])
        return functools.partial(fn, **kwargs) if node.get("_partial_", False) else fn(**kwargs)
    if isinstance(node, Mapping):
        return AttrDict({k: instantiate(v) for k, v in node.items()})
    if isinstance(node, list):
        return [instantiate(v) for v in node]
    return node


_INTERP = re.compile(r"^\$\{(\.+)([A-Za-z0-9_]+)\}$")


def _resolve(node: Any, parents):
    if isinstance(node, dict):
        for k in list(node):
            node[k] = _resolve(node[k], parents + [node])
        return node
    if isinstance(node, list):
        return [_resolve(v, parents + [node]) for v in node]
    if isinstance(node, str):
        m = _INTERP.match(node)
        if m:
            up = len(m.group(1))  # '.' = same node, '..' = parent of the containing list/dict, ...
            scope = parents[-up] if up <= len(parents) else parents[0]
            return scope[m.group(2)]
    return node


def load_yaml_tree(path: str) -> dict:
    """Loads one config file, merging the files named by its `defaults:` list from the sibling group directories."""
    with open(path) as f:
        data = yaml.safe_load(f) or {}
    base = os.path.dirname(path)
    out: dict = {}
    for item in data.pop("defaults", []) or []:
        if isinstance(item, dict):
            for group, name in item.items():
                sub = os.path.join(base, group, name if name.endswith(".yaml") else name + ".yaml")
                out[group] = load_yaml_tree(sub)
    out.update(data)
    return out


def load_model_config(path: str) -> AttrDict:
    """e.g. load_model_config('<configs>/model/gcpnet_nms.yaml') -> AttrDict with layer_class, model_cfg, module_cfg,
    layer_cfg (with mp_cfg) -- the arguments the reference's LitModules receive from Hydra."""
    tree = _resolve(load_yaml_tree(path), [])
    tree.pop("optimizer", None)
    tree.pop("scheduler", None)
    return as_cfg(tree)


def default_module_cfg(**over) -> AttrDict:
    """Values of configs/model/module_cfg/gcp_module_nms.yaml, with this package's GCP2 as `selected_GCP`."""
    from .gcpnet import GCP2

    cfg = AttrDict(
        selected_GCP=functools.partial(GCP2), norm_x_diff=True, scalar_gate=0, vector_gate=True, vector_residual=False,
        vector_frame_residual=False, frame_gate=False, sigma_frame_gate=False, scalar_nonlinearity="relu",
        vector_nonlinearity=None, nonlinearities=["relu", None], bottleneck=4, vector_linear=True, vector_identity=True,
        default_vector_residual=False, default_bottleneck=4, node_positions_weight=1.0, ablate_frame_updates=False,
        ablate_scalars=False, ablate_vectors=False, ablate_x_force_update=True, enable_e3_equivariance=False)
    cfg.update(over)
    if "scalar_nonlinearity" in over or "vector_nonlinearity" in over:
        cfg["nonlinearities"] = [cfg["scalar_nonlinearity"], cfg["vector_nonlinearity"]]
    return cfg


def default_layer_cfg(**over) -> AttrDict:
    """Values of configs/model/layer_cfg/gcp_interaction_layer_nms.yaml + mp_cfg/gcp_mp_nms.yaml."""
    mp = AttrDict(edge_encoder=False, edge_gate=False, num_message_layers=8, message_residual=0,
                  message_ff_multiplier=1, self_message=True, use_residual_message_gcp=True)
    cfg = AttrDict(pre_norm=False, num_feedforward_layers=2, dropout=0.1, nonlinearity_slope=1e-2, mp_cfg=mp)
    for k, v in over.items():
        (mp if k in mp else cfg)[k] = v
    return cfg
