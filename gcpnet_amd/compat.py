"""Zero-edit switch for the reference tree: `install_aliases()` makes the reference's own dotted names --
`src.models.components.gcpnet.{GCP, GCP2, GCP3, GCPMLPDecoder, GCPEmbedding, GCPMessagePassing, GCPInteractions, GCPInteractions2,
get_GCP_with_custom_cfg}` and `src.models.components.{ScalarVector, GCPLayerNorm, GCPDropout, VectorDropout, centralize,
decentralize, localize}` -- resolve to this package, so that neither the Hydra `_target_` strings
(configs/model/gcpnet_nms.yaml:3-6, module_cfg/gcp_module_nms.yaml:1-4) nor the LitModules' import lines
(src/models/gcpnet_nms_module.py:17-18, gcpnet_lba_module.py:19-20) need editing.  Call it once before `hydra.utils.instantiate`
(one line at the top of src/train.py / src/eval.py, or from a `sitecustomize`).

When the reference's modules are importable their other contents (tasks this package does not cover) stay reachable: only the
names above are rebound.  When they are not (this image: no PyG / Lightning), empty stand-in modules carrying just these names
are registered in `sys.modules`."""
from __future__ import annotations

import importlib
import sys
import types

GCPNET_NAMES = ("GCP", "GCP2", "GCP3", "GCPMLPDecoder", "GCPEmbedding", "GCPMessagePassing", "GCPInteractions", "GCPInteractions2", "get_GCP_with_custom_cfg")
COMPONENT_NAMES = ("ScalarVector", "GCPLayerNorm", "GCPDropout", "VectorDropout", "centralize", "decentralize", "localize")


def _module(name: str) -> types.ModuleType:
    try:
        return importlib.import_module(name)
    except Exception:  # the reference (or one of its dependencies) is not importable here: a stand-in with our names only
        mod = sys.modules.get(name)
        if mod is None:
            mod = types.ModuleType(name)
            mod.__path__ = []  # (a package, so that sub-modules can hang below it)
            sys.modules[name] = mod
            parent, _, leaf = name.rpartition(".")
            if parent:
                setattr(_module(parent), leaf, mod)
        return mod


def install_aliases() -> None:
    import gcpnet_amd
    from gcpnet_amd import components, gcpnet

    comp = _module("src.models.components")
    gn = _module("src.models.components.gcpnet")
    for name in GCPNET_NAMES:
        setattr(gn, name, getattr(gcpnet, name))
    for name in COMPONENT_NAMES:
        obj = getattr(components, name, None) or getattr(gcpnet_amd, name)
        setattr(comp, name, obj)
        setattr(gn, name, obj)  # (the reference's gcpnet.py re-exports what it imports from components)
