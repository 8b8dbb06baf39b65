"""Stub modules that let the reference's hot-path files import in the authoring container.

Only used by ``tests/golden/gen_fixtures.py`` (authoring-time fixture generation; the GPU box never sees the reference).  Nothing here ships
arithmetic of its own except ``torch_scatter.scatter``, which restates the documented semantics of
pytorch-scatter 2.0.9 (``environment.yaml:209`` of the reference): sum by ``index`` into ``dim_size`` rows,
``mean`` = sum / clamp(count, 1).

The reference itself is never copied: it is imported from ``/root/reference`` where it lies.
"""
import copy as _copy
import sys
import types

import torch

REFERENCE_ROOT = "/root/reference"


def _scatter(src, index, dim=0, out=None, dim_size=None, reduce="sum"):
    assert dim == 0 and out is None
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() else 0
    res = torch.zeros((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    res = res.index_add(0, index, src)
    if reduce in ("sum", "add"):
        return res
    if reduce == "mean":
        cnt = torch.zeros(dim_size, dtype=src.dtype, device=src.device)
        cnt = cnt.index_add(0, index, torch.ones_like(index, dtype=src.dtype))
        cnt = cnt.clamp(min=1)
        return res / cnt.view((-1,) + (1,) * (src.dim() - 1))
    raise NotImplementedError(reduce)


class _Bag:
    """Attribute bag standing in for torch_geometric.data.{Data,Batch}."""

    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)

    def __getitem__(self, k):
        return getattr(self, k)

    def __setitem__(self, k, v):
        setattr(self, k, v)


class DictConfig(dict):
    """Attribute dict with the shallow ``copy()`` behaviour the reference relies on (gcpnet.py:867,1001)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def __copy__(self):
        return DictConfig(self)


class _OmegaConf:
    @staticmethod
    def to_container(cfg, throw_on_missing=False, resolve=True):
        return dict(cfg)


def _identity_decorator(f=None, **kw):
    if f is None:
        return lambda g: g
    return f


class _TensorTypeMeta(type):
    def __getitem__(cls, item):
        return torch.Tensor


class _TensorType(metaclass=_TensorTypeMeta):
    pass


def _subgraph(subset, edge_index, edge_attr=None, relabel_nodes=False, num_nodes=None):
    """torch_geometric.utils.subgraph (PyG 2.x), documented semantics restated for an index-tensor `subset`: keeps the edges whose
    two end points are both in the subset (original order), optionally relabels node ids to their position in the subset."""
    import torch

    n = int(num_nodes) if num_nodes is not None else int(max(int(edge_index.max()) + 1 if edge_index.numel() else 0,
                                                             int(subset.max()) + 1 if subset.numel() else 0))
    node_mask = torch.zeros(n, dtype=torch.bool)
    node_mask[subset] = True
    keep = node_mask[edge_index[0]] & node_mask[edge_index[1]]
    ei = edge_index[:, keep]
    if relabel_nodes:
        relabel = torch.zeros(n, dtype=torch.long)
        relabel[subset] = torch.arange(subset.shape[0])
        ei = relabel[ei]
    return ei, (edge_attr[keep] if edge_attr is not None else None)


RADIUS_GRAPH_SELECT = "refuse"  # "first": torch_cluster's index-order walk for inputs that exceed the cap (fx_lba_features_capped)


def _radius_graph_first(x, r, batch, max_num_neighbors):
    """torch_cluster 1.6.0, `radius_graph(x, r, batch, loop=False, max_num_neighbors)` as its CUDA kernel computes it, restated:
    torch_cluster/radius.py calls `radius(x, x, r, batch, batch, max_num_neighbors + 1)` and removes the self loops afterwards;
    csrc/cuda/radius_cuda.cu's `radius_kernel` visits, for every target, the nodes of its graph in ascending index order, takes a
    node when dist^2 < r^2 (strict; fp32 arithmetic there, float64 here: generators keep every pair away from the boundary) and
    stops at the cap.  edge_index = [source; target], grouped by target, sources ascending."""
    import torch

    n = x.shape[0]
    d2 = torch.cdist(x.double(), x.double()) ** 2
    ok = d2 < float(r) * float(r)
    assert bool(((d2 - float(r) ** 2).abs() > 1e-6).all()), "a pair on the cutoff sphere"
    if batch is not None:
        ok &= batch.view(-1, 1) == batch.view(1, -1)
    rows, cols = [], []
    for i in range(n):
        cand = torch.nonzero(ok[i]).flatten()[: max_num_neighbors + 1]  # ascending ids, self included, cap + 1
        cand = cand[cand != i]
        rows.append(cand)
        cols.append(torch.full_like(cand, i))
    return torch.stack((torch.cat(rows), torch.cat(cols)))


def _radius_graph(x, r, batch=None, loop=False, max_num_neighbors=32, flow="source_to_target", num_workers=1):
    """torch_cluster.radius_graph (1.6.0), documented semantics restated by brute force: edge (j -> i) for every pair of the same
    graph with |x_j - x_i| <= r, j != i (loop=False), at most `max_num_neighbors` per target i; edge_index = [source j; target i],
    grouped by target.  WHICH neighbours survive the cap is implementation-defined in torch_cluster (an unsorted nanoflann search);
    this stub refuses inputs that hit the cap, so fixtures built with it do not depend on that choice."""
    import torch

    assert flow == "source_to_target" and not loop
    if RADIUS_GRAPH_SELECT == "first":
        return _radius_graph_first(x, r, batch, max_num_neighbors)
    n = x.shape[0]
    d = torch.cdist(x.double(), x.double())
    ok = d <= float(r)
    ok.fill_diagonal_(False)
    if batch is not None:
        ok &= batch.view(-1, 1) == batch.view(1, -1)
    assert int(ok.sum(0).max()) <= max_num_neighbors, "the fixture graph must stay below the neighbour cap"
    col, row = torch.nonzero(ok.t(), as_tuple=True)  # grouped by target (col), sources ascending
    return torch.stack((row, col))


def install():
    """Insert the stub modules into ``sys.modules`` and put the reference on ``sys.path``."""
    if "src.models.components.gcpnet" in sys.modules:
        return

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    mod("torch_scatter", scatter=_scatter)
    tg = mod("torch_geometric")
    tg.data = mod("torch_geometric.data", Batch=_Bag, Data=_Bag)
    tg.utils = mod("torch_geometric.utils", subgraph=_subgraph)
    tg.loader = mod("torch_geometric.loader", DataLoader=object)
    mod("torch_cluster", radius_graph=_radius_graph)
    mod("omegaconf", OmegaConf=_OmegaConf, DictConfig=DictConfig)
    mod("torchtyping", TensorType=_TensorType, patch_typeguard=lambda: None)
    mod("typeguard", typechecked=_identity_decorator)
    bp = mod("biopandas")
    bp.pdb = mod("biopandas.pdb", PandasPdb=object)
    a3 = mod("atom3d")
    a3.util = mod("atom3d.util", metrics=None)

    sys.dont_write_bytecode = True  # never write into /root/reference
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import src  # the reference's package (namespace root only)

    import logging

    utils = mod("src.utils", get_pylogger=lambda name=None: logging.getLogger(name or "ref"))
    src.utils = utils


def install_lightning_stubs():
    """Import stubs for the reference's LitModules (pytorch_lightning / torchmetrics are not in the image): `LightningModule` is a
    plain nn.Module whose `save_hyperparameters` records the constructor arguments of the calling frame as `self.hparams`;
    `torchmetrics.MeanMetric / CatMetric` are inert modules.  No arithmetic."""
    import inspect

    install()
    if "pytorch_lightning" in sys.modules:
        return

    class LightningModule(torch.nn.Module):
        def save_hyperparameters(self, *args, logger=True, ignore=None, **kw):
            frame = inspect.currentframe().f_back
            loc = dict(frame.f_locals)
            loc.pop("self", None)
            loc.pop("__class__", None)
            extra = loc.pop("kwargs", {}) or {}
            loc.update(extra)
            for k in ignore or []:
                loc.pop(k, None)
            self.hparams = DictConfig(loc)

    class _Metric(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

        def forward(self, *a, **k):
            return None

    m = types.ModuleType("pytorch_lightning")
    m.LightningModule = LightningModule
    sys.modules["pytorch_lightning"] = m
    tm = types.ModuleType("torchmetrics")
    tm.MeanMetric = tm.CatMetric = _Metric
    sys.modules["torchmetrics"] = tm
    sys.modules["torch_geometric.utils"].unbatch = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError("unbatch"))


def load_reference():
    """Returns (components module, gcpnet module, get_nonlinearity) of the real reference."""
    install()
    import src.models as ref_models
    import src.models.components as ref_components
    import src.models.components.gcpnet as ref_gcpnet

    return ref_components, ref_gcpnet, ref_models


def make_cfg(**overrides):
    """module_cfg of configs/model/module_cfg/gcp_module_nms.yaml (values verbatim, :1-37), with the real
    reference GCP2 as ``selected_GCP`` (what Hydra's ``_partial_`` instantiation produces, train.py:83-86)."""
    import functools

    _, ref_gcpnet, _ = load_reference()
    cfg = DictConfig(
        selected_GCP=functools.partial(ref_gcpnet.GCP2),
        norm_x_diff=True,
        scalar_gate=0,
        vector_gate=True,
        vector_residual=False,
        vector_frame_residual=False,
        frame_gate=False,
        sigma_frame_gate=False,
        scalar_nonlinearity="relu",
        vector_nonlinearity=None,
        nonlinearities=["relu", None],
        bottleneck=4,
        vector_linear=True,
        vector_identity=True,
        default_vector_residual=False,
        default_bottleneck=4,
        node_positions_weight=1.0,
        ablate_frame_updates=False,
        ablate_scalars=False,
        ablate_vectors=False,
        ablate_x_force_update=True,
        enable_e3_equivariance=False,
    )
    cfg.update(overrides)
    if "scalar_nonlinearity" in overrides or "vector_nonlinearity" in overrides:
        cfg["nonlinearities"] = [cfg["scalar_nonlinearity"], cfg["vector_nonlinearity"]]
    return cfg


def make_layer_cfg(**overrides):
    """layer_cfg of gcp_interaction_layer_nms.yaml + mp_cfg of gcp_mp_nms.yaml (verbatim values)."""
    mp = DictConfig(
        edge_encoder=False,
        edge_gate=False,
        num_message_layers=8,
        message_residual=0,
        message_ff_multiplier=1,
        self_message=True,
        use_residual_message_gcp=True,
    )
    cfg = DictConfig(pre_norm=False, num_feedforward_layers=2, dropout=0.1, nonlinearity_slope=1e-2, mp_cfg=mp)
    for k, v in overrides.items():
        if k in mp:
            mp[k] = v
        else:
            cfg[k] = v
    return cfg


Bag = _Bag
deepcopy = _copy.deepcopy
