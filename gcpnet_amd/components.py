"""Host-side mirror of the reference's `src/models/components/__init__.py` for the hot path: same names, same
argument meaning, arithmetic dispatched to the HIP kernels (gcpnet_amd.ops).  Citations are file:line in
/root/reference/src/models/components/__init__.py unless stated otherwise.
"""
from __future__ import annotations

from typing import Any, Callable, Optional, Tuple, Union

import torch
from torch import nn

from . import ops
from .ops import GatherPlan, GraphPlan


class ScalarVector(tuple):
    """(scalar [rows, s], vector [rows, V, 3]) pair with the reference's helper surface (:17-94)."""

    def __new__(cls, scalar, vector):
        return super().__new__(cls, (scalar, vector))

    def __getnewargs__(self):
        return self[0], self[1]

    scalar = property(lambda self: self[0])
    vector = property(lambda self: self[1])

    @staticmethod
    def _parts(other):
        return (other[0], other[1]) if isinstance(other, tuple) else (other.scalar, other.vector)

    def __add__(self, other):
        s, v = self._parts(other)
        return ScalarVector(self[0] + s, self[1] + v)

    def __mul__(self, other):
        if isinstance(other, tuple):
            return ScalarVector(self[0] * other[0], self[1] * other[1])
        return ScalarVector(self[0] * other, self[1] * other)

    def concat(self, others, dim=-1):
        """Scalars are joined on the last axis, vectors on the CHANNEL axis (dim %= scalar.ndim, :56-59)."""
        dim %= self[0].dim()
        parts = [self] + [ScalarVector(*o) for o in others]
        return torch.cat([p[0] for p in parts], dim=dim), torch.cat([p[1] for p in parts], dim=dim)

    def flatten(self):
        v = self[1]
        return torch.cat((self[0], v.reshape(v.shape[:-2] + (3 * v.shape[-2],))), dim=-1)

    @staticmethod
    def recover(x, vector_dim):
        cut = x.shape[-1] - 3 * vector_dim
        return ScalarVector(x[..., :cut], x[..., cut:].reshape(x.shape[:-1] + (vector_dim, 3)))

    def vs(self):
        return self[0], self[1]

    def idx(self, idx):
        return ScalarVector(self[0][idx], self[1][idx])

    def repeat(self, n, c=1, y=1):
        return ScalarVector(self[0].repeat(n, c), self[1].repeat(n, y, c))

    def clone(self):
        return ScalarVector(self[0].clone(), self[1].clone())

    def mask(self, node_mask):
        return ScalarVector(self[0] * node_mask[:, None], self[1] * node_mask[:, None, None])

    def __setitem__(self, key, value):
        self[0][key] = value[0]
        self[1][key] = value[1]

    def __repr__(self):
        return f"ScalarVector({self[0]}, {self[1]})"


def get_nonlinearity(nonlinearity: Optional[str] = None, slope: float = 1e-2, return_functional: bool = False) -> Any:
    """Name validation with the reference's error behaviour (src/models/__init__.py:42-57).  The activation itself is
    evaluated inside the HIP kernels; the returned callable exists for API parity (host glue such as the LBA head)."""
    import torch.nn.functional as F
    from functools import partial

    key = nonlinearity if nonlinearity is None else nonlinearity.lower().strip()
    table = {
        "relu": (F.relu, nn.ReLU), "leakyrelu": (partial(F.leaky_relu, negative_slope=slope), partial(nn.LeakyReLU, slope)),
        "selu": (F.selu, nn.SELU), "silu": (F.silu, nn.SiLU), "sigmoid": (torch.sigmoid, nn.Sigmoid),
    }
    if key is None:
        return nn.Identity()
    if key not in table:
        raise NotImplementedError(f"The nonlinearity {nonlinearity} is currently not implemented.")
    fn, mod = table[key]
    return fn if return_functional else mod()


def canonical_act(name: Optional[str]) -> Optional[str]:
    if name is None:
        return None
    key = name.lower().strip()
    if key not in ("relu", "leakyrelu", "selu", "silu", "sigmoid"):
        raise NotImplementedError(f"The nonlinearity {name} is currently not implemented.")
    return key


class VectorDropout(nn.Module):
    """Drops whole 3-vectors (:97-115): one Bernoulli draw per vector, survivors scaled by 1 / (1 - p); HIP kernel
    (ops.dropout, group = 3)."""

    def __init__(self, drop_rate):
        super().__init__()
        self.drop_rate = drop_rate

    def forward(self, x):
        if not self.training or self.drop_rate == 0:
            return x
        from . import ops

        return ops.dropout(x, self.drop_rate, group=3)


class ScalarDropout(nn.Dropout):
    """nn.Dropout with the mask / scale applied by the HIP kernel (ops.dropout); same attribute (`p`) and eval behaviour."""

    def forward(self, x):
        if not self.training or self.p == 0:
            return x
        from . import ops

        return ops.dropout(x, self.p, group=1)


class GCPDropout(nn.Module):
    """:118-135"""

    def __init__(self, drop_rate: float):
        super().__init__()
        self.scalar_dropout = ScalarDropout(drop_rate)
        self.vector_dropout = VectorDropout(drop_rate)

    @property
    def active(self):
        return self.training and self.scalar_dropout.p > 0

    def forward(self, x):
        if isinstance(x, torch.Tensor):
            return x if x.shape[0] == 0 else self.scalar_dropout(x)
        if x[0].shape[0] == 0 or x[1].shape[0] == 0:
            return x
        return ScalarVector(self.scalar_dropout(x[0]), self.vector_dropout(x[1]))


class GCPLayerNorm(nn.Module):
    """:138-167.  `forward(x, residual=None)` normalises `x + residual` in one kernel when a residual is given
    (the add in front of every norm in GCPInteractions.forward, gcpnet.py:1220-1226,1242-1246)."""

    def __init__(self, dims, eps: float = 1e-8):
        super().__init__()
        self.scalar_dims, self.vector_dims = dims
        self.scalar_norm = nn.LayerNorm(self.scalar_dims)
        self.eps = eps
        if eps != 1e-8:
            raise NotImplementedError("GCPLayerNorm: only the reference default eps=1e-8 is built into the kernel")

    def forward(self, x, residual=None):
        is_sv = not isinstance(x, torch.Tensor)
        s = x[0] if is_sv else x
        if s.shape[0] == 0 or (is_sv and x[1].shape[0] == 0):
            return x if residual is None else (ScalarVector(*x) + residual if is_sv else x + residual)
        g, b = self.scalar_norm.weight, self.scalar_norm.bias
        if not self.vector_dims:
            rs = None if residual is None else (residual if isinstance(residual, torch.Tensor) else residual[0])
            out, _ = ops.layernorm(s, None, g, b, s_b=rs)
            return out
        rs, rv = (None, None) if residual is None else (residual[0], residual[1])
        so, vo = ops.layernorm(s, x[1], g, b, s_b=rs, v_b=rv)
        return ScalarVector(so, vo)


def _plan_for_batch(batch_index: torch.Tensor) -> GatherPlan:
    return GatherPlan.get(batch_index)


def edge_mask_of(edge_index: torch.Tensor, node_mask: torch.Tensor) -> torch.Tensor:
    """:230, :296, :347 -- an edge takes part when both of its end points are unmasked."""
    return node_mask[edge_index[0]] & node_mask[edge_index[1]]


def mask_frames(frames: torch.Tensor, edge_index: torch.Tensor, node_mask: Optional[torch.Tensor]) -> torch.Tensor:
    """What `node_mask` does inside scalarize (:295-302) and vectorize (:346-357): masked edges contribute zeros and still count
    in the node-row means, i.e. their frames act as zero frames (whatever the caller stored there: the masked localize writes
    +inf).  The GCP kernels then run unchanged on the result."""
    if node_mask is None:
        return frames
    return torch.where(edge_mask_of(edge_index, node_mask)[:, None, None], frames, torch.zeros_like(frames))


def centralize(batch, key: str, batch_index: torch.Tensor, node_mask: Optional[torch.Tensor] = None):
    """:171-200: per-graph centroid by segmented mean, subtract the gathered centroid.  Masked (:177-193): the centroid of the
    unmasked nodes; masked rows of the centred copy are +inf."""
    x = batch[key]
    if node_mask is not None:
        idx = torch.nonzero(node_mask).squeeze(1)
        xs, bs = x.index_select(0, idx).contiguous(), batch_index.index_select(0, idx)
        plan = GatherPlan(bs, int(batch_index.max()) + 1 if batch_index.numel() else 0)
        centroid = ops.segment_reduce(xs, plan, mean=True)
        centered_sel = ops.axpy(xs, ops.gather_rows(centroid, plan), -1.0)
        return centroid, torch.full_like(x, float("inf")).index_copy(0, idx, centered_sel)
    plan = _plan_for_batch(batch_index)
    centroid = ops.segment_reduce(x, plan, mean=True)
    centered = ops.axpy(x, ops.gather_rows(centroid, plan), -1.0)
    return centroid, centered


def decentralize(batch, key: str, batch_index: torch.Tensor, entities_centroid: torch.Tensor,
                 node_mask: Optional[torch.Tensor] = None):
    """:204-217.  Masked (:211-213, as written there): `batch_index` must list the graphs of the UNMASKED nodes (or no node be
    masked); masked rows of the result are +inf."""
    x = batch[key]
    if node_mask is not None:
        idx = torch.nonzero(node_mask).squeeze(1)
        if batch_index.shape[0] != idx.shape[0]:
            raise RuntimeError(f"decentralize(node_mask=...): {idx.shape[0]} unmasked nodes but {batch_index.shape[0]} graph "
                               "indices (the reference adds entities_centroid[batch_index] to batch[key][node_mask], :212)")
        plan = GatherPlan(batch_index, entities_centroid.shape[0])
        moved = ops.axpy(x.index_select(0, idx).contiguous(), ops.gather_rows(entities_centroid, plan), 1.0)
        return torch.full_like(x, float("inf")).index_copy(0, idx, moved)
    plan = GatherPlan.get(batch_index, entities_centroid.shape[0])
    return ops.axpy(x, ops.gather_rows(entities_centroid, plan), 1.0)


def localize(x: torch.Tensor, edge_index: torch.Tensor, norm_x_diff: bool = True,
             node_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """:221-269.  Frames are constants of the step (no gradient), as in the reference's use.  With a node mask the frames of the
    masked edges are +inf (:232-236, :262-264)."""
    frames = ops.localize(x, GraphPlan.get(edge_index, x.shape[0]), norm_x_diff)
    if node_mask is not None:
        frames = torch.where(edge_mask_of(edge_index, node_mask)[:, None, None], frames, torch.full_like(frames, float("inf")))
    return frames


def is_identity(nonlinearity: Optional[Union[Callable, nn.Module]] = None):
    """:396-397"""
    return nonlinearity is None or isinstance(nonlinearity, nn.Identity)
