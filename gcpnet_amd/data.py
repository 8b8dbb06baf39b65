"""Input side of the hot path on the GPU: the NMS featuriser
(src/datamodules/components/nms_dataset.py:23-61, helper.py:16-59) and the radius-graph builder (atom3d_dataset.py:110-112
recipe).  HIP kernels do the arithmetic and the neighbour search; index preprocessing (cell ids, one sort, prefix sums) is torch
plumbing, as for the CSR plans."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import torch

from . import _lib
from ._lib import check
from .ops import _p, _req, _stream

Tensor = torch.Tensor


def nms_featurize(x: Tensor, vel: Tensor, edge_attr: Tensor, edge_index: Tensor, batch: Optional[Tensor] = None,
                  d_max: float = 4.5, num_rbf: int = 16) -> Dict[str, Tensor]:
    """h [N,1], chi [N,3,3], e [E, A + num_rbf], xi [E,1,3] of a (collated) batch of n-body graphs, as
    NMSDataset._featurize_as_graph builds them graph by graph (nms_dataset.py:180-206)."""
    lib = _lib.load()
    x, vel, edge_attr = _req(x, "x"), _req(vel, "vel"), _req(edge_attr, "edge_attr")
    n, e_cnt, a = x.shape[0], edge_index.shape[1], edge_attr.shape[1]
    row, col = edge_index[0].to(torch.int32).contiguous(), edge_index[1].to(torch.int32).contiguous()
    f32 = dict(dtype=torch.float32, device=x.device)
    e = torch.empty((e_cnt, a + num_rbf), **f32)
    xi = torch.empty((e_cnt, 1, 3), **f32)
    check(lib.gcpnet_nms_edge_features(e_cnt, _p(x), _p(row), _p(col), _p(edge_attr), a, float(d_max), int(num_rbf), _p(e), _p(xi),
                                       _stream()), "nms_edge_features")
    h, chi = torch.empty((n, 1), **f32), torch.empty((n, 3, 3), **f32)
    b32 = batch.to(torch.int32).contiguous() if batch is not None else None
    check(lib.gcpnet_nms_node_features(n, _p(vel), _p(x), _p(b32), _p(h), _p(chi), _stream()), "nms_node_features")
    return dict(h=h, chi=chi, e=e, xi=xi)


def radius_graph(x: Tensor, r: float = 4.5, max_num_neighbors: int = 32, batch: Optional[Tensor] = None) -> Tensor:
    """edge_index [2, E] (int64; row = neighbour, col = node; sorted by col, neighbours of a node in ascending distance): for
    every node its `max_num_neighbors` nearest other nodes of the same graph within `r`, no self loops -- the edge list
    gcpnet_amd.synthetic.radius_graph builds with scipy's cKDTree, bit for bit.  Cell list with cell edge r."""
    lib = _lib.load()
    x = _req(x, "x")
    n = x.shape[0]
    dev = x.device
    if n == 0:
        return torch.zeros((2, 0), dtype=torch.int64, device=dev)
    if max_num_neighbors > 64:
        raise ValueError("radius_graph: at most 64 neighbours per node")
    b = batch.long() if batch is not None else torch.zeros(n, dtype=torch.long, device=dev)
    n_graphs = int(b.max()) + 1
    lo = torch.full((n_graphs, 3), float("inf"), device=dev).scatter_reduce(0, b.unsqueeze(1).expand(n, 3), x, "amin")
    cell = torch.floor((x - lo[b]) / float(r)).long().clamp_(min=0)
    nx, ny, nz = (int(v) + 1 for v in cell.max(dim=0).values.tolist())
    ncell = nx * ny * nz
    key = b * ncell + (cell[:, 2] * ny + cell[:, 1]) * nx + cell[:, 0]
    key_sorted, order = torch.sort(key, stable=True)
    counts = torch.bincount(key_sorted, minlength=n_graphs * ncell)
    start = torch.zeros(n_graphs * ncell + 1, dtype=torch.int64, device=dev)
    start[1:] = torch.cumsum(counts, 0)
    xs = x[order].contiguous()
    nbr = torch.empty((n, max_num_neighbors), dtype=torch.int32, device=dev)
    cnt = torch.empty((n,), dtype=torch.int32, device=dev)
    # (held in variables until the launch is enqueued: a temporary's memory goes back to the caching allocator at once)
    order32, cell32, start32 = order.to(torch.int32), key_sorted.to(torch.int32), start.to(torch.int32)
    check(lib.gcpnet_radius_graph(n, _p(xs), _p(order32), _p(cell32), _p(start32), nx, ny, nz, float(r), int(max_num_neighbors),
                                  _p(nbr), _p(cnt), _stream()), "radius_graph")
    keep = nbr >= 0
    col = torch.arange(n, device=dev).unsqueeze(1).expand(n, max_num_neighbors)[keep]
    return torch.stack((nbr[keep].long(), col))
