"""Seeded synthetic inputs for the benchmark configurations of BASELINE.json / SURVEY.md section 8d:
radius graphs with the reference's recipe (r = 4.5, at most K neighbours, no self loops, both directions present when
both nodes select each other; src/datamodules/components/atom3d_dataset.py:110-112) and N(0,1) features
(src/models/__init__.py:104-115).  Graph construction is host-side data preparation (scipy KD-tree)."""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch


def radius_graph(n_nodes: int, max_neighbors: int, seed: int = 0, radius: float = 4.5, expected_in_radius: float = 60.0):
    """Returns (x [N,3] float32, edge_index [2,E] int64 sorted by col).  Points are uniform in a cube sized so that a
    ball of `radius` holds ~`expected_in_radius` points, i.e. every node saturates `max_neighbors`; edge (row=j, col=i)
    means j is one of the `max_neighbors` nearest nodes of i within the radius (torch_cluster.radius_graph order)."""
    from scipy.spatial import cKDTree

    rng = np.random.default_rng(seed)
    ball = 4.0 / 3.0 * math.pi * radius ** 3
    side = (n_nodes * ball / expected_in_radius) ** (1.0 / 3.0)
    x = rng.uniform(0.0, side, size=(n_nodes, 3)).astype(np.float32)
    tree = cKDTree(x)
    dist, nbr = tree.query(x, k=max_neighbors + 1, distance_upper_bound=radius)
    dist, nbr = dist[:, 1:], nbr[:, 1:]  # drop self
    ok = np.isfinite(dist)
    col = np.repeat(np.arange(n_nodes), max_neighbors).reshape(n_nodes, max_neighbors)[ok]
    row = nbr[ok]
    edge_index = torch.from_numpy(np.stack((row, col)).astype(np.int64))
    return torch.from_numpy(x), edge_index


def make_inputs(n_nodes: int, max_neighbors: int, node_dims=(128, 16), edge_dims=(32, 4), seed: int = 0) -> Dict[str, torch.Tensor]:
    x, edge_index = radius_graph(n_nodes, max_neighbors, seed)
    g = torch.Generator().manual_seed(seed)
    e = edge_index.shape[1]
    return dict(
        x=x - x.mean(0, keepdim=True), edge_index=edge_index,
        h=torch.randn(n_nodes, node_dims[0], generator=g), chi=torch.randn(n_nodes, node_dims[1], 3, generator=g),
        e=torch.randn(e, edge_dims[0], generator=g), xi=torch.randn(e, edge_dims[1], 3, generator=g),
    )


def gcp_macs(si: int, vi: int, so: int, vo: int, bottleneck: int = 4) -> int:
    """Multiply-accumulates of one GCP2 row:
    3*vi*H + 9*vi + 27 + (si+H+9)*so + 3*H*vo + so*vo with H = vi // bottleneck."""
    H = vi // bottleneck if bottleneck > 1 else max(vi, vo)
    return 3 * vi * H + 9 * vi + 27 + (si + H + 9) * so + 3 * H * vo + so * vo


def layer_flops(n_nodes: int, n_edges: int, node_dims=(128, 16), edge_dims=(32, 4), n_msg: int = 8) -> Dict[str, float]:
    """Algorithmic FLOPs (2 x MAC) of one GCPInteractions layer, forward; fwd+bwd = 3x."""
    s, v = node_dims
    es, ev = edge_dims
    m_e = gcp_macs(2 * s + es, 2 * v + ev, s, v) + (n_msg - 1) * gcp_macs(s, v, s, v)
    m_n = gcp_macs(s, v, 4 * s, 2 * v) + gcp_macs(4 * s, 2 * v, s, v)
    fwd = 2.0 * (n_edges * m_e + n_nodes * m_n)
    return dict(mac_per_edge=m_e, mac_per_node=m_n, fwd=fwd, fwd_bwd=3.0 * fwd)
