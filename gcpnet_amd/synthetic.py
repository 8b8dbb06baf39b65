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


def nbody_batch(n_graphs: int, n_body: int, seed: int, h_dim: int = 1, chi_dim: int = 3, e_dim: int = 17, xi_dim: int = 1):
    """Collated batch of fully-connected n-body graphs with random stand-in features of the NMS shapes
    (src/datamodules/components/nms_dataset.py:23-61: h [N,1], chi [N,3,3], e [E,17], xi [E,1,3]; block-diagonal edge_index as
    torch_geometric's Batch.from_data_list builds it) and a next-frame position label."""
    g = torch.Generator().manual_seed(seed)
    idx = torch.arange(n_body)
    r, c = torch.meshgrid(idx, idx, indexing="ij")
    keep = r != c
    offs = (torch.arange(n_graphs) * n_body).repeat_interleave(int(keep.sum()))
    ei = torch.stack((r[keep].repeat(n_graphs) + offs, c[keep].repeat(n_graphs) + offs))
    n, e = n_graphs * n_body, ei.shape[1]
    x = torch.randn(n, 3, generator=g) * 2 + 1.5
    return dict(h=torch.randn(n, h_dim, generator=g), chi=torch.randn(n, chi_dim, 3, generator=g), e=torch.randn(e, e_dim, generator=g),
                xi=torch.randn(e, xi_dim, 3, generator=g), x=x, edge_index=ei, batch=torch.arange(n_graphs).repeat_interleave(n_body),
                label=x + 0.3 * torch.randn(n, 3, generator=g))


def radius_batch(n_graphs: int, atoms: int, max_neighbors: int, seed: int):
    """Collated batch of independent radius graphs (r = 4.5, <= max_neighbors; atom3d_dataset.py:110-129 recipe) with the LBA
    feature shapes: integer atom types h [N], chi [N,2,3], e [E,16], xi [E,1,3], one label per graph."""
    g = torch.Generator().manual_seed(seed)
    xs, eis, bidx, off = [], [], [], 0
    for i in range(n_graphs):
        n = atoms + (i % 5) * (atoms // 10)
        x, ei = radius_graph(n, max_neighbors, seed=seed * 1000 + i, expected_in_radius=40.0)
        xs.append(x)
        eis.append(ei + off)
        bidx.append(torch.full((n,), i, dtype=torch.long))
        off += n
    x, ei = torch.cat(xs), torch.cat(eis, dim=1)
    n, e = x.shape[0], ei.shape[1]
    return dict(h=torch.randint(0, 9, (n,), generator=g), chi=torch.randn(n, 2, 3, generator=g), e=torch.randn(e, 16, generator=g),
                xi=torch.randn(e, 1, 3, generator=g), x=x, edge_index=ei, batch=torch.cat(bidx), label=torch.randn(n_graphs, generator=g))


def model_batch(config: str, seed: int = 0):
    """(batch, model_cfg, kind, label) of the model-level BASELINE configurations: c1 / c4 = NMS small (5-body) / small_20body,
    100 graphs per batch, gcp_model_nms.yaml dims; c3 = ATOM3D-LBA, 16 pocket-sized radius graphs, gcp_model_lba.yaml dims."""
    if config in ("c1", "c4"):
        n_body = 5 if config == "c1" else 20
        b = nbody_batch(100, n_body, seed)
        cfg = dict(h_input_dim=1, chi_input_dim=3, e_input_dim=17, xi_input_dim=1, h_hidden_dim=64, chi_hidden_dim=16,
                   e_hidden_dim=32, xi_hidden_dim=4, num_encoder_layers=4, dropout=0.0)
        label = (f"NMS model step(): 100 fully-connected {n_body}-body graphs = {b['h'].shape[0]} nodes / {b['edge_index'].shape[1]} "
                 f"edges per GPU, (64,16) hidden, 4 GCPInteractions layers with position updates, MSE loss, fwd+bwd")
        return b, cfg, "nms", label
    if config == "c3":
        b = radius_batch(16, 400, 32, seed)
        cfg = dict(chi_input_dim=2, e_input_dim=16, xi_input_dim=1, h_hidden_dim=100, chi_hidden_dim=16, e_hidden_dim=32,
                   xi_hidden_dim=4, output_dim=1, output_scale_factor=2, num_encoder_layers=8, dropout=0.0, dense_dropout=0.0)
        label = (f"LBA model step(): 16 radius graphs (r=4.5, K<=32) = {b['h'].shape[0]} nodes / {b['edge_index'].shape[1]} edges per "
                 f"GPU, (100,16) hidden, 8 GCPInteractions layers + invariant projection + graph-mean readout + dense head, MSE loss")
        return b, cfg, "lba", label
    raise ValueError(config)


def reorder_nodes(inputs: Dict[str, torch.Tensor], perm: torch.Tensor) -> Dict[str, torch.Tensor]:
    """The same graph with node k renamed to the position of k in `perm` (new node j = old node perm[j]): per-node tensors are
    permuted, edge_index relabelled and re-sorted by target (per-edge tensors follow).  Used with parallel.spatial_order to give
    the synthetic graphs the id locality real structures have."""
    perm = perm.to(inputs["edge_index"].device)
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(perm.numel(), device=perm.device)
    ei = inv[inputs["edge_index"]]
    order = torch.argsort(ei[1], stable=True)
    n, e = perm.numel(), ei.shape[1]
    out = {}
    for k, v in inputs.items():
        if k == "edge_index":
            out[k] = ei[:, order]
        elif torch.is_tensor(v) and v.shape[:1] == (n,) and k not in ("e", "xi"):
            out[k] = v[perm]
        elif torch.is_tensor(v) and v.shape[:1] == (e,):
            out[k] = v[order]
        else:
            out[k] = v
    return out
