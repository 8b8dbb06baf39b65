"""Input side of the hot path on the GPU: the NMS featuriser
(src/datamodules/components/nms_dataset.py:23-61, helper.py:16-59), the ATOM3D / LBA featuriser (atom3d_dataset.py:42-149), the
radius-graph builder (atom3d_dataset.py:110-112 recipe) and the collation of per-graph samples into one batch
(torch_geometric `Batch.from_data_list` semantics).  HIP kernels do the arithmetic and the neighbour search; index preprocessing
(cell ids, one sort, prefix sums, concatenation) is torch plumbing, as for the CSR plans."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Mapping, Optional, Sequence

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


def radius_graph(x: Tensor, r: float = 4.5, max_num_neighbors: int = 32, batch: Optional[Tensor] = None, select: str = "nearest") -> Tensor:
    """edge_index [2, E] (int64; row = neighbour, col = node; sorted by col, neighbours of a node in ascending distance): for
    every node its `max_num_neighbors` nearest other nodes of the same graph within `r`, no self loops -- the edge list
    gcpnet_amd.synthetic.radius_graph builds with scipy's cKDTree, bit for bit.  Cell list with cell edge r.

    select="first": torch_cluster 1.6.0's choice when a node has more than `max_num_neighbors` nodes in range (the reference's
    `radius_graph(..., max_num_neighbors=32)`, atom3d_dataset.py:110-112) as its CUDA kernel makes it -- the LOWEST node ids instead
    of the nearest, distance test strict, the self loop removed after the cap was applied (a node whose own id is not among its
    first max_num_neighbors + 1 in-range ids keeps max_num_neighbors + 1 edges: upstream's behaviour, kept); a node's neighbours
    ascending by id.  (torch_cluster's CPU path walks a nanoflann k-d tree unsorted: that order is implementation-defined and not
    restated.)  Below the cap both modes give the same edge SET."""
    lib = _lib.load()
    x = _req(x, "x")
    n = x.shape[0]
    dev = x.device
    if n == 0:
        return torch.zeros((2, 0), dtype=torch.int64, device=dev)
    if select not in ("nearest", "first"):
        raise ValueError("radius_graph: select is 'nearest' or 'first'")
    first = select == "first"
    if max_num_neighbors + int(first) > 64:
        raise ValueError("radius_graph: at most 64 neighbours per node (63 with select='first')")
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
    width = max_num_neighbors + int(first)
    nbr = torch.empty((n, width), dtype=torch.int32, device=dev)
    cnt = torch.empty((n,), dtype=torch.int32, device=dev)
    # (held in variables until the launch is enqueued: a temporary's memory goes back to the caching allocator at once)
    order32, cell32, start32 = order.to(torch.int32), key_sorted.to(torch.int32), start.to(torch.int32)
    fn = lib.gcpnet_radius_graph_first if first else lib.gcpnet_radius_graph
    check(fn(n, _p(xs), _p(order32), _p(cell32), _p(start32), nx, ny, nz, float(r), int(max_num_neighbors), _p(nbr), _p(cnt), _stream()),
          "radius_graph")
    keep = nbr >= 0
    col = torch.arange(n, device=dev).unsqueeze(1).expand(n, width)[keep]
    return torch.stack((nbr[keep].long(), col))


# ---- ATOM3D / LBA (src/datamodules/components/atom3d_dataset.py) -----------------------------------------------------------
ATOM_TYPES: Dict[str, int] = {"H": 0, "C": 1, "N": 2, "O": 3, "F": 4, "S": 5, "Cl": 6, "CL": 6, "P": 7}  # :20-30
NUM_ATOM_TYPES = 9  # the eight elements above + "anything else" = 8 (:36-37)


def element_mapping(elements: Sequence[str]) -> Tensor:
    """`_element_mapping` (atom3d_dataset.py:35-37) over a column of element symbols -> int64 atom types (host-side: strings)."""
    return torch.tensor([ATOM_TYPES.get(e, 8) for e in elements], dtype=torch.long)


def lba_featurize(coords: Tensor, atom_types: Tensor, n_ligand: Optional[int] = None, edge_cutoff: float = 4.5, num_rbf: int = 16,
                  max_num_neighbors: int = 32, batch: Optional[Tensor] = None, edge_index: Optional[Tensor] = None,
                  select: str = "nearest") -> Dict[str, Tensor]:
    """`BaseTransform.__call__` / `LBATransform.__call__` (atom3d_dataset.py:101-149) for one structure -- or, with `batch`, for
    the concatenated atoms of several -- on the GPU: radius graph (r = edge_cutoff, <= max_num_neighbors in-edges per atom, no
    self loops), e = 16 RBFs of the edge length with D_max = edge_cutoff (`_edge_features`, :42-62; `_rbf`, helper.py:29-49), xi =
    unit(x_row - x_col) [E, 1, 3], h = atom types (int64), chi = forward / backward orientations along the atom order [N, 2, 3]
    (`_node_features`, :65-84; `_orientations`, helper.py:52-59); `lig_flag` marks the last `n_ligand` atoms (:146-148).

    Neighbour selection when an atom has MORE than `max_num_neighbors` atoms within the cutoff: by default this builder keeps the
    nearest ones (deterministic, order-independent); `select="first"` keeps what torch_cluster 1.6.0's `radius_graph` keeps when it
    walks a structure's atoms in index order (its CUDA kernel; pinned by the fixture `lba_features_capped`, whose stub restates that
    walk) -- see radius_graph.  torch_cluster's CPU path (an unsorted nanoflann search) is not reproducible without that library.
    Below the cap the edge SETS are identical; the edge order here is col-sorted with each atom's neighbours by ascending distance
    ("nearest") or id ("first").  Pass `edge_index` to featurise a given graph instead."""
    lib = _lib.load()
    coords = _req(coords, "coords")
    n = coords.shape[0]
    dev = coords.device
    if edge_index is None:
        edge_index = radius_graph(coords, r=edge_cutoff, max_num_neighbors=max_num_neighbors, batch=batch, select=select)
    e_cnt = edge_index.shape[1]
    row, col = edge_index[0].to(torch.int32).contiguous(), edge_index[1].to(torch.int32).contiguous()
    f32 = dict(dtype=torch.float32, device=dev)
    e = torch.empty((e_cnt, num_rbf), **f32)
    xi = torch.empty((e_cnt, 1, 3), **f32)
    check(lib.gcpnet_nms_edge_features(e_cnt, _p(coords), _p(row), _p(col), None, 0, float(edge_cutoff), int(num_rbf), _p(e), _p(xi),
                                       _stream()), "lba edge features")
    chi = torch.empty((n, 2, 3), **f32)
    b32 = batch.to(torch.int32).contiguous() if batch is not None else None
    check(lib.gcpnet_orientations(n, _p(coords), _p(b32), _p(chi), _stream()), "orientations")
    out = dict(h=atom_types.to(device=dev, dtype=torch.long), chi=chi, e=e, xi=xi, x=coords, edge_index=edge_index)
    if n_ligand is not None:
        if batch is not None:
            raise ValueError("lba_featurize: n_ligand marks the tail of ONE structure; featurise per structure, then collate()")
        flag = torch.zeros(n, dtype=torch.bool, device=dev)
        if n_ligand > 0:
            flag[n - int(n_ligand):] = True
        out["lig_flag"] = flag
    return out


def collate(graphs: Sequence[Mapping[str, Tensor]]) -> Dict[str, Tensor]:
    """torch_geometric 2.1 `Batch.from_data_list` for plain dict samples (what the reference's DataLoader does to the `Data`
    objects its transforms return, atom3d_dataset.py:122-129): every tensor attribute with at least one dimension is
    concatenated along dim 0 in list order -- per-node, per-edge and per-graph tensors alike (`Data.__cat_dim__` = 0) --, except
    attributes whose name contains "index" (`edge_index`), which are concatenated along the LAST dimension with each graph's
    entries incremented by the number of nodes before it (`Data.__inc__`: block-diagonal adjacency); 0-dim tensors and Python
    numbers (LBA's `label`) become one entry each of a [G] tensor.  Adds `batch` [N] (graph id per node) and `ptr` [G + 1]
    (node offsets).  A graph's node count is the first dimension of `x` (else `h`), as PyG infers `num_nodes`."""
    if not graphs:
        raise ValueError("collate: empty list")
    counts = []
    for g in graphs:
        ref = g["x"] if "x" in g else g["h"]
        counts.append(int(ref.shape[0]))
    offs = [0]
    for c in counts:
        offs.append(offs[-1] + c)
    dev = (graphs[0]["x"] if "x" in graphs[0] else graphs[0]["h"]).device
    out: Dict[str, Tensor] = {}
    for k in graphs[0].keys():
        vals = [g[k] for g in graphs]
        if not torch.is_tensor(vals[0]):
            out[k] = torch.tensor(vals, device=dev)
        elif vals[0].dim() == 0:
            out[k] = torch.stack(list(vals))
        elif "index" in k:
            out[k] = torch.cat([v + o for v, o in zip(vals, offs[:-1])], dim=-1)
        else:
            out[k] = torch.cat(list(vals), dim=0)
    out["batch"] = torch.repeat_interleave(torch.arange(len(graphs), device=dev), torch.tensor(counts, device=dev))
    out["ptr"] = torch.tensor(offs, dtype=torch.long, device=dev)
    return out
