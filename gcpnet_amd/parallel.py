"""Multi-GPU data path: one process per GPU, graphs sharded by whole graphs, one RCCL all-reduce per step.

The reference scales only by Lightning DDP replicas (configs/trainer/default.yaml:8, configs/trainer/ddp.yaml:4-8):
independent graph batches per rank plus a gradient all-reduce.  Graphs in a batch are block-diagonal, so sharding
whole graphs needs no halo and no data-path collective; the only exchange is the weight-gradient mean (NMS model
1.8 MB, LBA 7.3 MB of fp32), sent as ONE flat bucket: on the xGMI mesh this size is latency-bound, so a single
collective per step beats per-tensor calls.  `backend="nccl"` is RCCL on ROCm; the CPU tests use gloo.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Sequence

import torch
import torch.distributed as dist


class GradAllReducer:
    """Averages the gradients of `params` across ranks through one persistent flat buffer."""

    def __init__(self, params: Iterable[torch.nn.Parameter], group=None):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.group = group
        from . import ops  # this reducer reads the gradients AFTER the backward pass: the weight-gradient stream stays usable

        ops.SIDE_STREAM_UNDER_DISTRIBUTED = True
        n = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(n, dtype=ref.dtype, device=ref.device)
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()

    @torch.no_grad()
    def all_reduce_sum(self):
        """As all_reduce_mean, with a SUM: for ranks that hold disjoint shares of ONE graph's rows (ShardedGraph), where the
        weight gradient of the whole graph is the sum of the ranks' gradients."""
        self.all_reduce_mean(_mean=False)

    @torch.no_grad()
    def all_reduce_mean(self, _mean: bool = True):
        """Gradients -> flat bucket (one multi-tensor copy), ONE all-reduce (averaging inside RCCL), and the parameters'
        .grad become views of the bucket (no copy back): three launches per step instead of two per parameter tensor."""
        world = dist.get_world_size(self.group)
        dst, src = [], []
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():  # (already the bucket view when the caller accumulates in place)
                dst.append(v)
                src.append(p.grad)
        if dst:
            torch._foreach_copy_(dst, src)
        if not _mean:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        elif dist.get_backend(self.group) == "nccl":
            dist.all_reduce(self.flat, op=dist.ReduceOp.AVG, group=self.group)
        else:  # gloo has no AVG
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
            self.flat.div_(world)
        for p, v in zip(self.params, self.views):
            p.grad = v


def shard_graph_batch(batch: Dict[str, torch.Tensor], rank: int, world: int) -> Dict[str, torch.Tensor]:
    """Returns rank's share of a collated batch of graphs (fields h, chi, e, xi, x, edge_index, batch[, label]):
    a contiguous range of whole graphs balanced by EDGE count (compute is proportional to edges), with node indices,
    edge_index and the batch vector re-based to the local range.  Requires nodes grouped by graph (PyG collation)."""
    bidx, ei = batch["batch"], batch["edge_index"]
    n_graphs = int(bidx.max()) + 1
    edge_graph = bidx[ei[1]]
    epg = torch.bincount(edge_graph, minlength=n_graphs).double()
    cum = torch.cumsum(epg, 0)
    total = float(cum[-1]) if n_graphs else 0.0
    # graph g goes to the rank whose edge-quantile interval contains the midpoint of g's edge range
    mid = cum - epg / 2
    owner = torch.clamp((mid / max(total, 1.0) * world).floor().long(), max=world - 1)
    mine = owner == rank
    node_mask = mine[bidx]
    edge_mask = mine[edge_graph]
    node_ids = torch.nonzero(node_mask).flatten()
    remap = torch.full((bidx.shape[0],), -1, dtype=torch.long, device=bidx.device)
    remap[node_ids] = torch.arange(node_ids.numel(), device=bidx.device)
    graph_ids = torch.nonzero(mine).flatten()
    gmap = torch.full((n_graphs,), -1, dtype=torch.long, device=bidx.device)
    gmap[graph_ids] = torch.arange(graph_ids.numel(), device=bidx.device)
    out = {}
    for k, v in batch.items():
        if k == "edge_index":
            out[k] = remap[ei[:, edge_mask]]
        elif k == "batch":
            out[k] = gmap[bidx[node_mask]]
        elif torch.is_tensor(v) and v.shape[:1] == bidx.shape[:1]:
            out[k] = v[node_mask]
        elif torch.is_tensor(v) and v.shape[:1] == ei.shape[1:2]:
            out[k] = v[edge_mask]
        elif torch.is_tensor(v) and v.shape[:1] == (n_graphs,):
            out[k] = v[mine]
        else:
            out[k] = v
    return out


# ==============================================================================================================
# ONE large graph across ranks: strong scaling of BASELINE configs[1] / configs[4]
# ==============================================================================================================
class ShardedGraph:
    """Partition of one graph by TARGET-node ranges for message passing over several GPUs.

    Edges are taken in col-sorted order and cut into `world` pieces of (almost) equal EDGE count, snapped to node boundaries:
    rank r owns the nodes [n0, n1) and all their in-edges (col in range) -- the rows of the message kernels and of the
    aggregation -- so compute per rank is proportional to its edges.  Messages need the features of the SOURCE nodes
    (h[row], chi[row]), which may live anywhere: per layer one all-gather of the node features forward and the matching
    reduce-scatter of their gradients backward (`all_gather_rows`, an autograd Function); at configs[4] size that is 100 000 x
    352 floats = 141 MB per layer, ~17.6 MB per peer over 7 xGMI links in parallel.  Everything else is local: per-edge features and
    frames, aggregation (scatter over the rank's own col range), node-level GCPs (their `scalarize(node_inputs=True)` needs the
    mean frame over each local node's OUT-edges, built once from the replicated positions and index arrays).  Weights are
    replicated; their gradients go through GradAllReducer with a SUM (each rank holds a disjoint share of the rows).
    Bytes per rank and layer: it sends its [n_local, s + 3V] rows to 7 peers and receives the other ranks' rows (forward), the
    reverse for the gradients (backward); `bench.py --dry-run-world N` prints the figures for a configuration.

    This class is index bookkeeping + collectives only (device-agnostic; the CPU tests drive it with the oracle)."""

    def __init__(self, edge_index: torch.Tensor, n_nodes: int, rank: int, world: int, group=None, halo: bool = False):
        """`halo`: exchange only the rows a rank's in-edges actually reference (all-to-all of per-peer row lists built here, once,
        from the replicated edge list -- no set-up communication) instead of all-gathering the whole table; pays when the node ids
        are spatially sorted (`spatial_order`), where the halo is a surface term."""
        self.rank, self.world, self.group, self.n_nodes, self.halo = rank, world, group, int(n_nodes), bool(halo)
        row, col = edge_index[0].long(), edge_index[1].long()
        n_edges = int(col.shape[0])
        if n_edges > 1 and not bool((col[1:] >= col[:-1]).all()):
            self.edge_perm = torch.argsort(col, stable=True)
            row, col = row[self.edge_perm], col[self.edge_perm]
        else:
            self.edge_perm = None
        counts = torch.bincount(col, minlength=self.n_nodes)
        ptr = torch.zeros(self.n_nodes + 1, dtype=torch.long, device=col.device)
        ptr[1:] = torch.cumsum(counts, 0)
        targets = torch.tensor([(k * n_edges) // world for k in range(world + 1)], dtype=torch.long, device=col.device)
        bounds = torch.searchsorted(ptr, targets, right=False).clamp(max=self.n_nodes)  # first node whose edges start at/after the cut
        bounds[0], bounds[-1] = 0, self.n_nodes
        self.bounds = [int(b) for b in torch.cummax(bounds, 0).values.tolist()]
        self.n0, self.n1 = self.bounds[rank], self.bounds[rank + 1]
        self.e0, self.e1 = int(ptr[self.n0]), int(ptr[self.n1])
        self.node_counts = [self.bounds[k + 1] - self.bounds[k] for k in range(world)]
        self.max_nodes = max(self.node_counts)
        self.edge_counts = [int(ptr[self.bounds[k + 1]] - ptr[self.bounds[k]]) for k in range(world)]
        # local in-edges three ways: with GLOBAL node ids (frames are built from the replicated positions), with TABLE ids (the
        # gathers read the all-gathered feature table, see _gather: rank k's nodes sit at rows k * max_nodes ..., so that the
        # collective's output buffer IS the table and neither direction needs a re-packing pass) and with local target ids
        # (aggregation)
        self.edge_index_global = torch.stack((row[self.e0:self.e1], col[self.e0:self.e1]))
        bnd = torch.tensor(self.bounds, dtype=torch.long, device=col.device)
        owner = torch.bucketize(self.edge_index_global, bnd[1:-1], right=True)
        self.edge_index = owner * self.max_nodes + (self.edge_index_global - bnd[owner])
        self.col_local = col[self.e0:self.e1] - self.n0
        self.table_rows = world * self.max_nodes
        self.table_slice = slice(rank * self.max_nodes, rank * self.max_nodes + (self.n1 - self.n0))
        if self.halo:
            # table = [own rows | halo rows]; the halo = the remote source nodes of the local in-edges, ascending by global id,
            # which is also grouped by owner (owners hold contiguous id ranges).  Rank q's send list for rank r is
            # halo(r) restricted to q's range -- every rank derives all of them from the replicated, col-sorted edge list.
            src_owner = torch.bucketize(row, bnd[1:-1], right=True)   # owner of every edge's SOURCE node
            dst_owner = torch.bucketize(col, bnd[1:-1], right=True)   # ... of its TARGET node = the rank that holds the edge
            cross = src_owner != dst_owner
            pair = dst_owner[cross] * self.n_nodes + row[cross]       # (consumer rank, source node), de-duplicated and sorted
            pair = torch.unique(pair)
            cons, gid = pair // self.n_nodes, pair % self.n_nodes
            prod = torch.bucketize(gid, bnd[1:-1], right=True)        # the rank that owns (and sends) the row
            mine = cons == rank
            self.halo_ids = gid[mine]                                  # what this rank receives, in table order
            self.recv_counts = torch.bincount(prod[mine], minlength=world).tolist()
            send = prod == rank
            send_cons, send_gid = cons[send], gid[send]                # sorted by consumer, then id: the all-to-all's send order
            self.send_index = send_gid - self.n0                       # local row of every piece this rank sends
            self.send_counts = torch.bincount(send_cons, minlength=world).tolist()
            n_loc = self.n1 - self.n0
            loc = self.edge_index_global[0]
            remote = (loc < self.n0) | (loc >= self.n1)
            pos = torch.searchsorted(self.halo_ids, loc.clamp(min=0))
            src_t = torch.where(remote, n_loc + pos, loc - self.n0)
            self.edge_index = torch.stack((src_t, self.col_local))
            self.table_rows = n_loc + int(self.halo_ids.numel())
            self.table_slice = slice(0, n_loc)
        # out-edges of the local nodes (row in range), for the node-level mean frames: [2, E_out] with LOCAL row ids
        out_mask = (row >= self.n0) & (row < self.n1)
        self.out_edge_index_global = torch.stack((row[out_mask], col[out_mask]))
        self.out_row_local = row[out_mask] - self.n0

    def to(self, device) -> "ShardedGraph":
        """Moves the index tensors (they are built where `edge_index` lives) to `device`."""
        for name in ("edge_index", "edge_index_global", "col_local", "out_edge_index_global", "out_row_local", "edge_perm", "halo_ids",
                     "send_index"):
            t = getattr(self, name, None)
            if t is not None:
                setattr(self, name, t.to(device))
        return self

    @property
    def n_local(self) -> int:
        return self.n1 - self.n0

    def local_nodes(self, t: torch.Tensor) -> torch.Tensor:
        return t[self.n0:self.n1]

    def local_edges(self, t: torch.Tensor) -> torch.Tensor:
        """Rows of a per-edge tensor (given in the caller's edge order) that belong to this rank, in col-sorted order."""
        if self.edge_perm is not None:
            t = t[self.edge_perm.to(t.device)]
        return t[self.e0:self.e1]

    def halo_nodes(self) -> torch.Tensor:
        """Global ids of the REMOTE source nodes this rank's in-edges reference (what a halo exchange would have to fetch instead
        of the whole table; `bench.py --dry-run-world` prints its size next to the all-gather's)."""
        src = torch.unique(self.edge_index_global[0])
        return src[(src < self.n0) | (src >= self.n1)]

    # ---- collectives -----------------------------------------------------------------------------------------------
    def _collectives(self) -> bool:
        """One rank WITH a process group still goes through the collectives (tests/test_rccl_world1.py runs the RCCL branches that
        way on a one-GPU box); one rank without torch.distributed does not."""
        return self.world > 1 or (dist.is_available() and dist.is_initialized())

    def _gather(self, local: torch.Tensor) -> torch.Tensor:
        """[n_local, D] -> the feature table [world * max_nodes, D]: rank k's rows at k * max_nodes ... (padding rows are zero and
        never referenced: `edge_index` holds table ids).  ONE copy of the local rows into their slot of the output buffer and an
        in-place all-gather; no concatenation afterwards."""
        D = local.shape[1]
        if self.halo:
            return self._halo_forward(local)
        buf = local.new_empty((self.table_rows, D))
        own = buf[self.rank * self.max_nodes: (self.rank + 1) * self.max_nodes]
        own[: self.n_local].copy_(local)
        if self.n_local < self.max_nodes:
            own[self.n_local:].zero_()
        if not self._collectives():
            return buf
        if dist.get_backend(self.group) == "nccl":
            dist.all_gather_into_tensor(buf, own, group=self.group)  # (in place: `own` is this rank's slot of `buf`)
        else:
            chunks = [buf[k * self.max_nodes: (k + 1) * self.max_nodes] for k in range(self.world)]
            dist.all_gather(chunks, own.clone(), group=self.group)  # gloo: list form, each chunk a view of the table
        return buf

    def _scatter_sum(self, full: torch.Tensor) -> torch.Tensor:
        """Sum over ranks of the table-shaped gradient `full` [world * max_nodes, D]; this rank's rows returned.  The table layout
        is the reduce-scatter's input layout: no re-packing."""
        if self.halo:
            return self._halo_backward(full)
        if not self._collectives():
            return full[self.table_slice].contiguous()
        D = full.shape[1]
        if dist.get_backend(self.group) == "nccl":
            out = full.new_empty((self.max_nodes, D))
            dist.reduce_scatter_tensor(out, full, op=dist.ReduceOp.SUM, group=self.group)
            return out[: self.n_local].contiguous()
        total = full.clone()  # gloo has no reduce-scatter
        dist.all_reduce(total, op=dist.ReduceOp.SUM, group=self.group)
        return total[self.table_slice].contiguous()

    # ---- halo exchange: only the rows the peers' in-edges reference ---------------------------------------------------------
    def _all_to_all(self, recv: torch.Tensor, send: torch.Tensor, recv_counts, send_counts) -> None:
        """recv rows grouped by source rank <- send rows grouped by destination rank: one all_to_all_single (RCCL; gloo in the
        CPU tests implements it too)."""
        if not self._collectives():
            return
        dist.all_to_all_single(recv, send, output_split_sizes=list(recv_counts), input_split_sizes=list(send_counts), group=self.group)

    def _halo_forward(self, local: torch.Tensor) -> torch.Tensor:
        n_loc, D = self.n_local, local.shape[1]
        table = local.new_empty((self.table_rows, D))
        table[:n_loc].copy_(local)
        send = local.index_select(0, self.send_index)  # rows grouped by consumer rank
        self._all_to_all(table[n_loc:], send, self.recv_counts, self.send_counts)  # (a row slice of the table: contiguous)
        return table

    def _halo_backward(self, g_table: torch.Tensor) -> torch.Tensor:
        """The gradient rows of the halo go back to their owners, which add them to their own rows' gradients."""
        n_loc = self.n_local
        g_local = g_table[:n_loc].clone()
        back = g_table.new_zeros((int(self.send_index.numel()), g_table.shape[1]))
        self._all_to_all(back, g_table[n_loc:].contiguous(), self.send_counts, self.recv_counts)
        if back.shape[0] == 0:
            return g_local
        if back.is_cuda:
            # ONE segmented sum over the whole receive buffer, accumulated onto the local rows' own gradients: the rows that came
            # back for local node j are added in buffer order (= consumer-rank order) by one lane -- deterministic, no atomics,
            # one launch whatever the number of peers (gcpnet_segment_reduce; until round 5: an ATen index_add_ per peer)
            from . import _lib, ops

            plan = getattr(self, "_halo_plan", None)
            if plan is None or plan.idx.device != back.device:
                plan = self._halo_plan = ops.GatherPlan(self.send_index.to(back.device), n_loc)
            D = back.shape[1]
            _lib.check(_lib.load().gcpnet_segment_reduce(plan.n_src, ops._p(plan.seg_ptr), ops._p(plan.perm), ops._p(back), D, D, 0,
                                                         ops._p(g_local), D, 1, ops._stream()), "segment_reduce(halo)")
            return g_local
        # host tensors (the gloo tests, where the oracle stands in for the kernels): one index_add_ per consumer rank -- inside a
        # consumer's block the local rows are distinct, and the blocks are added in rank order: the same sum in every run
        off = 0
        for k in range(self.world):
            c = self.send_counts[k]
            if c:
                g_local.index_add_(0, self.send_index[off:off + c], back[off:off + c])
            off += c
        return g_local

    def all_gather_rows(self, local: torch.Tensor) -> torch.Tensor:
        """[n_local, D] -> the table [world * max_nodes, D] over all ranks (rows by TABLE id: `edge_index`, `table_slice`);
        backward = reduce-scatter (sum) of the gradient rows to their owners."""
        return _AllGatherRows.apply(local, self)


def spatial_order(x: torch.Tensor, cell: float = 4.5) -> torch.Tensor:
    """Permutation that sorts nodes along a Morton (Z-order) curve over cells of edge `cell` (the radius-graph cutoff): nodes that
    are close in space get close ids, so that a contiguous node range is a compact region and the SOURCE nodes of its in-edges that
    live on other ranks (the halo) are a surface term instead of "almost everybody" (`bench.py --dry-run-world` prints both).
    Data preparation, like the CSR sort: new_id[perm[k]] = k; apply with `x[perm]`, `h[perm]`, `inv[edge_index]`."""
    c = torch.floor((x - x.min(dim=0).values) / float(cell)).long().clamp_(min=0, max=(1 << 20) - 1)
    code = torch.zeros(x.shape[0], dtype=torch.long, device=x.device)
    for b in range(20):
        for d in range(3):
            code |= ((c[:, d] >> b) & 1) << (3 * b + d)
    return torch.argsort(code, stable=True)


class _AllGatherRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, local, sg: ShardedGraph):
        ctx.sg = sg
        return sg._gather(local.contiguous())

    @staticmethod
    def backward(ctx, g_full):
        return ctx.sg._scatter_sum(g_full.contiguous()), None


def sharded_interactions_forward(layer, node_rep, edge_rep, sg: ShardedGraph, frames, node_frames, node_pos=None):
    """`GCPInteractions.forward` (components/gcpnet.py:1161-1262, unmasked) on one rank's share of a ShardedGraph.

    node_rep = (h, chi) of the LOCAL nodes, edge_rep = (e, xi) and `frames` of the LOCAL in-edges (sg.local_edges order),
    `node_frames` [n_local, 3, 3] the local nodes' mean out-edge frames.  Returns the updated local (h, chi) [and positions]."""
    from . import ops
    from .components import ScalarVector
    from .gcpnet import _sv_add

    h, chi = node_rep
    n_loc, s, v = h.shape[0], h.shape[1], chi.shape[1]
    node_rep = ScalarVector(h, chi)
    if layer.pre_norm:
        node_rep = layer.gcp_norm[0](node_rep)
    # ---- the one exchange of the layer: every rank's node features, flattened [s | V x 3] (ScalarVector.flatten) ----------
    flat = torch.cat((node_rep[0], node_rep[1].reshape(n_loc, 3 * v)), dim=1)
    full = sg.all_gather_rows(flat)
    h_full, chi_full = full[:, :s], full[:, s:].reshape(-1, v, 3)
    mp = layer.interaction
    plan = ops.GatherPlan(sg.col_local, n_loc) if not hasattr(sg, "_col_plan") else sg._col_plan
    sg._col_plan = plan
    mean = mp.reduce_function == "mean"
    full_rep = ScalarVector(h_full.contiguous(), chi_full.contiguous())
    done = False
    if ops.FUSE_AGGREGATION and not mp.use_scalar_message_attention:  # (the aggregation inside the chain's Function, as unsharded)
        m, done = mp._fused_messages(full_rep, ScalarVector(*edge_rep), sg.edge_index, frames, agg=(plan, mean))
    else:
        m = mp._messages(full_rep, ScalarVector(*edge_rep), sg.edge_index, frames)
    if done:
        hidden = m
    else:
        agg_s = ops.segment_reduce(m[0], plan, mean)
        agg_v = ops.segment_reduce(m[1].reshape(m[1].shape[0], 3 * v), plan, mean).reshape(n_loc, v, 3)
        hidden = ScalarVector(agg_s, agg_v)
    if layer.gcp_dropout[0].active:
        hidden = layer.gcp_dropout[0](hidden)
    node_rep = (layer.gcp_norm[1] if layer.pre_norm else layer.gcp_norm[0])(node_rep, residual=hidden)

    def node_gcp(module, rep):  # GCP2.forward(node_inputs=True) on local rows with the precomputed mean frames
        if getattr(module, "enable_e3_equivariance", False) and not getattr(module, "ablate_frame_updates", False):
            # |.| of the x_cross projections is taken per out-edge BEFORE the mean (GCP2._forward_node_e3): the mean out-edge
            # frame used here would silently give other numbers
            raise NotImplementedError("sharded_interactions_forward: enable_e3_equivariance on node rows")
        out = module.apply_rows([rep[0]], [None], [rep[1]], [None], node_frames)
        return ScalarVector(*out) if isinstance(out, tuple) else out

    hidden = node_rep
    for module in layer.feedforward_network:
        hidden = node_gcp(module, hidden)
    if layer.gcp_dropout[1].active:
        hidden = layer.gcp_dropout[1](hidden)
    node_rep = _sv_add(node_rep, hidden) if layer.pre_norm else layer.gcp_norm[1](node_rep, residual=hidden)
    if not layer.updating_node_positions:
        return node_rep
    rep = node_rep
    for gcp in layer.node_position_update_network:
        rep = node_gcp(gcp, rep)
    upd = rep[1].reshape(n_loc, 3)
    if not layer.ablate_x_force_update:
        # inter-node force term (reference gcpnet.py:1143-1153): coef[e] = W3 act(phi_i(h)[row] + phi_j(h)[col]) on the rank's
        # in-edges -- phi_i of the SOURCE nodes comes through a second all-gather (an [n_local, s] table per layer), phi_j of the
        # targets is local (placed in the rank's slot of a table-shaped buffer so that one index space serves both) -- then the
        # mean over each local node's in-edges
        hv = rep[0]
        A = sg.all_gather_rows(ops.linear(hv, layer.phi_force_i.weight, layer.phi_force_i.bias))
        Bl = ops.linear(hv, layer.phi_force_j.weight, layer.phi_force_j.bias)
        B = _PlaceRows.apply(Bl, sg)
        plan_t = ops.GraphPlan.get(sg.edge_index, sg.table_rows)
        force = ops.edge_force(A, B, layer.phi_force_ij[1].weight, frames, plan_t, layer.force_act, layer.force_slope)
        upd = ops.axpy(upd, ops.segment_reduce(force, plan, True), 1.0)
    return node_rep, ops.axpy_clamp(node_pos, upd, float(layer.node_positions_weight), -100.0, 100.0)


class _PlaceRows(torch.autograd.Function):
    """local [n_local, D] -> table-shaped [world * max_nodes, D] with the local rows in this rank's slot and zeros elsewhere
    (no communication); backward = the slot of the gradient."""

    @staticmethod
    def forward(ctx, local, sg):
        ctx.sg = sg
        buf = local.new_zeros((sg.table_rows, local.shape[1]))
        buf[sg.table_slice] = local
        return buf

    @staticmethod
    def backward(ctx, g):
        return g[ctx.sg.table_slice].contiguous(), None
