"""N > 1 path on CPU: two gloo ranks.  Checks (1) the flat-bucket gradient all-reduce against a single-process mean,
(2) that sharding a collated batch by whole graphs + all-reducing a readout reproduces the unsharded result (graphs are
block-diagonal, so no data-path collective is needed -- SURVEY.md section 8e), using the oracle as the per-rank model."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gcpnet_amd.parallel import GradAllReducer, ShardedGraph, shard_graph_batch
from tests.helpers import Fixture


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, fn, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def _run(fn, world=2):
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, fn, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)  # (a spawned interpreter's first `import torch` on a fresh or busy box can take minutes)
    codes = [p.exitcode for p in procs]
    for p in procs:
        if p.exitcode is None:
            p.kill()
    assert codes == [0] * world, codes
    return dict(ret)


def _grad_job(rank, world):
    torch.manual_seed(0)
    lin = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3))
    x = torch.arange(20, dtype=torch.float32).reshape(4, 5) * (rank + 1)
    lin(x).square().mean().backward()
    local = [p.grad.clone() for p in lin.parameters()]
    GradAllReducer(lin.parameters()).all_reduce_mean()
    return [p.grad.clone() for p in lin.parameters()], local


def test_grad_all_reducer_matches_mean_of_ranks():
    out = _run(_grad_job)
    mean = [(a + b) / 2 for a, b in zip(out[0][1], out[1][1])]
    for r in (0, 1):
        for g, m in zip(out[r][0], mean):
            assert torch.allclose(g, m, atol=1e-6)


def _shard_job(rank, world):
    from oracle import gcp_oracle as O

    f = Fixture("model_lba_small")
    batch = {k: v for k, v in f.i.items()}
    part = shard_graph_batch(batch, rank, world)
    cfg, lc = O.default_module_cfg(), O.default_layer_cfg(num_message_layers=4)
    n_local = int(part["batch"].max()) + 1 if part["batch"].numel() else 0
    pred = O.lba_forward(f.p, part, cfg, lc, 2)["pred"].reshape(-1) if n_local else torch.zeros(0)
    # readout all-reduce: sum of per-graph predictions and graph count (what a loss/metric reduction needs)
    t = torch.tensor([pred.sum().item(), float(n_local)], dtype=torch.float64)
    dist.all_reduce(t)
    return pred, part["edge_index"].shape[1], t.tolist()


def test_graph_sharding_reproduces_unsharded_readout():
    from oracle import gcp_oracle as O

    f = Fixture("model_lba_small")
    full = O.lba_forward(f.p, f.i, O.default_module_cfg(), O.default_layer_cfg(num_message_layers=4), 2)["pred"]
    out = _run(_shard_job)
    joined = torch.cat((out[0][0], out[1][0]))
    assert joined.shape == full.shape
    assert torch.allclose(joined, full, atol=1e-5, rtol=1e-4)
    assert out[0][1] + out[1][1] == f.i["edge_index"].shape[1]  # every edge lives on exactly one rank
    assert abs(out[0][1] - out[1][1]) <= f.i["edge_index"].shape[1] // 2  # balanced by edges
    assert out[0][2] == out[1][2] and abs(out[0][2][0] - full.sum().item()) < 1e-4 and out[0][2][1] == full.numel()


def test_shard_graph_batch_single_process():
    f = Fixture("model_nms_small")
    b = f.i
    parts = [shard_graph_batch(b, r, 3) for r in range(3)]
    assert sum(p["h"].shape[0] for p in parts) == b["h"].shape[0]
    for p in parts:
        assert int(p["edge_index"].max()) < p["h"].shape[0] and int(p["edge_index"].min()) >= 0
        assert torch.equal(torch.unique(p["batch"]), torch.arange(int(p["batch"].max()) + 1))


# ---- ONE graph split by target-node ranges: per-layer all-gather of node features forward,
#      reduce-scatter of their gradients backward; the oracle plays the per-rank kernels -----------------------------------
def _sharded_case(skew=False):
    """Seeded graph (not col-sorted: the sharding sorts), two GCPInteractions layers' parameters, inputs, loss weights.
    skew: a third of the edges point at six hub nodes and a stretch of nodes has no in-edge at all -- the equal-EDGE cut then gives
    the ranks very different node counts (one of them a handful of hubs), i.e. uneven table slots and empty peers in the halo lists."""
    import gcpnet_amd as G
    from oracle import gcp_oracle as O
    from tests.helpers import rand_graph

    n, e, dims = 90, 700, (24, 8)
    torch.manual_seed(5)
    layers = torch.nn.ModuleList(G.GCPInteractions(dims, (32, 4), cfg=G.default_module_cfg(scalar_nonlinearity="silu"),
                                                   layer_cfg=G.default_layer_cfg(num_message_layers=3), dropout=0.0) for _ in range(2))
    P = {k: v.detach().clone() for k, v in layers.state_dict().items()}
    ei, x = rand_graph(n, e, 6)
    if skew:
        gs = torch.Generator().manual_seed(8)
        col = ei[1].clone()
        hub = torch.randperm(e, generator=gs)[: e // 3]
        col[hub] = torch.randint(40, 46, (hub.numel(),), generator=gs)
        quiet = (col >= 60) & (col < 75)  # nodes 60..74 keep no in-edge
        col[quiet] = torch.randint(0, 40, (int(quiet.sum()),), generator=gs)
        col = torch.where(col == ei[0], (col + 1) % 40, col)
        ei = torch.stack((ei[0], col))
    g = torch.Generator().manual_seed(7)
    ins = dict(h=torch.randn(n, dims[0], generator=g), chi=torch.randn(n, dims[1], 3, generator=g),
               e=torch.randn(e, 32, generator=g), xi=torch.randn(e, 4, 3, generator=g))
    lw = dict(h=torch.randn(n, dims[0], generator=g), chi=torch.randn(n, dims[1], 3, generator=g))
    cfg = O.default_module_cfg(scalar_nonlinearity="silu", nonlinearities=("silu", None))
    lcfg = O.default_layer_cfg(num_message_layers=3)
    return P, ei, x, ins, lw, cfg, lcfg, n


def _oracle_sharded_layer(O, P, pre, h, chi, e_loc, xi_loc, sg, frames_loc, out_ei_local, out_frames, cfg, lcfg):
    """One post-norm GCPInteractions layer on a rank's share, mirroring gcpnet_amd.parallel.sharded_interactions_forward."""
    n_loc, s, v = h.shape[0], h.shape[1], chi.shape[1]
    full = sg.all_gather_rows(torch.cat((h, chi.reshape(n_loc, 3 * v)), dim=1))
    h_full, chi_full = full[:, :s], full[:, s:].reshape(-1, v, 3)
    (rs, rv), _ = O.message_passing(P, pre + "interaction.", h_full, chi_full, e_loc, xi_loc, sg.edge_index, frames_loc, cfg,
                                    lcfg["mp_cfg"], return_messages=True)
    h, chi = h + rs[sg.table_slice], chi + rv[sg.table_slice]  # (all in-edges of the local nodes are local: their mean is complete)
    h, chi = O.gcp_layer_norm(P, pre + "gcp_norm.0.", h, chi)
    no_res = dict(cfg, vector_residual=False)
    kws = [O._gcp_kwargs(no_res, nonlinearities=tuple(cfg["nonlinearities"])), O._gcp_kwargs(no_res, nonlinearities=(None, None))]
    fs, fv = h, chi
    for k, kw in enumerate(kws):  # node-level GCPs: scalarize(node_inputs=True) over the local nodes' OUT-edges
        fs, fv = O.gcp2(P, f"{pre}feedforward_network.{k}.", fs, fv, out_ei_local, out_frames, node_inputs=True, **kw)
    return O.gcp_layer_norm(P, pre + "gcp_norm.1.", h + fs, chi + fv)


def _sharded_job_halo(rank, world):
    return _sharded_job(rank, world, halo=True)


def _sharded_job_skew(rank, world):
    return _sharded_job(rank, world, halo=False, skew=True)


def _sharded_job_halo_skew(rank, world):
    return _sharded_job(rank, world, halo=True, skew=True)


def _sharded_job(rank, world, halo=False, skew=False):
    from oracle import gcp_oracle as O

    P, ei, x, ins, lw, cfg, lcfg, n = _sharded_case(skew)
    P = {k: v.clone().requires_grad_() for k, v in P.items()}
    sg = ShardedGraph(ei, n, rank, world, halo=halo)
    frames_loc = O.localize(x, sg.edge_index_global)  # (positions are replicated, by global id)
    out_frames = O.localize(x, sg.out_edge_index_global)
    out_ei_local = torch.stack((sg.out_row_local, sg.out_edge_index_global[1]))
    h = sg.local_nodes(ins["h"]).clone().requires_grad_()
    chi = sg.local_nodes(ins["chi"]).clone().requires_grad_()
    e_loc = sg.local_edges(ins["e"]).clone().requires_grad_()
    xi_loc = sg.local_edges(ins["xi"]).clone().requires_grad_()
    hh, cc = h, chi
    for i in range(2):
        hh, cc = _oracle_sharded_layer(O, P, f"{i}.", hh, cc, e_loc, xi_loc, sg, frames_loc, out_ei_local, out_frames, cfg, lcfg)
    loss = (hh * sg.local_nodes(lw["h"])).sum() + (cc * sg.local_nodes(lw["chi"])).sum()
    loss.backward()
    params = [p for p in P.values()]
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    GradAllReducer(params).all_reduce_sum()
    return dict(n0=sg.n0, n1=sg.n1, e0=sg.e0, e1=sg.e1, edges=sg.edge_counts, perm=sg.edge_perm,
                h=hh.detach(), chi=cc.detach(), dh=h.grad, dchi=chi.grad, de=e_loc.grad, dxi=xi_loc.grad,
                w={k: v.grad.clone() for k, v in P.items()})


@pytest.mark.parametrize("halo,world,skew", [(False, 2, False), (True, 2, False), (False, 4, False), (True, 4, False),
                                             (False, 4, True), (True, 4, True), (False, 8, True), (True, 8, True)])
def test_sharded_graph_matches_unsharded(halo, world, skew):
    """halo=False: all-gather of the whole feature table; halo=True: all-to-all of the rows the peers' in-edges reference.  2 and 4
    ranks, and 8 (one per GPU of a node: the world size the driver's scaling run uses) on the skewed graph; `skew`: hub nodes and
    nodes without in-edges, so that the equal-edge cut leaves the ranks with very uneven node ranges."""
    from oracle import gcp_oracle as O

    P, ei, x, ins, lw, cfg, lcfg, n = _sharded_case(skew)
    P = {k: v.clone().requires_grad_() for k, v in P.items()}
    ci = {k: v.clone().requires_grad_() for k, v in ins.items()}
    fr = O.localize(x, ei)
    hh, cc = ci["h"], ci["chi"]
    for i in range(2):
        hh, cc = O.gcp_interactions(P, f"{i}.", hh, cc, ci["e"], ci["xi"], ei, fr, cfg, lcfg)
    ((hh * lw["h"]).sum() + (cc * lw["chi"]).sum()).backward()
    job = {(False, False): _sharded_job, (True, False): _sharded_job_halo, (False, True): _sharded_job_skew,
           (True, True): _sharded_job_halo_skew}[(halo, skew)]
    out = _run(job, world)
    assert out[0]["n0"] == 0 and out[world - 1]["n1"] == n and all(out[r]["n1"] == out[r + 1]["n0"] for r in range(world - 1))
    assert sum(out[0]["edges"]) == ei.shape[1]
    if skew:  # the node ranges really are uneven (a rank of hubs next to ranks of ordinary nodes)
        sizes = [out[r]["n1"] - out[r]["n0"] for r in range(world)]
        assert max(sizes) >= 4 * max(1, min(sizes)), sizes
    else:
        assert max(out[0]["edges"]) - min(out[0]["edges"]) <= 40  # equal-edge cut
    perm = out[0]["perm"]
    de, dxi = ci["e"].grad[perm], ci["xi"].grad[perm]  # (the shards hold their edges in col-sorted order)
    tol = dict(atol=2e-6, rtol=1e-5)
    for r in range(world):
        o = out[r]
        sl, es = slice(o["n0"], o["n1"]), slice(o["e0"], o["e1"])
        assert torch.allclose(o["h"], hh.detach()[sl], **tol) and torch.allclose(o["chi"], cc.detach()[sl], **tol)
        assert torch.allclose(o["dh"], ci["h"].grad[sl], **tol) and torch.allclose(o["dchi"], ci["chi"].grad[sl], **tol)
        assert torch.allclose(o["de"], de[es], **tol) and torch.allclose(o["dxi"], dxi[es], **tol)
        for k, v in P.items():
            want = v.grad if v.grad is not None else torch.zeros_like(v)
            assert torch.allclose(o["w"][k], want, atol=2e-6 * max(1.0, float(want.abs().max())), rtol=1e-5), k


# ---- data preparation of the halo mode: spatial order and node renaming (no process group) --------------------------------------
def test_spatial_order_is_a_permutation_that_makes_ranges_compact():
    from gcpnet_amd.parallel import ShardedGraph, spatial_order
    from gcpnet_amd.synthetic import make_inputs, reorder_nodes

    host = make_inputs(4000, 12, (8, 2), (4, 1), seed=3)
    perm = spatial_order(host["x"])
    assert sorted(perm.tolist()) == list(range(4000))
    re = reorder_nodes(host, perm)
    ei = re["edge_index"]
    assert bool((ei[1][1:] >= ei[1][:-1]).all()), "edges stay sorted by target"
    # the renamed graph is the same graph: identical multiset of (edge length, edge feature row sum), identical node features per position
    d0 = (host["x"][host["edge_index"][0]] - host["x"][host["edge_index"][1]]).norm(dim=1) + host["e"].sum(1)
    d1 = (re["x"][ei[0]] - re["x"][ei[1]]).norm(dim=1) + re["e"].sum(1)
    assert torch.allclose(d0.sort().values, d1.sort().values)
    assert torch.equal(re["h"], host["h"][perm]) and torch.equal(re["x"], host["x"][perm])
    # ... and a contiguous node range now has a small halo: compare the remote sources of rank 1 of 4 before and after
    before = ShardedGraph(host["edge_index"], 4000, 1, 4).halo_nodes().numel()
    after = ShardedGraph(ei, 4000, 1, 4).halo_nodes().numel()
    assert after < before // 2, (before, after)


def test_halo_lists_are_consistent_across_ranks():
    """Every rank derives all send / receive lists from the replicated edge list: what rank q sends to rank r must be what rank r
    expects from rank q, row for row (no process group needed to check that)."""
    from gcpnet_amd.parallel import ShardedGraph
    from gcpnet_amd.synthetic import make_inputs

    host = make_inputs(1500, 8, (8, 2), (4, 1), seed=5)
    world = 3
    sgs = [ShardedGraph(host["edge_index"], 1500, r, world, halo=True) for r in range(world)]
    for r, sg in enumerate(sgs):
        assert sg.table_rows == sg.n_local + sg.halo_ids.numel() and sum(sg.recv_counts) == sg.halo_ids.numel()
        assert sg.recv_counts[r] == 0 and sg.send_counts[r] == 0
        # table ids of the edge sources point at the right global node
        table_gid = torch.cat((torch.arange(sg.n0, sg.n1), sg.halo_ids))
        assert torch.equal(table_gid[sg.edge_index[0]], sg.edge_index_global[0])
        ro = 0
        for q in range(world):
            want = sg.halo_ids[ro:ro + sg.recv_counts[q]]          # global ids rank r receives from rank q, in order
            so = sum(sgs[q].send_counts[:r])
            sent = sgs[q].send_index[so:so + sgs[q].send_counts[r]] + sgs[q].n0
            assert torch.equal(want, sent), (r, q)
            ro += sg.recv_counts[q]
