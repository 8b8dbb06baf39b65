"""The product's sharded-graph path on the GPU (gcpnet_amd.parallel.ShardedGraph + sharded_interactions_forward): two ranks
share cuda:0 and talk over gloo (RCCL refuses two ranks on one device; the collective calls are the same); the result --
outputs, input gradients, summed weight gradients -- must equal the unsharded GCPInteractions stack on the same GPU."""
import os

import pytest
import torch

from tests.test_parallel_gloo import _run

pytestmark = pytest.mark.gpu


def _case():
    import gcpnet_amd as G
    from tests.helpers import rand_graph

    n, e, dims = 500, 6000, (128, 16)
    torch.manual_seed(5)
    layers = torch.nn.ModuleList(G.GCPInteractions(dims, (32, 4), cfg=G.default_module_cfg(scalar_nonlinearity="silu"),
                                                   layer_cfg=G.default_layer_cfg(), dropout=0.0) for _ in range(2))
    ei, x = rand_graph(n, e, 6)
    g = torch.Generator().manual_seed(7)
    ins = dict(h=torch.randn(n, dims[0], generator=g), chi=torch.randn(n, dims[1], 3, generator=g),
               e=torch.randn(e, 32, generator=g), xi=torch.randn(e, 4, 3, generator=g))
    lw = dict(h=torch.randn(n, dims[0], generator=g), chi=torch.randn(n, dims[1], 3, generator=g))
    return layers, ei, x, ins, lw, n


def _sharded_gpu_job_halo(rank, world):
    return _sharded_gpu_job(rank, world, halo=True)


def _sharded_gpu_job(rank, world, halo=False):
    import gcpnet_amd as G
    from gcpnet_amd import ops
    from gcpnet_amd.parallel import GradAllReducer, ShardedGraph, sharded_interactions_forward

    torch.cuda.set_device(0)
    layers, ei, x, ins, lw, n = _case()
    layers = layers.cuda().eval()
    sg = ShardedGraph(ei, n, rank, world, halo=halo)
    xg = x.cuda()
    sg.to("cuda")
    frames = G.localize(xg, sg.edge_index_global)
    fr_out = G.localize(xg, sg.out_edge_index_global)
    node_frames = ops.segment_reduce(fr_out.reshape(-1, 9), ops.GatherPlan(sg.out_row_local, sg.n_local),
                                     mean=True).reshape(sg.n_local, 3, 3)
    loc = dict(h=sg.local_nodes(ins["h"]), chi=sg.local_nodes(ins["chi"]), e=sg.local_edges(ins["e"]), xi=sg.local_edges(ins["xi"]))
    loc = {k: v.cuda().clone().requires_grad_() for k, v in loc.items()}
    h, chi = loc["h"], loc["chi"]
    for layer in layers:
        h, chi = sharded_interactions_forward(layer, (h, chi), (loc["e"], loc["xi"]), sg, frames, node_frames)
    ((h * sg.local_nodes(lw["h"]).cuda()).sum() + (chi * sg.local_nodes(lw["chi"]).cuda()).sum()).backward()
    params = [p for p in layers.parameters()]
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    GradAllReducer(params).all_reduce_sum()
    torch.cuda.synchronize()
    return dict(n0=sg.n0, n1=sg.n1, e0=sg.e0, e1=sg.e1, perm=sg.edge_perm.cpu(), h=h.detach().cpu(), chi=chi.detach().cpu(),
                d={k: v.grad.cpu() for k, v in loc.items()}, w={k: p.grad.cpu() for k, p in layers.named_parameters()})


@pytest.mark.parametrize("halo", [False, True])
def test_sharded_gpu_forward_backward_matches_unsharded(halo):
    layers, ei, x, ins, lw, n = _case()
    layers = layers.cuda().eval()
    import gcpnet_amd as G

    gi = {k: v.cuda().requires_grad_() for k, v in ins.items()}
    fr = G.localize(x.cuda(), ei.cuda())
    h, chi = gi["h"], gi["chi"]
    for layer in layers:
        h, chi = layer((h, chi), (gi["e"], gi["xi"]), ei.cuda(), fr)
    ((h * lw["h"].cuda()).sum() + (chi * lw["chi"].cuda()).sum()).backward()
    ref = dict(h=h.detach().cpu(), chi=chi.detach().cpu(), d={k: v.grad.cpu() for k, v in gi.items()},
               w={k: p.grad.cpu() for k, p in layers.named_parameters()})
    out = _run(_sharded_gpu_job_halo if halo else _sharded_gpu_job)
    perm = out[0]["perm"]

    def ok(a, b, name):
        tol = 2e-5 * max(1.0, float(b.abs().max()))
        assert torch.allclose(a, b, atol=tol, rtol=1e-4), f"{name}: max diff {(a - b).abs().max():.3e} (scale {b.abs().max():.3e})"

    for r in (0, 1):
        o = out[r]
        sl, es = slice(o["n0"], o["n1"]), slice(o["e0"], o["e1"])
        ok(o["h"], ref["h"][sl], "h"); ok(o["chi"], ref["chi"][sl], "chi")
        ok(o["d"]["h"], ref["d"]["h"][sl], "dh"); ok(o["d"]["chi"], ref["d"]["chi"][sl], "dchi")
        ok(o["d"]["e"], ref["d"]["e"][perm][es], "de"); ok(o["d"]["xi"], ref["d"]["xi"][perm][es], "dxi")
        for k, want in ref["w"].items():
            ok(o["w"][k], want, k)


# ---- position updates with the inter-node force term on the sharded path (reference gcpnet.py:1143-1153) -----------------------
def _force_case():
    import gcpnet_amd as G
    from tests.helpers import rand_graph

    n, e, dims = 400, 5000, (64, 16)
    torch.manual_seed(15)
    layers = torch.nn.ModuleList(
        G.GCPInteractions(dims, (32, 4), cfg=G.default_module_cfg(scalar_nonlinearity="silu", ablate_x_force_update=False),
                          layer_cfg=G.default_layer_cfg(num_message_layers=4), dropout=0.0, updating_node_positions=True) for _ in range(2))
    with torch.no_grad():
        for layer in layers:  # (the reference initialises the last force Linear with gain 0.001: redrawn so that the term is visible)
            layer.phi_force_ij[1].weight.normal_(0, 0.2)
    ei, x = rand_graph(n, e, 16)
    g = torch.Generator().manual_seed(17)
    ins = dict(h=torch.randn(n, dims[0], generator=g), chi=torch.randn(n, dims[1], 3, generator=g),
               e=torch.randn(e, 32, generator=g), xi=torch.randn(e, 4, 3, generator=g))
    lw = dict(h=torch.randn(n, dims[0], generator=g), chi=torch.randn(n, dims[1], 3, generator=g), x=torch.randn(n, 3, generator=g))
    return layers, ei, x, ins, lw, n


def _sharded_force_job(rank, world):
    import gcpnet_amd as G
    from gcpnet_amd import ops
    from gcpnet_amd.parallel import GradAllReducer, ShardedGraph, sharded_interactions_forward

    torch.cuda.set_device(0)
    layers, ei, x, ins, lw, n = _force_case()
    layers = layers.cuda().eval()
    sg = ShardedGraph(ei, n, rank, world, halo=True).to("cuda")  # (the halo exchange: also under the force term's second gather)
    xg = x.cuda()
    frames = G.localize(xg, sg.edge_index_global)  # frames are constants of the step, built from the INPUT positions
    fr_out = G.localize(xg, sg.out_edge_index_global)
    node_frames = ops.segment_reduce(fr_out.reshape(-1, 9), ops.GatherPlan(sg.out_row_local, sg.n_local), mean=True).reshape(sg.n_local, 3, 3)
    loc = dict(h=sg.local_nodes(ins["h"]), chi=sg.local_nodes(ins["chi"]), e=sg.local_edges(ins["e"]), xi=sg.local_edges(ins["xi"]))
    loc = {k: v.cuda().clone().requires_grad_() for k, v in loc.items()}
    h, chi, pos = loc["h"], loc["chi"], sg.local_nodes(xg)
    for layer in layers:
        (h, chi), pos = sharded_interactions_forward(layer, (h, chi), (loc["e"], loc["xi"]), sg, frames, node_frames, node_pos=pos)
    loss = (h * sg.local_nodes(lw["h"]).cuda()).sum() + (chi * sg.local_nodes(lw["chi"]).cuda()).sum() + (pos * sg.local_nodes(lw["x"]).cuda()).sum()
    loss.backward()
    params = [p for p in layers.parameters()]
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    GradAllReducer(params).all_reduce_sum()
    torch.cuda.synchronize()
    return dict(n0=sg.n0, n1=sg.n1, e0=sg.e0, e1=sg.e1, perm=sg.edge_perm.cpu(), h=h.detach().cpu(), x=pos.detach().cpu(),
                d={k: v.grad.cpu() for k, v in loc.items()}, w={k: p.grad.cpu() for k, p in layers.named_parameters()})


def test_sharded_position_update_with_force_term_matches_unsharded():
    import gcpnet_amd as G

    layers, ei, x, ins, lw, n = _force_case()
    layers = layers.cuda().eval()
    gi = {k: v.cuda().requires_grad_() for k, v in ins.items()}
    fr = G.localize(x.cuda(), ei.cuda())
    h, chi, pos = gi["h"], gi["chi"], x.cuda()
    for layer in layers:
        (h, chi), pos = layer((h, chi), (gi["e"], gi["xi"]), ei.cuda(), fr, node_pos=pos)
    ((h * lw["h"].cuda()).sum() + (chi * lw["chi"].cuda()).sum() + (pos * lw["x"].cuda()).sum()).backward()
    ref = dict(h=h.detach().cpu(), x=pos.detach().cpu(), d={k: v.grad.cpu() for k, v in gi.items()},
               w={k: p.grad.cpu() for k, p in layers.named_parameters()})
    assert float(ref["w"]["0.phi_force_ij.1.weight"].abs().max()) > 0
    out = _run(_sharded_force_job)
    perm = out[0]["perm"]

    def ok(a, b, name):
        tol = 2e-5 * max(1.0, float(b.abs().max()))
        assert torch.allclose(a, b, atol=tol, rtol=1e-4), f"{name}: max diff {(a - b).abs().max():.3e} (scale {b.abs().max():.3e})"

    for r in (0, 1):
        o = out[r]
        sl, es = slice(o["n0"], o["n1"]), slice(o["e0"], o["e1"])
        ok(o["h"], ref["h"][sl], "h"); ok(o["x"], ref["x"][sl], "x")
        ok(o["d"]["h"], ref["d"]["h"][sl], "dh"); ok(o["d"]["chi"], ref["d"]["chi"][sl], "dchi")
        ok(o["d"]["e"], ref["d"]["e"][perm][es], "de"); ok(o["d"]["xi"], ref["d"]["xi"][perm][es], "dxi")
        for k, want in ref["w"].items():
            ok(o["w"][k], want, k)


@pytest.mark.parametrize("extra", [[], ["--shard", "graph"], ["--shard", "graph", "--allgather"]], ids=["batch", "graph-halo", "graph-allgather"])
def test_bench_two_ranks_on_one_gpu_prints_its_line(extra):
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one process per rank), rehearsed on ONE GPU over gloo:
    every rank must reach the closing barrier -- a collective that only rank 0 executes (round 4: the saved-activation measurement
    ran a sharded forward on rank 0 alone) hangs the job.  BENCH_WATCHDOG_S turns a hang into a stack dump and a non-zero exit."""
    import json
    import socket
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = dict(os.environ, BENCH_SHARE_GPU="1", BENCH_DIST_BACKEND="gloo", BENCH_WATCHDOG_S="150")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--nodes", "2000",
           "--no-cpu-baseline", "--no-c5-block", "--no-other-configs"] + extra
    r = subprocess.run(cmd, cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["value"] > 0
    assert line["scaling"] == ("strong" if extra else "weak")
    assert line["rccl"]["world"] == 2 and line["rccl"]["fallback"] is None


def test_bench_gpus_2_without_a_launcher_relaunches_itself():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment (the shape of the driver's 1-GPU command with another N) must not
    die on an assertion: it re-executes itself under torch.distributed.run.  Rehearsed on one GPU over gloo."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(BENCH_SHARE_GPU="1", BENCH_DIST_BACKEND="gloo", BENCH_WATCHDOG_S="150")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--nodes", "2000",
           "--no-cpu-baseline", "--no-c5-block", "--no-other-configs"]
    r = subprocess.run(cmd, cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["rccl"]["world"] == 2 and line["rccl"]["fallback"] is None
