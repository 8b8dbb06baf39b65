#!/usr/bin/env python
"""Benchmark of the GCP message-passing hot path on MI355X (contract: see the task statement / DESIGN.md).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

step      = one forward + scalar loss + backward (to inputs and weights) of a stack of `--layers` GCPInteractions
            layers over one synthetic radius graph resident in HBM (BASELINE.json configs[1]: 10 000 nodes /
            160 000 edges, (s, V) = (128, 16), edge dims (32, 4), GCP2, 8 message GCPs + 2 feed-forward GCPs, post-norm,
            dropout 0); N > 1: every rank holds its own graph of that size (graphs shard by whole graphs, weak
            scaling) and the weight gradients are all-reduced over RCCL each step.
value     = edges processed per second by the whole job = n_gpus * E * layers * steps / max-over-ranks time.
roofline  = the dominant kernel (the GCP2 backward data kernel on edge rows), timed live with HIP events on the
            stream it is launched on: algorithmic FLOPs per launch / average launch time vs the fp32 MFMA peak.
cpu_baseline = the oracle (pure PyTorch on the host cores) on one step of the same stack and inputs.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--nodes", type=int, default=10000)
    ap.add_argument("--neighbors", type=int, default=16)
    ap.add_argument("--sdim", type=int, default=128)
    ap.add_argument("--vdim", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--step-only", action="store_true", help="profiling runs: only the timed steps (no per-kernel timing, no CPU baseline)")
    ap.add_argument("--cpu-layers", type=int, default=1, help="layers of the stack the CPU baseline runs (bounded sample)")
    return ap.parse_args()


def kernel_roofline(G, ops, layer, inputs, frames, edge_index, n_edges, sdim, vdim, iters=20):
    """Times the dominant kernels alone on the current stream (HIP events via torch.cuda.Event, which records on
    torch's current stream -- the stream every gcpnet kernel is launched on)."""
    from gcpnet_amd.synthetic import gcp_macs

    block = layer.interaction.message_fusion[1]  # a residual message GCP: (s,V)->(s,V) on E rows
    g = torch.Generator(device="cuda").manual_seed(0)
    s = torch.randn(n_edges, sdim, device="cuda", generator=g)
    v = torch.randn(n_edges, vdim, 3, device="cuda", generator=g)
    ds = torch.randn(n_edges, sdim, device="cuda", generator=g)
    dv = torch.randn(n_edges, vdim, 3, device="cuda", generator=g)
    spec = ops.Gcp2Spec(si=sdim, vi=vdim, so=sdim, vo=vdim, hidden=block.hidden_dim, use_frames=True, act_s=block.act_s,
                        act_v=block.act_v, slope=1e-2, vmode=block._vmode(), vector_residual=False, e3=False,
                        s_plans=[None], v_plans=[None], residual=True, pack_cache={})
    w = tuple(None if t is None else t.detach() for t in block._weights())
    s.requires_grad_()
    out_s, out_v = ops.gcp2(spec, [s], [v], frames, w)  # forward with saved tensors
    fn = out_s.grad_fn
    saved = fn.saved_tensors
    pack, s_pre, gate = saved[-3], saved[-2], saved[-1]

    def timeit(f):
        for _ in range(3):
            f()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(iters):
            f()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / iters * 1e-3

    with torch.no_grad():
        t_fwd = timeit(lambda: ops.gcp2(spec, [s.detach()], [v], frames, w))
        res = {}

        def bwd():
            res["r"] = ops.gcp2_backward_data(spec, n_edges, [s.detach()], [v], frames, w, pack, s_pre, gate, ds, dv)

        t_bwd = timeit(bwd)
        scr = res["r"][2]
        t_tn = timeit(lambda: ops.gcp2_weight_grads(spec, n_edges, [s.detach()], s_pre, scr))
    flops = 2.0 * n_edges * gcp_macs(sdim, vdim, sdim, vdim)
    return dict(t_fwd=t_fwd, t_bwd=t_bwd, t_tn=t_tn, flops=flops)


def main():
    args = parse()
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local_rank)
    import torch.distributed as dist

    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import gcpnet_amd as G
    from gcpnet_amd import ops
    from gcpnet_amd.parallel import GradAllReducer
    from gcpnet_amd.synthetic import layer_flops, make_inputs

    node_dims, edge_dims = (args.sdim, args.vdim), (32, 4)
    host = make_inputs(args.nodes, args.neighbors, node_dims, edge_dims, seed=rank)  # each rank: its own graph
    n_edges = host["edge_index"].shape[1]
    dev = {k: v.cuda() for k, v in host.items()}
    torch.manual_seed(0)  # identical replicated weights on every rank
    cfg, lcfg = G.default_module_cfg(), G.default_layer_cfg()
    layers = torch.nn.ModuleList(
        G.GCPInteractions(node_dims, edge_dims, cfg=cfg, layer_cfg=lcfg, dropout=0.0) for _ in range(args.layers)).cuda()
    layers.train()
    params = [p for p in layers.parameters()]
    reducer = GradAllReducer(params) if world > 1 else None
    frames = G.localize(dev["x"], dev["edge_index"])
    ins = {k: dev[k].clone().requires_grad_() for k in ("h", "chi", "e", "xi")}

    def step():
        for p in params:
            p.grad = None
        for t in ins.values():
            t.grad = None
        h, chi = ins["h"], ins["chi"]
        for layer in layers:
            h, chi = layer((h, chi), (ins["e"], ins["xi"]), dev["edge_index"], frames)
        loss = h.square().mean() + chi.square().mean()
        loss.backward()
        if reducer is not None:
            reducer.all_reduce_mean()
        return loss

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.isfinite(loss).item(), "non-finite loss"

    if rank == 0 and args.step_only:
        print(json.dumps({"ms_per_step": elapsed / args.steps * 1e3, "steps": args.steps, "warmup": args.warmup}))
    elif rank == 0:
        value = world * n_edges * args.layers * args.steps / elapsed
        fl = layer_flops(args.nodes, n_edges, node_dims, edge_dims)
        kr = kernel_roofline(G, ops, layers[0], ins, frames, dev["edge_index"], n_edges, args.sdim, args.vdim)
        times = {"gcp2_fwd_kernel<4>": kr["t_fwd"], "gcp2_bwd_kernel<4,4>": kr["t_bwd"], "tn_gemm_kernel(+reduce)": kr["t_tn"]}
        dom = max(times, key=times.get)  # each of the three does ~the same algorithmic work: 2 * E * gcp_macs FLOP
        achieved = kr["flops"] / times[dom] / 1e12
        out = {
            "metric": "processed edges/sec (GCP fwd+bwd)", "value": value, "unit": "edges/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"synthetic radius graph r=4.5 K={args.neighbors}: {args.nodes} nodes / {n_edges} edges per GPU, "
                            f"(s,V)=({args.sdim},{args.vdim}), edge dims (32,4), {args.layers} GCPInteractions layers "
                            f"(GCP2, 8 message GCPs, 2 FF GCPs, post-norm, dropout 0), fwd+loss+bwd to inputs and weights",
                "n_nodes": args.nodes, "n_edges": n_edges, "layers": args.layers,
                "parallelism": f"graphs sharded 1 per GPU x{world}, RCCL all-reduce of weight grads" if world > 1 else "single GPU",
            },
            "whole_step": {
                "algorithmic_tflops_per_s": fl["fwd_bwd"] * args.layers * args.steps / elapsed / 1e12,
                "frac_of_fp32_mfma_peak": fl["fwd_bwd"] * args.layers * args.steps / elapsed / 1e12 / PEAK_FP32_MFMA_TFLOPS,
            },
            "roofline": {
                "kernel": dom + " on one residual message GCP (s,V)->(s,V), E rows", "bound": "mfma",
                "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / PEAK_FP32_MFMA_TFLOPS, "traffic": None,
                "avg_launch_ms": times[dom] * 1e3, "flop_per_launch": kr["flops"],
                "all_kernels_ms": {k: v * 1e3 for k, v in times.items()},
                "all_kernels_tflops": {k: kr["flops"] / v / 1e12 for k, v in times.items()},
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(layers, host, args)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(layers, host, args):
    """The oracle on the host cores, on a bounded sample of the workload: one fwd+bwd step of ONE layer of the stack
    (same weights) on a graph built by the same recipe with 1/4 of the nodes (same degree, same feature widths), so
    that the default run stays within a few minutes.  Throughput per edge and layer is what is compared."""
    from gcpnet_amd.synthetic import make_inputs
    from oracle import gcp_oracle as O

    cores = min(os.cpu_count() or 1, 32)  # more threads than this only adds contention on these small ops
    torch.set_num_threads(cores)
    sample = make_inputs(max(args.nodes // 4, 64), args.neighbors, (args.sdim, args.vdim), (32, 4), seed=1234)
    P = {k: v.detach().cpu().clone().requires_grad_() for k, v in layers.state_dict().items()}
    ins = {k: sample[k].clone().requires_grad_() for k in ("h", "chi", "e", "xi")}
    ei = sample["edge_index"]
    fr = O.localize(sample["x"], ei)
    cfg, lcfg = O.default_module_cfg(), O.default_layer_cfg()

    def one():
        h, chi = O.gcp_interactions(P, "0.", ins["h"], ins["chi"], ins["e"], ins["xi"], ei, fr, cfg, lcfg)
        (h.square().mean() + chi.square().mean()).backward()

    one()  # warm-up (allocator, thread pool)
    t0 = time.perf_counter()
    reps = 2
    for _ in range(reps):
        one()
    dt = (time.perf_counter() - t0) / reps
    return {"value": ei.shape[1] / dt, "unit": "edges/s", "cores": cores, "kind": "port",
            "sample": f"fwd+bwd of 1 GCPInteractions layer (same weights) on a {sample['h'].shape[0]}-node / "
                      f"{ei.shape[1]}-edge graph of the same recipe, torch CPU {cores} threads, {dt:.2f} s per step, "
                      f"mean of {reps} after 1 warm-up"}


if __name__ == "__main__":
    main()
