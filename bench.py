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
roofline  = the dominant kernel (forward / backward / weight-gradient GEMM of the 7-block ResGCP message chain on edge
            rows, whichever is slowest), timed live with HIP events on the stream it is launched on: algorithmic FLOPs per
            launch / average launch time vs the fp32 MFMA peak; its algorithmic HBM bytes / time is reported next to it.
aggregate_kernel = the scatter-mean (segmented reduction) kernel against the HBM roofline.
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


def kernel_roofline(G, ops, layer, frames, n_edges, sdim, vdim, iters=20):
    """Times the three kernels that carry the step -- forward, backward and weight-gradient GEMM of the 7-block ResGCP message
    chain of one layer, on E edge rows -- each alone, with HIP events recorded on the stream the kernels are launched on
    (torch's current stream).  All three do the same algorithmic work per launch: 7 * 2 * E * gcp_macs FLOP."""
    from gcpnet_amd.synthetic import gcp_macs

    blocks = list(layer.interaction.message_fusion[1:])  # the residual message GCPs: (s,V)->(s,V) on E rows
    n = len(blocks)
    g = torch.Generator(device="cuda").manual_seed(0)
    s = torch.randn(n_edges, sdim, device="cuda", generator=g).requires_grad_()
    v = torch.randn(n_edges, vdim, 3, device="cuda", generator=g).requires_grad_()
    ds = torch.randn(n_edges, sdim, device="cuda", generator=g)
    dv = torch.randn(n_edges, vdim, 3, device="cuda", generator=g)
    specs = [b.make_spec([None], [None], residual=True) for b in blocks]
    ws = [tuple(None if t is None else t.detach().requires_grad_() for t in b._weights()) for b in blocks]

    def timeit(f):
        """Median launch duration, one HIP event pair per launch (a launch that has to grow the caching allocator's pool --
        ~1.4 GB of saved activations per forward launch -- takes milliseconds on the host and must not enter the figure)."""
        for _ in range(5):
            f()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
        torch.cuda.synchronize()
        for a, b in ev:
            a.record()
            f()
            b.record()
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in ev)
        return ts[len(ts) // 2] * 1e-3

    keep = {}

    def fwd():
        keep.pop("out", None)  # (the previous launch's saved activations go back to the allocator first)
        keep["out"] = ops.gcp2_chain(specs, s, v, frames, ws)  # training mode: s_pre / gates / states are saved

    t_fwd = timeit(fwd)
    s0, v0, ws_, packs, outs = keep["out"][0].grad_fn.state
    ins = [(s0, v0) if k == 0 else (outs[k - 1][0], outs[k - 1][1]) for k in range(n)]
    with torch.no_grad():
        def bwd():
            keep["bwd"] = ops.gcp2_chain_backward_data(specs, n_edges, ins, outs, frames, ws_, packs, ds, dv, [True] * n)

        t_bwd = timeit(bwd)
        assert keep["bwd"] is not None, "the chain backward kernel does not cover this shape"
        scrs = keep["bwd"][2]

        def tn():
            ops.run_weight_grad_jobs([ops._WeightGradJob(specs[k], n_edges, [ins[k][0]], outs[k][2], scrs[k]) for k in range(n)])

        t_tn = timeit(tn)
    flops = n * 2.0 * n_edges * gcp_macs(sdim, vdim, sdim, vdim)
    H = blocks[0].hidden_dim
    row_s, row_v = 4 * sdim, 12 * vdim
    ext, gate = 4 * ((H + 9 + 3) // 4 * 4), 4 * vdim
    # algorithmic HBM bytes per launch (every tensor a kernel must read or write once, per edge row)
    bytes_fwd = n_edges * ((row_s + row_v + 36) + n * (row_s + row_v + row_s + gate))
    bytes_bwd = n_edges * (2 * (row_s + row_v) + 36 + n * ((row_s + row_v + gate) + (row_s + gate + ext)))
    bytes_tn = n_edges * n * ((row_s + row_s + ext) + (gate + row_s))
    return dict(t_fwd=t_fwd, t_bwd=t_bwd, t_tn=t_tn, flops=flops, n_blocks=n,
                bytes=dict(fwd=bytes_fwd, bwd=bytes_bwd, tn=bytes_tn))


def _pmc_file():
    try:
        with open(os.path.join(ROOT, "profiles", "r01_e_traffic.json")) as f:
            return json.load(f)
    except OSError:
        return {}


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/README.md); None when not collected."""
    return _pmc_file().get(kernel)


def pmc_mfma_busy(kernel):
    """Fraction of SIMD cycles with the MFMA pipe busy, from the committed SQ_VALU_MFMA_BUSY_CYCLES pass; None when not collected."""
    return _pmc_file().get("mfma_busy_frac", {}).get(kernel)


def aggregate_roofline(G, ops, n_nodes, n_edges, width, label, iters=30):
    """The gather / aggregate kernel (scatter-mean of the edge messages [E, s + 3V] onto their target nodes, gcpnet.py:946)
    against the HBM roofline: algorithmic bytes = (E + N) * width * 4 (+ the CSR pointers)."""
    from gcpnet_amd.synthetic import make_inputs
    g = torch.Generator(device="cuda").manual_seed(1)
    col = torch.sort(torch.randint(0, n_nodes, (n_edges,), device="cuda", generator=g)).values
    plan = ops.GatherPlan(col, n_nodes)
    msg = torch.randn(n_edges, width, device="cuda", generator=g)
    for _ in range(3):
        ops.segment_reduce(msg, plan, True)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        ops.segment_reduce(msg, plan, True)
    b.record()
    torch.cuda.synchronize()
    t = a.elapsed_time(b) / iters * 1e-3
    nbytes = 4.0 * width * (n_edges + n_nodes) + 4.0 * (n_nodes + 1)
    return {"kernel": "segment_reduce_kernel<mean>", "workload": label, "bound": "hbm", "achieved": nbytes / t / 1e9,
            "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": nbytes / t / 1e9 / PEAK_HBM_GBS, "avg_launch_ms": t * 1e3,
            "bytes_per_launch": nbytes}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # test hooks (a 1-GPU box can rehearse the N > 1 code path): BENCH_SHARE_GPU=1 puts every rank on cuda:0,
    # BENCH_DIST_BACKEND=gloo replaces RCCL (which refuses two ranks on one device)
    if os.environ.get("BENCH_SHARE_GPU") == "1":
        local_rank = 0
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    import torch.distributed as dist

    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

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
        kr = kernel_roofline(G, ops, layers[0], frames, n_edges, args.sdim, args.vdim)
        times = {"gcp2_chain_fwd_kernel": kr["t_fwd"], "gcp2_chain_bwd_kernel": kr["t_bwd"],
                 "tn_gemm_dma_kernel(+reduce)": kr["t_tn"]}
        kbytes = {"gcp2_chain_fwd_kernel": kr["bytes"]["fwd"], "gcp2_chain_bwd_kernel": kr["bytes"]["bwd"],
                  "tn_gemm_dma_kernel(+reduce)": kr["bytes"]["tn"]}
        dom = max(times, key=times.get)  # each of the three does the same algorithmic work: n_blocks * 2 * E * gcp_macs FLOP
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
                "kernel": f"{dom} on the {kr['n_blocks']}-block residual message chain (s,V)->(s,V) of one layer, E rows",
                "bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / PEAK_FP32_MFMA_TFLOPS, "traffic": pmc_traffic(dom), "mfma_busy_frac_pmc": pmc_mfma_busy(dom),
                "avg_launch_ms": times[dom] * 1e3, "flop_per_launch": kr["flops"],
                "algorithmic_hbm_gbs": kbytes[dom] / times[dom] / 1e9,
                "algorithmic_hbm_frac": kbytes[dom] / times[dom] / 1e9 / PEAK_HBM_GBS,
                "all_kernels_ms": {k: v * 1e3 for k, v in times.items()},
                "all_kernels_tflops": {k: kr["flops"] / v / 1e12 for k, v in times.items()},
                "all_kernels_algorithmic_hbm_gbs": {k: kbytes[k] / v / 1e9 for k, v in times.items()},
            },
            # the gather / aggregate kernel against the HBM roofline, at this run's size and at BASELINE configs[4]'s
            # (100k nodes / 1M edges, (256,32): 1.5 GB per launch, far beyond the 256 MB Infinity Cache)
            "aggregate_kernel": aggregate_roofline(G, ops, args.nodes, n_edges, args.sdim + 3 * args.vdim,
                                                   f"{args.nodes} nodes / {n_edges} edges, width s+3V = {args.sdim + 3 * args.vdim}"),
            "aggregate_kernel_c5": aggregate_roofline(G, ops, 100000, 1000000, 256 + 96,
                                                      "100000 nodes / 1000000 edges, width s+3V = 352", iters=10),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(layers, host, args)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    # orderly teardown: nothing in flight and no module-level stream objects left for interpreter shutdown to destroy
    # after the HIP runtime has gone
    torch.cuda.synchronize()
    ops._side_pending.clear()
    ops._side_streams.clear()
    sys.stdout.flush()


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
