#!/usr/bin/env python
"""Benchmark of the GCP message-passing hot path on MI355X (contract: see the task statement / DESIGN.md).

    python bench.py [--config c2] --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W [--shard graph]

--config  the BASELINE.json configuration that is timed (named in config.workload):
            c2 (default) configs[1]: synthetic radius graph, 10 000 nodes / ~160 000 edges, (s,V) = (128,16), 4 GCPInteractions layers
            c5           configs[4]: 100 000 nodes / 1 000 000 edges, (256,32), 4 layers (fits one GPU: ~75 GB of saved activations)
            c1 / c4      NMS model step() on 100 fully-connected 5-body / 20-body graphs ((64,16), 4 layers, position updates)
            c3           LBA model step() on 16 radius graphs ((100,16), 8 layers, readout head)
step      = one forward + scalar loss + backward (to inputs and weights): of the layer stack with a random linear functional of the
            outputs as the loss (c2, c5), or the model's own step() = forward + MSELoss + backward (c1, c3, c4).  Inputs are
            resident in HBM.  N > 1: every rank holds its own graph / batch of that size (graphs shard whole, weak scaling) and
            the weight gradients are all-reduced over RCCL each step; `--shard graph` (c2, c5) instead splits ONE graph by
            target-node ranges (strong scaling, gcpnet_amd.parallel.ShardedGraph).
value     = edges processed per second by the whole job = total edges * layers * steps / max-over-ranks wall time of the K steps.
roofline  = the dominant kernel of the step (forward / backward / weight-gradient GEMM of the 7-block ResGCP message chain on the
            edge rows, whichever takes longest), timed live with HIP events on the stream it is launched on: algorithmic FLOPs per
            launch / median launch time vs the fp32 MFMA peak; its algorithmic HBM bytes / time is reported next to it.
aggregate_kernel = the scatter-mean (segmented reduction) kernel against the HBM roofline.
cpu_baseline = the oracle (pure PyTorch on the host cores) on the same stack, weights and inputs (full c2 workload: 1 warm-up + 3
            timed steps; larger configurations: a bounded sample, stated), plus `parity_check`: the GPU outputs and input gradients
            of that same step against the oracle's.
other_configs = (default run, one GPU) the model-step configurations c1 / c4 / c3, eager and as a hipGraph replay: median ms/step
            and edges/s each; c5_single_gpu = 10 steps at configs[4] size.  Every BASELINE configuration that fits one GPU is thus in
            the one JSON line.  aggregate_product_path = the aggregation as the product launches it (two reductions).
roofline.peak_f16x3_equiv = the ceiling of the 16-bit matrix pipe for fp32 products computed as three fp16 MFMAs (2 500 / 3 TFLOP/s: the
            arithmetic of the big products since round 6, csrc/gcp_f16x2.h); peak_bf16x6_equiv = the same for the six-product bf16 form
            of rounds 3 - 5 (2 500 / 6), kept for comparison with the earlier lines.
--dry-run-world N = no timing: the N-rank bookkeeping of `--shard graph` on one GPU (nodes / edges / halo per rank, MB per layer).
"""
import argparse
import gc
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
PEAK_HBM_GBS = 8000.0
# fp32 products that run on the bf16 matrix pipe as three-term splits cost six bf16 MFMAs each (csrc/gcp_bf16x3.h): the dense bf16
# peak (2 500 TFLOP/s) / 6 is the ceiling of THAT pipe in fp32-equivalent FLOPs -- reported next to the fp32 MFMA peak, which stays
# `roofline.peak` (the kernels mix both forms)
PEAK_BF16X6_EQUIV_TFLOPS = 2500.0 / 6.0
# round 6: two fp16 terms per operand, three MFMAs per product block (csrc/gcp_f16x2.h) -- the pipe's ceiling for that form
PEAK_F16X3_EQUIV_TFLOPS = 2500.0 / 3.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", choices=["c1", "c2", "c3", "c4", "c5"], default="c2")
    ap.add_argument("--shard", choices=["batch", "graph"], default="batch",
                    help="N > 1: 'batch' = one graph per rank (weak scaling); 'graph' = one graph split by node ranges (strong scaling)")
    ap.add_argument("--halo", dest="halo", action="store_true", default=True,
                    help="--shard graph (the default there): nodes renumbered along a Morton curve, only the halo rows exchanged "
                         "(all-to-all): 30 - 45x fewer bytes than the all-gather at configs[4] size (profiles/r03_c5_dry_run_world8.json)")
    ap.add_argument("--allgather", dest="halo", action="store_false",
                    help="--shard graph: all-gather the whole node-feature table per layer (reduce-scatter of its gradient) instead")
    ap.add_argument("--layers", type=int, default=None)
    ap.add_argument("--nodes", type=int, default=None)
    ap.add_argument("--neighbors", type=int, default=None)
    ap.add_argument("--sdim", type=int, default=None)
    ap.add_argument("--vdim", type=int, default=None)
    ap.add_argument("--no-c5-block", action="store_true", help="skip the configs[4]-size measurement of the default run")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the c1 / c3 / c4 blocks of the default run")
    ap.add_argument("--dry-run-world", type=int, default=0, metavar="N",
                    help="no timing: execute the N-rank index bookkeeping of `--shard graph` for this configuration on one GPU and print "
                         "per-rank node / edge counts and the bytes each rank exchanges per layer")
    ap.add_argument("--hip-graph", action="store_true", help="capture the step in a hipGraph and time replays (launch-bound configs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--step-only", action="store_true", help="profiling runs: only the timed steps (no per-kernel timing, no CPU baseline)")
    ap.add_argument("--cpu-layers", type=int, default=1, help="layers of the stack the CPU baseline runs (bounded sample)")
    args = ap.parse_args()
    preset = {"c2": (10000, 16, 128, 16, 4), "c5": (100000, 10, 256, 32, 4)}.get(args.config, (10000, 16, 128, 16, 4))
    for name, val in zip(("nodes", "neighbors", "sdim", "vdim", "layers"), preset):
        if getattr(args, name) is None:
            setattr(args, name, val)
    return args


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
    flops = n * 2.0 * n_edges * gcp_macs(sdim, vdim, sdim, vdim)
    H = blocks[0].hidden_dim
    row_s, row_v = 4 * sdim, 12 * vdim
    ext, gate = 4 * ((H + 9 + 3) // 4 * 4), 4 * vdim
    # algorithmic HBM bytes per launch (every tensor a kernel must read or write once, per edge row)
    # (sign masks, round 6: the forward writes one bit per s_pre element beside s_pre, the backward reads those bits INSTEAD of s_pre
    # when the chain's activations are piecewise linear)
    signed = bool(ops.CHAIN_SIGN_MASKS and all(sp.act_s in ops._PWL_ACTS and sp.act_v in ops._PWL_ACTS for sp in specs)
                  and sdim % 64 == 0 and sdim <= 128)
    bits = row_s // 32 if signed else 0
    # (round 6: with the sign masks and an identity gate activation nothing reads s_pre any more: the forward does not store it and the
    # gate weight gradients come from dgate^T [s | ext], the scalar_out gradient's own second operand -- ops.CHAIN_SKIP_S_PRE)
    no_pre = bool(signed and ops.CHAIN_SKIP_S_PRE and ops.GATE_GRADS_FROM_INPUTS and all(sp.act_v is None for sp in specs))
    bytes_fwd = n_edges * ((row_s + row_v + 36) + n * (row_s + row_v + (0 if no_pre else row_s) + gate + bits))
    bytes_bwd = n_edges * (2 * (row_s + row_v) + 36 + n * (((bits if signed else row_s) + row_v + gate) + (row_s + gate + ext)))
    bytes_tn = n_edges * n * ((row_s + row_s + ext) + (gate + (row_s + ext if no_pre else row_s)))
    fwd_name = "gcp_wg_fwd_kernel" if ops.WG_STATS["fwd_chain"] > 0 else "gcp2_chain_fwd_kernel"
    times, kbytes, kflops = {fwd_name: t_fwd}, {fwd_name: bytes_fwd}, {fwd_name: flops}
    s0, v0, ws_, packs, outs = keep["out"][0].grad_fn.state
    ins = [(s0, v0) if k == 0 else (outs[k - 1][0], outs[k - 1][1]) for k in range(n)]
    with torch.no_grad():
        probe = ops.gcp2_chain_backward_data(specs, n_edges, ins, outs, frames, ws_, packs, ds, dv, [True] * n) \
            if sdim <= 128 else None
    if probe is not None:  # (s <= 128: one launch of the wave-per-tile chain kernel + the weight-gradient GEMMs)
        with torch.no_grad():
            def bwd():
                keep["bwd"] = ops.gcp2_chain_backward_data(specs, n_edges, ins, outs, frames, ws_, packs, ds, dv, [True] * n)

            times["gcp2_chain_bwd_kernel"] = timeit(bwd)
            scrs = keep["bwd"][2]

            def tn():
                ops.run_weight_grad_jobs([ops._WeightGradJob(specs[k], n_edges, [ins[k][0]], outs[k][2], scrs[k], w=ws_[k]) for k in range(n)])

            times["tn_pipe_kernel(+reduce)"] = timeit(tn)
        kbytes.update({"gcp2_chain_bwd_kernel": bytes_bwd, "tn_pipe_kernel(+reduce)": bytes_tn})
        kflops.update({"gcp2_chain_bwd_kernel": flops, "tn_pipe_kernel(+reduce)": flops})
    else:  # wider chains: block by block through the workgroup backward kernel, weight-gradient GEMMs included (2x the FLOPs)
        del probe

        def bwd_all():
            fwd()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            torch.autograd.backward(list(keep["out"]), [ds, dv])
            b.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b) * 1e-3

        ts = sorted(bwd_all() for _ in range(max(3, iters // 4)))
        name = f"gcp_wg_bwd_kernel x{n} + weight-gradient GEMMs"
        times[name], kbytes[name], kflops[name] = ts[len(ts) // 2], bytes_bwd + bytes_tn, 2.0 * flops
    keep.clear()
    return dict(times=times, bytes=kbytes, flops=kflops, n_blocks=n)


PMC_FILE = "profiles/r06_traffic.json"  # HBM bytes / MFMA-busy share per launch from committed rocprofv3 --pmc passes


def _pmc_file():
    try:
        with open(os.path.join(ROOT, PMC_FILE)) as f:
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


def gather_roofline(G, ops, n_nodes, n_edges, width, label, iters=30):
    """The gather kernel (rows of a node-level table [N, width] copied to their edges: the message assembly of gcpnet.py:907-917 in its
    unfused form, and the adjoint of the aggregation) against the HBM roofline: algorithmic bytes = (N + E) * width * 4 + the index."""
    g = torch.Generator(device="cuda").manual_seed(4)
    col = torch.sort(torch.randint(0, n_nodes, (n_edges,), device="cuda", generator=g)).values
    plan = ops.GatherPlan(col, n_nodes)
    tab = torch.randn(n_nodes, width, device="cuda", generator=g)
    for _ in range(3):
        ops._gather_rows_raw(tab, plan, None)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        ops._gather_rows_raw(tab, plan, None)
    b.record()
    torch.cuda.synchronize()
    t = a.elapsed_time(b) / iters * 1e-3
    nbytes = 4.0 * width * (n_edges + n_nodes) + 4.0 * n_edges
    return {"kernel": "gather_rows_kernel", "workload": label, "bound": "hbm", "achieved": nbytes / t / 1e9, "peak": PEAK_HBM_GBS,
            "unit": "GB/s", "frac": nbytes / t / 1e9 / PEAK_HBM_GBS, "avg_launch_ms": t * 1e3, "bytes_per_launch": nbytes}


def count_device_kernels(step):
    """Launches of one step by family, from torch's profiler (one extra step, outside every timed region): how many of them are ATen
    kernels (autograd's gradient sums, fills, the index preprocessing) as opposed to this library's."""
    try:
        from torch.profiler import ProfilerActivity, profile

        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            step()
            torch.cuda.synchronize()
        total, aten, names = 0, 0, {}
        for ev in prof.key_averages():
            if getattr(ev, "device_type", None) is None or "cuda" not in str(ev.device_type).lower():
                continue
            n = int(ev.count)
            total += n
            nm = ev.key
            if nm.startswith(("at::", "void at::", "__amd_rocclr", "rocprim", "void rocprim")) or "at::native" in nm:
                aten += n
                short = nm.split("<")[0][-60:]
                names[short] = names.get(short, 0) + n
        return {"launches_per_step": total, "aten_or_runtime_kernels_per_step": aten, "aten_by_name": dict(sorted(names.items(), key=lambda kv: -kv[1])[:6])}
    except Exception as exc:  # noqa: BLE001 -- a diagnostic, never worth the line
        return {"error": f"{type(exc).__name__}: {exc}"[:200]}


def aggregate_product_path(G, ops, n_nodes, n_edges, sdim, vdim, iters=30):
    """The aggregation as GCPMessagePassing.forward launches it (gcpnet_amd/gcpnet.py: the chain kernel leaves the scalar and the
    vector part of the messages in two tensors, so the scatter-mean is TWO segmented reductions, [E, s] and [E, 3V]) against the
    HBM roofline, next to the single [E, s + 3V] launch of `aggregate_kernel`."""
    g = torch.Generator(device="cuda").manual_seed(2)
    col = torch.sort(torch.randint(0, n_nodes, (n_edges,), device="cuda", generator=g)).values
    plan = ops.GatherPlan(col, n_nodes)
    ms = torch.randn(n_edges, sdim, device="cuda", generator=g)
    mv = torch.randn(n_edges, 3 * vdim, device="cuda", generator=g)

    def both():
        ops.segment_reduce(ms, plan, True)
        ops.segment_reduce(mv, plan, True)

    for _ in range(3):
        both()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        both()
    b.record()
    torch.cuda.synchronize()
    t = a.elapsed_time(b) / iters * 1e-3
    nbytes = 4.0 * (sdim + 3 * vdim) * (n_edges + n_nodes) + 2 * 4.0 * (n_nodes + 1)
    return {"kernel": "segment_reduce_kernel<mean> x2 (scalars [E,s], vectors [E,3V]): the launches of the product path",
            "achieved": nbytes / t / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": nbytes / t / 1e9 / PEAK_HBM_GBS,
            "ms_for_both_launches": t * 1e3, "bytes": nbytes}


def model_cpu_baseline(wl, kind, batch, n_layers):
    """cpu_baseline of a model configuration (c1 / c4 / c3): the oracle's forward + loss + backward of the SAME model (state_dict of
    the timed one) on the SAME batch, on the host cores: 1 warm-up + 3 timed iterations; no optimizer step on the CPU
    side (the fused Adam update is a negligible share of the GPU step and has no oracle counterpart to time)."""
    from oracle import gcp_oracle as O

    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    P = {k: v.detach().cpu().clone().requires_grad_() for k, v in wl["model"].state_dict().items() if v.is_floating_point()}
    b = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in batch.items()}
    fwd = O.nms_forward if kind == "nms" else O.lba_forward
    cfg, lcfg = O.default_module_cfg(), O.default_layer_cfg()

    def one():
        for t in P.values():
            t.grad = None
        out = fwd(P, b, cfg, lcfg, n_layers)
        pred = out["x"] if kind == "nms" else out["pred"].reshape(-1)
        loss = torch.nn.functional.mse_loss(pred, b["label"].reshape(pred.shape))
        loss.backward()
        return float(loss.detach())

    one()
    reps = 3 if b["edge_index"].shape[1] * n_layers < 500000 else 1  # (the LBA batch: ~15 s per iteration)
    t0 = time.perf_counter()
    for _ in range(reps):
        lv = one()
    dt = (time.perf_counter() - t0) / reps
    n_e = b["edge_index"].shape[1]
    return {"value": n_e * n_layers / dt, "unit": "edges/s", "cores": cores, "kind": "port",
            "sample": f"the full timed batch: forward + MSE loss + backward of the {kind.upper()} model (oracle, same weights and batch; no "
                      f"optimizer step), {n_e} edges x {n_layers} layers, torch CPU {cores} threads, {dt:.3f} s per step, mean of {reps} after 1 warm-up",
            "loss": lv}


def other_configs_block(G, ops, args):
    """The BASELINE configurations that are model steps -- c1 (NMS 5-body), c4 (NMS 20-body), c3 (LBA) -- on this one GPU, eager
    and as a hipGraph replay of the captured step (these are launch-bound: ~700 launches for 2 000 .. 250 000 edges): median
    ms/step of HIP-event pairs, edges/s = edges x layers / step.  A step here is the reference's whole training step: forward +
    loss + backward + Adam update, all of it inside the captured graph."""
    import copy
    from gcpnet_amd.graphs import GraphedStep

    out = {}
    for cfg in ("c1", "c4", "c3"):
        a = copy.copy(args)
        a.config = cfg
        wl = build_model_workload(a, 0, 1, G, ops)
        steps = 20
        _, med = timed_steps(wl["step"], steps, 5, 1, None)
        rec = {"workload": wl["label"], "n_edges": wl["n_edges"], "layers": wl["n_layers"], "steps": steps, "warmup": 5,
               "eager_ms_per_step_median": med, "eager_edges_per_s": wl["n_edges"] * wl["n_layers"] / (med * 1e-3)}
        if not args.no_cpu_baseline:
            try:
                rec["cpu_baseline"] = model_cpu_baseline(wl, wl["kind"], wl["batch"], wl["n_layers"])
            except Exception as exc:  # noqa: BLE001 -- must not cost the run its headline line
                rec["cpu_baseline_error"] = f"{type(exc).__name__}: {exc}"[:300]
        try:
            graphed = GraphedStep(wl["fwd_bwd"], warmup=3, optimizer=wl["optimizer"])  # forward + backward + Adam in ONE graph
            _, gmed = timed_steps(graphed, steps, 3, 1, None)
            rec.update({"hipgraph_ms_per_step_median": gmed, "hipgraph_edges_per_s": wl["n_edges"] * wl["n_layers"] / (gmed * 1e-3)})
            del graphed
        except Exception as exc:  # (a capture failure must not cost the run its headline line)
            rec["hipgraph_error"] = f"{type(exc).__name__}: {exc}"[:300]
        if cfg == "c3":  # the LBA step's dominant kernels: the (100,16) residual message chain of one layer on the batch's edge rows
            try:
                layer = wl["model"].interaction_layers[0]
                n_e = wl["n_edges"]
                fr = torch.randn(n_e, 3, 3, device="cuda", generator=torch.Generator(device="cuda").manual_seed(2))
                kr = kernel_roofline(G, ops, layer, fr, n_e, 100, 16, iters=10)
                dom = min(kr["times"], key=lambda k_: kr["flops"][k_] / kr["times"][k_])
                ach = kr["flops"][dom] / kr["times"][dom] / 1e12
                rec["roofline"] = {"kernel": f"{dom} on the {kr['n_blocks']}-block residual message chain (100,16)->(100,16) of one layer, {n_e} rows",
                                   "bound": "mfma", "achieved": ach, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_FP32_MFMA_TFLOPS,
                                   "median_launch_ms": kr["times"][dom] * 1e3, "flop_per_launch": kr["flops"][dom],
                                   "note": "FLOPs of the unpadded (100,16) block (SURVEY.md 8d gcp()); the kernels run the width padded to 128",
                                   "all_kernels_ms": {k_: v_ * 1e3 for k_, v_ in kr["times"].items()},
                                   "all_kernels_tflops": {k_: kr["flops"][k_] / v_ / 1e12 for k_, v_ in kr["times"].items()},
                                   "traffic": None}
            except Exception as exc:  # noqa: BLE001 -- must not cost the run its headline line
                rec["roofline_error"] = f"{type(exc).__name__}: {exc}"[:300]
        out[cfg] = rec
        del wl
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    return out


def dry_run_world(args, G, ops):
    """`--dry-run-world N`: the index bookkeeping of the one-graph sharding (gcpnet_amd.parallel.ShardedGraph) for N ranks, executed
    here rank by rank on one GPU, with the volume each rank would exchange per GCPInteractions layer: all-gather of node features
    forward + reduce-scatter of their gradients backward (full table or only the halo: the source nodes its in-edges actually
    reference), and the flat-bucket all-reduce of the weight gradients once per step."""
    from gcpnet_amd.parallel import ShardedGraph
    from gcpnet_amd.synthetic import make_inputs

    n = args.dry_run_world
    host = make_inputs(args.nodes, args.neighbors, (args.sdim, args.vdim), (32, 4), seed=0)
    ei = host["edge_index"].cuda()
    width = args.sdim + 3 * args.vdim
    layer = G.GCPInteractions((args.sdim, args.vdim), (32, 4), cfg=G.default_module_cfg(), layer_cfg=G.default_layer_cfg(), dropout=0.0)
    n_params = sum(p.numel() for p in layer.parameters())
    # the same graph with its nodes renumbered along a Morton curve (gcpnet_amd.parallel.spatial_order): what a halo exchange would
    # have to move if the ids were spatially sorted, as they are for real structures
    from gcpnet_amd.parallel import spatial_order
    perm = spatial_order(host["x"].cuda())
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(perm.numel(), device=perm.device)
    ei_sorted = inv[ei]
    ei_sorted = ei_sorted[:, torch.argsort(ei_sorted[1], stable=True)]
    ranks = []
    for r in range(n):
        sg = ShardedGraph(ei, args.nodes, r, n)
        halo = sg.halo_nodes()
        halo_sorted = ShardedGraph(ei_sorted, args.nodes, r, n).halo_nodes()
        ranks.append({"rank": r, "nodes": sg.n_local, "in_edges": sg.e1 - sg.e0, "out_edges": int(sg.out_row_local.shape[0]),
                      "halo_nodes": int(halo.numel()),
                      "allgather_recv_MB_per_layer": (args.nodes - sg.n_local) * width * 4 / 1e6,
                      "halo_recv_MB_per_layer": int(halo.numel()) * width * 4 / 1e6,
                      "halo_nodes_morton_sorted": int(halo_sorted.numel()),
                      "halo_recv_MB_per_layer_morton_sorted": int(halo_sorted.numel()) * width * 4 / 1e6,
                      "send_MB_per_layer_full_table": sg.n_local * width * 4 * (n - 1) / 1e6})
    edges = [x["in_edges"] for x in ranks]
    out = {"dry_run_world": n, "config": args.config, "n_nodes": args.nodes, "n_edges": int(ei.shape[1]), "row_bytes": width * 4,
           "ranks": ranks, "edge_imbalance_max_over_mean": max(edges) / (sum(edges) / n),
           "weight_grad_allreduce_MB_per_step": n_params * 4 * args.layers / 1e6,
           "note": "xGMI: 7 links x ~153 GB/s per GPU, point to point; the all-gather of a layer moves allgather_recv_MB into each rank "
                   "over its 7 links in parallel; forward + backward (reduce-scatter) = twice that per layer"}
    print(json.dumps(out))


def build_layer_workload(args, rank, world, G, ops):
    """c2 / c5: a stack of GCPInteractions layers on one synthetic radius graph per rank (or, with --shard graph, on this rank's
    target-node range of ONE graph).  Returns a dict with the step function and bookkeeping."""
    from gcpnet_amd.parallel import GradAllReducer, ShardedGraph, sharded_interactions_forward
    from gcpnet_amd.synthetic import make_inputs

    node_dims, edge_dims = (args.sdim, args.vdim), (32, 4)
    sharded = world > 1 and args.shard == "graph"
    host = make_inputs(args.nodes, args.neighbors, node_dims, edge_dims, seed=0 if sharded else rank)
    if sharded and args.halo:
        from gcpnet_amd.parallel import spatial_order
        from gcpnet_amd.synthetic import reorder_nodes

        host = reorder_nodes(host, spatial_order(host["x"]))
    n_edges_global = host["edge_index"].shape[1]
    torch.manual_seed(0)  # identical replicated weights on every rank
    cfg, lcfg = G.default_module_cfg(), G.default_layer_cfg()
    layers = torch.nn.ModuleList(
        G.GCPInteractions(node_dims, edge_dims, cfg=cfg, layer_cfg=lcfg, dropout=0.0) for _ in range(args.layers)).cuda()
    layers.train()
    params = [p for p in layers.parameters()]
    reducer = GradAllReducer(params) if world > 1 else None
    g = torch.Generator().manual_seed(1)  # the loss: a random linear functional of the outputs (a squared loss behind the
    lw = dict(h=torch.randn(args.nodes, node_dims[0], generator=g),  # final GCPLayerNorm has a vanishing gradient)
              chi=torch.randn(args.nodes, node_dims[1], 3, generator=g))
    if sharded:
        sg = ShardedGraph(host["edge_index"], args.nodes, rank, world, halo=args.halo)
        x = host["x"].cuda()
        sg.to("cuda")
        frames = G.localize(x, sg.edge_index_global)
        fr_out = G.localize(x, sg.out_edge_index_global)
        node_frames = ops.segment_reduce(fr_out.reshape(-1, 9), ops.GatherPlan(sg.out_row_local, sg.n_local),
                                         mean=True).reshape(sg.n_local, 3, 3)
        ins = {"h": sg.local_nodes(host["h"]), "chi": sg.local_nodes(host["chi"]), "e": sg.local_edges(host["e"]),
               "xi": sg.local_edges(host["xi"])}
        ins = {k: v.cuda().clone().requires_grad_() for k, v in ins.items()}
        lw = {k: sg.local_nodes(v).cuda() for k, v in lw.items()}
        n_edges_rank = sg.e1 - sg.e0
    else:
        dev = {k: v.cuda() for k, v in host.items()}
        frames = G.localize(dev["x"], dev["edge_index"])
        ins = {k: dev[k].clone().requires_grad_() for k in ("h", "chi", "e", "xi")}
        lw = {k: v.cuda() for k, v in lw.items()}
        n_edges_rank = n_edges_global

    def forward():
        h, chi = ins["h"], ins["chi"]
        for layer in layers:
            if sharded:
                h, chi = sharded_interactions_forward(layer, (h, chi), (ins["e"], ins["xi"]), sg, frames, node_frames)
            else:
                h, chi = layer((h, chi), (ins["e"], ins["xi"]), dev["edge_index"], frames)
        return h, chi

    def step():
        for p in params:
            p.grad = None
        for t in ins.values():
            t.grad = None
        h, chi = forward()
        loss = (h * lw["h"]).sum() + (chi * lw["chi"]).sum()
        loss.backward()
        if reducer is not None:
            reducer.all_reduce_sum() if sharded else reducer.all_reduce_mean()
        return loss

    total_edges = n_edges_global if sharded else world * n_edges_global
    label = (f"synthetic radius graph r=4.5 K={args.neighbors}: {args.nodes} nodes / {n_edges_global} edges "
             f"{'in total, split by target-node ranges over the GPUs' if sharded else 'per GPU'}, (s,V)=({args.sdim},{args.vdim}), "
             f"edge dims (32,4), {args.layers} GCPInteractions layers (GCP2, 8 message GCPs, 2 FF GCPs, post-norm, dropout 0), "
             f"fwd + linear-functional loss + bwd to inputs and weights")
    return dict(step=step, forward=forward, layers=layers, ins=ins, host=host, frames=frames, lw=lw, n_edges=n_edges_global,
                total_edges=total_edges, n_layers=args.layers, label=label, sharded=sharded, n_edges_rank=n_edges_rank,
                scaling="strong" if sharded else "weak")


def build_model_workload(args, rank, world, G, ops):
    """c1 / c4: NMS model step() on 100 fully-connected n-body graphs; c3: LBA model step() on 16 radius graphs."""
    from gcpnet_amd.parallel import GradAllReducer
    from gcpnet_amd.synthetic import model_batch

    torch.manual_seed(0)
    batch, model_cfg, kind, label = model_batch(args.config, seed=rank)
    Model = G.GCPNetNMS if kind == "nms" else G.GCPNetLBA
    model = Model(model_cfg=model_cfg, module_cfg=G.default_module_cfg(), layer_cfg=G.default_layer_cfg()).cuda().train()
    params = [p for p in model.parameters()]
    reducer = GradAllReducer(params) if world > 1 else None
    dev = {k: v.cuda() for k, v in batch.items()}
    n_edges = batch["edge_index"].shape[1]
    # the reference's training step ends in Adam (configs/model/gcpnet_nms.yaml:8-12, gcpnet_lba.yaml): one fused launch, step count
    # on the device, so that the whole step -- optimizer included -- can be captured
    opt = G.FusedAdam(params, lr=1e-4, capturable=True)

    def fwd_bwd():
        for p in params:
            p.grad = None
        b = G.Batch(**dev)
        loss, _, _ = model.step(b)
        loss.backward()
        if reducer is not None:
            reducer.all_reduce_mean()
        return loss

    def step():
        loss = fwd_bwd()
        opt.step()
        return loss

    return dict(step=step, fwd_bwd=fwd_bwd, optimizer=opt, n_edges=n_edges, total_edges=world * n_edges,
                n_layers=model_cfg["num_encoder_layers"], label=label + " + Adam update (FusedAdam, one launch)",
                sharded=False, scaling="weak", model=model, kind=kind, batch=batch)


def timed_steps(step, steps, warmup, world, dist):
    """W warm-up steps, then K steps between barrier + synchronize on both sides (wall clock, MAX over ranks) -- plus one HIP
    event pair per step on the compute stream, whose median is reported next to the mean."""
    for _ in range(warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    # (the eager step is as much host work as GPU work -- tools/host_enqueue_time.py: 10.6 ms of Python per 11.2 ms configs[1] step --
    # so a generation-2 garbage collection inside the loop shows up as a 10 - 30 ms outlier step: collect before, none inside, as
    # `timeit` does)
    gc.collect()
    gc_was_on = gc.isenabled()
    gc.disable()
    try:
        t0 = time.perf_counter()
        for a, b in ev:
            a.record()
            loss = step()
            b.record()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
    finally:
        if gc_was_on:
            gc.enable()
    if world > 1:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    per = sorted(a.elapsed_time(b) for a, b in ev)
    assert torch.isfinite(loss).item(), "non-finite loss"
    return elapsed, per[len(per) // 2]


def saved_activation_bytes(wl):
    """Device memory one training-mode forward of the stack keeps for its backward (allocator bytes held while the outputs live),
    per layer."""
    torch.cuda.synchronize()
    before = torch.cuda.memory_allocated()
    out = wl["forward"]()
    torch.cuda.synchronize()
    held = torch.cuda.memory_allocated() - before
    del out
    return held / wl["n_layers"]


def c5_block(G, ops, args):
    """A short measurement at BASELINE configs[4] size on this one GPU (100 000 nodes / 1 000 000 edges, (256,32), 4 layers),
    reported inside the default line: the configuration north_star's target sentence is written on."""
    import copy
    from gcpnet_amd.synthetic import layer_flops

    a = copy.copy(args)
    a.config, a.nodes, a.neighbors, a.sdim, a.vdim, a.layers, a.shard = "c5", 100000, 10, 256, 32, 4, "batch"
    wl = build_layer_workload(a, 0, 1, G, ops)
    # (two warm-up steps: the first ones grow the caching allocator's pool by ~75 GB of saved activations)
    k = 10
    elapsed, med = timed_steps(wl["step"], k, 2, 1, None)
    fl = layer_flops(a.nodes, wl["n_edges"], (256, 32), (32, 4))["fwd_bwd"] * a.layers
    saved = saved_activation_bytes(wl)
    out = {"saved_activation_bytes_per_layer": saved, "workload": wl["label"], "n_edges": wl["n_edges"], "steps": k, "warmup": 2, "ms_per_step": elapsed / k * 1e3,
           "ms_per_step_median": med, "edges_per_s": wl["n_edges"] * a.layers * k / elapsed,
           "algorithmic_tflops_per_s": fl * k / elapsed / 1e12,
           "frac_of_fp32_mfma_peak": fl * k / elapsed / 1e12 / PEAK_FP32_MFMA_TFLOPS,
           "frac_of_fp32_mfma_peak_median_step": fl / (med * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS}
    # the memory route (ops.CHAIN_RECOMPUTE: the message chain recomputed in the backward, SURVEY.md 8d's byte model): what it holds
    # and what it costs, same graph, same weights
    try:
        prev = ops.CHAIN_RECOMPUTE
        ops.CHAIN_RECOMPUTE = True
        torch.cuda.empty_cache()
        el_r, med_r = timed_steps(wl["step"], 5, 2, 1, None)
        out["recompute_route"] = {"switch": "GCPNET_CHAIN_RECOMPUTE=1 / ops.CHAIN_RECOMPUTE", "saved_activation_bytes_per_layer": saved_activation_bytes(wl),
                                  "ms_per_step": el_r / 5 * 1e3, "ms_per_step_median": med_r, "steps": 5, "warmup": 2}
    except Exception as exc:  # noqa: BLE001 -- must not cost the run its headline line
        out["recompute_route_error"] = f"{type(exc).__name__}: {exc}"[:300]
    finally:
        ops.CHAIN_RECOMPUTE = prev
        torch.cuda.empty_cache()
    if not args.no_cpu_baseline:
        try:  # ONE layer on a 1/10-size graph of the same recipe, labelled as such; with its own parity check
            out["cpu_baseline"], out["parity_check"] = cpu_baseline(wl, a)
        except Exception as exc:  # noqa: BLE001 -- must not cost the run its headline line
            out["cpu_baseline_error"] = f"{type(exc).__name__}: {exc}"[:300]
    del wl
    torch.cuda.empty_cache()
    try:
        out["roofline"] = c5_kernel_roofline(G, ops, 1000000, 256, 32)
    except Exception as exc:  # noqa: BLE001 -- must not cost the run its headline line
        out["roofline_error"] = f"{type(exc).__name__}: {exc}"[:300]
    torch.cuda.empty_cache()
    return out


def c5_kernel_roofline(G, ops, rows, sdim, vdim, iters=10):
    """The dominant kernel of the configs[4] step -- the backward (data path) of ONE residual message GCP (256,32)->(256,32) on
    10^6 edge rows, `gcp_wg_bwd_kernel`, 28 launches per 4-layer step -- alone, HIP events on the launch stream: 2 * rows *
    gcp_macs FLOP per launch against the fp32 MFMA peak; counter traffic / MFMA-busy share from the committed PMC passes."""
    from gcpnet_amd.synthetic import gcp_macs

    g = torch.Generator(device="cuda").manual_seed(3)
    block = G.GCP2((sdim, vdim), (sdim, vdim), nonlinearities=("relu", None), bottleneck=4).cuda()
    spec = block.make_spec([None], [None], residual=True)
    w = tuple(None if t is None else t.detach() for t in block._weights())
    s = torch.randn(rows, sdim, device="cuda", generator=g).requires_grad_()
    v = torch.randn(rows, vdim, 3, device="cuda", generator=g)
    fr = torch.randn(rows, 3, 3, device="cuda", generator=g)
    ds = torch.randn(rows, sdim, device="cuda", generator=g)
    dv = torch.randn(rows, vdim, 3, device="cuda", generator=g)
    out_s, _ = ops.gcp2(spec, [s], [v], fr, w)
    saved = out_s.grad_fn.saved_tensors
    pack, s_pre, gate = saved[-3], saved[-2], saved[-1]
    if getattr(out_s.grad_fn, "s_pre_tb", False):  # (a single block through the workgroup kernels saves s_pre tile-blocked: rows for this probe)
        s_pre = ops.TileBlocked(rows, sdim, s_pre.device, owner=s_pre, offset=0, n=s_pre.numel()).to_rows()
    keep = {}
    # as the step launches it for a block inside the chain: s_pre, the state gradient on both sides, ds_pre and the block's input
    # scalars (an operand of the weight-gradient GEMM only) in the tile-blocked layout; and with row-major tensors for comparison
    tb = ops.CHAIN_TILE_BLOCKED
    if tb:
        s_tb, s_pre_tb, ds_tb = (ops.TileBlocked.from_rows(t_) for t_ in (s.detach(), s_pre, ds))

    def bwd_rows():
        keep["r"] = ops.gcp2_backward_data(spec, rows, [s.detach()], [v], fr, w, pack, s_pre, gate, ds, dv, need_w=True)

    def bwd_tb():
        keep["r"] = ops.gcp2_backward_data(spec, rows, [s_tb], [v], fr, w, pack, s_pre_tb, gate, ds_tb, dv, need_w=True, tb_out=True)

    def median_ms(fn):
        with torch.no_grad():
            for _ in range(3):
                fn()
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
            torch.cuda.synchronize()
            for a, b in ev:
                a.record()
                fn()
                b.record()
            torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in ev)
        return ts[len(ts) // 2]

    t_rows = median_ms(bwd_rows)
    t = (median_ms(bwd_tb) if tb else t_rows) * 1e-3
    flops = 2.0 * rows * gcp_macs(sdim, vdim, sdim, vdim)
    H = block.hidden_dim
    # algorithmic bytes per row: reads s_pre, d(s_out), d(v_out), v_in, gate, frames; writes d(s_in), d(v_in), ds_pre, dgate, ext
    nbytes = rows * (4.0 * (2 * sdim + 2 * 3 * vdim + vdim + 9) + 4.0 * (2 * sdim + 3 * vdim + vdim + (H + 9 + 3) // 4 * 4))
    name = "gcp_wg_bwd_kernel (256,32)"
    achieved = flops / t / 1e12
    return {"kernel": f"gcp_wg_bwd_kernel: backward (data path) of one residual message GCP ({sdim},{vdim})->({sdim},{vdim}) on {rows} rows, "
                      "28 launches per 4-layer configs[4] step",
            "bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_MFMA_TFLOPS,
            "frac_of_bf16x6_equiv": achieved / PEAK_BF16X6_EQUIV_TFLOPS, "frac_of_f16x3_equiv": achieved / PEAK_F16X3_EQUIV_TFLOPS,
            "median_launch_ms": t * 1e3,
            "layout": "tile-blocked s_pre / state gradient / ds_pre, as inside the chain" if tb else "row-major",
            "row_major_launch_ms": t_rows, "flop_per_launch": flops,
            "algorithmic_bytes_per_launch": nbytes, "algorithmic_hbm_gbs": nbytes / t / 1e9,
            "traffic": pmc_traffic(name), "traffic_source": PMC_FILE if pmc_traffic(name) is not None else None,
            "mfma_busy_frac_pmc": pmc_mfma_busy(name)}


def main():
    args = parse()
    if os.environ.get("BENCH_WATCHDOG_S"):  # diagnosis of a hung multi-rank run: every thread's Python stack on stderr after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["BENCH_WATCHDOG_S"]), exit=True)
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: become the launcher (one rank per GPU over RCCL, the contract's command)
        import socket

        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stderr.write(f"[bench] --gpus {args.gpus} without WORLD_SIZE: re-launching under torch.distributed.run\n")
        os.execv(sys.executable, cmd)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with `python -m torch.distributed.run --nproc-per-node {args.gpus} "
                         f"bench.py --gpus {args.gpus} ...` (or run `python bench.py --gpus {args.gpus}` without WORLD_SIZE set: it re-launches itself)")
    # test hooks (a 1-GPU box can rehearse the N > 1 code path): BENCH_SHARE_GPU=1 puts every rank on cuda:0,
    # BENCH_DIST_BACKEND=gloo replaces RCCL (which refuses two ranks on one device)
    if os.environ.get("BENCH_SHARE_GPU") == "1":
        local_rank = 0
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    import torch.distributed as dist

    # what the process group actually is goes into the line (`rccl`): world size, backend and the (rank, device) pairs collected
    # with an all_gather -- i.e. a collective over every rank has run before anything is timed.  If the group cannot be set up
    # or that collective raises, the run falls back to independent replicas (no collective, rank 0 reports N x its own rate)
    # and says so.
    rccl_info = {"world": world, "backend": None, "ranks_seen": [[0, local_rank]], "fallback": None}
    if world > 1:
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)
            seen = [torch.zeros(2, dtype=torch.int64, device="cuda") for _ in range(world)]
            dist.all_gather(seen, torch.tensor([rank, torch.cuda.current_device()], dtype=torch.int64, device="cuda"))
            probe = torch.ones(1, device="cuda")
            dist.all_reduce(probe)
            assert int(probe.item()) == world
            rccl_info.update(backend=dist.get_backend(), world=dist.get_world_size(), ranks_seen=[[int(t[0]), int(t[1])] for t in seen])
        except Exception as exc:  # noqa: BLE001 -- any failure of the collective layer ends up in the JSON line, not in a dead run
            rccl_info["fallback"] = f"independent replicas, no collective ({type(exc).__name__}: {str(exc)[:200]})"
            sys.stderr.write(f"[bench rank {rank}] process group unusable, falling back to replicas: {exc}\n")
            try:
                if dist.is_initialized():
                    dist.destroy_process_group()
            except Exception:  # noqa: BLE001
                pass
            if args.shard == "graph":  # a one-graph run cannot degrade to replicas: that would be a different measurement
                raise SystemExit(f"[bench rank {rank}] --shard graph needs the collective layer and it is unusable: {exc}")
            world_job, world = world, 1  # this process now behaves like a single-GPU run and REPORTS itself as one (n_gpus = 1)
        else:
            world_job = world
    else:
        world_job = 1

    import gcpnet_amd as G
    from gcpnet_amd import ops
    from gcpnet_amd.synthetic import layer_flops

    from gcpnet_amd import _lib as _glib
    if _glib.load().gcpnet_debug_knobs_compiled():
        raise SystemExit("libgcpnet_hip.so was built with -DGCP_DEBUG_KNOBS (measurement knobs that change results): no bench line from it")
    is_stack = args.config in ("c2", "c5")
    if args.dry_run_world:
        assert is_stack and world == 1, "--dry-run-world: c2 / c5, one process"
        dry_run_world(args, G, ops)
        return
    wl = (build_layer_workload if is_stack else build_model_workload)(args, rank, world, G, ops)
    step_fn = wl["step"]
    if args.hip_graph:
        from gcpnet_amd.graphs import GraphedStep

        assert world == 1, "--hip-graph: single-GPU runs (collectives are not captured here)"
        step_fn = (GraphedStep(wl["fwd_bwd"], warmup=max(args.warmup, 3), optimizer=wl["optimizer"]) if "optimizer" in wl else
                   GraphedStep(wl["step"], warmup=max(args.warmup, 3)))
    elapsed, median_ms = timed_steps(step_fn, args.steps, args.warmup, world, dist)

    if rank == 0 and args.step_only:
        print(json.dumps({"ms_per_step": elapsed / args.steps * 1e3, "ms_per_step_median": median_ms, "steps": args.steps,
                          "warmup": args.warmup, "config": args.config}))
    elif rank == 0:
        value = wl["total_edges"] * wl["n_layers"] * args.steps / elapsed
        out = {
            # (process group unusable: the ranks ran unsynchronised replicas -- the line is this rank's own single-GPU measurement;
            # what N such replicas would add up to is an extrapolation and lives in rccl.unsynchronised_replica_extrapolation)
            "metric": "processed edges/sec (GCP fwd+bwd)", "value": value, "unit": "edges/s",
            "n_gpus": 1 if rccl_info["fallback"] else world_job,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "ms_per_step_median": median_ms,
            "higher_is_better": True, "scaling": wl["scaling"], "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            # every switch of this package that the environment sets (the shipped library honours none that changes results)
            "env": {k: v for k, v in sorted(os.environ.items()) if k.startswith(("GCPNET_", "BENCH_"))},
            "rccl": dict(rccl_info, **({"requested_gpus": world_job, "unsynchronised_replica_extrapolation": value * world_job}
                                       if rccl_info["fallback"] else {})),
            "config": {
                "workload": f"{args.config}: {wl['label']}", "n_edges": wl["n_edges"], "layers": wl["n_layers"],
                "launch": "hipGraph replay of the captured step" if args.hip_graph else "eager launches",
                "parallelism": ("single GPU" if world == 1 else
                                (f"one graph split by target-node ranges over {world} GPUs: per layer "
                                 + ("all-to-all of the halo rows (Morton-ordered nodes) forward and backward" if args.halo else
                                    "all-gather of node features / reduce-scatter of their gradients")
                                 + " + all-reduce (sum) of weight grads, RCCL" if wl["sharded"] else
                                 f"one graph (batch) per GPU x{world}, RCCL all-reduce (mean) of weight grads")),
            },
        }
        if is_stack:
            node_dims = (args.sdim, args.vdim)
            fl = layer_flops(args.nodes, wl["n_edges"], node_dims, (32, 4))["fwd_bwd"] * args.layers
            per_job = fl * (1 if wl["sharded"] else world)
            # (a forward of the one-graph sharding holds collectives: rank 0 alone must not run one -- every other rank is already
            # at the closing barrier.  Found by the two-rank rehearsal: BENCH_SHARE_GPU=1 BENCH_DIST_BACKEND=gloo, BENCH_WATCHDOG_S)
            out["saved_activation_bytes_per_layer"] = None if wl["sharded"] else saved_activation_bytes(wl)
            out["whole_step"] = {"algorithmic_tflops_per_s": per_job * args.steps / elapsed / 1e12,
                                 "frac_of_fp32_mfma_peak": per_job * args.steps / elapsed / 1e12 / PEAK_FP32_MFMA_TFLOPS / world}
        if is_stack and not wl["sharded"]:
            kr = kernel_roofline(G, ops, wl["layers"][0], wl["frames"], wl["n_edges"], args.sdim, args.vdim)
            times, kbytes = kr["times"], kr["bytes"]
            dom = min(times, key=lambda k: kr["flops"][k] / times[k])  # the kernel furthest below its roofline
            achieved = kr["flops"][dom] / times[dom] / 1e12
            out["roofline"] = {
                "kernel": f"{dom} on the {kr['n_blocks']}-block residual message chain (s,V)->(s,V) of one layer, E rows",
                "bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / PEAK_FP32_MFMA_TFLOPS, "peak_bf16x6_equiv": PEAK_BF16X6_EQUIV_TFLOPS,
                "frac_of_bf16x6_equiv": achieved / PEAK_BF16X6_EQUIV_TFLOPS,
                "peak_f16x3_equiv": PEAK_F16X3_EQUIV_TFLOPS, "frac_of_f16x3_equiv": achieved / PEAK_F16X3_EQUIV_TFLOPS,
                "traffic": pmc_traffic(dom),
                "traffic_source": PMC_FILE if pmc_traffic(dom) is not None else None,
                "traffic_note": (_pmc_file().get(dom + "_parts") or {}).get("note"),
                "arithmetic": ("fp32 results; the large products of the chain kernels (scalar_out forward, W^T ds_pre backward; at (256,32) the "
                               "same two in the workgroup kernels and the 256 x 288 weight-gradient GEMM) run on the fp16 matrix pipe with both "
                               "operands split into two fp16 terms under exact power-of-two scales and three products kept, fp32 "
                               "accumulation (error <= 3*2^-22 of sum|a b|: csrc/gcp_f16x2.h, tests/test_f16x2.py, tests/test_bf16x3.py); "
                               "the gate Linear and the 128 x 160 GEMM form keep three bf16 terms / six products (3*2^-24: csrc/gcp_bf16x3.h); "
                               "every other product is v_mfma_f32_32x32x2_f32.  `peak` stays the fp32 MFMA peak"),
                "mfma_busy_frac_pmc": pmc_mfma_busy(dom),
                "median_launch_ms": times[dom] * 1e3, "flop_per_launch": kr["flops"][dom],
                "algorithmic_hbm_gbs": kbytes[dom] / times[dom] / 1e9,
                "algorithmic_hbm_frac": kbytes[dom] / times[dom] / 1e9 / PEAK_HBM_GBS,
                "all_kernels_ms": {k: v * 1e3 for k, v in times.items()},
                "all_kernels_tflops": {k: kr["flops"][k] / v / 1e12 for k, v in times.items()},
                "all_kernels_algorithmic_hbm_gbs": {k: kbytes[k] / v / 1e9 for k, v in times.items()},
                "all_kernels_traffic": {k: pmc_traffic(k) for k in times},
                "all_kernels_algorithmic_bytes": {k: float(kbytes[k]) for k in times},
            }
            width = args.sdim + 3 * args.vdim
            # the gather / aggregate kernel against the HBM roofline, at this run's size and at BASELINE configs[4]'s
            # (100k nodes / 1M edges, (256,32): 1.5 GB per launch, far beyond the 256 MB Infinity Cache)
            out["aggregate_kernel"] = aggregate_roofline(G, ops, args.nodes, wl["n_edges"], width,
                                                         f"{args.nodes} nodes / {wl['n_edges']} edges, width s+3V = {width}")
            out["aggregate_kernel_c5"] = aggregate_roofline(G, ops, 100000, 1000000, 256 + 96,
                                                            "100000 nodes / 1000000 edges, width s+3V = 352", iters=10)
            out["aggregate_product_path"] = aggregate_product_path(G, ops, args.nodes, wl["n_edges"], args.sdim, args.vdim)
            out["aggregate_product_path_c5"] = aggregate_product_path(G, ops, 100000, 1000000, 256, 32, iters=10)
            out["gather_kernel_c5"] = gather_roofline(G, ops, 100000, 1000000, 256 + 96, "100000 nodes / 1000000 edges, width s+3V = 352", iters=10)
            # north_star's ">= 40 % of the HBM roofline on the message / aggregate kernel at 100k nodes / 1M edges": inside `roofline`
            ag, ga = out["aggregate_kernel_c5"], out["gather_kernel_c5"]
            out["roofline"]["hbm_kernel"] = {"kernel": ag["kernel"], "workload": ag["workload"], "frac": ag["frac"], "achieved": ag["achieved"],
                                             "peak": ag["peak"], "unit": "GB/s", "bytes": ag["bytes_per_launch"], "avg_launch_ms": ag["avg_launch_ms"]}
            out["roofline"]["hbm_kernel_gather"] = {"kernel": ga["kernel"], "workload": ga["workload"], "frac": ga["frac"], "achieved": ga["achieved"],
                                                    "peak": ga["peak"], "unit": "GB/s", "bytes": ga["bytes_per_launch"], "avg_launch_ms": ga["avg_launch_ms"]}
            if world == 1:  # (a step of a multi-rank job holds a collective: rank 0 alone must not run one)
                out["kernels_per_step"] = count_device_kernels(step_fn)
            if world == 1 and not args.no_cpu_baseline:
                out["cpu_baseline"], out["parity_check"] = cpu_baseline(wl, args)
            if world == 1 and args.config == "c2" and not (args.no_c5_block and args.no_other_configs):
                wl.clear()
                torch.cuda.empty_cache()
                if not args.no_other_configs:
                    out["other_configs"] = other_configs_block(G, ops, args)
                if not args.no_c5_block:
                    out["c5_single_gpu"] = c5_block(G, ops, args)
                    # the configuration north_star's target sentence is written on, at the top level of the line as well
                    out["roofline_c5"] = out["c5_single_gpu"].get("roofline")
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


def cpu_baseline(wl, args):
    """The oracle on the host cores: the SAME stack, weights, graph, inputs and loss as the timed GPU step (configs[1]: the full
    workload, 1 warm-up + 3 timed steps; configs[4]: one layer on a 1/10-size graph of the same recipe, 1 timed step, stated in
    `sample`), and the parity check that goes with it: GPU outputs and input gradients of that step against the oracle's."""
    from gcpnet_amd.synthetic import make_inputs
    from oracle import gcp_oracle as O

    cores = min(os.cpu_count() or 1, 32)  # more threads than this only adds contention on these small ops
    torch.set_num_threads(cores)
    layers, n_layers = wl["layers"], wl["n_layers"]
    full = args.config == "c2"
    if full:
        sample, lw, reps, what = wl["host"], {k: v.cpu() for k, v in wl["lw"].items()}, 3, "the full timed workload"
    else:
        sample = make_inputs(max(args.nodes // 10, 64), args.neighbors, (args.sdim, args.vdim), (32, 4), seed=1234)
        g = torch.Generator().manual_seed(1)
        lw = dict(h=torch.randn(sample["h"].shape, generator=g), chi=torch.randn(sample["chi"].shape, generator=g))
        n_layers, reps, what = 1, 1, "ONE layer of the stack on a 1/10-size graph of the same recipe"
    P = {k: v.detach().cpu().clone().requires_grad_() for k, v in layers.state_dict().items()}
    ins = {k: sample[k].clone().requires_grad_() for k in ("h", "chi", "e", "xi")}
    ei = sample["edge_index"]
    fr = O.localize(sample["x"], ei)
    cfg, lcfg = O.default_module_cfg(), O.default_layer_cfg()

    def one():
        for t in ins.values():
            t.grad = None
        for t in P.values():
            t.grad = None
        h, chi = ins["h"], ins["chi"]
        for i in range(n_layers):
            h, chi = O.gcp_interactions(P, f"{i}.", h, chi, ins["e"], ins["xi"], ei, fr, cfg, lcfg)
        ((h * lw["h"]).sum() + (chi * lw["chi"]).sum()).backward()
        return h.detach(), chi.detach()

    one()  # warm-up (allocator, thread pool)
    t0 = time.perf_counter()
    for _ in range(reps):
        ch, cchi = one()
    dt = (time.perf_counter() - t0) / reps
    base = {"value": ei.shape[1] * n_layers / dt, "unit": "edges/s", "cores": cores, "kind": "port",
            "sample": f"{what}: fwd+bwd of {n_layers} GCPInteractions layer(s) (same weights, inputs and loss) on "
                      f"{sample['h'].shape[0]} nodes / {ei.shape[1]} edges, torch CPU {cores} threads, {dt:.2f} s per step, "
                      f"mean of {reps} after 1 warm-up"}
    # ---- parity of the timed GPU path against this very CPU run: outputs element-wise at 1e-5 of their
    #      scale; input gradients in the relative L2 sense (the shipped config is ReLU: isolated sign flips of pre-activations
    #      within round-off of zero move single rows, see tests/helpers.as_accurate)
    import gcpnet_amd as G
    gi = {k: sample[k].cuda().requires_grad_() for k in ("h", "chi", "e", "xi")}
    gfr = G.localize(sample["x"].cuda(), ei.cuda())
    h, chi = gi["h"], gi["chi"]
    for p_ in layers.parameters():
        p_.grad = None
    for i in range(n_layers):
        h, chi = layers[i]((h, chi), (gi["e"], gi["xi"]), ei.cuda(), gfr)
    ((h * lw["h"].cuda()).sum() + (chi * lw["chi"].cuda()).sum()).backward()
    torch.cuda.synchronize()
    # weight gradients of the FIRST layer (it sees the gradient of everything behind it): every parameter tensor against the oracle's,
    # relative L2 (a step that skipped or corrupted a weight-gradient GEMM cannot pass)
    wg_err = {}
    for name, p_ in layers[0].named_parameters():
        ref = P["0." + name].grad
        if ref is None or p_.grad is None:
            wg_err[name] = float("inf") if (ref is None) != (p_.grad is None) else 0.0
            continue
        wg_err[name] = float((p_.grad.cpu().double() - ref.double()).norm() / ref.double().norm().clamp(min=1e-30))
    worst_w = max(wg_err, key=wg_err.get)
    fwd_err = max(float((h.detach().cpu() - ch).abs().max() / ch.abs().max().clamp(min=1.0)),
                  float((chi.detach().cpu() - cchi).abs().max() / cchi.abs().max().clamp(min=1.0)))
    fwd_abs = max(float((h.detach().cpu() - ch).abs().max()), float((chi.detach().cpu() - cchi).abs().max()))
    # element-wise view of the input gradients (ReLU configuration: a pre-activation within round-off of zero flips in one of the two
    # fp32 evaluations and moves the rows it reaches -- tests/helpers.as_accurate, the ReLU census of tests/test_full_size.py): the
    # largest element-wise difference in units of the tensor's scale, and the share of ROWS holding an element off by more than 1e-4 of it
    el_err, row_frac = {}, {}
    for k in gi:
        ref = ins[k].grad
        dlt = (gi[k].grad.cpu() - ref).abs()
        sc = float(ref.abs().max().clamp(min=1e-30))
        el_err[k] = float(dlt.max()) / sc
        row_frac[k] = float((dlt.reshape(dlt.shape[0], -1).max(dim=1).values > 1e-4 * sc).float().mean())
    grad_err = {k: float((gi[k].grad.cpu().double() - ins[k].grad.double()).norm() / ins[k].grad.double().norm().clamp(min=1e-30))
                for k in gi}
    parity = {"against": "the cpu_baseline run above (oracle, fp32)", "forward_max_abs_err_over_scale": fwd_err,
              "forward_max_abs_err": fwd_abs, "forward_tol": 1e-5, "input_grad_rel_l2_err": grad_err, "input_grad_tol": 1e-3,
              "input_grad_max_elementwise_err_over_scale": el_err, "input_grad_rows_off_by_more_than_1e-4_of_scale": row_frac,
              "layer0_weight_grad_rel_l2_err_max": wg_err[worst_w], "layer0_weight_grad_worst": worst_w,
              "layer0_weight_grads_checked": len(wg_err), "weight_grad_tol": 2e-3,
              "ok": bool(fwd_err <= 1e-5 and max(grad_err.values()) <= 1e-3 and wg_err[worst_w] <= 2e-3)}
    assert parity["ok"], f"GPU path disagrees with the CPU oracle: {parity}"
    return base, parity


if __name__ == "__main__":
    main()
