"""Is a stretch of the eager step bound by the GPU or by the host?  Times the step with one family of launches REMOVED (the outputs stay
unwritten: results are garbage, only the clock is read) and compares the drop with that family's rocprofv3 time per step: a drop of about
the kernels' duration = the GPU was the bound there; no drop = the host (or another stream) was.
usage: skip_kernels_probe.py [c2|c5] [steps]      families: segment_reduce, layernorm_bwd reductions (via a 1-row call), small adds"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import gcpnet_amd as G  # noqa: E402
from gcpnet_amd import ops  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
sys.argv = [sys.argv[0], "--config", cfg]
args = bench.parse()
wl = bench.build_layer_workload(args, 0, 1, G, ops)


def timed(label):
    for _ in range(5):
        wl["step"]()
    torch.cuda.synchronize()
    ts = []
    for _ in range(steps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        wl["step"]()
        b.record()
        ts.append((a, b))
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ts)
    print(f"{cfg} {label:48s} median {ms[len(ms) // 2]:8.3f} ms   min {ms[0]:8.3f}", flush=True)


timed("as shipped")
real_sr = ops._segment_reduce_raw


def no_segment_reduce(x, col0, D, ld, plan, mean):
    return torch.empty((plan.n_src, D), dtype=torch.float32, device=x.device)


ops._segment_reduce_raw = no_segment_reduce
timed("without the _segment_reduce_raw launches")
ops._segment_reduce_raw = real_sr
timed("as shipped (again)")
real_sub = ops._side_submit
ops._side_submit = lambda fn, keep: None
real_run = ops.run_weight_grad_jobs
ops.run_weight_grad_jobs = lambda jobs, in_backward_of_leaves=False: None
timed("without any weight-gradient launch (second stream empty)")
ops._segment_reduce_raw = no_segment_reduce
timed("without both")
