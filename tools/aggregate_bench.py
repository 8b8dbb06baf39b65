"""The aggregate kernel (scatter-mean of the edge messages [E, s + 3V] onto their target nodes, reference gcpnet.py:939-947) ALONE, at
BASELINE configs[4] size: every `segment_reduce_kernel` launch of this process is that one call site, so a rocprofv3 --kernel-trace
--stats summary of it gives the kernel's average duration directly (in a step's summary five call sites share the kernel name).
    rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python tools/aggregate_bench.py
prints the HIP-event median, the algorithmic bytes (E + N) x width x 4 and the fraction of the 8 TB/s HBM peak."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gcpnet_amd import ops  # noqa: E402

n_nodes, n_edges, width = 100000, 1000000, 256 + 96
g = torch.Generator(device="cuda").manual_seed(1)
col = torch.sort(torch.randint(0, n_nodes, (n_edges,), device="cuda", generator=g)).values
plan = ops.GatherPlan(col, n_nodes)
msg = torch.randn(n_edges, width, device="cuda", generator=g)
for _ in range(5):
    ops._segment_reduce_raw(msg, 0, width, width, plan, True)
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
torch.cuda.synchronize()
for a, b in ev:
    a.record()
    ops._segment_reduce_raw(msg, 0, width, width, plan, True)
    b.record()
torch.cuda.synchronize()
ts = sorted(a.elapsed_time(b) for a, b in ev)
t = ts[len(ts) // 2] * 1e-3
by = (n_edges + n_nodes) * width * 4.0
print(f"segment_reduce_kernel<mean>, {n_nodes} nodes / {n_edges} edges, width {width}: median {t * 1e6:.1f} us per launch, "
      f"{by / 1e9:.3f} GB algorithmic -> {by / t / 1e9:.0f} GB/s = {by / t / 8e12:.3f} of the 8 TB/s HBM peak (35 launches in this process)")
