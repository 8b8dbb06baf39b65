"""Which host-side operations an eager step consists of: torch.profiler CPU activities over K steps of a bench workload, self CPU time
per operator / autograd node (the library's own launches appear under the autograd Function that makes them).
usage: host_ops_profile.py [c2|c5|...] [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402
import gcpnet_amd as G  # noqa: E402
from gcpnet_amd import ops  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
sys.argv = [sys.argv[0], "--config", cfg]
args = bench.parse()
wl = (bench.build_layer_workload if cfg in ("c2", "c5") else bench.build_model_workload)(args, 0, 1, G, ops)
for _ in range(5):
    wl["step"]()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU]) as prof:
    for _ in range(steps):
        wl["step"]()
    torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda e: -e.self_cpu_time_total)
tot = sum(e.self_cpu_time_total for e in rows)
print(f"{cfg}: {steps} steps, total self CPU time {tot / steps / 1e3:.2f} ms/step (profiler on: slower than the plain step)")
for e in rows[:45]:
    print(f"  {e.self_cpu_time_total / steps / 1e3:8.3f} ms/step self  {e.cpu_time_total / steps / 1e3:8.3f} total  {e.count / steps:7.1f} calls/step  {e.key[:90]}")
