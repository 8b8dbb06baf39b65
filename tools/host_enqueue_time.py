"""Is the step host-bound?  K eager steps: wall time until the host has enqueued all of them (no synchronize) against wall time until
the GPU has finished them, plus the per-step host time of forward / backward.  If "enqueued" ~= "finished" the host is the bound.
usage: host_enqueue_time.py [c2|c5|...] [steps] [extra bench.py flags]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import gcpnet_amd as G  # noqa: E402
from gcpnet_amd import ops  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
sys.argv = [sys.argv[0], "--config", cfg] + sys.argv[3:]
args = bench.parse()
wl = (bench.build_layer_workload if cfg in ("c2", "c5") else bench.build_model_workload)(args, 0, 1, G, ops)
for _ in range(5):
    wl["step"]()
torch.cuda.synchronize()
if os.environ.get("HOST_GC_OFF"):  # (as bench.py times its steps: one collection before, none inside)
    import gc
    gc.collect()
    gc.disable()
stamps = []
t0 = time.perf_counter()
for _ in range(steps):
    wl["step"]()
    stamps.append(time.perf_counter())
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
per = [b - a for a, b in zip([t0] + stamps[:-1], stamps)]
print(f"{cfg} {' '.join(sys.argv[3:])}: host enqueued {steps} steps in {(t1 - t0) / steps * 1e3:.2f} ms/step; GPU finished after "
      f"{(t2 - t0) / steps * 1e3:.2f} ms/step; host per step min {min(per) * 1e3:.2f} median {sorted(per)[len(per) // 2] * 1e3:.2f} max {max(per) * 1e3:.2f} ms")
if "forward" in wl:
    torch.cuda.synchronize()
    tf = tb = 0.0
    for _ in range(steps):
        a = time.perf_counter()
        h, chi = wl["forward"]()
        loss = (h * wl["lw"]["h"]).sum() + (chi * wl["lw"]["chi"]).sum()
        b = time.perf_counter()
        loss.backward()
        c = time.perf_counter()
        tf += b - a
        tb += c - b
    torch.cuda.synchronize()
    print(f"   host time of forward (+ loss) {tf / steps * 1e3:.2f} ms, of backward() {tb / steps * 1e3:.2f} ms")
