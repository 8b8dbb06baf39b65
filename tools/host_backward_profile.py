"""Host time of every autograd Function of gcpnet_amd.ops in one step (forward and backward bodies timed with perf_counter; the
backward bodies run on the autograd engine's thread, where cProfile started on the main thread does not see them).  Run on a tiny
graph (--nodes 200) the GPU work vanishes and the step time IS the host's time.
usage: host_backward_profile.py [c2|c5|c1|c4|c3] [steps] [extra bench.py flags]"""
import cProfile
import collections
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import gcpnet_amd as G  # noqa: E402
from gcpnet_amd import ops  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
sys.argv = [sys.argv[0], "--config", cfg] + sys.argv[3:]
args = bench.parse()
acc = collections.defaultdict(lambda: [0, 0.0])
depth = [0]
prof = cProfile.Profile() if os.environ.get("HOST_PROFILE_BODIES") else None  # function-level profile of the bodies (adds overhead)


def wrap(cls, name):
    fn = getattr(cls, name)

    def timed(*a, **k):
        t = time.perf_counter()
        depth[0] += 1
        if prof is not None and depth[0] == 1 and name == "backward":
            prof.enable()
        try:
            return fn(*a, **k)
        finally:
            depth[0] -= 1
            if prof is not None and depth[0] == 0 and name == "backward":
                prof.disable()
            r = acc[f"{cls.__name__}.{name}"]
            r[0] += 1
            r[1] += time.perf_counter() - t
    setattr(cls, name, staticmethod(timed))


for obj in list(vars(ops).values()):
    if isinstance(obj, type) and issubclass(obj, torch.autograd.Function) and obj is not torch.autograd.Function:
        wrap(obj, "forward")
        wrap(obj, "backward")
wl = (bench.build_layer_workload if cfg in ("c2", "c5") else bench.build_model_workload)(args, 0, 1, G, ops)
for _ in range(5):
    wl["step"]()
torch.cuda.synchronize()
acc.clear()
t0 = time.perf_counter()
for _ in range(steps):
    wl["step"]()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / steps
print(f"{cfg}: {wall * 1e3:.2f} ms per step (wall, timers on); inside Function bodies (nested calls counted in both):")
tot = 0.0
for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"  {t / steps * 1e3:8.3f} ms/step  {n / steps:6.1f} calls/step  {t / n * 1e6:8.1f} us/call  {k}")
if prof is not None:
    st = pstats.Stats(prof)
    st.sort_stats("tottime").print_stats(45)
    st.sort_stats("cumulative").print_stats(45)
