"""Host-side profile of one model step (launch-bound configurations): cProfile over K eager steps, top functions by own time.
usage: pyprofile_step.py [c1|c4|c3|c2|c5] [steps] [extra bench.py flags, e.g. --nodes 200]"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import gcpnet_amd as G  # noqa: E402
from gcpnet_amd import ops  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c1"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
sys.argv = [sys.argv[0], "--config", cfg] + sys.argv[3:]
args = bench.parse()
wl = (bench.build_layer_workload if cfg in ("c2", "c5") else bench.build_model_workload)(args, 0, 1, G, ops)
for _ in range(5):
    wl["step"]()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    wl["step"]()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(35)
st.sort_stats("cumulative").print_stats(30)
