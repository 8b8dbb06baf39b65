"""Which call sites of a step still reach a library GEMM (torch.matmul / addmm / mm)?  Prints each distinct call stack (innermost
frames inside gcpnet_amd) with operand shapes and a count.  usage: find_library_gemms.py [c2|c5|c3|...]"""
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import gcpnet_amd as G  # noqa: E402
from gcpnet_amd import ops  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c5"
sys.argv = [sys.argv[0], "--config", cfg]
args = bench.parse()
wl = bench.build_layer_workload(args, 0, 1, G, ops) if cfg in ("c2", "c5") else bench.build_model_workload(args, 0, 1, G, ops)
wl["step"]()
seen = collections.Counter()
for name in ("matmul", "addmm", "mm", "bmm"):
    orig = getattr(torch, name)

    def hook(*a, _orig=orig, _name=name, **k):
        st = [f"{os.path.basename(f.filename)}:{f.lineno}:{f.name}" for f in traceback.extract_stack()[:-1] if "gcpnet_amd" in f.filename][-3:]
        shapes = [tuple(t.shape) for t in a if torch.is_tensor(t)]
        seen[(_name, " <- ".join(reversed(st)), str(shapes))] += 1
        return _orig(*a, **k)

    setattr(torch, name, hook)
wl["step"]()
torch.cuda.synchronize()
for (name, where, shapes), c in seen.most_common():
    print(f"{c:4d} x torch.{name} {shapes}  at {where}")
if not seen:
    print("no library GEMM call from Python in this step")
