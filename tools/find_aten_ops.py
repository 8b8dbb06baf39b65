"""Which ATen operators does one step still run on the device, and from where?  A TorchDispatchMode counts every op that touches a
device tensor by (op, innermost gcpnet_amd / bench frames).  usage: find_aten_ops.py [c1|c4|c3|c2|c5]"""
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

import bench  # noqa: E402
import gcpnet_amd as G  # noqa: E402
from gcpnet_amd import ops  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c4"
sys.argv = [sys.argv[0], "--config", cfg]
args = bench.parse()
wl = bench.build_layer_workload(args, 0, 1, G, ops) if cfg in ("c2", "c5") else bench.build_model_workload(args, 0, 1, G, ops)
for _ in range(2):
    wl["step"]()
seen = collections.Counter()
SKIP = ("aten.view", "aten._unsafe_view", "aten.detach", "aten.t.", "aten.transpose", "aten.slice", "aten.select", "aten.expand", "aten.alias",
        "aten.unsqueeze", "aten.squeeze", "aten.as_strided", "aten.empty", "aten.reshape", "aten.permute", "aten.split", "aten._reshape_alias",
        "aten.unbind", "aten.narrow", "aten.is_", "aten.sym_", "aten.stride", "aten.size")


class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            dev = any(torch.is_tensor(a) and a.is_cuda for a in args) or any(
                isinstance(a, (list, tuple)) and any(torch.is_tensor(b) and b.is_cuda for b in a) for a in args)
            kd = (kwargs or {}).get("device")
            dev = dev or (kd is not None and "cuda" in str(kd))  # (factories: zeros / full / arange on the device have no tensor argument)
            if dev:
                st = [f"{os.path.basename(f.filename)}:{f.lineno}:{f.name}" for f in traceback.extract_stack()[:-1]
                      if "gcpnet_amd" in f.filename or f.filename.endswith("bench.py")][-2:]
                shp = next((tuple(a.shape) for a in args if torch.is_tensor(a)), args[0] if args and isinstance(args[0], (list, tuple)) else None)
                seen[(name, " <- ".join(reversed(st)) or "(autograd engine)", str(shp))] += 1
        return func(*args, **(kwargs or {}))


with Mode():
    wl["step"]()
torch.cuda.synchronize()
total = sum(seen.values())
print(f"{total} device ATen ops in one {cfg} step")
for (name, where, shp), c in seen.most_common(60):
    print(f"{c:4d} x {name:34s} {shp:22s} {where}")
