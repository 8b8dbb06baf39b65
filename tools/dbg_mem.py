"""Memory held across repeated chain forwards at configs[4] size (looking for reference cycles that delay frees)."""
import gc
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gcpnet_amd as G  # noqa: E402
from gcpnet_amd import ops  # noqa: E402

E, sdim, vdim = 1000000, 256, 32
layer = G.GCPInteractions((sdim, vdim), (32, 4), cfg=G.default_module_cfg(), layer_cfg=G.default_layer_cfg(), dropout=0.0).cuda()
blocks = list(layer.interaction.message_fusion[1:])
g = torch.Generator(device="cuda").manual_seed(0)
s = torch.randn(E, sdim, device="cuda", generator=g).requires_grad_()
v = torch.randn(E, vdim, 3, device="cuda", generator=g).requires_grad_()
fr = torch.randn(E, 3, 3, device="cuda", generator=g)
specs = [b.make_spec([None], [None], residual=True) for b in blocks]
ws = [tuple(None if t is None else t.detach().requires_grad_() for t in b._weights()) for b in blocks]
keep = {}
gb = lambda: torch.cuda.memory_allocated() / 2**30
print("start", gb())
for i in range(4):
    keep.pop("out", None)
    keep["out"] = ops.gcp2_chain(specs, s, v, fr, ws)
    torch.cuda.synchronize()
    print("after fwd", i, gb())
keep.clear()
print("cleared", gb())
gc.collect()
print("after gc", gb())
