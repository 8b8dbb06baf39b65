"""Which kernel family runs the feed-forward blocks of a configuration, and why not the workgroup kernels when it is not them:
wraps the C entry points, logs every GCPNET_E_UNSUPPORTED with the call's dims.  usage: diag_wg_support.py [sdim vdim rows]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import gcpnet_amd as G  # noqa: E402
from gcpnet_amd import _lib, ops  # noqa: E402

s, v, rows = (int(a) for a in (sys.argv[1:4] + ["256", "32", "3200"][len(sys.argv) - 1:]))
lib = _lib.load()
for name in ("gcpnet_wg_forward", "gcpnet_wg_backward", "gcpnet_wg_backward_plan", "gcpnet_gcp2_forward", "gcpnet_gcp2_backward",
             "gcpnet_gcp2_chain_forward", "gcpnet_gcp2_chain_backward", "gcpnet_gcp2_headchain_forward"):
    fn = getattr(lib, name)

    def wrap(*a, _fn=fn, _name=name):
        rc = _fn(*a)
        print(f"    {_name} -> {rc}", flush=True)
        return rc

    setattr(lib, name, wrap)

for label, din, dout, acts in (("FF0", (s, v), (4 * s, 2 * v), ("relu", None)), ("FF1", (4 * s, 2 * v), (s, v), (None, None))):
    print(f"== {label} {din} -> {dout}, {rows} node rows", flush=True)
    torch.manual_seed(0)
    mod = G.GCP2(din, dout, nonlinearities=acts, bottleneck=4).cuda()
    x = (torch.randn(rows, din[0], device="cuda").requires_grad_(), torch.randn(rows, din[1], 3, device="cuda").requires_grad_())
    ei = torch.stack((torch.arange(rows), torch.arange(rows))).cuda()
    fr = torch.randn(rows, 3, 3, device="cuda")
    print("  forward:", flush=True)
    out = mod(x, ei, fr, node_inputs=True)
    print("  backward:", flush=True)
    (out[0].sum() + out[1].sum()).backward()
    torch.cuda.synchronize()
print(ops.WG_STATS)
