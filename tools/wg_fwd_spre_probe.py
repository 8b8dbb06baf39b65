"""What the s_pre store costs the (256,32) chain forward: the 7-block workgroup forward launch on 10^6 rows as shipped, and with the
blocks' s_pre pointers NULL (the kernel's inference form of that output; everything else still stored).  Forward only, HIP events.
usage: wg_fwd_spre_probe.py [rows]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import gcpnet_amd as G  # noqa: E402
from gcpnet_amd import ops  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
S, V, n = 256, 32, 7
torch.manual_seed(0)
mods = [G.GCP2((S, V), (S, V), nonlinearities=("relu", None), bottleneck=4).cuda() for _ in range(n)]
specs = [m.make_spec([None], [None], residual=True) for m in mods]
g = torch.Generator(device="cuda").manual_seed(0)
s0 = torch.randn(rows, S, device="cuda", generator=g).requires_grad_()
v0 = torch.randn(rows, V, 3, device="cuda", generator=g).requires_grad_()
fr = torch.randn(rows, 3, 3, device="cuda", generator=g)
ws = [m._weights() for m in mods]
orig = ops._wg_block


def no_spre(spec, w, s_out, v_out, s_pre, gate, residual, keep):
    blk = orig(spec, w, s_out, v_out, s_pre, gate, residual, keep)
    blk.s_pre = None
    return blk


def run(label, patched, what="s_pre"):
    ops._wg_block = no_spre if patched else orig
    ts = []
    for i in range(8):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        out = ops.gcp2_chain(specs, s0, v0, fr, ws)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
        del out
    ts = sorted(ts[2:])
    print(f"{label}: median {ts[len(ts) // 2]:.3f} ms per 7-block forward launch on {rows} rows (incl. the Function's host work)")
    ops._wg_block = orig


run("shipped", False)
run("s_pre pointers NULL", True)
run("shipped", False)
run("s_pre pointers NULL", True)
