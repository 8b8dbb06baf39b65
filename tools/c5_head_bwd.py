"""Times the backward launches of one GCPMessagePassing layer at configs[4] size, split into the first message GCP (head) and the
chain blocks, for the tuning knobs GCPNET_WG_BWD_NW / GCPNET_WG_BWD_NOFUSE.  usage: python tools/c5_head_bwd.py [nodes] [K] [sdim vdim]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gcpnet_amd as G  # noqa: E402
from gcpnet_amd import ops  # noqa: E402
from gcpnet_amd.synthetic import make_inputs  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dims = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (256, 32)
ops.FORCE_WG_CHAIN_BACKWARD = True  # (so <= 128 would otherwise take the wave-per-tile chain kernel)
host = make_inputs(N, K, dims, (32, 4), seed=0)
dev = {k: v.cuda() for k, v in host.items()}
torch.manual_seed(0)
mp = G.GCPMessagePassing(dims, dims, (32, 4), cfg=G.default_module_cfg(), mp_cfg=G.default_layer_cfg().mp_cfg).cuda()
fr = G.localize(dev["x"], dev["edge_index"])
ins = {k: dev[k].clone().requires_grad_() for k in ("h", "chi", "e", "xi")}
times = {"head": [], "chain": []}
orig = ops._wg_backward


import ctypes as C  # noqa: E402
from gcpnet_amd import _lib  # noqa: E402

lib = _lib.load()
ntiles = (host["edge_index"].shape[1] + 31) // 32
buf = torch.zeros(ntiles * 8, dtype=torch.int64, device="cuda")
buf_chain = torch.zeros(ntiles * 8, dtype=torch.int64, device="cuda")
state = {"chain_seen": 0}
BL = ["loads + barrier", "P1 recompute", "P2 epilogue adjoint", "P3 gate adj + ds_pre", "P4 W^T ds_pre", "P5/P6 weight grads", "P7-P9"]


def timed(spec, *a, **kw):
    a0, b0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    head = bool(spec.add_plans)
    stamped = head or state["chain_seen"] == 0
    if not head:
        state["chain_seen"] += 1
    if stamped:
        tb = buf if head else buf_chain
        tb.zero_()
        lib.gcpnet_debug_set_phase_timing(C.c_void_p(tb.data_ptr()), ntiles)
    a0.record()
    r = orig(spec, *a, **kw)
    b0.record()
    if stamped:
        lib.gcpnet_debug_set_phase_timing(None, 0)
    times["head" if spec.add_plans else "chain"].append((a0, b0))
    return r


ops._wg_backward = timed
if os.environ.get("NO_SIDE_STREAM"):
    ops.set_weight_grad_stream(False)  # (then nothing else shares the CUs with the timed launches)
for it in range(3):
    for t in ins.values():
        t.grad = None
    for p in mp.parameters():
        p.grad = None
    for v in times.values():
        v.clear()
    state["chain_seen"] = 0
    out = mp((ins["h"], ins["chi"]), (ins["e"], ins["xi"]), dev["edge_index"], fr)
    (out[0].sum() + out[1].sum()).backward()
    torch.cuda.synchronize()
print("env:", {k: v for k, v in os.environ.items() if k.startswith("GCPNET_WG")})
for k, v in times.items():
    ts = [a.elapsed_time(b) for a, b in v]
    print(f"  {k}: {len(ts)} launches, {sum(ts) / max(len(ts), 1):.2f} ms each (incl. reduces / scratch allocation)")
for name, tb in (("head", buf), ("first chain block", buf_chain)):
    t = tb.view(ntiles, 8).cpu().double()
    t = t[(t[:, 7] > 0) & (t[:, 0] > 0)][:512]
    d = t[:, 1:8] - t[:, :7]
    print(f"{name} backward, mid-range tile of {t.shape[0]} workgroups: tile total median {(t[:, 7] - t[:, 0]).median().item():.0f} cycles")
    for i, lab in enumerate(BL):
        print(f"   {lab:24s} median {d[:, i].median().item():9.0f}  mean {d[:, i].mean().item():9.0f}  max {d[:, i].max().item():9.0f}")
