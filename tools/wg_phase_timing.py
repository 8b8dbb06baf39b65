"""Per-phase s_memtime profile of the workgroup forward kernel on a chain of 7 residual message GCPs (stamps: wave 0, last block).
usage: python tools/wg_phase_timing.py [n_edges] [sdim] [vdim]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gcpnet_amd as G  # noqa: E402
from gcpnet_amd import _lib, ops  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 160000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 128
V = int(sys.argv[3]) if len(sys.argv) > 3 else 16
lib = _lib.load()
torch.manual_seed(0)
g = torch.Generator(device="cuda").manual_seed(0)
s = torch.randn(E, S, device="cuda", generator=g)
v = torch.randn(E, V, 3, device="cuda", generator=g)
fr = torch.randn(E, 3, 3, device="cuda", generator=g)
ntiles = (E + 31) // 32
buf = torch.zeros(ntiles * 8, dtype=torch.int64, device="cuda")
mods = [G.GCP2((S, V), (S, V), nonlinearities=("relu", None), bottleneck=4).cuda() for _ in range(7)]
specs = [m.make_spec([None], [None], residual=True) for m in mods]
ws = [tuple(None if t is None else t.detach() for t in m._weights()) for m in mods]
LABELS = ["prologue", "B1 wait", "K loop (MFMA)", "bias/gate/stores", "GP + B2 wait", "X write + epilogue", "B3 + v_out store"]


def report(name):
    torch.cuda.synchronize()
    t = buf.view(ntiles, 8).cpu().double()
    d = t[:, 1:8] - t[:, :7]
    tot = t[:, 7] - t[:, 0]
    print(f"{name}: {ntiles} tiles, block total median {tot.median().item():.0f} ticks (s_memtime: 100 MHz ticks? see ratio below)")
    for i, lab in enumerate(LABELS):
        print(f"   {lab:24s} median {d[:, i].median().item():9.0f}  mean {d[:, i].mean().item():9.0f}  max {d[:, i].max().item():9.0f}")


for need_grad in (False, True):
    sx = s.clone().requires_grad_(need_grad)
    for _ in range(2):
        ops.gcp2_chain(specs, sx, v, fr, ws)
    buf.zero_()
    lib.gcpnet_debug_set_phase_timing(C.c_void_p(buf.data_ptr()), ntiles)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    ops.gcp2_chain(specs, sx, v, fr, ws)
    b.record()
    torch.cuda.synchronize()
    lib.gcpnet_debug_set_phase_timing(None, 0)
    print(f"chain forward launch {a.elapsed_time(b) * 1e3:.0f} us (need_grad={need_grad}), wg launches so far: {ops.WG_STATS}")
    report(f"wg chain fwd, last block, need_grad={need_grad}")

# ---- backward of one residual block (fused weight gradients), stamps of wave 0 on the last tile of every workgroup ----------
wreq = [mods[0]._weights()]  # (parameters: the weight gradients are needed, so the fused mode runs)
sx = s.clone().requires_grad_(True)
BL = ["loads + barrier", "P1 recompute", "P2 epilogue adjoint", "P3 gate adj + ds_pre", "P4 W^T ds_pre", "P5/P6 weight grads", "P7-P9 prologue adj, small w"]
for it in range(3):
    out_s, out_v = ops.gcp2_chain(specs[:1], sx, v, fr, wreq)
    ds, dv = torch.randn_like(out_s), torch.randn_like(out_v)
    buf.zero_()
    if it == 2:
        lib.gcpnet_debug_set_phase_timing(C.c_void_p(buf.data_ptr()), ntiles)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    ops.FORCE_WG_CHAIN_BACKWARD = True
    torch.autograd.backward([out_s, out_v], [ds, dv])
    b.record()
    torch.cuda.synchronize()
lib.gcpnet_debug_set_phase_timing(None, 0)
print(f"one-block backward (incl. reduces) {a.elapsed_time(b) * 1e3:.0f} us; wg launches: {ops.WG_STATS}")
t = buf.view(ntiles, 8).cpu().double()
t = t[(t[:, 7] > 0) & (t[:, 0] > 0)][:512]
if os.environ.get("WG_STAMPS_RAW"):  # (-DGCP_WG_STAMP_TAIL=4 builds: slot w = arrival of wave w at one barrier)
    rel = t - t.min(dim=1, keepdim=True).values
    print("arrival of wave w after the first wave (cycles), median over workgroups:", [round(float(x)) for x in rel.median(dim=0).values])
    print("spread first -> last, median:", float((t.max(dim=1).values - t.min(dim=1).values).median()))
    sys.exit(0)
d = t[:, 1:8] - t[:, :7]
print(f"workgroups with stamps: {t.shape[0]}, tile total median {(t[:, 7] - t[:, 0]).median().item():.0f} cycles")
for i, lab in enumerate(BL):
    print(f"   {lab:28s} median {d[:, i].median().item():9.0f}  mean {d[:, i].mean().item():9.0f}  max {d[:, i].max().item():9.0f}")
