"""Per-phase s_memtime profile of the GCP2 forward / backward kernels on one residual message GCP (s,V)->(s,V).
usage: python tools/phase_timing.py [n_edges] [sdim] [vdim]"""
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
block = G.GCP2((S, V), (S, V), nonlinearities=("relu", None), bottleneck=4).cuda()
g = torch.Generator(device="cuda").manual_seed(0)
s = torch.randn(E, S, device="cuda", generator=g).requires_grad_()
v = torch.randn(E, V, 3, device="cuda", generator=g)
fr = torch.randn(E, 3, 3, device="cuda", generator=g)
ds, dv = torch.randn(E, S, device="cuda", generator=g), torch.randn(E, V, 3, device="cuda", generator=g)
spec = ops.Gcp2Spec(si=S, vi=V, so=S, vo=V, hidden=block.hidden_dim, use_frames=True, act_s="relu", act_v=None, slope=1e-2,
                    vmode=1, vector_residual=False, e3=False, s_plans=[None], v_plans=[None], residual=True, pack_cache={})
w = tuple(None if t is None else t.detach() for t in block._weights())
out_s, out_v = ops.gcp2(spec, [s], [v], fr, w)
saved = out_s.grad_fn.saved_tensors
pack, s_pre, gate = saved[-3], saved[-2], saved[-1]
if getattr(out_s.grad_fn, "s_pre_tb", False):  # (a single block through the workgroup kernels saves s_pre tile-blocked)
    s_pre = ops.TileBlocked(E, S, s_pre.device, owner=s_pre, offset=0, n=s_pre.numel()).to_rows()
ntiles = (E + 31) // 32
buf = torch.zeros(ntiles * 8, dtype=torch.int64, device="cuda")


def report(name, nst, labels):
    torch.cuda.synchronize()
    t = buf.view(ntiles, 8).cpu().double()
    d = t[:, 1:nst] - t[:, : nst - 1]
    tot = t[:, nst - 1] - t[:, 0]
    span = t[:, nst - 1].max() - t[:, 0].min()
    print(f"{name}: {ntiles} tiles, whole-launch span {span.item():.0f} ticks, tile total median {tot.median().item():.0f}")
    for i, lab in enumerate(labels):
        print(f"   {lab:28s} median {d[:, i].median().item():9.0f}  mean {d[:, i].mean().item():9.0f}  max {d[:, i].max().item():9.0f}")
    starts = (t[:, 0] - t[:, 0].min())
    q = torch.quantile(starts, torch.tensor([0.1, 0.25, 0.5, 0.75, 0.9], dtype=torch.float64))
    print("   tile start offsets (10/25/50/75/90 %):", [f"{x:.0f}" for x in q.tolist()])


with torch.no_grad():
    for _ in range(2):
        ops.gcp2(spec, [s.detach()], [v], fr, w)
    lib.gcpnet_debug_set_phase_timing(C.c_void_p(buf.data_ptr()), ntiles)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    ops.gcp2(spec, [s.detach()], [v], fr, w)
    b.record()
    torch.cuda.synchronize()
    print(f"fwd launch {a.elapsed_time(b) * 1e3:.0f} us (inference: no s_pre/gate stores)")
    report("fwd(no save)", 7, ["load tile", "vector prologue", "mfma loop", "s_out stage+store", "s_pre/gate gemm", "vector epilogue"])
    buf.zero_()
    lib.gcpnet_debug_set_phase_timing(None, 0)
s2 = s.detach().clone().requires_grad_()
lib.gcpnet_debug_set_phase_timing(C.c_void_p(buf.data_ptr()), ntiles)
ops.gcp2(spec, [s2], [v], fr, w)
report("fwd(training)", 7, ["load tile", "vector prologue", "mfma loop", "s_out stage+store", "s_pre/gate gemm", "vector epilogue"])
buf.zero_()
with torch.no_grad():
    ops.gcp2_backward_data(spec, E, [s.detach()], [v], fr, w, pack, s_pre, gate, ds, dv)
    if os.environ.get("GCP_BWD_FINE"):  # library built with GCPNET_HIPCC_EXTRA=-DGCP_BWD_FINE
        report("bwd(fine)", 8, ["tile loads committed", "small weights staged", "s_pre/d_s_out requested", "vh recompute + scratch",
                                "frame scalars", "transposed v copy", "vector epilogue adjoint"])
    else:
        report("bwd", 5, ["load + recompute vh", "vector epilogue adjoint", "ds_pre + W^T ds (mfma)", "vector prologue adjoint"])
lib.gcpnet_debug_set_phase_timing(None, 0)

# ---- register-resident chain of 7 residual blocks (stamps taken on the LAST block: steady state) ------------------------
mods = [G.GCP2((S, V), (S, V), nonlinearities=("relu", None), bottleneck=4).cuda() for _ in range(7)]
specs = [m.make_spec([None], [None], residual=True) for m in mods]
ws = [tuple(None if t is None else t.detach() for t in m._weights()) for m in mods]
for need_grad in (False, True):
    sx = s.detach().clone().requires_grad_(need_grad)
    buf.zero_()
    with torch.set_grad_enabled(need_grad):
        ops.gcp2_chain(specs, sx, v, fr, ws)
        lib.gcpnet_debug_set_phase_timing(C.c_void_p(buf.data_ptr()), ntiles)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ops.gcp2_chain(specs, sx, v, fr, ws)
        b.record()
        torch.cuda.synchronize()
        lib.gcpnet_debug_set_phase_timing(None, 0)
    print(f"chain x7 launch {a.elapsed_time(b) * 1e3:.0f} us (need_grad={need_grad})")
    t = buf.view(ntiles, 8).cpu().double()
    tot = t[:, 7] - t[:, 0]
    print(f"   last block total median {tot.median().item():.0f} ticks")
    for i, lab in enumerate(["[vh|vf] = Wdf v (mfma)", "norms, frame scalars -> LDS", "bias + mfma (state)", "mfma (norms/frames)",
                             "gate gemm", "s_pre/s_out stores + x update", "vector epilogue + stores"]):
        d = t[:, i + 1] - t[:, i]
        print(f"   {lab:30s} median {d.median().item():9.0f}  mean {d.mean().item():9.0f}")

# ---- backward of the same chain (stamps 2..6 taken on the FIRST block = last loop iteration: steady state) ---------------
sx = s.detach().clone().requires_grad_(True)
vx = v.detach().clone().requires_grad_(True)
for m in mods:
    for q in m.parameters():
        q.requires_grad_(True)
ws_g = [m._weights() for m in mods]
o_s, o_v = ops.gcp2_chain(specs, sx, vx, fr, ws_g)
buf.zero_()
torch.cuda.synchronize()
lib.gcpnet_debug_set_phase_timing(C.c_void_p(buf.data_ptr()), ntiles)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
torch.autograd.backward([o_s, o_v], [ds, dv])
b.record()
torch.cuda.synchronize()
lib.gcpnet_debug_set_phase_timing(None, 0)
print(f"chain x7 backward (data kernel + weight-gradient GEMMs) {a.elapsed_time(b) * 1e3:.0f} us")
t = buf.view(ntiles, 8).cpu().double()
print(f"   chain bwd kernel: tile total median {(t[:, 7] - t[:, 0]).median().item():.0f} ticks")
for i, lab in enumerate(["prologue (loads of the last block)", "blocks n-1 .. 1", "A-C: recompute, epilogue adjoint, 1st partials (block 0)",
                         "D-E: gate adjoint, ds_pre, W^T ds_pre (mfma)", "F: vector prologue adjoint", "G: 2nd partials, commit next",
                         "end"]):
    d = t[:, i + 1] - t[:, i]
    print(f"   {lab:45s} median {d.median().item():9.0f}  mean {d.mean().item():9.0f}")
