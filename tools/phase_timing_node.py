"""Per-phase s_memtime profile of the generic GCP2 kernels on the node-level feed-forward blocks of GCPInteractions
(FF0 (s,V)->(4s,2V), FF1 (4s,2V)->(s,V); few rows, so about one wave per CU).
usage: python tools/phase_timing_node.py [n_nodes] [sdim] [vdim]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gcpnet_amd as G  # noqa: E402
from gcpnet_amd import _lib, ops  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 128
V = int(sys.argv[3]) if len(sys.argv) > 3 else 16
lib = _lib.load()
torch.manual_seed(0)
ntiles = (N + 31) // 32
buf = torch.zeros(ntiles * 8, dtype=torch.int64, device="cuda")
g = torch.Generator(device="cuda").manual_seed(0)


def report(name, nst, labels):
    torch.cuda.synchronize()
    t = buf.view(ntiles, 8).cpu().double()
    d = t[:, 1:nst] - t[:, : nst - 1]
    tot = t[:, nst - 1] - t[:, 0]
    span = t[:, nst - 1].max() - t[:, 0].min()
    print(f"{name}: {ntiles} tiles, whole-launch span {span.item():.0f} ticks (shader clock), tile total median {tot.median().item():.0f}")
    for i, lab in enumerate(labels):
        print(f"   {lab:28s} median {d[:, i].median().item():9.0f}  mean {d[:, i].mean().item():9.0f}  max {d[:, i].max().item():9.0f}")
    buf.zero_()


FWD = ["load tile", "vector prologue", "mfma loop", "s_out stage+store", "s_pre/gate gemm", "vector epilogue"]
for name, din, dout, acts in (("FF0", (S, V), (4 * S, 2 * V), ("relu", None)), ("FF1", (4 * S, 2 * V), (S, V), (None, None))):
    block = G.GCP2(din, dout, nonlinearities=acts, bottleneck=4).cuda()
    s = torch.randn(N, din[0], device="cuda", generator=g)
    v = torch.randn(N, din[1], 3, device="cuda", generator=g)
    fr = torch.randn(N, 3, 3, device="cuda", generator=g)
    ds, dv = torch.randn(N, dout[0], device="cuda", generator=g), torch.randn(N, dout[1], 3, device="cuda", generator=g)
    spec = block.make_spec([None], [None])
    w = tuple(None if t is None else t.detach() for t in block._weights())
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    for it in range(3):
        sx = s.clone().requires_grad_()
        vx = v.clone().requires_grad_()
        if it == 2:
            lib.gcpnet_debug_set_phase_timing(C.c_void_p(buf.data_ptr()), ntiles)
        ev[0].record()
        out = ops.gcp2(spec, [sx], [vx], fr, w)
        ev[1].record()
        if it == 2:
            report(f"{name} fwd(training) {din}->{dout}", 7, FWD)
        ev[2].record()
        torch.autograd.backward(out, (ds, dv))
        ev[3].record()
        torch.cuda.synchronize()
    print(f"{name}: fwd {ev[0].elapsed_time(ev[1]) * 1e3:.0f} us, bwd (data + weight grads) {ev[2].elapsed_time(ev[3]) * 1e3:.0f} us")
    report(f"{name} bwd", 5, ["load + recompute vh", "vector epilogue adjoint", "ds_pre + W^T ds (mfma)", "vector prologue adjoint"])
    lib.gcpnet_debug_set_phase_timing(None, 0)
