"""Per-phase s_memtime profile of the generic GCP2 forward / backward kernels on the first message GCP of a layer:
[h_row | e | h_col] -> (s, V), gathered inputs, project-then-gather for the node scalars.  usage: python tools/phase_timing_gcp0.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gcpnet_amd as G  # noqa: E402
from gcpnet_amd import _lib, ops  # noqa: E402
from gcpnet_amd.synthetic import make_inputs  # noqa: E402

lib = _lib.load()
host = make_inputs(10000, 16, (128, 16), (32, 4), seed=0)
dev = {k: v.cuda() for k, v in host.items()}
E = dev["edge_index"].shape[1]
torch.manual_seed(0)
layer = G.GCPInteractions((128, 16), (32, 4), cfg=G.default_module_cfg(), layer_cfg=G.default_layer_cfg(), dropout=0.0).cuda()
frames = G.localize(dev["x"], dev["edge_index"])
first = layer.interaction.message_fusion[0]
plan = ops.GraphPlan.get(dev["edge_index"], 10000)
h, chi = dev["h"].clone().requires_grad_(), dev["chi"].clone().requires_grad_()
e, xi = dev["e"].clone().requires_grad_(), dev["xi"].clone().requires_grad_()
ntiles = (E + 31) // 32
buf = torch.zeros(ntiles * 8, dtype=torch.int64, device="cuda")


def report(name, nst, labels):
    torch.cuda.synchronize()
    t = buf.view(ntiles, 8).cpu().double()
    d = t[:, 1:nst] - t[:, : nst - 1]
    print(f"{name}: tile total median {(t[:, nst - 1] - t[:, 0]).median().item():.0f} ticks")
    for i, lab in enumerate(labels):
        print(f"   {lab:28s} median {d[:, i].median().item():9.0f}  mean {d[:, i].mean().item():9.0f}")


def run():
    return first.apply_rows([h, e, h], [plan.row, None, plan.col], [chi, xi, chi], [plan.row, None, plan.col], frames)


for _ in range(2):
    m = run()
lib.gcpnet_debug_set_phase_timing(C.c_void_p(buf.data_ptr()), ntiles)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
m = run()
b.record()
torch.cuda.synchronize()
print(f"forward (projections + kernel) {a.elapsed_time(b) * 1e3:.0f} us")
report("fwd(training)", 7, ["load tile", "vector prologue", "mfma loop", "s_out stage+store", "s_pre/gate gemm", "vector epilogue"])
buf.zero_()
a.record()
torch.autograd.backward([m[0], m[1]], [torch.randn_like(m[0]), torch.randn_like(m[1])])
b.record()
torch.cuda.synchronize()
print(f"backward (kernel + weight-gradient GEMMs + projections) {a.elapsed_time(b) * 1e3:.0f} us")
if os.environ.get("GCP_BWD_FINE"):  # library built with GCPNET_HIPCC_EXTRA=-DGCP_BWD_FINE
    report("bwd(fine)", 8, ["tile loads committed", "small weights staged", "s_pre/d_s_out requested", "vh recompute + scratch",
                            "frame scalars", "transposed v copy", "vector epilogue adjoint"])
else:
    report("bwd", 5, ["load + recompute vh", "vector epilogue adjoint", "ds_pre + W^T ds (mfma)", "vector prologue adjoint"])
lib.gcpnet_debug_set_phase_timing(None, 0)
