"""Debug: GCPInteractions2 layer grads with the workgroup backward on / off; prints per-tensor differences."""
import functools, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gcpnet_amd as G
from gcpnet_amd import ops
from oracle import gcp_oracle as O
from tests.helpers import rand_graph

torch.manual_seed(21)
n, e, dims = 600, 6000, (128, 16)
upd = False
kw = dict(use_scalar_message_attention=True, aggregate_with_row=not upd, num_feedforward_layers=2 if upd else 1)
cfg = G.default_module_cfg(selected_GCP=functools.partial(G.GCP3), scalar_nonlinearity="silu")
layer = G.GCPInteractions2(dims, (32, 4), cfg=cfg, layer_cfg=G.default_layer_cfg(**kw), dropout=0.0, updating_node_positions=upd).cuda().eval()
ei, x = rand_graph(n, e, 22, sort_by_col=True)
fr = O.localize(x, ei)
g = torch.Generator().manual_seed(23)
ins = dict(h=torch.randn(n, dims[0], generator=g), chi=torch.randn(n, dims[1], 3, generator=g),
           e=torch.randn(e, 32, generator=g), xi=torch.randn(e, 4, 3, generator=g))
lw = None


def run():
    global lw
    gi = {k: t.clone().cuda().requires_grad_() for k, t in ins.items()}
    for p in layer.parameters():
        p.grad = None
    got = layer((gi["h"], gi["chi"]), (gi["e"], gi["xi"]), ei.cuda(), fr.cuda())
    gl = list(got)
    if lw is None:
        lw = [torch.randn(t.shape, generator=g).cuda() for t in gl]
    sum((t * w).sum() for t, w in zip(gl, lw)).backward()
    torch.cuda.synchronize()
    out = {"out_" + str(i): t.detach().cpu() for i, t in enumerate(gl)}
    out.update({"d_" + k: t.grad.cpu() for k, t in gi.items()})
    out.update({"w_" + k: p.grad.cpu().clone() for k, p in layer.named_parameters() if p.grad is not None})
    return out


ops.USE_WG_BACKWARD = False
ref = run()
ops.USE_WG_BACKWARD = True
before = dict(ops.WG_STATS)
alt = run()
print("wg launches:", {k: ops.WG_STATS[k] - before[k] for k in before})
for k in ref:
    d = (ref[k] - alt[k]).abs().max().item()
    s = ref[k].abs().max().item()
    flag = "  <-----" if d > 1e-4 * max(s, 1e-6) else ""
    print(f"{k:70s} max diff {d:.3e} scale {s:.3e}{flag}")
