"""The memory route of a ResGCP chain (ops.CHAIN_RECOMPUTE / GCPNET_CHAIN_RECOMPUTE=1; SURVEY.md section 7 step 6, the loop of
components/gcpnet.py:921-924): the forward keeps only the chain's inputs, the backward runs the chain's forward again.  Same launches on
the same inputs, so outputs and every gradient must equal the plain route's bit for bit -- at (128,16) (wave-per-tile kernels), at
(256,32) (workgroup kernels, block-by-block backward) and with the aggregation fused into the chain Function -- and the forward must
hold an order of magnitude less memory."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _layer_run(G, ops, dims, n, e, recompute, act="relu"):
    from tests.helpers import rand_graph

    torch.manual_seed(5)
    layer = G.GCPInteractions(dims, (32, 4), cfg=G.default_module_cfg(scalar_nonlinearity=act), layer_cfg=G.default_layer_cfg(),
                              dropout=0.0).cuda().train()
    ei, x = rand_graph(n, e, 7, sort_by_col=True)
    g = torch.Generator().manual_seed(1)
    ins = {k: v.cuda().requires_grad_() for k, v in dict(h=torch.randn(n, dims[0], generator=g), chi=torch.randn(n, dims[1], 3, generator=g),
                                                         e=torch.randn(e, 32, generator=g), xi=torch.randn(e, 4, 3, generator=g)).items()}
    lw = (torch.randn(n, dims[0], generator=g).cuda(), torch.randn(n, dims[1], 3, generator=g).cuda())
    fr = G.localize(x.cuda(), ei.cuda())
    saved = ops.CHAIN_RECOMPUTE
    try:
        ops.CHAIN_RECOMPUTE = recompute
        torch.cuda.synchronize()
        before = torch.cuda.memory_allocated()
        h, chi = layer((ins["h"], ins["chi"]), (ins["e"], ins["xi"]), ei.cuda(), fr)
        torch.cuda.synchronize()
        held = torch.cuda.memory_allocated() - before
        ((h * lw[0]).sum() + (chi * lw[1]).sum()).backward()
        torch.cuda.synchronize()
    finally:
        ops.CHAIN_RECOMPUTE = saved
    out = dict(h=h.detach().clone(), chi=chi.detach().clone())
    out.update({"d" + k: v.grad.clone() for k, v in ins.items()})
    out.update({"w." + k: p.grad.clone() for k, p in layer.named_parameters() if p.grad is not None})
    return out, held


@pytest.mark.parametrize("dims,n,e,act", [((128, 16), 700, 9000, "relu"), ((128, 16), 300, 4000, "silu"), ((256, 32), 400, 5000, "relu"),
                                          ((100, 16), 300, 3500, "relu")], ids=["128x16", "128x16-silu", "256x32", "100x16"])
def test_recompute_route_is_bit_identical_and_holds_less(dims, n, e, act):
    import gcpnet_amd as G
    from gcpnet_amd import ops

    plain, held_plain = _layer_run(G, ops, dims, n, e, False, act)
    rec, held_rec = _layer_run(G, ops, dims, n, e, True, act)
    assert plain.keys() == rec.keys()
    for k in plain:
        assert torch.equal(plain[k], rec[k]), f"{k}: the recompute route differs by {float((plain[k] - rec[k]).abs().max()):.3e}"
    # the chain's saved activations are ~85 % of what a layer's forward holds on the plain route
    assert held_rec < 0.45 * held_plain, (held_rec, held_plain)
