"""d vector_out_scale.weight from the block's inputs (gcp2_wgrad_job_t.gate_lin, include/gcpnet_hip.h) and a chain forward that does
not store s_pre (ops.CHAIN_SKIP_S_PRE): with an identity gate activation -- `nonlinearities: [relu, None]`, every shipped configuration,
components/gcpnet.py:345-347 -- s_pre = [s | norms | frame scalars] W^T + b, so dgate^T s_pre = (dgate^T [s | ...]) W^T + (sum dgate) b^T.
Everything except the gate weight gradient must stay bit-identical; that one must agree to fp32 round-off with the s_pre form and
with the oracle; and the forward must hold 512 bytes per row and block less."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(G, ops, dims, n, e, skip, lin):
    from tests.helpers import rand_graph

    torch.manual_seed(5)
    layer = G.GCPInteractions(dims, (32, 4), cfg=G.default_module_cfg(), layer_cfg=G.default_layer_cfg(), dropout=0.0).cuda().train()
    ei, x = rand_graph(n, e, 7, sort_by_col=True)
    g = torch.Generator().manual_seed(1)
    ins = {k: v.cuda().requires_grad_() for k, v in dict(h=torch.randn(n, dims[0], generator=g), chi=torch.randn(n, dims[1], 3, generator=g),
                                                         e=torch.randn(e, 32, generator=g), xi=torch.randn(e, 4, 3, generator=g)).items()}
    lw = (torch.randn(n, dims[0], generator=g).cuda(), torch.randn(n, dims[1], 3, generator=g).cuda())
    fr = G.localize(x.cuda(), ei.cuda())
    saved = ops.CHAIN_SKIP_S_PRE, ops.GATE_GRADS_FROM_INPUTS
    try:
        ops.CHAIN_SKIP_S_PRE, ops.GATE_GRADS_FROM_INPUTS = skip, lin
        torch.cuda.synchronize()
        before = torch.cuda.memory_allocated()
        h, chi = layer((ins["h"], ins["chi"]), (ins["e"], ins["xi"]), ei.cuda(), fr)
        torch.cuda.synchronize()
        held = torch.cuda.memory_allocated() - before
        ((h * lw[0]).sum() + (chi * lw[1]).sum()).backward()
        torch.cuda.synchronize()
    finally:
        ops.CHAIN_SKIP_S_PRE, ops.GATE_GRADS_FROM_INPUTS = saved
    out = dict(h=h.detach().clone(), chi=chi.detach().clone())
    out.update({"d" + k: v.grad.clone() for k, v in ins.items()})
    out.update({"w." + k: p.grad.clone() for k, p in layer.named_parameters() if p.grad is not None})
    return out, held


@pytest.mark.parametrize("dims,n,e", [((128, 16), 700, 9000), ((64, 16), 300, 4000), ((100, 16), 300, 3500)], ids=["128x16", "64x16", "100x16"])
def test_gate_gradients_from_inputs_and_no_stored_s_pre(dims, n, e):
    import gcpnet_amd as G
    from gcpnet_amd import ops

    old, held_old = _run(G, ops, dims, n, e, skip=False, lin=False)
    lin, held_lin = _run(G, ops, dims, n, e, skip=False, lin=True)
    new, held_new = _run(G, ops, dims, n, e, skip=True, lin=True)
    assert old.keys() == lin.keys() == new.keys()
    n_gate = 0
    for k in old:
        if "message_fusion" in k and "vector_out_scale.weight" in k and ".0." not in k.split("message_fusion")[1][:3]:
            n_gate += 1
            scale = float(old[k].abs().max())
            for other in (lin, new):
                err = float((old[k] - other[k]).abs().max())
                assert err <= 2e-5 * scale, f"{k}: {err:.3e} (scale {scale:.3e})"
            assert torch.equal(lin[k], new[k]), k
        else:
            assert torch.equal(old[k], lin[k]) and torch.equal(old[k], new[k]), f"{k} changed"
    assert n_gate == 7
    # 7 blocks x 4 so bytes per edge row less
    assert held_old - held_new >= 0.9 * 7 * e * 4 * ((dims[0] + 31) // 32 * 32), (held_old, held_new)


def test_both_weight_gradients_of_a_block_from_one_product(monkeypatch):
    """GCPNET_TN_MID=1 (opt-in): the job builder of gcpnet_gcp2_weight_grads puts [ds_pre | dgate]^T [s | norms | frame scalars | 1] of a
    gated (128,16) block into ONE product on the five-wave 160 x 160 kernel (gcp_tn_problem_t.m_split).  Same sums in another kernel:
    data gradients bit-identical, every weight gradient within fp32 round-off of the two-product route."""
    import gcpnet_amd as G
    from gcpnet_amd import ops

    two, _ = _run(G, ops, (128, 16), 700, 9000, skip=True, lin=True)
    monkeypatch.setenv("GCPNET_TN_MID", "1")
    one, _ = _run(G, ops, (128, 16), 700, 9000, skip=True, lin=True)
    assert two.keys() == one.keys()
    for k in two:
        if k.startswith("w."):
            scale = max(float(two[k].abs().max()), 1e-6)
            err = float((two[k] - one[k]).abs().max())
            assert err <= 2e-5 * scale, f"{k}: {err:.3e} (scale {scale:.3e})"
        else:
            assert torch.equal(two[k], one[k]), f"{k} changed"
