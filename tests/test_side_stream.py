"""GPU tests of the hazards around the weight-gradient (side) stream and the caches keyed on tensor identity -- the advisor findings
of rounds 1 and 2, at sizes where the GPU is busy long enough for a race to show:

  * gradient accumulation (`backward()` twice without zeroing) and `zero_grad(set_to_none=False)` on the full configs[1] graph;
  * the fused workgroup backward's gradients are ADOPTED by autograd (not cloned before the side stream has written them), with the
    allocator poisoned;
  * a weight used by two autograd Functions of one graph (`autoregressive_forward` calls `interaction` twice; a module applied
    twice) falls back to the caller's stream and gives the same gradients as with the side stream off;
  * GCP3 `feedforward_out` (two launches sharing weights) at 160 000 rows against the oracle;
  * weights updated through `p.data` + `ops.invalidate_packs()`;
  * a frames tensor dropped and recreated at the same address (`GraphPlan.node_frames`).
Every comparison between two routes of the same arithmetic is BITWISE: the kernels use fixed summation orders."""
import functools
import gc

import pytest
import torch

from oracle import gcp_oracle as O
from tests.helpers import close, poison_allocations

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import gcpnet_amd

    return gcpnet_amd


@pytest.fixture(scope="module")
def c2(G):
    """One GCPInteractions layer on the full configs[1] graph (10 000 nodes / ~160 000 edges, (128,16)), inputs on the GPU."""
    from gcpnet_amd.synthetic import make_inputs

    ins = make_inputs(10000, 16, node_dims=(128, 16), seed=0)
    ei, x = ins.pop("edge_index"), ins.pop("x")
    fr = O.localize(x, ei).cuda()
    torch.manual_seed(1)
    layer = G.GCPInteractions((128, 16), (32, 4), cfg=G.default_module_cfg(), layer_cfg=G.default_layer_cfg(), dropout=0.0).cuda().eval()
    dev = {k: t.cuda() for k, t in ins.items()}
    g = torch.Generator().manual_seed(2)
    lw = (torch.randn(10000, 128, generator=g).cuda(), torch.randn(10000, 16, 3, generator=g).cuda())
    return layer, dev, ei.cuda(), fr, lw


def _step(layer, dev, ei, fr, lw):
    h, chi = layer((dev["h"], dev["chi"]), (dev["e"], dev["xi"]), ei, fr)
    ((h * lw[0]).sum() + (chi * lw[1]).sum()).backward()


def _grads(layer):
    torch.cuda.synchronize()
    return {k: p.grad.clone() for k, p in layer.named_parameters() if p.grad is not None}


def _same_bits(a: dict, b: dict, scale=None):
    assert a.keys() == b.keys()
    bad = [k for k in a if not torch.equal(a[k] if scale is None else a[k] * scale, b[k])]
    assert not bad, f"differ bitwise: {bad[:6]} ({len(bad)} of {len(a)})"


@pytest.mark.parametrize("side", [True, False])
def test_backward_twice_without_zeroing_full_c2(G, c2, side):
    """Gradient accumulation: the second backward finds .grad set, so its weight gradients must be complete on the caller's stream
    when autograd adds them (ops._side_stream_ok).  2 x g is exact in fp32, so the sum must equal twice the single gradient bitwise."""
    from gcpnet_amd import ops

    layer = c2[0]
    ops.set_weight_grad_stream(side)
    try:
        layer.zero_grad(set_to_none=True)
        _step(*c2)
        single = _grads(layer)
        layer.zero_grad(set_to_none=True)
        _step(*c2)
        _step(*c2)  # (no zeroing in between)
        _same_bits(single, _grads(layer), scale=2.0)
        # zero_grad(set_to_none=False): .grad is a zero tensor, the step accumulates into it
        layer.zero_grad(set_to_none=False)
        _step(*c2)
        _same_bits(single, _grads(layer))
    finally:
        ops.set_weight_grad_stream(True)


def test_side_stream_equals_main_stream_full_c2(G, c2):
    from gcpnet_amd import ops

    layer = c2[0]
    out = []
    for side in (False, True, True):
        ops.set_weight_grad_stream(side)
        layer.zero_grad(set_to_none=True)
        _step(*c2)
        out.append(_grads(layer))
    ops.set_weight_grad_stream(True)
    _same_bits(out[0], out[1])
    _same_bits(out[0], out[2])


def test_fused_backward_gradients_are_adopted_not_cloned(G):
    """a gradient tensor that is still referenced from Python when AccumulateGrad receives it is cloned on
    the caller's stream -- before the side stream's reduction has written it.  The fused workgroup backward must hand its buffers
    over: .grad IS the buffer the reduction writes, and with the allocator poisoned (fresh memory = NaN) the values are right."""
    from gcpnet_amd import ops

    rows = 160000
    torch.manual_seed(5)
    mod = G.GCP2((128, 16), (128, 1), nonlinearities=("relu", None), bottleneck=4).cuda()  # the position-update block: fused form
    g = torch.Generator().manual_seed(6)
    s, v = torch.randn(rows, 128, generator=g).cuda(), torch.randn(rows, 16, 3, generator=g).cuda()
    fr = torch.randn(rows, 3, 3, generator=g).cuda()
    ei = torch.stack((torch.arange(rows), torch.arange(rows))).cuda()
    ls, lv = torch.randn(rows, 128, generator=g).cuda(), torch.randn(rows, 1, 3, generator=g).cuda()

    def step():
        mod.zero_grad(set_to_none=True)
        so, vo = mod((s.clone().requires_grad_(), v.clone().requires_grad_()), ei, fr)
        ((so * ls).sum() + (vo * lv).sum()).backward()

    ops.set_weight_grad_stream(False)
    step()
    want = _grads(mod)
    ops.set_weight_grad_stream(True)
    restore = poison_allocations()
    ops._DEBUG_GRAD_PTRS = []
    try:
        before = ops.WG_STATS["bwd"]
        step()
        assert ops.WG_STATS["bwd"] == before + 1
        ptrs = set(ops._DEBUG_GRAD_PTRS)
        assert ptrs, "the fused workgroup backward did not run"
        assert mod.scalar_out.weight.grad.data_ptr() in ptrs, "scalar_out.weight.grad is a copy, not the kernel's buffer"
        assert mod.vector_out_scale.weight.grad.data_ptr() in ptrs
        got = _grads(mod)
    finally:
        restore()
        ops._DEBUG_GRAD_PTRS = None
    for k in want:
        assert bool(torch.isfinite(got[k]).all()), f"{k}: uninitialised memory in the gradient"
    _same_bits(want, got)


def test_module_applied_twice_in_one_graph(G):
    """the same leaf weights feed two autograd Functions; the engine sums the two gradients inside the
    backward pass, on the caller's stream -- both must be complete there (ops._note_uses)."""
    from gcpnet_amd import ops

    rows = 120000
    torch.manual_seed(7)
    mod = G.GCP2((128, 16), (128, 16), nonlinearities=("silu", None), bottleneck=4).cuda()
    g = torch.Generator().manual_seed(8)
    s, v = torch.randn(rows, 128, generator=g).cuda(), torch.randn(rows, 16, 3, generator=g).cuda()
    fr = torch.randn(rows, 3, 3, generator=g).cuda()
    ei = torch.stack((torch.arange(rows), torch.arange(rows))).cuda()
    ls, lv = torch.randn(rows, 128, generator=g).cuda(), torch.randn(rows, 16, 3, generator=g).cuda()

    def step():
        mod.zero_grad(set_to_none=True)
        a = mod((s, v), ei, fr)
        b = mod((a[0], a[1]), ei, fr)  # second use of the same weights
        ((b[0] * ls).sum() + (b[1] * lv).sum()).backward()
        return _grads(mod)

    ops.set_weight_grad_stream(False)
    want = step()
    ops.set_weight_grad_stream(True)
    restore = poison_allocations()
    try:
        got = step()
        again = step()
    finally:
        restore()
    _same_bits(want, got)
    _same_bits(want, again)


def test_autoregressive_forward_shares_interaction_weights(G):
    """`autoregressive_forward` (reference gcpnet.py:1066-1116) runs `interaction` twice: same check on a GPU-bound graph."""
    from gcpnet_amd import ops
    from gcpnet_amd.synthetic import make_inputs

    ins = make_inputs(6000, 16, node_dims=(128, 16), seed=4)
    ei, x = ins.pop("edge_index"), ins.pop("x")
    fr = O.localize(x, ei).cuda()
    torch.manual_seed(9)
    layer = G.GCPInteractions((128, 16), (32, 4), cfg=G.default_module_cfg(), layer_cfg=G.default_layer_cfg(), dropout=0.0,
                              autoregressive=True).cuda().eval()
    dev = {k: t.cuda() for k, t in ins.items()}
    g = torch.Generator().manual_seed(10)
    ar = (torch.randn(6000, 128, generator=g).cuda(), torch.randn(6000, 16, 3, generator=g).cuda())
    lw = (torch.randn(6000, 128, generator=g).cuda(), torch.randn(6000, 16, 3, generator=g).cuda())
    eid = ei.cuda()

    def step():
        layer.zero_grad(set_to_none=True)
        h, chi = layer((dev["h"], dev["chi"]), (dev["e"], dev["xi"]), eid, fr, node_rep_regressive=ar)
        ((h * lw[0]).sum() + (chi * lw[1]).sum()).backward()
        return _grads(layer)

    ops.set_weight_grad_stream(False)
    want = step()
    ops.set_weight_grad_stream(True)
    restore = poison_allocations()
    try:
        got = step()
    finally:
        restore()
    _same_bits(want, got)


def test_gcp3_feedforward_out_160k_rows_vs_oracle(G):
    """GCP3 `feedforward_out` (reference gcpnet.py:529-533): two launches sharing vector_down / vector_down_frames
    (Gcp2Spec.shared_weights keeps their gradients on the caller's stream), at a GPU-bound size, weight gradients against the oracle."""
    rows = 160000
    torch.manual_seed(11)
    mod = G.GCP3((64, 16), (64, 16), nonlinearities=("silu", "silu"), feedforward_out=True, bottleneck=4).cuda()
    g = torch.Generator().manual_seed(12)
    s, v = torch.randn(rows, 64, generator=g), torch.randn(rows, 16, 3, generator=g)
    fr = torch.randn(rows, 3, 3, generator=g)
    ei = torch.stack((torch.arange(rows), torch.arange(rows)))
    P = {k: t.detach().cpu().double().requires_grad_() for k, t in mod.state_dict().items()}
    ws, wv = O.gcp2(P, "", s.double(), v.double(), ei, fr.double(), nonlinearities=("silu", "silu"), scalar_out_nonlinearity="silu")
    ls, lv = torch.randn(rows, 64, generator=g), torch.randn(rows, 16, 3, generator=g)
    ((ws * ls.double()).sum() + (wv * lv.double()).sum()).backward()
    restore = poison_allocations()
    try:
        outs = []
        for _ in range(2):
            mod.zero_grad(set_to_none=True)
            gs, gv = mod((s.cuda(), v.cuda()), ei.cuda(), fr.cuda())
            ((gs * ls.cuda()).sum() + (gv * lv.cuda()).sum()).backward()
            outs.append(_grads(mod))
    finally:
        restore()
    _same_bits(outs[0], outs[1])
    close(gs.detach().cpu(), ws.detach().float(), atol=2e-5 * float(ws.detach().abs().max()), rtol=1e-5)
    for k, t in outs[0].items():
        want = P[k].grad.float()
        close(t.cpu(), want, atol=1e-4 * float(want.abs().max()), rtol=1e-4)  # (sums over 160 000 rows in fp32 against float64)


def test_weights_updated_through_p_data_need_invalidate_packs(G):
    """The packed-weight caches are keyed on (data_ptr, _version); an update through `p.data` changes neither.  Documented
    contract (ops.invalidate_packs): call it after such an update -- without it the stale image is used."""
    from gcpnet_amd import ops

    rows = 512
    torch.manual_seed(13)
    mod = G.GCP2((64, 16), (64, 16), nonlinearities=("silu", None), bottleneck=4).cuda()
    g = torch.Generator().manual_seed(14)
    s, v, fr = torch.randn(rows, 64, generator=g).cuda(), torch.randn(rows, 16, 3, generator=g).cuda(), torch.randn(rows, 3, 3, generator=g).cuda()
    ei = torch.stack((torch.arange(rows), torch.arange(rows))).cuda()
    with torch.no_grad():
        y0 = mod((s, v), ei, fr)[0].clone()
        for p in mod.parameters():  # an SGD-style step through .data: no version bump, same storage
            p.data.add_(0.05 * torch.randn(p.shape, generator=g).cuda())
        stale = mod((s, v), ei, fr)[0].clone()
        ops.invalidate_packs()
        fresh = mod((s, v), ei, fr)[0].clone()
    assert not torch.equal(fresh, y0)
    P = {k: t.detach().cpu() for k, t in mod.state_dict().items()}
    want, _ = O.gcp2(P, "", s.cpu(), v.cpu(), ei.cpu(), fr.cpu(), nonlinearities=("silu", None))
    close(fresh.cpu(), want, atol=1e-5 * float(want.abs().max()), rtol=1e-5)
    # without the call the packed image of scalar_out / vector_out_scale is the OLD one (the small vector weights and the biases
    # are read from the tensors themselves): a result that belongs to neither weight set -- this is what the contract is about
    stale_err = float((stale.cpu() - want).abs().max())
    assert stale_err > 1e-2 * float(want.abs().max()), "the cache noticed a p.data update by itself: invalidate_packs()'s contract is outdated"
    # an update through the tensor itself (what torch.optim does) bumps the version and needs no call
    with torch.no_grad():
        for p in mod.parameters():
            p.add_(0.05 * torch.randn(p.shape, generator=g).cuda())
        again = mod((s, v), ei, fr)[0]
    P = {k: t.detach().cpu() for k, t in mod.state_dict().items()}
    want, _ = O.gcp2(P, "", s.cpu(), v.cpu(), ei.cpu(), fr.cpu(), nonlinearities=("silu", None))
    close(again.cpu(), want, atol=1e-5 * float(want.abs().max()), rtol=1e-5)


def test_node_frames_cache_follows_the_tensor_object_not_its_address(G):
    """GraphPlan.node_frames caches the mean out-edge frames per frames TENSOR (weak reference + version): a frames tensor that
    is dropped and recreated -- the caching allocator hands back the same address -- must be recomputed."""
    from gcpnet_amd.ops import GraphPlan
    from tests.helpers import rand_graph

    n, e = 300, 4000
    ei, x = rand_graph(n, e, 15)
    eid = ei.cuda()
    plan = GraphPlan.get(eid, n)
    seen = []
    for k in range(3):
        fr = O.localize(x * (1.0 + k), ei, norm_x_diff=False).cuda()
        got = plan.node_frames(fr).clone()
        want = O.scatter(O.localize(x * (1.0 + k), ei, norm_x_diff=False).reshape(e, 9), ei[0], n, "mean").reshape(n, 3, 3)
        close(got.cpu(), want, atol=1e-5 * float(want.abs().max()), rtol=1e-5)
        seen.append(fr.data_ptr())
        del fr
        gc.collect()
    assert len(set(seen)) < 3, "the allocator never reused the address: the test did not exercise the hazard"
    # in-place refill through torch bumps the version: recomputed as well
    fr = O.localize(x, ei, norm_x_diff=False).cuda()
    a = plan.node_frames(fr).clone()
    fr.mul_(2.0)
    b = plan.node_frames(fr)
    close(b.cpu(), (2.0 * a).cpu(), atol=1e-6 * float(a.abs().max()), rtol=1e-6)
