"""Tail split of the wave-per-tile chain backward (gcpnet_gcp2_chain_backward_split, include/gcpnet_hip.h): a tile's chain run as two
workgroups that hand d(s) / d(V) over through d_s_in / d_v_in and a flag word must give bit-identical results to one workgroup per
tile -- for every cut, for the plain and the gathered (aggregation-fused) form, and under a REVERSED workgroup order, where the
second halves are dispatched first and either take a tile over or wait for a first half that is running (no dispatch order is assumed
by the protocol)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import gcpnet_amd
    return gcpnet_amd


def _chain(G, dims, rows, act):
    from gcpnet_amd import ops

    torch.manual_seed(3)
    layer = G.GCPInteractions(dims, (32, 4), cfg=G.default_module_cfg(scalar_nonlinearity=act), layer_cfg=G.default_layer_cfg(),
                              dropout=0.0).cuda()
    blocks = list(layer.interaction.message_fusion[1:])
    specs = [b.make_spec([None], [None], residual=True) for b in blocks]
    ws = [tuple(None if t is None else t.detach().requires_grad_() for t in b._weights()) for b in blocks]
    g = torch.Generator(device="cuda").manual_seed(0)
    s = torch.randn(rows, dims[0], device="cuda", generator=g).requires_grad_()
    v = torch.randn(rows, dims[1], 3, device="cuda", generator=g).requires_grad_()
    x = torch.randn(rows, 3, 3, device="cuda", generator=g)
    frames = x / x.norm(dim=-1, keepdim=True)
    saved = ops.CHAIN_SKIP_S_PRE
    try:
        ops.CHAIN_SKIP_S_PRE = False  # (these tests run the backward with and without the sign masks: s_pre must exist)
        out = ops.gcp2_chain(specs, s, v, frames, ws)
    finally:
        ops.CHAIN_SKIP_S_PRE = saved
    s0, v0, ws_, packs, outs = out[0].grad_fn.state
    n = len(blocks)
    ins = [(s0, v0) if k == 0 else (outs[k - 1][0], outs[k - 1][1]) for k in range(n)]
    return ops, specs, ins, outs, frames, ws_, packs, n, g, out


def _flatten(res):
    from gcpnet_amd import ops

    d_s, d_v, scrs = res
    out = [d_s, d_v]
    for t in scrs:
        for k in ("ds_pre", "ext", "dgate", "w_part"):
            if k in t:
                r = t[k]
                if isinstance(r, ops.TileBlocked):
                    out.append(r.data.clone())
                else:
                    off = (r.ptr - r.owner.data_ptr()) // 4
                    out.append(r.owner[off:off + r.shape[0] * r.shape[1]].clone())
    return out


@pytest.mark.parametrize("gathered", [False, True], ids=["plain", "gathered"])
@pytest.mark.parametrize("dims,act", [((128, 16), "relu"), ((64, 8), "silu"), ((100, 16), "relu")], ids=["128x16", "64x8-silu", "100x16-padded"])
def test_forced_splits_are_bit_identical(G, dims, act, gathered):
    from gcpnet_amd import _lib

    rows = 1000 + 13
    ops, specs, ins, outs, frames, ws_, packs, n, g, _keep = _chain(G, dims, rows, act)
    lib = _lib.load()
    if gathered:
        n_seg = 97
        idx = torch.sort(torch.randint(0, n_seg, (rows,), device="cuda", generator=g)).values.to(torch.int32)
        plan = ops.GatherPlan(idx, n_seg)
        ds = torch.randn(n_seg, dims[0], device="cuda", generator=g)
        dv = torch.randn(n_seg, dims[1], 3, device="cuda", generator=g)
        agg = (plan, True)
    else:
        ds = torch.randn(rows, dims[0], device="cuda", generator=g)
        dv = torch.randn(rows, dims[1], 3, device="cuda", generator=g)
        agg = None

    def run():
        with torch.no_grad():
            res = ops.gcp2_chain_backward_data(specs, rows, ins, outs, frames, ws_, packs, ds, dv, [True] * n, out_agg=agg)
        torch.cuda.synchronize()
        assert res is not None
        return _flatten(res)

    saved = ops.CHAIN_TAIL_SPLIT
    try:
        ops.CHAIN_TAIL_SPLIT = False
        want = run()
        ops.CHAIN_TAIL_SPLIT = True
        tiles = (rows + 31) // 32
        for n_split, k_split, rev in [(1, 1, 0), (7, 3, 0), (tiles, n - 1, 0), (tiles, 4, 0), (tiles, 2, 1), (5, 5, 1), (tiles + 9, 1, 1)]:
            lib.gcpnet_debug_force_chain_split(n_split, k_split, rev)
            for rep in range(2):  # (the flag words are zeroed by every call)
                got = run()
                assert len(got) == len(want)
                for i, (a, b) in enumerate(zip(got, want)):
                    assert torch.equal(a, b), f"split ({n_split}, {k_split}, rev {rev}) rep {rep}: tensor {i} differs by {float((a - b).abs().max()):.3e}"
    finally:
        lib.gcpnet_debug_force_chain_split(-1, 0, 0)
        ops.CHAIN_TAIL_SPLIT = saved


def test_planned_split_at_configs1_size_is_bit_identical_and_used(G):
    """159 913 rows = 4 998 tiles on 2 048 wave slots: the planner must cut tiles (the case it was built for), results unchanged."""
    from gcpnet_amd import _lib

    rows = 159913
    ops, specs, ins, outs, frames, ws_, packs, n, g, _keep = _chain(G, (128, 16), rows, "relu")
    lib = _lib.load()
    sp0 = specs[0]
    assert lib.gcpnet_gcp2_chain_backward_flags(rows, n, sp0.si, sp0.vi, sp0.so, sp0.vo, sp0.hidden, int(sp0.use_frames)) > 0
    ds = torch.randn(rows, 128, device="cuda", generator=g)
    dv = torch.randn(rows, 16, 3, device="cuda", generator=g)

    def run():
        with torch.no_grad():
            res = ops.gcp2_chain_backward_data(specs, rows, ins, outs, frames, ws_, packs, ds, dv, [True] * n)
        torch.cuda.synchronize()
        return _flatten(res)

    saved = ops.CHAIN_TAIL_SPLIT
    try:
        ops.CHAIN_TAIL_SPLIT = False
        want = run()
        ops.CHAIN_TAIL_SPLIT = True
        for rep in range(3):
            got = run()
            for i, (a, b) in enumerate(zip(got, want)):
                assert torch.equal(a, b), f"rep {rep}: tensor {i} differs"
    finally:
        ops.CHAIN_TAIL_SPLIT = saved
