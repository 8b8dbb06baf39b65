"""GPU parity of the multi-wave workgroup kernels (gcpnet_amd/csrc/gcp_wg_*.hip) against the CPU oracle, with the
wave-per-tile kernels as a second opinion.  Every case asserts that the workgroup path actually ran (ops.WG_STATS)."""
import pytest
import torch

from oracle import gcp_oracle as O
from tests.helpers import as_accurate, close, rand_graph

pytestmark = pytest.mark.gpu

FWD = dict(atol=1e-5, rtol=1e-5)
GRAD = dict(atol=1e-5, rtol=1e-4)


@pytest.fixture(scope="module")
def G():
    import gcpnet_amd

    return gcpnet_amd


def lin_loss(outs, ws):
    return sum((t * w).sum() for t, w in zip(outs, ws))


SINGLE = [
    # rows, in dims, out dims, bottleneck, nonlinearities, extra kwargs
    (1000, (128, 16), (512, 32), 4, ("relu", None), {}),            # FF0 at C2 dims: two tiles per wave, 8 waves
    (333, (512, 32), (128, 16), 4, (None, None), {}),               # FF1 at C2 dims: K = 529
    (97, (256, 32), (1024, 64), 4, ("relu", None), {}),             # FF0 at C5 dims: two output groups, two tiles of gate outputs
    (130, (64, 16), (64, 48), 4, ("silu", None), {}),               # 32 < vo <= 64 with four waves (vo = 48: a partial second gate tile)
    (75, (160, 24), (192, 64), 4, ("relu", "sigmoid"), {}),         # ... with eight waves, one output tile per wave
    (70, (100, 16), (100, 16), 4, ("silu", "sigmoid"), {}),         # LBA width: so = 100 (partial last tile)
    (129, (128, 16), (128, 1), 4, ("relu", None), {}),              # position update: one output vector
    (64, (1, 3), (64, 16), 1, (None, None), {}),                    # NMS node embedding: si = 1, H = 16
    (45, (17, 1), (32, 4), 1, ("relu", None), {}),                  # NMS edge embedding
    (200, (64, 16), (64, 16), 4, ("leakyrelu", None), dict(vector_residual=True, enable_e3_equivariance=True)),
    (150, (64, 8), (96, 8), 2, ("silu", "sigmoid"), dict(vector_gate=False)),        # self gate
    (150, (64, 8), (96, 8), 2, ("selu", "sigmoid"), {}),                              # selu / sigmoid gate input
    (90, (40, 12), (72, 24), 4, ("relu", None), dict(ablate_frame_updates=True)),
    (31, (128, 16), (128, 0), 4, ("relu", None), {}),               # scalar-only output
    (600, (256, 32), (128, 0), 4, ("silu", None), {}),              # scalar-only output, eight waves, split-K last tile
    (32, (3, 12), (100, 24), 2, ("sigmoid", "selu"), dict(vector_gate=False)),  # found by tests/sweep_gcp2.py: si % 4 != 0, vo = 24
    (600, (128, 32), (128, 16), 4, ("silu", None), {}),             # vi != vo, fused weight gradients, not residual
    (77, (256, 32), (256, 32), 4, ("relu", None), {}),              # C5 message block
]


@pytest.mark.parametrize("rows,din,dout,bott,acts,kw", SINGLE)
def test_single_block_vs_oracle(G, rows, din, dout, bott, acts, kw):
    from gcpnet_amd import ops

    torch.manual_seed(rows + din[0])
    mod = G.GCP2(din, dout, nonlinearities=acts, bottleneck=bott, **kw).cuda()
    g = torch.Generator().manual_seed(rows + 1)
    ei = torch.stack((torch.arange(rows), torch.arange(rows)))
    fr = torch.randn(rows, 3, 3, generator=g)
    s = torch.randn(rows, din[0], generator=g).requires_grad_()
    v = torch.randn(rows, din[1], 3, generator=g).requires_grad_()
    P = {k: t.detach().cpu().clone().requires_grad_() for k, t in mod.state_dict().items()}
    okw = {k: kw[k] for k in ("vector_gate", "vector_residual", "ablate_frame_updates", "enable_e3_equivariance") if k in kw}
    want = O.gcp2(P, "", s, v, ei, fr, nonlinearities=acts, vector_output_dim=dout[1], **okw)
    want = want if isinstance(want, tuple) else (want,)
    sg, vg = s.detach().cuda().requires_grad_(), v.detach().cuda().requires_grad_()
    before = dict(ops.WG_STATS)
    got = mod((sg, vg), ei.cuda(), fr.cuda())
    got = tuple(got) if isinstance(got, tuple) else (got,)
    expect_wg = not (dout[1] > 64 and kw.get("vector_gate", True))  # (gated blocks with more than 64 output vectors: wave kernels)
    assert ops.WG_STATS["fwd"] == before["fwd"] + int(expect_wg), "the workgroup forward kernel did not run"
    scale = max(1.0, float(want[0].abs().max()))
    for a, b in zip(got, want):
        close(a.detach().cpu(), b.detach(), atol=1e-5 * scale, rtol=1e-5)
    ws = [torch.randn(t.shape, generator=g) for t in want]
    lin_loss(want, ws).backward()
    lin_loss(got, [w.cuda() for w in ws]).backward()
    gs = max(1.0, float(s.grad.abs().max()))
    close(sg.grad.cpu(), s.grad, atol=2e-5 * gs, rtol=1e-4)
    close(vg.grad.cpu(), v.grad, atol=2e-5 * max(1.0, float(v.grad.abs().max())), rtol=1e-4)
    for k, p in mod.named_parameters():
        if P[k].grad is None:
            continue
        close(p.grad.cpu(), P[k].grad, atol=2e-5 * max(1.0, float(P[k].grad.abs().max())), rtol=2e-4)


@pytest.mark.parametrize("wg_bwd", [False, True], ids=["chain-bwd-default", "chain-bwd-wg"])
@pytest.mark.parametrize("act", ["silu", "relu"])
@pytest.mark.parametrize("n,e,dims,blocks", [(300, 2500, (128, 16), 8), (200, 1500, (256, 32), 8), (150, 1100, (64, 16), 4),
                                             (33, 64, (128, 32), 8),  # found by tests/sweep_layers.py: V = 32 at so <= 128
                                             (90, 700, (100, 16), 3),   # scalar width not a multiple of 32: the padded form of
                                             (120, 900, (36, 8), 3)],   # the chain backward (four / two 32-wide tiles)
                         ids=["C2-dims", "C5-dims", "NMS-dims", "V32-dims", "LBA-dims", "pad36-dims"])
def test_message_chain_vs_oracle(G, n, e, dims, blocks, act, wg_bwd):
    """GCPMessagePassing (first message GCP after project-then-gather + ResGCP chain + aggregation) through the workgroup
    kernels: chain in one launch at every hidden size, including (256, 32) where the wave-per-tile chain kernels do not apply."""
    from gcpnet_amd import ops

    torch.manual_seed(21)
    cfg = G.default_module_cfg(scalar_nonlinearity=act)
    lcfg = G.default_layer_cfg(num_message_layers=blocks)
    mp = G.GCPMessagePassing(dims, dims, (32, 4), cfg=cfg, mp_cfg=lcfg.mp_cfg).cuda()
    ei, x = rand_graph(n, e, 22, sort_by_col=True)
    fr = O.localize(x, ei)
    g = torch.Generator().manual_seed(23)
    ins = dict(h=torch.randn(n, dims[0], generator=g), chi=torch.randn(n, dims[1], 3, generator=g),
               e=torch.randn(e, 32, generator=g), xi=torch.randn(e, 4, 3, generator=g))
    P = {k: t.detach().cpu().clone().requires_grad_() for k, t in mp.state_dict().items()}
    ci = {k: t.clone().requires_grad_() for k, t in ins.items()}
    ocfg = O.default_module_cfg(scalar_nonlinearity=act, nonlinearities=(act, None))
    olcfg = O.default_layer_cfg(num_message_layers=blocks)
    ws_, wv_ = O.message_passing(P, "", ci["h"], ci["chi"], ci["e"], ci["xi"], ei, fr, ocfg, olcfg["mp_cfg"])
    # the same in float64: what both fp32 evaluations are measured against where ReLU kinks forbid an element-wise bound
    P64 = {k: t.detach().double().requires_grad_() for k, t in P.items()}
    c64 = {k: t.double().requires_grad_() for k, t in ins.items()}
    ws64, wv64 = O.message_passing(P64, "", c64["h"], c64["chi"], c64["e"], c64["xi"], ei, fr.double(), ocfg, olcfg["mp_cfg"])
    gi = {k: t.cuda().requires_grad_() for k, t in ins.items()}
    before = dict(ops.WG_STATS)
    saved, saved_fwd = ops.FORCE_WG_CHAIN_BACKWARD, ops.PREFER_WAVE_CHAIN_FORWARD
    ops.FORCE_WG_CHAIN_BACKWARD = wg_bwd
    ops.PREFER_WAVE_CHAIN_FORWARD = False  # (this test is about the workgroup kernels; the wave-per-tile route: test_bf16x3, parity tests)
    out = mp((gi["h"], gi["chi"]), (gi["e"], gi["xi"]), ei.cuda(), fr.cuda())
    assert ops.WG_STATS["fwd_chain"] == before["fwd_chain"] + 1, "the chain did not run in the workgroup kernel"
    if dims != (128, 32):  # (V = 32 with 4-channel edge vectors: the first message GCP takes the wave-per-tile kernel)
        assert ops.WG_STATS["fwd"] >= before["fwd"] + 1, "the first message GCP did not run in the workgroup kernel"
    sc = max(1.0, float(ws_.abs().max()))
    close(out[0].detach().cpu(), ws_.detach(), atol=1e-5 * sc, rtol=1e-5)
    close(out[1].detach().cpu(), wv_.detach(), atol=1e-5 * sc, rtol=1e-5)
    ls, lv = torch.randn(ws_.shape, generator=g), torch.randn(wv_.shape, generator=g)
    ((ws_ * ls).sum() + (wv_ * lv).sum()).backward()
    ((ws64 * ls.double()).sum() + (wv64 * lv.double()).sum()).backward()
    # conditioning of THIS problem instance: the float64 gradients after perturbing inputs and weights by fp32-round-off-sized
    # relative noise (1e-6).  Where a ReLU pre-activation sits that close to zero the exact gradient itself jumps; no fp32
    # evaluation can be asked to be closer to the unperturbed one than that.
    sens = {}
    if act == "relu":
        gp = torch.Generator().manual_seed(99)
        nz = lambda t: (t.detach() * (1 + 1e-6 * torch.randn(t.shape, generator=gp, dtype=torch.float64))).requires_grad_()
        Pp, cp = {k: nz(t) for k, t in P64.items()}, {k: nz(t) for k, t in c64.items()}
        wsp, wvp = O.message_passing(Pp, "", cp["h"], cp["chi"], cp["e"], cp["xi"], ei, fr.double(), ocfg, olcfg["mp_cfg"])
        ((wsp * ls.double()).sum() + (wvp * lv.double()).sum()).backward()
        sens = {k: float((cp[k].grad - c64[k].grad).norm()) for k in cp}
        sens.update({k: float((Pp[k].grad - P64[k].grad).norm()) for k in Pp if Pp[k].grad is not None})
    try:
        ((out[0] * ls.cuda()).sum() + (out[1] * lv.cuda()).sum()).backward()
    finally:
        ops.FORCE_WG_CHAIN_BACKWARD, ops.PREFER_WAVE_CHAIN_FORWARD = saved, saved_fwd
    if wg_bwd or dims[0] > 128:
        need = blocks - (1 if dims == (128, 32) else 0)  # (V = 32: the first message GCP is outside the workgroup kernels)
        assert ops.WG_STATS["bwd"] >= before["bwd"] + need, "the chain backward did not run in the workgroup kernel"

    def grads_close(a, b, b64, name):
        if act == "silu":  # smooth: element-wise
            close(a, b, atol=2e-5 * float(b.abs().max()), rtol=1e-4)
        else:  # relu: as accurate as the CPU fp32 path, measured against float64 (helpers.as_accurate)
            as_accurate(a, b, b64, name, abs_floor=4.0 * sens.get(name, 0.0))

    for k in ins:
        grads_close(gi[k].grad.cpu(), ci[k].grad, c64[k].grad, k)
    for k, p in mp.named_parameters():
        grads_close(p.grad.cpu(), P[k].grad, P64[k].grad, k)


def test_wg_and_wave_kernels_agree(G):
    """Same layer through both kernel families: results equal to fp32 round-off."""
    from gcpnet_amd import ops
    from tests.test_gpu_parity import _layer_run

    torch.manual_seed(5)
    n, e = 700, 9000
    layer = G.GCPInteractions((128, 16), (32, 4), cfg=G.default_module_cfg(scalar_nonlinearity="silu"),
                              layer_cfg=G.default_layer_cfg(), dropout=0.0).cuda().eval()
    ei, x = rand_graph(n, e, 12, sort_by_col=True)
    fr = O.localize(x, ei)
    g = torch.Generator().manual_seed(17)
    ins = dict(h=torch.randn(n, 128, generator=g), chi=torch.randn(n, 16, 3, generator=g),
               e=torch.randn(e, 32, generator=g), xi=torch.randn(e, 4, 3, generator=g))
    ref = _layer_run(G, layer, ins, ei, fr)
    again = _layer_run(G, layer, ins, ei, fr)
    for k in ref:
        assert torch.equal(ref[k], again[k]), f"{k} differs between two identical runs"
    saved = ops.USE_WG_KERNELS
    try:
        ops.USE_WG_KERNELS = False
        alt = _layer_run(G, layer, ins, ei, fr)
    finally:
        ops.USE_WG_KERNELS = saved
    for k in ref:
        close(alt[k], ref[k], atol=2e-5 * max(1.0, float(ref[k].abs().max())), rtol=1e-4)


@pytest.mark.parametrize("mean", [True, False], ids=["mean", "sum"])
@pytest.mark.parametrize("dims", [(128, 16), (64, 8), (100, 16)], ids=["128x16", "64x8", "100x16-padded"])
def test_aggregation_inside_the_chain_function_is_bit_identical(G, dims, mean):
    """ops.gcp2_chain(..., agg=(plan, mean)) -- the chain's Function returns the segment mean / sum itself and its backward kernel reads
    the node-level gradient tables through the edge -> node index (gcpnet_gcp2_chain_backward_gathered) -- against the same chain
    followed by ops.segment_reduce as its own Function: same kernels forward, the same fp32 product scale x gradient backward, so
    outputs, input gradients and weight gradients must agree BITWISE.  Rows not a multiple of 32, a node without in-edges."""
    from gcpnet_amd import ops

    torch.manual_seed(3)
    n, e = 57, 1000 + 13
    s, v = dims
    mods = [G.GCP2((s, v), (s, v), nonlinearities=("silu", "silu"), bottleneck=4).cuda() for _ in range(3)]
    g = torch.Generator().manual_seed(4)
    col = torch.randint(0, n - 1, (e,), generator=g).sort().values  # (node n - 1 receives nothing)
    plan = ops.GatherPlan(col.cuda(), n)
    fr = torch.randn(e, 3, 3, generator=g).cuda()
    s0, v0 = torch.randn(e, s, generator=g).cuda(), torch.randn(e, v, 3, generator=g).cuda()
    ls, lv = torch.randn(n, s, generator=g).cuda(), torch.randn(n, v, 3, generator=g).cuda()
    specs = [m.make_spec([None], [None], residual=True) for m in mods]
    res = []
    for fused in (True, False):
        for m in mods:
            m.zero_grad(set_to_none=True)
        a, b = s0.clone().requires_grad_(), v0.clone().requires_grad_()
        ws = [m._weights() for m in mods]
        if fused:
            o_s, o_v = ops.gcp2_chain(specs, a, b, fr, ws, agg=(plan, mean))
        else:
            m_s, m_v = ops.gcp2_chain(specs, a, b, fr, ws)
            o_s = ops.segment_reduce(m_s, plan, mean)
            o_v = ops.segment_reduce(m_v.reshape(e, 3 * v), plan, mean).reshape(n, v, 3)
        ((o_s * ls).sum() + (o_v * lv).sum()).backward()
        torch.cuda.synchronize()
        res.append([o_s.detach(), o_v.detach(), a.grad, b.grad] + [p.grad for m in mods for p in m.parameters()])
    assert float(res[0][0][n - 1].abs().max()) == 0.0 and float(res[0][2].abs().max()) > 0
    for x, y in zip(*res):
        assert torch.equal(x, y)


@pytest.mark.gpu
@pytest.mark.parametrize("rows", [1000 + 13, 64], ids=["ragged", "two-tiles"])
def test_wide_chain_tile_blocked_route_is_bit_identical(G, rows):
    """A (256,32) ResGCP chain runs in the workgroup kernels both ways (forward: one launch, backward: block by block, weight gradients
    through gcpnet_tn_gemm).  With ops.CHAIN_TILE_BLOCKED its s_pre, its intermediate states, the state gradient between the blocks and
    ds_pre travel tile-blocked (gcp_wg_block_t.s_*_tb, gcp_wg_bwd_args_t.tb); the arithmetic is the same, so outputs, input gradients
    and weight gradients must agree BITWISE with the row-major route.  Rows not a multiple of 32: the padding rows of the last tile."""
    from gcpnet_amd import ops

    torch.manual_seed(5)
    s, v = 256, 32
    mods = [G.GCP2((s, v), (s, v), nonlinearities=("silu", "silu"), bottleneck=4).cuda() for _ in range(3)]
    g = torch.Generator().manual_seed(6)
    fr = torch.randn(rows, 3, 3, generator=g).cuda()
    s0, v0 = torch.randn(rows, s, generator=g).cuda(), torch.randn(rows, v, 3, generator=g).cuda()
    ls, lv = torch.randn(rows, s, generator=g).cuda(), torch.randn(rows, v, 3, generator=g).cuda()
    specs = [m.make_spec([None], [None], residual=True) for m in mods]
    res, routes = [], []
    saved = ops.CHAIN_TILE_BLOCKED
    try:
        for tb in (True, False):
            ops.CHAIN_TILE_BLOCKED = tb
            for m in mods:
                m.zero_grad(set_to_none=True)
            a, b = s0.clone().requires_grad_(), v0.clone().requires_grad_()
            before = dict(ops.WG_STATS)
            o_s, o_v = ops.gcp2_chain(specs, a, b, fr, [m._weights() for m in mods])
            routes.append(bool(getattr(o_s.grad_fn, "wtb", False)))
            ((o_s * ls).sum() + (o_v * lv).sum()).backward()
            torch.cuda.synchronize()
            assert ops.WG_STATS["fwd_chain"] == before["fwd_chain"] + 1 and ops.WG_STATS["bwd"] == before["bwd"] + 3
            res.append([o_s.detach(), o_v.detach(), a.grad, b.grad] + [p.grad for m in mods for p in m.parameters()])
    finally:
        ops.CHAIN_TILE_BLOCKED = saved
    assert routes == [True, False], routes
    for x, y in zip(*res):
        assert torch.equal(x, y)


@pytest.mark.gpu
def test_block_backward_with_tile_blocked_tensors_matches_row_major(G):
    """gcpnet_wg_backward with gcp_wg_bwd_args_t.tb = s_pre | d_s_out | d_s_in | ds_pre on tensors converted by TileBlocked.from_rows:
    the input gradients (converted back by to_rows) and every scratch operand equal the row-major launch bit for bit -- the layout
    of include/gcpnet_hip.h as the host writes it is the layout the kernel addresses.  1000 + 13 rows: a partial last tile."""
    from gcpnet_amd import ops

    torch.manual_seed(7)
    rows, s, v = 1013, 256, 32
    blk = G.GCP2((s, v), (s, v), nonlinearities=("silu", "silu"), bottleneck=4).cuda()
    spec = blk.make_spec([None], [None], residual=True)
    w = tuple(None if t is None else t.detach() for t in blk._weights())
    g = torch.Generator().manual_seed(8)
    x, vv = torch.randn(rows, s, generator=g).cuda().requires_grad_(), torch.randn(rows, v, 3, generator=g).cuda()
    fr, ds, dv = torch.randn(rows, 3, 3, generator=g).cuda(), torch.randn(rows, s, generator=g).cuda(), torch.randn(rows, v, 3, generator=g).cuda()
    out_s, _ = ops.gcp2(spec, [x], [vv], fr, w)
    pack, s_pre, gate = out_s.grad_fn.saved_tensors[-3:]
    assert out_s.grad_fn.s_pre_tb  # (the single block's own s_pre is saved tile-blocked: rows for the reference launch)
    s_pre = ops.TileBlocked(rows, s, s_pre.device, owner=s_pre, offset=0, n=s_pre.numel()).to_rows()
    with torch.no_grad():
        a_s, a_v, a_t = ops.gcp2_backward_data(spec, rows, [x.detach()], [vv], fr, w, pack, s_pre, gate, ds, dv, need_w=True)
        tbs = [ops.TileBlocked.from_rows(t) for t in (x.detach(), s_pre, ds)]
        assert torch.equal(tbs[1].to_rows(), s_pre)
        b_s, b_v, b_t = ops.gcp2_backward_data(spec, rows, [tbs[0]], [vv], fr, w, pack, tbs[1], gate, tbs[2], dv, need_w=True, tb_out=True)
    torch.cuda.synchronize()
    assert isinstance(b_s, ops.TileBlocked) and isinstance(b_t["ds_pre"], ops.TileBlocked) and "fused" not in b_t
    assert torch.equal(b_s.to_rows(), a_s) and torch.equal(b_v, a_v)
    assert torch.equal(b_t["ds_pre"].to_rows(), a_t["ds_pre"])
    for k in ("ext", "dgate", "w_part"):
        assert torch.equal(b_t[k], a_t[k])


@pytest.mark.gpu
def test_stale_workgroup_images_are_rebuilt_together_and_equal_single_packs(G):
    """ops._pack_wg: when one block's packed image is stale, the images of every other known block whose weights have moved on are
    rebuilt by the same gcpnet_wg_pack_multi launch (a training step after the optimizer update).  The batched images must be the
    single-launch images bit for bit, a block whose weights did NOT change must keep its image, and a weight changed again after the
    batch must miss again."""
    from gcpnet_amd import ops

    torch.manual_seed(9)
    mods = [G.GCP2((64, 16), (64, 16), nonlinearities=("silu", "silu"), bottleneck=4).cuda() for _ in range(3)]
    specs = [m.make_spec([None], [None]) for m in mods]
    first = [ops._pack_wg(sp, m._weights()) for sp, m in zip(specs, mods)]
    assert all(ops._pack_wg(sp, m._weights()) is p for sp, m, p in zip(specs, mods, first))  # cached
    with torch.no_grad():
        for m in mods[:2]:  # (module 2 keeps its weights)
            m.scalar_out.weight.add_(0.25)
    p0 = ops._pack_wg(specs[0], mods[0]._weights())  # miss: rebuilds 0 and, in the same launch, 1
    assert p0 is not first[0]
    c1 = mods[1]._pack_cache
    assert c1["wg_pack"] is not first[1] and c1["wg_key"] == ops._wg_pack_key(None, mods[1].scalar_out.weight, mods[1].vector_out_scale.weight)
    assert ops._pack_wg(specs[1], mods[1]._weights()) is c1["wg_pack"]  # a hit now
    assert ops._pack_wg(specs[2], mods[2]._weights()) is first[2]
    saved = ops.BATCH_WG_PACKS
    try:
        ops.BATCH_WG_PACKS = False
        ops.invalidate_packs()
        single = [ops._pack_wg(sp, m._weights()) for sp, m in zip(specs, mods)]
    finally:
        ops.BATCH_WG_PACKS = saved
    torch.cuda.synchronize()
    assert torch.equal(single[0], p0) and torch.equal(single[1], c1["wg_pack"]) and torch.equal(single[2], first[2])
    with torch.no_grad():
        mods[1].scalar_out.weight.mul_(2.0)
    assert ops._pack_wg(specs[1], mods[1]._weights()) is not single[1]


@pytest.mark.gpu
@pytest.mark.parametrize("invalidations", [1, 2], ids=["one-invalidation-per-step", "two-per-step"])
def test_batched_repack_is_one_launch_in_every_consecutive_round(G, invalidations, monkeypatch):
    """the ride-along filter of ops._pack_wg must hold over CONSECUTIVE optimizer steps (a block whose image was
    built ahead by another block's miss still counts as "used in that step"), and when the epoch advances more than once per step
    (FusedAdam plus an EMA swap).  Four rounds of update -> invalidate -> use all blocks: one gcpnet_wg_pack_multi launch per round."""
    from gcpnet_amd import _lib, ops

    torch.manual_seed(10)
    mods = [G.GCP2((64, 16), (64, 16), nonlinearities=("silu", "silu"), bottleneck=4).cuda() for _ in range(4)]
    specs = [m.make_spec([None], [None]) for m in mods]
    lib = _lib.load()
    real = lib.gcpnet_wg_pack_multi
    calls = []

    def counted(n, jobs, stream):
        calls.append(int(n))
        return real(n, jobs, stream)

    monkeypatch.setattr(lib, "gcpnet_wg_pack_multi", counted)
    for sp, m in zip(specs, mods):  # round 0: every block packs on its own first use
        ops._pack_wg(sp, m._weights())
    for rnd in range(4):
        with torch.no_grad():
            for m in mods:
                m.scalar_out.weight.add_(0.125)
        for _ in range(invalidations):
            ops.invalidate_packs()
        calls.clear()
        packs = [ops._pack_wg(sp, m._weights()) for sp, m in zip(specs, mods)]
        assert calls == [len(mods)], f"round {rnd}: launches {calls}"
        assert all(ops._pack_wg(sp, m._weights()) is p for sp, m, p in zip(specs, mods, packs))
    # a block that is no longer called (a frozen teacher) stops riding along after one step
    for rnd in range(2):
        with torch.no_grad():
            for m in mods:
                m.scalar_out.weight.add_(0.125)
        ops.invalidate_packs()
        calls.clear()
        for sp, m in zip(specs[:3], mods[:3]):
            ops._pack_wg(sp, m._weights())
        assert calls == ([4] if rnd == 0 else [3]), f"teacher round {rnd}: {calls}"


@pytest.mark.gpu
def test_zero_row_linears_and_unaligned_rows_matmul(G):
    """with the library-GEMM fallbacks gone, the legal zero-row case (a ShardedGraph rank without local nodes) must
    still return empty outputs and zero weight gradients, and _rows_matmul_small must take a channel count that is not a multiple
    of 4 through the workgroup kernel (padded for the launch)."""
    from gcpnet_amd import ops

    torch.manual_seed(11)
    w = torch.randn(24, 40, device="cuda", requires_grad=True)
    b = torch.randn(24, device="cuda", requires_grad=True)
    x = torch.zeros(0, 40, device="cuda", requires_grad=True)
    y = ops.linear(x, w, b)
    assert y.shape == (0, 24)
    y.sum().backward()
    assert x.grad.shape == (0, 40) and torch.count_nonzero(w.grad) == 0 and torch.count_nonzero(b.grad) == 0
    w2 = torch.randn(24, 40, device="cuda", requires_grad=True)
    x2 = torch.zeros(0, 40, device="cuda", requires_grad=True)
    p = ops._Project.apply(x2, w2)
    assert p.shape == (0, 24)
    p.sum().backward()
    assert torch.count_nonzero(w2.grad) == 0
    # K * J > 4096 with J % 4 != 0
    xs = torch.randn(70, 96, device="cuda")
    ws = torch.randn(96, 50, device="cuda")
    got = ops._rows_matmul_small(xs, ws)
    want = (xs.double() @ ws.double()).float()
    assert got.shape == (70, 50)
    assert torch.allclose(got, want, atol=1e-4, rtol=1e-5)
