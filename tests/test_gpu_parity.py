"""GPU parity: the HIP path (through the C ABI) against the golden vectors of the reference and against the CPU
oracle on seeded inputs.  Tolerances: forward 1e-5 absolute on O(1) activations (north_star), gradients 1e-4
relative (+1e-5 absolute) -- fp32 everywhere, only the summation order differs (MFMA k-ordered fma chain vs BLAS)."""
import pytest
import torch

from oracle import gcp_oracle as O
from tests.helpers import Fixture, close, rand_graph

pytestmark = pytest.mark.gpu

FWD = dict(atol=1e-5, rtol=1e-5)
GRAD = dict(atol=1e-5, rtol=1e-4)


@pytest.fixture(scope="module")
def G():
    import gcpnet_amd

    return gcpnet_amd


def dev(t):
    return t.cuda() if torch.is_tensor(t) else t


def sq_loss(*ts):
    return sum((t * t).mean() for t in ts if t is not None and t.numel())


# ------------------------------------------------------------------------------------------------------------
def test_geometry(G):
    f = Fixture("geometry")
    x, ei = f.i["x"].cuda(), f.i["edge_index"].cuda()
    close(G.localize(x, ei, True).cpu(), f.o["frames"], **FWD)
    close(G.localize(x, ei, False).cpu(), f.o["frames_raw"], **FWD)
    bag = G.Batch(x=x)
    cen, xc = G.centralize(bag, "x", f.i["batch"].cuda())
    close(cen.cpu(), f.o["centroid"], **FWD)
    close(xc.cpu(), f.o["x_centered"], **FWD)
    back = G.decentralize(G.Batch(x=xc), "x", f.i["batch"].cuda(), cen)
    close(back.cpu(), f.o["x_back"], **FWD)


@pytest.mark.parametrize("n_src,rows,D,sort", [(50, 700, 176, True), (50, 700, 176, False), (7, 33, 9, False),
                                                (300, 1000, 50, False), (10, 0, 8, True)])
def test_segment_reduce_and_gather(G, n_src, rows, D, sort):
    from gcpnet_amd import ops

    g = torch.Generator().manual_seed(rows + D)
    idx = torch.randint(0, n_src, (rows,), generator=g)
    if sort:
        idx = torch.sort(idx).values
    x = torch.randn(rows, D, generator=g).requires_grad_()
    for mean in (True, False):
        want = O.scatter(x, idx, n_src, "mean" if mean else "sum")
        gw, = torch.autograd.grad((want * want).sum(), x)
        xg = x.detach().cuda().requires_grad_()
        plan = ops.GatherPlan(idx.cuda(), n_src)
        got = ops.segment_reduce(xg, plan, mean)
        close(got.detach().cpu(), want.detach(), atol=1e-5, rtol=1e-5)
        gg, = torch.autograd.grad((got * got).sum(), xg)
        close(gg.cpu(), gw, atol=1e-5, rtol=1e-4)
    src = torch.randn(n_src, D, generator=g)
    plan = ops.GatherPlan(idx.cuda(), n_src)
    close(ops.gather_rows(src.cuda(), plan).cpu(), src[idx], atol=0, rtol=0)


def test_layernorm(G):
    f = Fixture("layernorm")
    ln = G.GCPLayerNorm((64, 16)).cuda()
    ln.load_state_dict(f.p)
    s, v = f.i["s"].cuda().requires_grad_(), f.i["v"].cuda().requires_grad_()
    so, vo = ln(G.ScalarVector(s, v))
    close(so.detach().cpu(), f.o["s"], **FWD)
    close(vo.detach().cpu(), f.o["v"], **FWD)
    g1 = torch.Generator().manual_seed(33)
    w1 = torch.randn(40, 64, generator=g1)
    w2 = torch.randn(40, 16, 3, generator=torch.Generator().manual_seed(34))
    sq_loss(so * w1.cuda(), vo * w2.cuda()).backward()
    close(s.grad.cpu(), f.g["s"], **GRAD)
    close(v.grad.cpu(), f.g["v"], **GRAD)
    close(ln.scalar_norm.weight.grad.cpu(), f.g["w.scalar_norm.weight"], **GRAD)
    close(ln.scalar_norm.bias.grad.cpu(), f.g["w.scalar_norm.bias"], **GRAD)


def test_layernorm_residual_vs_oracle(G):
    g = torch.Generator().manual_seed(5)
    s, sb = torch.randn(37, 100, generator=g), torch.randn(37, 100, generator=g)
    v, vb = torch.randn(37, 16, 3, generator=g), torch.randn(37, 16, 3, generator=g)
    ln = G.GCPLayerNorm((100, 16)).cuda()
    P = {k: t.cpu() for k, t in ln.state_dict().items()}
    ws, wv = O.gcp_layer_norm(P, "", s + sb, v + vb)
    gs, gv = ln(G.ScalarVector(s.cuda(), v.cuda()), residual=(sb.cuda(), vb.cuda()))
    close(gs.cpu(), ws, **FWD)
    close(gv.cpu(), wv, **FWD)


# ------------------------------------------------------------------------------------------------------------
GCP2_CASES = {
    "gcp2_edge_msg0": dict(nonlinearities=("relu", None), bottleneck=4),
    "gcp2_edge_res": dict(nonlinearities=("relu", None), bottleneck=4),
    "gcp2_node_ff0": dict(nonlinearities=("relu", None), bottleneck=4),
    "gcp2_node_ff1": dict(nonlinearities=(None, None), bottleneck=4),
    "gcp2_node_scalar_only": dict(nonlinearities=("relu", None)),
    "gcp2_no_vector_in": dict(nonlinearities=("relu", None)),
    "gcp2_node_posupd": dict(nonlinearities=("relu", None), bottleneck=4),
    "gcp2_silu_sigmoid": dict(nonlinearities=("silu", "sigmoid"), bottleneck=2),
    "gcp2_selfgate": dict(nonlinearities=("silu", "sigmoid"), vector_gate=False),
    "gcp2_ablate_frames": dict(nonlinearities=("relu", None), bottleneck=4, ablate_frame_updates=True),
    "gcp2_vres_e3": dict(nonlinearities=("leakyrelu", None), bottleneck=4, vector_residual=True,
                         enable_e3_equivariance=True),                                               # node rows, E(3) |.| per edge
    "gcp2_frame_gate": dict(nonlinearities=("silu", "silu"), bottleneck=2, frame_gate=True),        # node rows (mean frames)
    "gcp2_frame_gate_edge": dict(nonlinearities=("relu", "sigmoid"), frame_gate=True),              # edge rows
    "gcp3_edge_default": dict(bottleneck=4, cls="GCP3"),
    "gcp3_node_default": dict(bottleneck=2, cls="GCP3"),
    "gcp3_feedforward": dict(bottleneck=4, cls="GCP3", feedforward_out=True),
    "gcp3_feedforward_node": dict(bottleneck=2, cls="GCP3", feedforward_out=True, nonlinearities=(None, None)),
    "gcp3_feedforward_scalar": dict(cls="GCP3", feedforward_out=True, scalar_out_nonlinearity="relu"),
}


@pytest.mark.parametrize("name", sorted(GCP2_CASES))
def test_gcp2_golden(G, name):
    f = Fixture(name)
    kw = dict(GCP2_CASES[name])
    cls = getattr(G, kw.pop("cls", "GCP2"))
    mod = cls(tuple(int(d) for d in f.m["in_dims"]), tuple(int(d) for d in f.m["out_dims"]), **kw).cuda()
    mod.load_state_dict(f.p)
    s = f.i["s"].cuda().requires_grad_()
    ei, fr = f.i["edge_index"].cuda(), f.i["frames"].cuda()
    node_inputs = bool(f.m["node_inputs"])
    if "v" in f.i:
        v = f.i["v"].cuda().requires_grad_()
        out = mod((s, v), ei, fr, node_inputs=node_inputs)
    else:
        v = None
        out = mod(s, ei, fr, node_inputs=node_inputs)
    outs = dict(s=out[0], v=out[1]) if isinstance(out, tuple) else dict(s=out)
    for k, t in outs.items():
        close(t.detach().cpu(), f.o[k], **FWD)
    sq_loss(*outs.values()).backward()
    close(s.grad.cpu(), f.g["s"], **GRAD)
    if v is not None:
        close(v.grad.cpu(), f.g["v"], **GRAD)
    for k, p in mod.named_parameters():
        if "w." + k in f.g:
            assert p.grad is not None, k
            close(p.grad.cpu(), f.g["w." + k], **GRAD)


GCP_CASES = {  # the original GCP block (reference gcpnet.py:30-249), fixtures from the reference
    "gcp_edge_default": dict(bottleneck=4),
    "gcp_node_default": dict(bottleneck=2, nonlinearities=("silu", "sigmoid")),
    "gcp_sigma_gate": dict(nonlinearities=("relu", "sigmoid"), sigma_frame_gate=True, vector_residual=True),
    "gcp_frame_gate": dict(nonlinearities=("silu", "silu"), bottleneck=2, frame_gate=True, vector_frame_residual=True),
    "gcp_selfgate_e3": dict(nonlinearities=("silu", "sigmoid"), vector_gate=False, enable_e3_equivariance=True),
    "gcp_node_e3": dict(nonlinearities=("silu", "sigmoid"), bottleneck=2, sigma_frame_gate=True, enable_e3_equivariance=True),
    "gcp_scalar_out": dict(nonlinearities=("relu", None)),
    "gcp_ablate_frames": dict(nonlinearities=("relu", None), bottleneck=4, ablate_frame_updates=True),
}


@pytest.mark.parametrize("name", sorted(GCP_CASES))
def test_gcp_original_golden(G, name):
    f = Fixture(name)
    mod = G.GCP(tuple(int(d) for d in f.m["in_dims"]), tuple(int(d) for d in f.m["out_dims"]), **GCP_CASES[name]).cuda()
    mod.load_state_dict(f.p)
    s, v = f.i["s"].cuda().requires_grad_(), f.i["v"].cuda().requires_grad_()
    out = mod((s, v), f.i["edge_index"].cuda(), f.i["frames"].cuda(), node_inputs=bool(f.m["node_inputs"]))
    outs = dict(s=out[0], v=out[1]) if isinstance(out, tuple) else dict(s=out)
    for k, t in outs.items():
        close(t.detach().cpu(), f.o[k], **FWD)
    sq_loss(*outs.values()).backward()
    close(s.grad.cpu(), f.g["s"], **GRAD)
    close(v.grad.cpu(), f.g["v"], **GRAD)
    n = 0
    for k, p in mod.named_parameters():
        if "w." + k in f.g:
            assert p.grad is not None, k
            close(p.grad.cpu(), f.g["w." + k], **GRAD)
            n += 1
    assert n >= 4


def test_gcp2_e3_edge_vs_oracle(G):
    """enable_e3_equivariance on edge rows + vector_residual + leakyrelu, against the oracle."""
    torch.manual_seed(3)
    mod = G.GCP2((24, 8), (24, 8), nonlinearities=("leakyrelu", None), bottleneck=4, vector_residual=True,
                 enable_e3_equivariance=True).cuda()
    ei, x = rand_graph(20, 70, 9)
    fr = O.localize(x, ei)
    g = torch.Generator().manual_seed(4)
    s, v = torch.randn(70, 24, generator=g).requires_grad_(), torch.randn(70, 8, 3, generator=g).requires_grad_()
    P = {k: t.detach().cpu().clone().requires_grad_() for k, t in mod.state_dict().items()}
    ws, wv = O.gcp2(P, "", s, v, ei, fr, nonlinearities=("leakyrelu", None), vector_residual=True,
                    enable_e3_equivariance=True)
    sg, vg = s.detach().cuda().requires_grad_(), v.detach().cuda().requires_grad_()
    gs, gv = mod((sg, vg), ei.cuda(), fr.cuda())
    close(gs.detach().cpu(), ws.detach(), **FWD)
    close(gv.detach().cpu(), wv.detach(), **FWD)
    sq_loss(ws, wv).backward()
    sq_loss(gs, gv).backward()
    close(sg.grad.cpu(), s.grad, **GRAD)
    close(vg.grad.cpu(), v.grad, **GRAD)
    for k, p in mod.named_parameters():
        close(p.grad.cpu(), P[k].grad, **GRAD)


@pytest.mark.parametrize("rows,dims_in,dims_out,bott", [(1, (5, 3), (7, 2), 1), (33, (100, 16), (100, 16), 4),
                                                         (97, (40, 12), (300, 24), 4), (64, (530, 8), (36, 4), 2),
                                                         (1000, (128, 16), (128, 16), 4), (50, (33, 4), (65, 5), 1)])
def test_gcp2_ragged_vs_oracle(G, rows, dims_in, dims_out, bott):
    """Odd sizes: partial tiles, widths that are not multiples of 4 / 32, several output and k groups."""
    torch.manual_seed(rows)
    mod = G.GCP2(dims_in, dims_out, nonlinearities=("silu", None), bottleneck=bott).cuda()
    g = torch.Generator().manual_seed(rows + 1)
    ei = torch.stack((torch.arange(rows), torch.arange(rows)))
    fr = torch.randn(rows, 3, 3, generator=g)
    s = torch.randn(rows, dims_in[0], generator=g).requires_grad_()
    v = torch.randn(rows, dims_in[1], 3, generator=g).requires_grad_()
    P = {k: t.detach().cpu().clone().requires_grad_() for k, t in mod.state_dict().items()}
    ws, wv = O.gcp2(P, "", s, v, ei, fr, nonlinearities=("silu", None))
    sg, vg = s.detach().cuda().requires_grad_(), v.detach().cuda().requires_grad_()
    gs, gv = mod((sg, vg), ei.cuda(), fr.cuda())
    close(gs.detach().cpu(), ws.detach(), atol=1e-5, rtol=1e-5)
    close(gv.detach().cpu(), wv.detach(), atol=1e-5, rtol=1e-5)
    sq_loss(ws, wv).backward()
    sq_loss(gs, gv).backward()
    close(sg.grad.cpu(), s.grad, **GRAD)
    close(vg.grad.cpu(), v.grad, **GRAD)
    for k, p in mod.named_parameters():
        close(p.grad.cpu(), P[k].grad, atol=2e-5, rtol=2e-4)


# ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["embedding_nms", "embedding_lba"])
def test_embedding_golden(G, name):
    f = Fixture(name)
    lba = name.endswith("lba")
    emb = G.GCPEmbedding((16, 1) if lba else (17, 1), (9, 2) if lba else (1, 3), (32, 4), (100, 16) if lba else (64, 16),
                         num_atom_types=9 if lba else 0, cfg=G.default_module_cfg()).cuda()
    emb.load_state_dict(f.p)
    b = G.Batch(**{k: v.cuda() for k, v in f.i.items()})
    b.f_ij = b.frames
    (h, chi), (e, xi) = emb(b)
    for k, t in dict(h=h, chi=chi, e=e, xi=xi).items():
        close(t.detach().cpu(), f.o[k], **FWD)


def _load_layer(G, f, upd, force=False):
    layer = G.GCPInteractions((64, 16), (32, 4), cfg=G.default_module_cfg(ablate_x_force_update=not force),
                              layer_cfg=G.default_layer_cfg(), dropout=0.0, updating_node_positions=upd).cuda()
    layer.load_state_dict(f.p)
    return layer.eval()


def test_message_passing_golden(G):
    f = Fixture("message_passing")
    mp = G.GCPMessagePassing((64, 16), (64, 16), (32, 4), cfg=G.default_module_cfg(),
                             mp_cfg=G.default_layer_cfg().mp_cfg).cuda()
    mp.load_state_dict(f.p)
    ins = {k: f.i[k].cuda().requires_grad_() for k in ("h", "chi", "e", "xi")}
    ei, fr = f.i["edge_index"].cuda(), f.i["frames"].cuda()
    msg = mp.message((ins["h"], ins["chi"]), (ins["e"], ins["xi"]), ei, fr)
    close(msg.detach().cpu(), f.o["messages"], atol=1e-5, rtol=1e-5)
    s, v = mp((ins["h"], ins["chi"]), (ins["e"], ins["xi"]), ei, fr)
    close(s.detach().cpu(), f.o["s"], atol=1e-5, rtol=1e-5)
    close(v.detach().cpu(), f.o["v"], atol=1e-5, rtol=1e-5)
    sq_loss(s, v).backward()
    for k, t in ins.items():
        close(t.grad.cpu(), f.g[k], atol=1e-5, rtol=5e-4)
    for k, p in mp.named_parameters():
        close(p.grad.cpu(), f.g["w." + k], atol=1e-5, rtol=5e-4)


@pytest.mark.parametrize("flag", ["ablate_scalars", "ablate_vectors"])
def test_message_passing_ablations_golden(G, flag):
    """ablate_scalars / ablate_vectors on the message path (reference gcpnet.py:416-417,466-467): the
    project-then-gather / chain routes must honour the flags -- reference fixture, forward and every gradient."""
    f = Fixture("message_passing_" + flag)
    mp = G.GCPMessagePassing((64, 16), (64, 16), (32, 4), cfg=G.default_module_cfg(**{flag: True}),
                             mp_cfg=G.default_layer_cfg().mp_cfg).cuda()
    mp.load_state_dict(f.p)
    ins = {k: f.i[k].cuda().requires_grad_() for k in ("h", "chi", "e", "xi")}
    ei, fr = f.i["edge_index"].cuda(), f.i["frames"].cuda()
    msg = mp.message((ins["h"], ins["chi"]), (ins["e"], ins["xi"]), ei, fr)
    close(msg.detach().cpu(), f.o["messages"], atol=1e-5, rtol=1e-5)
    s, v = mp((ins["h"], ins["chi"]), (ins["e"], ins["xi"]), ei, fr)
    close(s.detach().cpu(), f.o["s"], atol=1e-5, rtol=1e-5)
    close(v.detach().cpu(), f.o["v"], atol=1e-5, rtol=1e-5)
    if flag == "ablate_scalars":
        assert float(s.detach().abs().max()) == 0.0
    else:
        assert float(v.detach().abs().max()) == 0.0
    ((s * f.i["lw_s"].cuda()).sum() + (v * f.i["lw_v"].cuda()).sum() + sq_loss(s, v)).backward()
    for k, t in ins.items():
        g = torch.zeros_like(t) if t.grad is None else t.grad
        close(g.cpu(), f.g[k], atol=1e-5 * max(1.0, float(f.g[k].abs().max())), rtol=5e-4)
    for k, p in mp.named_parameters():
        g = torch.zeros_like(p) if p.grad is None else p.grad
        close(g.cpu(), f.g["w." + k], atol=1e-5 * max(1.0, float(f.g["w." + k].abs().max())), rtol=5e-4)


@pytest.mark.parametrize("name", ["interactions", "interactions_posupd", "interactions_force"])
def test_interactions_golden(G, name):
    f = Fixture(name)
    upd = name != "interactions"
    layer = _load_layer(G, f, upd, force=name == "interactions_force")
    ins = {k: f.i[k].cuda().requires_grad_() for k in ("h", "chi", "e", "xi")}
    ei, fr = f.i["edge_index"].cuda(), f.i["frames"].cuda()
    if upd:
        (h, chi), x = layer((ins["h"], ins["chi"]), (ins["e"], ins["xi"]), ei, fr, node_pos=f.i["x"].cuda())
        outs = dict(h=h, chi=chi, x=x)
    else:
        h, chi = layer((ins["h"], ins["chi"]), (ins["e"], ins["xi"]), ei, fr)
        outs = dict(h=h, chi=chi)
    for k, t in outs.items():
        close(t.detach().cpu(), f.o[k], atol=1e-5, rtol=1e-5)
    sq_loss(*outs.values()).backward()
    for k, t in ins.items():
        close(t.grad.cpu(), f.g[k], atol=1e-5, rtol=1e-3)
    for k, p in layer.named_parameters():
        if "w." + k in f.g:
            close(p.grad.cpu(), f.g["w." + k], atol=1e-5, rtol=1e-3)


# ---- masked / autoregressive call paths, fixtures from the reference -----------------------------------
def test_geometry_masked_golden(G):
    """centralize / localize with node_mask (components/__init__.py:177-193, 229-264): +inf on the masked rows / edges."""
    f = Fixture("geometry_masked")
    i = f.i
    m, ei, x = i["mask"].bool().cuda(), i["edge_index"].cuda(), i["x"].cuda()
    fr = G.localize(x, ei, node_mask=m).cpu()
    assert torch.equal(torch.isinf(fr), torch.isinf(f.o["frames"]))
    fin = torch.isfinite(f.o["frames"])
    close(fr[fin], f.o["frames"][fin], **FWD)
    cen, xc = G.centralize(G.Batch(x=x), "x", i["batch"].cuda(), node_mask=m)
    close(cen.cpu(), f.o["centroid"], **FWD)
    xc = xc.cpu()
    assert torch.equal(torch.isinf(xc), torch.isinf(f.o["x_centered"]))
    close(xc[i["mask"].bool()], f.o["x_centered"][i["mask"].bool()], **FWD)


def test_gcp2_masked_node_golden(G):
    f = Fixture("gcp2_masked_node")
    mod = G.GCP2((24, 8), (16, 4), nonlinearities=("silu", "sigmoid"), bottleneck=2).cuda()
    mod.load_state_dict(f.p)
    s, v = f.i["s"].cuda().requires_grad_(), f.i["v"].cuda().requires_grad_()
    out = mod((s, v), f.i["edge_index"].cuda(), f.i["frames"].cuda(), node_inputs=True, node_mask=f.i["mask"].bool().cuda())
    close(out[0].detach().cpu(), f.o["s"], **FWD)
    close(out[1].detach().cpu(), f.o["v"], **FWD)
    sq_loss(*out).backward()
    close(s.grad.cpu(), f.g["s"], **GRAD)
    close(v.grad.cpu(), f.g["v"], **GRAD)
    for k, p in mod.named_parameters():
        if "w." + k in f.g:
            close(p.grad.cpu(), f.g["w." + k], **GRAD)


@pytest.mark.parametrize("name", ["interactions_masked", "interactions_masked_posupd", "interactions_masked_all",
                                  "interactions_autoregressive"])
def test_interactions_masked_and_autoregressive_golden(G, name):
    """GCPInteractions with node_mask (sub-graph feed-forward, reference gcpnet.py:1201-1251) and autoregressive_forward
    (:1066-1116): outputs and every gradient against the reference's."""
    f = Fixture(name)
    auto = name == "interactions_autoregressive"
    upd = name == "interactions_masked_posupd"
    pre_norm = bool(int(f.m["pre_norm"])) if "pre_norm" in f.m else False
    layer = G.GCPInteractions((32, 8), (16, 4), cfg=G.default_module_cfg(), layer_cfg=G.default_layer_cfg(pre_norm=pre_norm),
                              dropout=0.0, autoregressive=auto, updating_node_positions=upd).cuda().eval()
    layer.load_state_dict(f.p)
    keys = ("h", "chi", "e", "xi") + (("h_reg", "chi_reg") if auto else ())
    ins = {k: f.i[k].cuda().requires_grad_() for k in keys}
    ei, fr = f.i["edge_index"].cuda(), f.i["frames"].cuda()
    kw = {}
    if "mask" in f.i:
        kw["node_mask"] = f.i["mask"].bool().cuda()
    if auto:
        kw["node_rep_regressive"] = (ins["h_reg"], ins["chi_reg"])
    # with a mask the reference writes the new rows INTO its node input and returns it (gcpnet.py:1248-1251): clones are fed, as
    # the fixture generator does, and the returned tensors must be those very objects (no pre-norm: the normed copies otherwise)
    masked = "mask" in f.i
    h_in, chi_in = (ins["h"].clone(), ins["chi"].clone()) if masked else (ins["h"], ins["chi"])
    if upd:
        (h, chi), x = layer((h_in, chi_in), (ins["e"], ins["xi"]), ei, fr, node_pos=f.i["x"].cuda(), **kw)
        outs = dict(h=h, chi=chi, x=x)
    else:
        h, chi = layer((h_in, chi_in), (ins["e"], ins["xi"]), ei, fr, **kw)
        outs = dict(h=h, chi=chi)
    if masked and not pre_norm:
        assert h is h_in and chi is chi_in, "the masked forward must update its node input in place"
        with pytest.raises(RuntimeError):  # (a leaf that requires grad cannot be written in place: same error as the reference)
            layer((ins["h"], ins["chi"]), (ins["e"], ins["xi"]), ei, fr, **({"node_pos": f.i["x"].cuda()} if upd else {}), **kw)
    for k, t in outs.items():
        t = t.detach().cpu()
        fin = torch.isfinite(f.o[k])
        assert torch.equal(torch.isfinite(t), fin), k
        close(t[fin], f.o[k][fin], atol=1e-5, rtol=1e-5)
    if upd:
        fin = torch.isfinite(outs["x"]).all(dim=1)
        loss = sq_loss(outs["h"], outs["chi"], outs["x"][fin])
    else:
        loss = sq_loss(*outs.values())
    loss.backward()
    for k, t in ins.items():
        close(t.grad.cpu(), f.g[k], atol=1e-5, rtol=1e-3)
    n = 0
    for k, p in layer.named_parameters():
        if "w." + k in f.g:
            assert p.grad is not None, k
            close(p.grad.cpu(), f.g["w." + k], atol=1e-5, rtol=1e-3)
            n += 1
    assert n > 50


@pytest.mark.parametrize("name,res", [("mlp_decoder", False), ("mlp_decoder_residual", True)])
def test_mlp_decoder_golden(G, name, res):
    f = Fixture(name)
    dec = G.GCPMLPDecoder(32, vocab_size=20, num_layers=3, residual_updates=res).cuda()
    dec.load_state_dict(f.p)
    h = f.i["h"].cuda().requires_grad_()
    logits, logp = dec(h)
    close(logits.detach().cpu(), f.o["logits"], **FWD)
    close(logp.detach().cpu(), f.o["log_probs"], **FWD)
    (sq_loss(logits) + (logp * f.i["lw"].cuda()).mean()).backward()
    close(h.grad.cpu(), f.g["h"], **GRAD)
    for k, p in dec.named_parameters():
        close(p.grad.cpu(), f.g["w." + k], **GRAD)


@pytest.mark.parametrize("rows,s", [(1, 4), (37, 128), (5000, 64), (9000, 260), (33, 1024)])
def test_row_gate_vs_torch(G, rows, s):
    """Scalar message gate kernel (reference gcpnet.py:932-934) against the same formula in fp64 on the CPU."""
    from gcpnet_amd import ops
    g = torch.Generator().manual_seed(rows + s)
    x = torch.randn(rows, s, generator=g)
    w, b = torch.randn(1, s, generator=g) / s ** 0.5, torch.randn(1, generator=g)
    dy = torch.randn(rows, s, generator=g)
    xc, wc, bc = (t.double().requires_grad_() for t in (x, w, b))
    yc = xc * torch.sigmoid(xc @ wc.t() + bc)
    yc.backward(dy.double())
    xg, wg, bg = (t.cuda().requires_grad_() for t in (x, w, b))
    yg = ops.row_gate(xg, wg, bg)
    yg.backward(dy.cuda())
    close(yg.detach().cpu(), yc.detach().float(), atol=1e-6, rtol=1e-5)
    close(xg.grad.cpu(), xc.grad.float(), atol=2e-6, rtol=1e-5)
    scale = max(1.0, float(wc.grad.abs().max()))
    close(wg.grad.cpu(), wc.grad.float(), atol=2e-5 * scale, rtol=1e-4)
    close(bg.grad.cpu(), bc.grad.float(), atol=2e-5 * scale, rtol=1e-4)


INTERACTIONS2 = {
    "interactions2_eq": dict(use_scalar_message_attention=True, aggregate_with_row=True, num_feedforward_layers=1),
    "interactions2_posupd": dict(use_scalar_message_attention=True, num_message_layers=4, num_feedforward_layers=2),
}


@pytest.mark.parametrize("name", sorted(INTERACTIONS2))
def test_interactions2_golden(G, name):
    """GCPInteractions2 (reference gcpnet.py:1265-1451) with GCP3 blocks: sum aggregation (over row), scalar message gate,
    two-layer scalar_out in the last feed-forward GCP, un-clamped position update."""
    import functools
    f = Fixture(name)
    upd = name == "interactions2_posupd"
    cfg = G.default_module_cfg(selected_GCP=functools.partial(G.GCP3))
    layer = G.GCPInteractions2((64, 16), (32, 4), cfg=cfg, layer_cfg=G.default_layer_cfg(**INTERACTIONS2[name]), dropout=0.0,
                               updating_node_positions=upd).cuda().eval()
    assert list(layer.state_dict()) == list(f.p)
    layer.load_state_dict(f.p)
    ins = {k: f.i[k].cuda().requires_grad_() for k in ("h", "chi", "e", "xi")}
    ei, fr = f.i["edge_index"].cuda(), f.i["frames"].cuda()
    if upd:
        (h, chi), x = layer((ins["h"], ins["chi"]), (ins["e"], ins["xi"]), ei, fr, node_pos=f.i["x"].cuda())
        outs = dict(h=h, chi=chi, x=x)
    else:
        h, chi = layer((ins["h"], ins["chi"]), (ins["e"], ins["xi"]), ei, fr)
        outs = dict(h=h, chi=chi)
    for k, t in outs.items():
        close(t.detach().cpu(), f.o[k], atol=1e-5, rtol=1e-5)
    sq_loss(*outs.values()).backward()
    for k, t in ins.items():
        close(t.grad.cpu(), f.g[k], atol=1e-5, rtol=1e-3)
    for k, p in layer.named_parameters():
        if "w." + k in f.g:
            close(p.grad.cpu(), f.g["w." + k], atol=1e-5, rtol=1e-3)


@pytest.mark.parametrize("upd", [False, True], ids=["eq", "posupd"])
def test_interactions2_large_vs_oracle(G, upd):
    """GCPInteractions2 with GCP3 blocks at chain-kernel dims (128,16) on a col-sorted 6000-edge graph, silu (smooth: tight
    element-wise agreement), fwd + bwd with a random linear functional of the outputs."""
    import functools
    torch.manual_seed(21)
    n, e, dims = 600, 6000, (128, 16)
    kw = dict(use_scalar_message_attention=True, aggregate_with_row=not upd, num_feedforward_layers=2 if upd else 1)
    cfg = G.default_module_cfg(selected_GCP=functools.partial(G.GCP3), scalar_nonlinearity="silu")
    layer = G.GCPInteractions2(dims, (32, 4), cfg=cfg, layer_cfg=G.default_layer_cfg(**kw), dropout=0.0,
                               updating_node_positions=upd).cuda().eval()
    ei, x = rand_graph(n, e, 22, sort_by_col=True)
    fr = O.localize(x, ei)
    g = torch.Generator().manual_seed(23)
    ins = dict(h=torch.randn(n, dims[0], generator=g), chi=torch.randn(n, dims[1], 3, generator=g),
               e=torch.randn(e, 32, generator=g), xi=torch.randn(e, 4, 3, generator=g))
    P = {k: t.detach().cpu().clone().requires_grad_() for k, t in layer.state_dict().items()}
    ci = {k: t.clone().requires_grad_() for k, t in ins.items()}
    ocfg = O.default_module_cfg(scalar_nonlinearity="silu", nonlinearities=("silu", None))
    want = O.gcp_interactions2(P, "", ci["h"], ci["chi"], ci["e"], ci["xi"], ei, fr, ocfg, O.default_layer_cfg(**kw),
                               node_pos=x if upd else None)
    gi = {k: t.cuda().requires_grad_() for k, t in ins.items()}
    got = layer((gi["h"], gi["chi"]), (gi["e"], gi["xi"]), ei.cuda(), fr.cuda(), node_pos=x.cuda() if upd else None)
    wl = [want[0][0], want[0][1], want[1]] if upd else list(want)
    gl = [got[0][0], got[0][1], got[1]] if upd else list(got)
    for a, b in zip(gl, wl):
        close(a.detach().cpu(), b.detach(), atol=1e-5, rtol=1e-5)
    lw = [torch.randn(t.shape, generator=g) for t in wl]
    sum((t * w).sum() for t, w in zip(wl, lw)).backward()
    sum((t * w.cuda()).sum() for t, w in zip(gl, lw)).backward()
    for k in ins:
        b = ci[k].grad
        close(gi[k].grad.cpu(), b, atol=2e-5 * float(b.abs().max()), rtol=1e-4)
    for k, p in layer.named_parameters():
        b = P[k].grad
        if b is not None and p.grad is not None:
            close(p.grad.cpu(), b, atol=2e-5 * max(float(b.abs().max()), 1e-6), rtol=1e-4)


def test_interactions_prenorm_silu_golden(G):
    f = Fixture("interactions_prenorm_silu")
    cfg = G.default_module_cfg(scalar_nonlinearity="silu", vector_nonlinearity="silu")
    lc = G.default_layer_cfg(pre_norm=True, num_feedforward_layers=3, num_message_layers=3)
    layer = G.GCPInteractions((16, 4), (8, 4), cfg=cfg, layer_cfg=lc, dropout=0.0).cuda().eval()
    layer.load_state_dict(f.p)
    i = {k: v.cuda() for k, v in f.i.items()}
    h, chi = layer((i["h"], i["chi"]), (i["e"], i["xi"]), i["edge_index"], i["frames"])
    close(h.cpu(), f.o["h"], atol=1e-5, rtol=1e-5)
    close(chi.cpu(), f.o["chi"], atol=1e-5, rtol=1e-5)


def _check_step_grads(f, model, leaves):
    """Gradients of step() against those the reference produced (fixture tag g): every parameter and float input."""
    params = dict(model.named_parameters())
    n = 0
    for k, want in f.g.items():
        got = params[k[2:]].grad if k.startswith("w.") else leaves[k].grad
        assert got is not None, k
        close(got.cpu(), want, atol=1e-6 + 2e-5 * float(want.abs().max()), rtol=1e-3)
        n += 1
    assert n > 50


def test_model_nms_golden(G):
    f = Fixture("model_nms_small")
    model_cfg = dict(h_input_dim=1, chi_input_dim=3, e_input_dim=17, xi_input_dim=1, h_hidden_dim=32, chi_hidden_dim=8,
                     e_hidden_dim=16, xi_hidden_dim=4, num_encoder_layers=2, dropout=0.0)
    model = G.GCPNetNMS(model_cfg=model_cfg, module_cfg=G.default_module_cfg(),
                        layer_cfg=G.default_layer_cfg(num_message_layers=4)).cuda().eval()
    model.load_state_dict(f.p)
    b = G.Batch(**{k: v.cuda() for k, v in f.i.items()})
    for k in ("h", "chi", "e", "xi"):
        setattr(b, k, getattr(b, k).requires_grad_())
    leaves = {k: getattr(b, k) for k in ("h", "chi", "e", "xi")}
    loss, _, _ = model.step(b)  # gcpnet_nms_module.py:153-158: forward + MSELoss(x_pred, label)
    for k in ("h", "chi", "e", "xi", "x", "f_ij"):
        close(getattr(b, k).detach().cpu(), f.o[k], atol=1e-4, rtol=1e-4)  # model-level tolerance of the reference's tests
    close(loss.detach().cpu(), f.o["loss"], atol=1e-6, rtol=1e-4)
    loss.backward()
    _check_step_grads(f, model, leaves)


def test_model_lba_golden(G):
    f = Fixture("model_lba_small")
    model_cfg = dict(chi_input_dim=2, e_input_dim=16, xi_input_dim=1, h_hidden_dim=20, chi_hidden_dim=4, e_hidden_dim=8,
                     xi_hidden_dim=4, output_dim=1, output_scale_factor=2, num_encoder_layers=2, dropout=0.0,
                     dense_dropout=0.1)
    model = G.GCPNetLBA(model_cfg=model_cfg, module_cfg=G.default_module_cfg(),
                        layer_cfg=G.default_layer_cfg(num_message_layers=4)).cuda().eval()
    model.load_state_dict(f.p)
    b = G.Batch(**{k: v.cuda() for k, v in f.i.items()})
    for k in ("chi", "e", "xi"):
        setattr(b, k, getattr(b, k).requires_grad_())
    leaves = {k: getattr(b, k) for k in ("chi", "e", "xi")}
    loss, pred, _ = model.step(b)  # gcpnet_lba_module.py:188-193
    close(b.h.detach().cpu(), f.o["h"], atol=1e-4, rtol=1e-4)
    close(b.chi.detach().cpu(), f.o["chi"], atol=1e-4, rtol=1e-4)
    close(pred.detach().cpu(), f.o["pred"], atol=1e-4, rtol=1e-4)
    close(loss.detach().cpu(), f.o["loss"], atol=1e-6, rtol=1e-4)
    loss.backward()
    _check_step_grads(f, model, leaves)


@pytest.mark.parametrize("act", ["silu", "relu"])
@pytest.mark.parametrize("n,e,dims", [(2000, 32000, (128, 16)), (600, 6000, (256, 32)), (600, 6000, (64, 8))],
                         ids=["C2-dims", "C5-dims", "small-V"])
def test_interactions_large_vs_oracle(G, n, e, dims, act):
    """Bench-shaped (but smaller) layers, col-sorted edges, edge dims (32,4), fwd + bwd: BASELINE configs[1] dims (128,16)
    (register-resident chain kernels), configs[4] dims (256,32) (two output groups: block-by-block kernels), and (64,8):
    H = 2, one register quad per vector quantity (the chain kernels' run-time-H instantiations)."""
    torch.manual_seed(11)
    layer = G.GCPInteractions(dims, (32, 4), cfg=G.default_module_cfg(scalar_nonlinearity=act), layer_cfg=G.default_layer_cfg(),
                              dropout=0.0).cuda().eval()
    ei, x = rand_graph(n, e, 12, sort_by_col=True)
    fr = O.localize(x, ei)
    g = torch.Generator().manual_seed(13)
    ins = dict(h=torch.randn(n, dims[0], generator=g), chi=torch.randn(n, dims[1], 3, generator=g),
               e=torch.randn(e, 32, generator=g), xi=torch.randn(e, 4, 3, generator=g))
    P = {k: t.detach().cpu().clone().requires_grad_() for k, t in layer.state_dict().items()}
    ci = {k: t.clone().requires_grad_() for k, t in ins.items()}
    wh, wc = O.gcp_interactions(P, "", ci["h"], ci["chi"], ci["e"], ci["xi"], ei, fr,
                                O.default_module_cfg(scalar_nonlinearity=act, nonlinearities=(act, None)), O.default_layer_cfg())
    gi = {k: t.cuda().requires_grad_() for k, t in ins.items()}
    gh, gc = layer((gi["h"], gi["chi"]), (gi["e"], gi["xi"]), ei.cuda(), fr.cuda())
    close(gh.detach().cpu(), wh.detach(), atol=1e-5, rtol=1e-5)
    close(gc.detach().cpu(), wc.detach(), atol=1e-5, rtol=1e-5)
    # a random linear functional of the outputs: |LayerNorm(x)|^2 is constant for gamma = 1, beta = 0, so a squared loss on
    # this post-norm layer would have (analytically) zero gradient through the scalar path and only compare round-off
    lh, lc = torch.randn(wh.shape, generator=g), torch.randn(wc.shape, generator=g)
    ((wh * lh).sum() + (wc * lc).sum()).backward()
    ((gh * lh.cuda()).sum() + (gc * lc.cuda()).sum()).backward()
    # silu: smooth, so the two fp32 implementations must agree tightly everywhere.  relu: a pre-activation within round-off of
    # zero (expected for a few of the ~1e7 units here) takes a different branch in the two implementations and changes the
    # gradient of its row by O(1/s) (and every weight gradient a little): agreement in the L2 sense, loose element-wise bound.
    def grads_close(a, b):
        scale = float(b.abs().max())
        if act == "silu":
            close(a, b, atol=2e-5 * scale, rtol=1e-4)
        else:
            rel_l2 = float((a.double() - b.double()).norm() / b.double().norm().clamp(min=1e-30))
            assert rel_l2 < 2e-3, f"relative L2 error {rel_l2:.2e}"
            close(a, b, atol=5e-2 * scale, rtol=0.0)

    for k in ins:
        grads_close(gi[k].grad.cpu(), ci[k].grad)
    for k, p in layer.named_parameters():
        grads_close(p.grad.cpu(), P[k].grad)


def _layer_run(G, layer, ins, ei, fr):
    gi = {k: t.clone().cuda().requires_grad_() for k, t in ins.items()}
    for p in layer.parameters():
        p.grad = None
    gh, gc = layer((gi["h"], gi["chi"]), (gi["e"], gi["xi"]), ei.cuda(), fr.cuda())
    sq_loss(gh, gc).backward()
    torch.cuda.synchronize()
    out = {"h": gh.detach().cpu(), "chi": gc.detach().cpu()}
    out.update({"d_" + k: t.grad.cpu() for k, t in gi.items()})
    out.update({"w_" + k: p.grad.cpu().clone() for k, p in layer.named_parameters()})
    return out


def test_execution_variants_agree_and_are_deterministic(G):
    """The host-side execution choices -- project-then-gather for gathered scalars / vectors, weight gradients on a second
    stream -- change summation order at most: every variant matches the default to fp32 round-off, and the default is
    bit-reproducible run to run (no atomics anywhere on the path)."""
    from gcpnet_amd import ops
    torch.manual_seed(5)
    n, e = 700, 9000
    layer = G.GCPInteractions((128, 16), (32, 4), cfg=G.default_module_cfg(), layer_cfg=G.default_layer_cfg(),
                              dropout=0.0).cuda().eval()
    ei, x = rand_graph(n, e, 12, sort_by_col=True)
    fr = O.localize(x, ei)
    g = torch.Generator().manual_seed(17)
    ins = dict(h=torch.randn(n, 128, generator=g), chi=torch.randn(n, 16, 3, generator=g),
               e=torch.randn(e, 32, generator=g), xi=torch.randn(e, 4, 3, generator=g))
    ref = _layer_run(G, layer, ins, ei, fr)
    again = _layer_run(G, layer, ins, ei, fr)
    for k in ref:
        assert torch.equal(ref[k], again[k]), f"{k} differs between two identical runs"
    saved = (ops.PROJECT_GATHERED_SCALARS, ops.PROJECT_GATHERED_VECTORS, ops.WEIGHT_GRADS_ON_SIDE_STREAM)
    try:
        for flags in [(False, False, True), (True, False, True), (True, True, False)]:
            ops.PROJECT_GATHERED_SCALARS, ops.PROJECT_GATHERED_VECTORS, ops.WEIGHT_GRADS_ON_SIDE_STREAM = flags
            alt = _layer_run(G, layer, ins, ei, fr)
            for k in ref:
                close(alt[k], ref[k], atol=2e-5 if k in ("h", "chi") else 1e-6, rtol=2e-3)
    finally:
        ops.PROJECT_GATHERED_SCALARS, ops.PROJECT_GATHERED_VECTORS, ops.WEIGHT_GRADS_ON_SIDE_STREAM = saved
