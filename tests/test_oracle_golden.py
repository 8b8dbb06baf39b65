"""Pins the CPU oracle (oracle/gcp_oracle.py) to golden vectors produced by the real reference
(tests/golden/gen_fixtures.py).  CPU only."""
import pytest
import torch

from oracle import gcp_oracle as O
from tests.helpers import Fixture, close

TOL = dict(atol=2e-6, rtol=2e-5)


def _grads(loss, tensors):
    return torch.autograd.grad(loss, tensors, allow_unused=True)


def sq_loss(*ts):
    return sum((t * t).mean() for t in ts if t is not None and t.numel())


def test_geometry():
    f = Fixture("geometry")
    x, ei = f.i["x"], f.i["edge_index"]
    fr = O.localize(x, ei, True)
    close(fr, f.o["frames"], **TOL)
    close(O.localize(x, ei, False), f.o["frames_raw"], **TOL)
    close(O.scalarize(f.i["vec_e"], ei, fr, False, False, 64), f.o["scalarize_edge"], **TOL)
    close(O.scalarize(f.i["vec_n"], ei, fr, True, False, 20), f.o["scalarize_node"], **TOL)
    close(O.scalarize(f.i["vec_e"], ei, fr, False, True, 64), f.o["scalarize_edge_e3"], **TOL)
    close(O.vectorize(f.i["gate_e"], ei, fr, False, 64), f.o["vectorize_edge"], **TOL)
    close(O.vectorize(f.i["gate_n"], ei, fr, True, 20), f.o["vectorize_node"], **TOL)
    c, xc = O.centralize(x, f.i["batch"])
    close(c, f.o["centroid"], **TOL)
    close(xc, f.o["x_centered"], **TOL)
    close(O.decentralize(xc, f.i["batch"], c), f.o["x_back"], **TOL)
    close(O.safe_norm(f.i["vec_e"], dim=-2), f.o["safe_norm"], **TOL)


GCP2_CASES = {
    "gcp2_edge_msg0": dict(nonlinearities=("relu", None)),
    "gcp2_edge_res": dict(nonlinearities=("relu", None)),
    "gcp2_node_ff0": dict(nonlinearities=("relu", None)),
    "gcp2_node_ff1": dict(nonlinearities=(None, None)),
    "gcp2_node_scalar_only": dict(nonlinearities=("relu", None)),
    "gcp2_no_vector_in": dict(nonlinearities=("relu", None), vector_output_dim=3),
    "gcp2_node_posupd": dict(nonlinearities=("relu", None)),
    "gcp2_silu_sigmoid": dict(nonlinearities=("silu", "sigmoid")),
    "gcp2_selfgate": dict(nonlinearities=("silu", "sigmoid"), vector_gate=False),
    "gcp2_vres_e3": dict(nonlinearities=("leakyrelu", None), vector_residual=True, enable_e3_equivariance=True),
    "gcp2_frame_gate": dict(nonlinearities=("silu", "silu"), frame_gate=True),
    "gcp2_frame_gate_edge": dict(nonlinearities=("relu", "sigmoid"), frame_gate=True),
    "gcp2_ablate_frames": dict(nonlinearities=("relu", None), ablate_frame_updates=True),
    # GCP3 without feedforward_out is GCP2 with silu defaults (reference gcpnet.py:471-700)
    "gcp3_edge_default": dict(nonlinearities=("silu", "silu")),
    "gcp3_node_default": dict(nonlinearities=("silu", "silu")),
    # feedforward_out: scalar_out = Linear -> act -> Linear (gcpnet.py:529-533, :552-556)
    "gcp3_feedforward": dict(nonlinearities=("silu", "silu")),
    "gcp3_feedforward_node": dict(nonlinearities=(None, None)),
    "gcp3_feedforward_scalar": dict(nonlinearities=("silu", "silu"), scalar_out_nonlinearity="relu"),
}


@pytest.mark.parametrize("name", sorted(GCP2_CASES))
def test_gcp2(name):
    f = Fixture(name)
    kw = GCP2_CASES[name]
    P = {k: v.clone().requires_grad_() for k, v in f.p.items()}
    s = f.i["s"].clone().requires_grad_()
    v = f.i["v"].clone().requires_grad_() if "v" in f.i else None
    out = O.gcp2(P, "", s, v, f.i["edge_index"], f.i["frames"], node_inputs=bool(f.m["node_inputs"]), **kw)
    outs = dict(s=out[0], v=out[1]) if isinstance(out, tuple) else dict(s=out)
    for k in outs:
        close(outs[k], f.o[k], **TOL)
    loss = sq_loss(*outs.values())
    names = ["s"] + (["v"] if v is not None else []) + ["w." + k for k in P]
    tens = [s] + ([v] if v is not None else []) + list(P.values())
    for n, g in zip(names, _grads(loss, tens)):
        if n in f.g:
            close(g, f.g[n], atol=2e-6, rtol=1e-4)
        else:
            assert g is None or float(g.abs().max()) == 0.0


GCP_CASES = {  # the original GCP block (gcpnet.py:30-249)
    "gcp_edge_default": dict(),
    "gcp_node_default": dict(nonlinearities=("silu", "sigmoid")),
    "gcp_sigma_gate": dict(nonlinearities=("relu", "sigmoid"), sigma_frame_gate=True, vector_residual=True),
    "gcp_frame_gate": dict(nonlinearities=("silu", "silu"), frame_gate=True, vector_frame_residual=True),
    "gcp_selfgate_e3": dict(nonlinearities=("silu", "sigmoid"), vector_gate=False, enable_e3_equivariance=True),
    "gcp_node_e3": dict(nonlinearities=("silu", "sigmoid"), sigma_frame_gate=True, enable_e3_equivariance=True),
    "gcp_scalar_out": dict(nonlinearities=("relu", None)),
    "gcp_ablate_frames": dict(nonlinearities=("relu", None), ablate_frame_updates=True),
}


@pytest.mark.parametrize("name", sorted(GCP_CASES))
def test_gcp_original(name):
    f = Fixture(name)
    P = {k: v.clone().requires_grad_() for k, v in f.p.items()}
    s = f.i["s"].clone().requires_grad_()
    v = f.i["v"].clone().requires_grad_()
    out = O.gcp(P, "", s, v, f.i["edge_index"], f.i["frames"], node_inputs=bool(f.m["node_inputs"]), **GCP_CASES[name])
    outs = dict(s=out[0], v=out[1]) if isinstance(out, tuple) else dict(s=out)
    for k in outs:
        close(outs[k], f.o[k], **TOL)
    loss = sq_loss(*outs.values())
    names = ["s", "v"] + ["w." + k for k in P]
    tens = [s, v] + list(P.values())
    for n, g in zip(names, _grads(loss, tens)):
        if n in f.g:
            close(g, f.g[n], atol=2e-6, rtol=1e-4)
        else:
            assert g is None or float(g.abs().max()) == 0.0


def test_layernorm():
    f = Fixture("layernorm")
    so, vo = O.gcp_layer_norm(f.p, "", f.i["s"], f.i["v"])
    close(so, f.o["s"], **TOL)
    close(vo, f.o["v"], **TOL)


@pytest.mark.parametrize("name", ["embedding_nms", "embedding_lba"])
def test_embedding(name):
    f = Fixture(name)
    i = f.i
    (h, chi), (e, xi) = O.gcp_embedding(f.p, "", i["h"], i["chi"], i["e"], i["xi"], i["edge_index"], i["frames"],
                                        O.default_module_cfg())
    for k, t in dict(h=h, chi=chi, e=e, xi=xi).items():
        close(t, f.o[k], **TOL)


def test_message_passing():
    f = Fixture("message_passing")
    i = f.i
    P = {k: v.clone().requires_grad_() for k, v in f.p.items()}
    ins = {k: i[k].clone().requires_grad_() for k in ("h", "chi", "e", "xi")}
    cfg, lc = O.default_module_cfg(), O.default_layer_cfg()
    (s, v), msg = O.message_passing(P, "", ins["h"], ins["chi"], ins["e"], ins["xi"], i["edge_index"], i["frames"],
                                    cfg, lc["mp_cfg"], return_messages=True)
    close(msg, f.o["messages"], atol=5e-6, rtol=5e-5)
    close(s, f.o["s"], atol=5e-6, rtol=5e-5)
    close(v, f.o["v"], atol=5e-6, rtol=5e-5)
    names = list(ins) + ["w." + k for k in P]
    for n, g in zip(names, _grads(sq_loss(s, v), list(ins.values()) + list(P.values()))):
        close(g, f.g[n], atol=5e-6, rtol=5e-4)


@pytest.mark.parametrize("flag", ["ablate_scalars", "ablate_vectors"])
def test_message_passing_ablations(flag):
    """ablate_scalars / ablate_vectors (reference gcpnet.py:416-417,466-467) through the whole message function."""
    f = Fixture("message_passing_" + flag)
    i = f.i
    P = {k: v.clone().requires_grad_() for k, v in f.p.items()}
    ins = {k: i[k].clone().requires_grad_() for k in ("h", "chi", "e", "xi")}
    cfg, lc = O.default_module_cfg(**{flag: True}), O.default_layer_cfg()
    (s, v), msg = O.message_passing(P, "", ins["h"], ins["chi"], ins["e"], ins["xi"], i["edge_index"], i["frames"],
                                    cfg, lc["mp_cfg"], return_messages=True)
    close(msg, f.o["messages"], atol=5e-6, rtol=5e-5)
    close(s, f.o["s"], atol=5e-6, rtol=5e-5)
    close(v, f.o["v"], atol=5e-6, rtol=5e-5)
    loss = (s * i["lw_s"]).sum() + (v * i["lw_v"]).sum() + sq_loss(s, v)
    names = list(ins) + ["w." + k for k in P]
    for n, g in zip(names, torch.autograd.grad(loss, list(ins.values()) + list(P.values()), allow_unused=True)):
        g = torch.zeros_like(f.g[n]) if g is None else g
        close(g, f.g[n], atol=5e-6 * max(1.0, float(f.g[n].abs().max())), rtol=5e-4)


@pytest.mark.parametrize("name", ["interactions", "interactions_posupd", "interactions_force"])
def test_interactions(name):
    f = Fixture(name)
    i = f.i
    P = {k: v.clone().requires_grad_() for k, v in f.p.items()}
    ins = {k: i[k].clone().requires_grad_() for k in ("h", "chi", "e", "xi")}
    cfg, lc = O.default_module_cfg(), O.default_layer_cfg()
    upd = name != "interactions"  # ("interactions_force": position update with the inter-node force term, :1143-1153)
    out = O.gcp_interactions(P, "", ins["h"], ins["chi"], ins["e"], ins["xi"], i["edge_index"], i["frames"], cfg, lc,
                             node_pos=i["x"] if upd else None)
    outs = dict(h=out[0][0], chi=out[0][1], x=out[1]) if upd else dict(h=out[0], chi=out[1])
    for k, t in outs.items():
        close(t, f.o[k], atol=5e-6, rtol=5e-5)
    names = list(ins) + ["w." + k for k in P]
    gr = _grads(sq_loss(*outs.values()), list(ins.values()) + list(P.values()))
    for n, g in zip(names, gr):
        if n in f.g:
            close(g, f.g[n], atol=5e-6, rtol=1e-3)


def test_geometry_masked():
    """node_mask branches of centralize / localize / scalarize / vectorize (components/__init__.py:177-193,229-264,294-300,346-357)."""
    f = Fixture("geometry_masked")
    i = f.i
    m, ei = i["mask"].bool(), i["edge_index"]
    fr = O.localize(i["x"], ei, node_mask=m)
    assert torch.equal(torch.isinf(fr), torch.isinf(f.o["frames"]))
    fin = torch.isfinite(f.o["frames"])
    close(fr[fin], f.o["frames"][fin], **TOL)
    cen, xc = O.centralize(i["x"], i["batch"], node_mask=m)
    close(cen, f.o["centroid"], **TOL)
    assert torch.equal(torch.isinf(xc), torch.isinf(f.o["x_centered"]))
    close(xc[m], f.o["x_centered"][m], **TOL)
    close(O.scalarize(i["vec_e"], ei, f.o["frames"], False, False, 96, node_mask=m), f.o["scalarize_edge"], **TOL)
    close(O.scalarize(i["vec_n"], ei, f.o["frames"], True, True, 24, node_mask=m), f.o["scalarize_node"], **TOL)
    close(O.vectorize(i["gate_e"], ei, f.o["frames"], False, 96, node_mask=m), f.o["vectorize_edge"], **TOL)
    close(O.vectorize(i["gate_n"], ei, f.o["frames"], True, 24, node_mask=m), f.o["vectorize_node"], **TOL)


def test_gcp2_masked_node():
    f = Fixture("gcp2_masked_node")
    P = {k: v.clone().requires_grad_() for k, v in f.p.items()}
    s, v = f.i["s"].clone().requires_grad_(), f.i["v"].clone().requires_grad_()
    out = O.gcp2(P, "", s, v, f.i["edge_index"], f.i["frames"], node_inputs=True, nonlinearities=("silu", "sigmoid"),
                 node_mask=f.i["mask"].bool())
    close(out[0], f.o["s"], **TOL)
    close(out[1], f.o["v"], **TOL)
    names = ["s", "v"] + ["w." + k for k in P]
    for n, g in zip(names, _grads(sq_loss(*out), [s, v] + list(P.values()))):
        if n in f.g:
            close(g, f.g[n], atol=2e-6, rtol=1e-4)


@pytest.mark.parametrize("name", ["interactions_masked", "interactions_masked_posupd", "interactions_masked_all",
                                  "interactions_autoregressive"])
def test_interactions_masked_and_autoregressive(name):
    """GCPInteractions with node_mask (sub-graph feed-forward, gcpnet.py:1201-1251) and autoregressive_forward (:1066-1116)."""
    f = Fixture(name)
    i = f.i
    P = {k: v.clone().requires_grad_() for k, v in f.p.items()}
    keys = ("h", "chi", "e", "xi") + (("h_reg", "chi_reg") if name == "interactions_autoregressive" else ())
    ins = {k: i[k].clone().requires_grad_() for k in keys}
    cfg = O.default_module_cfg()
    lc = O.default_layer_cfg(pre_norm=bool(int(f.m["pre_norm"]))) if "pre_norm" in f.m else O.default_layer_cfg()
    upd = name == "interactions_masked_posupd"
    mask = i["mask"].bool() if "mask" in i else None
    reg = (ins["h_reg"], ins["chi_reg"]) if name == "interactions_autoregressive" else None
    out = O.gcp_interactions(P, "", ins["h"], ins["chi"], ins["e"], ins["xi"], i["edge_index"], i["frames"], cfg, lc,
                             node_pos=i["x"] if upd else None, node_mask=mask, regressive=reg)
    outs = dict(h=out[0][0], chi=out[0][1], x=out[1]) if upd else dict(h=out[0], chi=out[1])
    for k, t in outs.items():
        fin = torch.isfinite(f.o[k])
        assert torch.equal(torch.isfinite(t), fin), k
        close(t[fin], f.o[k][fin], atol=5e-6, rtol=5e-5)
    if upd:
        fin = torch.isfinite(outs["x"]).all(dim=1)
        loss = sq_loss(outs["h"], outs["chi"], outs["x"][fin])
    else:
        loss = sq_loss(*outs.values())
    names = list(ins) + ["w." + k for k in P]
    for n, g in zip(names, _grads(loss, list(ins.values()) + list(P.values()))):
        if n in f.g:
            close(g, f.g[n], atol=5e-6, rtol=1e-3)


@pytest.mark.parametrize("name,res", [("mlp_decoder", False), ("mlp_decoder_residual", True)])
def test_mlp_decoder(name, res):
    f = Fixture(name)
    P = {k: v.clone().requires_grad_() for k, v in f.p.items()}
    h = f.i["h"].clone().requires_grad_()
    logits, logp = O.mlp_decoder(P, "", h, residual_updates=res)
    close(logits, f.o["logits"], **TOL)
    close(logp, f.o["log_probs"], **TOL)
    loss = sq_loss(logits) + (logp * f.i["lw"]).mean()
    for n, g in zip(["h"] + ["w." + k for k in P], _grads(loss, [h] + list(P.values()))):
        close(g, f.g[n], atol=2e-6, rtol=1e-4)


INTERACTIONS2 = {  # GCPInteractions2 as gcpnet_eq.yaml builds it (GCP3, gate, sum over row, 1 FF GCP); 2-FF variant + positions
    "interactions2_eq": dict(use_scalar_message_attention=True, aggregate_with_row=True, num_feedforward_layers=1),
    "interactions2_posupd": dict(use_scalar_message_attention=True, num_message_layers=4, num_feedforward_layers=2),
}


@pytest.mark.parametrize("name", sorted(INTERACTIONS2))
def test_interactions2(name):
    f = Fixture(name)
    i = f.i
    P = {k: v.clone().requires_grad_() for k, v in f.p.items()}
    ins = {k: i[k].clone().requires_grad_() for k in ("h", "chi", "e", "xi")}
    cfg, lc = O.default_module_cfg(), O.default_layer_cfg(**INTERACTIONS2[name])
    upd = name == "interactions2_posupd"
    out = O.gcp_interactions2(P, "", ins["h"], ins["chi"], ins["e"], ins["xi"], i["edge_index"], i["frames"], cfg, lc,
                              node_pos=i["x"] if upd else None)
    outs = dict(h=out[0][0], chi=out[0][1], x=out[1]) if upd else dict(h=out[0], chi=out[1])
    for k, t in outs.items():
        close(t, f.o[k], atol=5e-6, rtol=5e-5)
    names = list(ins) + ["w." + k for k in P]
    gr = _grads(sq_loss(*outs.values()), list(ins.values()) + list(P.values()))
    for n, g in zip(names, gr):
        if n in f.g:
            close(g, f.g[n], atol=5e-6, rtol=1e-3)


def test_interactions_prenorm_silu():
    f = Fixture("interactions_prenorm_silu")
    i = f.i
    cfg = O.default_module_cfg(scalar_nonlinearity="silu", vector_nonlinearity="silu", nonlinearities=("silu", "silu"))
    lc = O.default_layer_cfg(pre_norm=True, num_feedforward_layers=3, num_message_layers=3)
    h, chi = O.gcp_interactions(f.p, "", i["h"], i["chi"], i["e"], i["xi"], i["edge_index"], i["frames"], cfg, lc)
    close(h, f.o["h"], atol=5e-6, rtol=5e-5)
    close(chi, f.o["chi"], atol=5e-6, rtol=5e-5)


def _step_grads(f, forward, pred_key, float_inputs):
    """step() of the LitModules (gcpnet_nms_module.py:153-158, gcpnet_lba_module.py:188-193): MSELoss(pred, label), backward to
    the float inputs and every parameter; compared with the gradients the reference produced (fixture tag g)."""
    P = {k: t.clone().requires_grad_(t.is_floating_point()) for k, t in f.p.items()}
    ins = dict(f.i)
    for k in float_inputs:
        ins[k] = ins[k].clone().requires_grad_()
    out = forward(P, ins)
    loss = torch.nn.functional.mse_loss(out[pred_key], ins["label"])
    close(loss.detach(), f.o["loss"], atol=1e-6, rtol=1e-5)
    loss.backward()
    n = 0
    for k, want in f.g.items():
        got = P[k[2:]].grad if k.startswith("w.") else ins[k].grad
        assert got is not None, k
        close(got, want, atol=1e-6 + 1e-5 * float(want.abs().max()), rtol=1e-4)
        n += 1
    assert n > 50
    return out


def test_model_nms():
    f = Fixture("model_nms_small")
    out = _step_grads(f, lambda P, i: O.nms_forward(P, i, O.default_module_cfg(), O.default_layer_cfg(num_message_layers=4), 2),
                      "x", ("h", "chi", "e", "xi"))
    for k in ("h", "chi", "e", "xi", "x", "f_ij"):
        close(out[k].detach(), f.o[k], atol=1e-5, rtol=1e-4)


def test_model_lba():
    f = Fixture("model_lba_small")
    out = _step_grads(f, lambda P, i: O.lba_forward(P, i, O.default_module_cfg(), O.default_layer_cfg(num_message_layers=4), 2),
                      "pred", ("chi", "e", "xi"))
    for k in ("h", "chi", "pred"):
        close(out[k].detach(), f.o[k], atol=1e-5, rtol=1e-4)


def test_oracle_equivariance():
    """The property the reference's (disabled) tests assert, tests/test_gcpnet_equivariance.py:1773-1881."""
    f = Fixture("interactions_posupd")
    i = f.i
    cfg, lc = O.default_module_cfg(), O.default_layer_cfg()
    Q = O.random_rotation(3)
    x = i["x"]
    fr = O.localize(x, i["edge_index"])
    (h0, c0), x0 = O.gcp_interactions(f.p, "", i["h"], i["chi"], i["e"], i["xi"], i["edge_index"], fr, cfg, lc, node_pos=x)
    xr = x @ Q.t()
    (h1, c1), x1 = O.gcp_interactions(f.p, "", i["h"], i["chi"] @ Q.t(), i["e"], i["xi"] @ Q.t(), i["edge_index"],
                                      O.localize(xr, i["edge_index"]), cfg, lc, node_pos=xr)
    close(h1, h0, atol=1e-5, rtol=1e-4)
    close(c1, c0 @ Q.t(), atol=1e-5, rtol=1e-4)
    close(x1, x0 @ Q.t(), atol=1e-5, rtol=1e-4)


def test_nms_features_golden():
    """Oracle restatement of the NMS featuriser against the fixture built with the reference's real helper functions."""
    f = Fixture("nms_features")
    out = O.nms_features(f.i["x"], f.i["vel"], f.i["edge_attr"], f.i["edge_index"], f.i["batch"])
    for k in ("h", "chi", "e", "xi"):
        close(out[k], f.o[k], atol=1e-6, rtol=1e-6)


def test_radius_graph_golden():
    """Brute-force oracle against the scipy cKDTree edge list (the fixture): identical index arrays."""
    f = Fixture("radius_graph")
    ei = O.radius_graph(f.i["x"], f.i["batch"], float(f.m["radius"]), int(f.m["max_neighbors"]))
    assert torch.equal(ei, f.o["edge_index"])


def _by_col_row(ei):
    """Canonical edge order (target, then source): the order inside a target's neighbour list is implementation-defined."""
    key = ei[1] * (int(ei.max()) + 1) + ei[0]
    return torch.argsort(key)


def test_lba_features_capped_golden():
    """The neighbour cap (atom3d_dataset.py:110-112): the oracle's restatement of torch_cluster's index-order selection against the
    fixture the reference's real LBATransform produced on a dense structure (edge lists identical, order included)."""
    f = Fixture("lba_features_capped")
    x = f.i["x"]
    ei = O.radius_graph(x, torch.zeros(x.shape[0], dtype=torch.long), 4.5, 32, select="first")
    assert torch.equal(ei, f.o["edge_index"])
    deg = torch.bincount(ei[1], minlength=x.shape[0])
    assert int(deg.max()) == 33 and int((deg >= 32).sum()) > 50  # the cap binds, incl. upstream's 33-neighbour case
    near = O.radius_graph(x, torch.zeros(x.shape[0], dtype=torch.long), 4.5, 32)
    assert near.shape[1] != ei.shape[1] or not torch.equal(near[:, _by_col_row(near)], ei)  # "nearest" is another graph here
    out = O.lba_features(x, ei)
    close(out["e"], f.o["e"], atol=1e-6, rtol=1e-6)
    close(out["xi"], f.o["xi"], atol=1e-6, rtol=1e-6)
    close(out["chi"], f.o["chi"], atol=1e-6, rtol=1e-6)


def test_lba_features_golden():
    """Oracle restatement of the ATOM3D / LBA featuriser and of PyG collation against the fixture built by the reference's real
    LBATransform (atom3d_dataset.py:134-149) on two pocket + ligand structures."""
    f = Fixture("lba_features")
    graphs = []
    for k in range(2):
        x = f.i[f"x{k}"]
        ei = O.radius_graph(x, torch.zeros(x.shape[0], dtype=torch.long), 4.5, 32)
        pw, pg = _by_col_row(f.o[f"edge_index{k}"]), _by_col_row(ei)
        assert torch.equal(ei[:, pg], f.o[f"edge_index{k}"][:, pw])  # same edge SET (the fixture stays below the neighbour cap)
        out = O.lba_features(x, ei)
        close(out["e"][pg], f.o[f"e{k}"][pw], atol=1e-6, rtol=1e-6)
        close(out["xi"][pg], f.o[f"xi{k}"][pw], atol=1e-6, rtol=1e-6)
        close(out["chi"], f.o[f"chi{k}"], atol=1e-6, rtol=1e-6)
        n_lig = int(f.i[f"n_ligand{k}"])
        flag = torch.zeros(x.shape[0], dtype=torch.bool)
        flag[x.shape[0] - n_lig:] = True
        assert torch.equal(flag, f.o[f"lig_flag{k}"])
        graphs.append(dict(h=f.o[f"h{k}"], chi=f.o[f"chi{k}"], e=f.o[f"e{k}"], xi=f.o[f"xi{k}"], x=x, edge_index=f.o[f"edge_index{k}"],
                           lig_flag=f.o[f"lig_flag{k}"], label=f.o[f"label{k}"]))
    b = O.collate(graphs)
    for key in ("h", "chi", "e", "xi", "x", "lig_flag", "edge_index", "label", "batch"):
        assert torch.equal(b[key], f.o["batch_" + key].to(b[key].dtype)), key
    assert b["ptr"].tolist() == [0, graphs[0]["x"].shape[0], graphs[0]["x"].shape[0] + graphs[1]["x"].shape[0]]


@pytest.mark.parametrize("name", ["model_cpd_small", "model_cpd_direct"])
def test_cpd_model_golden(name):
    """Oracle restatement of the CPD task module (forward with teacher forcing / MLP decoder, cross-entropy step, every gradient,
    and the autoregressive sampling loop with an argmax sampler) against the fixtures generated by the reference's real
    GCPNetCPDLitModule (gcpnet_cpd_module.py:153-231, 281-360)."""
    f = Fixture(name)
    ar = bool(f.m["autoregressive"])
    P = {k: (v.clone().requires_grad_() if v.is_floating_point() else v) for k, v in f.p.items()}
    b = dict(f.i)
    for k in ("h", "chi", "e", "xi"):
        b[k] = b[k].clone().requires_grad_()
    cfg, lc = O.default_module_cfg(), O.default_layer_cfg(num_message_layers=int(f.m["num_message_layers"]))
    out = O.cpd_forward(P, b, cfg, lc, int(f.m["num_encoder_layers"]), int(f.m["num_decoder_layers"]), ar)
    preds = out["out"] if ar else out["out"][0]
    close(preds, f.o["preds"], atol=2e-5, rtol=1e-4)
    close(out["h"], f.o["h"], atol=1e-5, rtol=1e-4)
    loss = torch.nn.functional.cross_entropy(preds[b["mask"]], b["seq"][b["mask"]])
    close(loss, f.o["loss"], atol=1e-5, rtol=1e-5)
    loss.backward()
    for k in ("h", "chi", "e", "xi"):
        close(b[k].grad, f.g[k], atol=1e-6 + 2e-5 * float(f.g[k].abs().max()), rtol=1e-3)
    n = 0
    for k, p in P.items():
        if "w." + k in f.g:
            g = p.grad if p.grad is not None else torch.zeros_like(p)
            close(g, f.g["w." + k], atol=1e-6 + 2e-5 * float(f.g["w." + k].abs().max()), rtol=1e-3)
            n += 1
    assert n > 100
    if ar:
        with torch.no_grad():
            Pd = {k: v.detach() for k, v in P.items()}
            seqs = O.cpd_sample(Pd, f.i["h"], f.i["chi"], f.i["e"], f.i["xi"], f.i["edge_index"], f.o["sample_frames"], f.i["mask"], cfg, lc,
                                int(f.m["num_encoder_layers"]), int(f.m["num_decoder_layers"]), num_samples=2, temperature=0.1,
                                sampler=lambda lg: lg.argmax(-1))
        assert torch.equal(seqs, f.o["samples"].long())
