"""Generate golden input/output vectors by running the REAL reference (imported from /root/reference through
``ref_stubs``) on seeded inputs.  Authoring-container only; the outputs (``*.npz`` next to this file) are the
committed fixtures that pin ``oracle/gcp_oracle.py``.

    python tests/golden/gen_fixtures.py

Each fixture stores ``p/<state_dict key>``, ``i/<input>``, ``o/<output>``, ``g/<grad of input or weight>``.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_stubs  # noqa: E402

comp, gn, ref_models = ref_stubs.load_reference()
SV = comp.ScalarVector


def save(name, params=None, inputs=None, outputs=None, grads=None, meta=None):
    blob = {}
    for tag, d in (("p", params), ("i", inputs), ("o", outputs), ("g", grads)):
        for k, v in (d or {}).items():
            blob[f"{tag}/{k}"] = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
    for k, v in (meta or {}).items():
        blob[f"m/{k}"] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **blob)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB, {len(blob)} arrays")


def rand_graph(n, e, seed, no_self=True):
    g = torch.Generator().manual_seed(seed)
    row = torch.randint(0, n, (e,), generator=g)
    col = torch.randint(0, n, (e,), generator=g)
    if no_self:
        col = torch.where(col == row, (col + 1) % n, col)
    x = torch.randn(n, 3, generator=g)
    return torch.stack((row, col)), x


def randn(*shape, seed):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def sq_loss(*ts):
    return sum((t * t).mean() for t in ts if t is not None and t.numel())


# ------------------------------------------------------------------------------------------------------------
def fx_geometry():
    ei, x = rand_graph(20, 64, 1)
    f = comp.localize(x, ei, norm_x_diff=True)
    f_raw = comp.localize(x, ei, norm_x_diff=False)
    vec_e = randn(64, 3, 3, seed=2)
    vec_n = randn(20, 3, 3, seed=3)
    bidx = torch.tensor([0] * 7 + [1] * 8 + [2] * 5)
    bag = ref_stubs.Bag(x=x)
    cen, xc = comp.centralize(bag, "x", bidx)
    bag2 = ref_stubs.Bag(x=xc)
    back = comp.decentralize(bag2, "x", bidx, cen)
    gate_e, gate_n = randn(64, 9, seed=4), randn(20, 9, seed=5)
    save("geometry",
         inputs=dict(x=x, edge_index=ei, vec_e=vec_e, vec_n=vec_n, batch=bidx, gate_e=gate_e, gate_n=gate_n),
         outputs=dict(
             frames=f, frames_raw=f_raw,
             scalarize_edge=comp.scalarize(vec_e, ei, f, False, False, 64),
             scalarize_node=comp.scalarize(vec_n, ei, f, True, False, 20),
             scalarize_edge_e3=comp.scalarize(vec_e, ei, f, False, True, 64),
             vectorize_edge=comp.vectorize(gate_e, ei, f, False, 64),
             vectorize_node=comp.vectorize(gate_n, ei, f, True, 20),
             centroid=cen, x_centered=xc, x_back=back,
             safe_norm=comp.safe_norm(vec_e, dim=-2),
         ))


def run_gcp2(name, in_dims, out_dims, rows_are_nodes, seed, n=20, e=64, cls="GCP2", **kw):
    torch.manual_seed(seed)
    mod = getattr(gn, cls)(SV(*in_dims), SV(*out_dims), **kw)
    ei, x = rand_graph(n, e, seed + 100)
    frames = comp.localize(x, ei)
    rows = n if rows_are_nodes else e
    s = randn(rows, in_dims[0], seed=seed + 1).requires_grad_()
    if in_dims[1]:
        v = randn(rows, in_dims[1], 3, seed=seed + 2).requires_grad_()
        out = mod((s, v), ei, frames, node_inputs=rows_are_nodes)
    else:
        v = None
        out = mod(s, ei, frames, node_inputs=rows_are_nodes)
    outs = dict(s=out[0], v=out[1]) if isinstance(out, tuple) else dict(s=out)
    loss = sq_loss(*outs.values())
    loss.backward()
    grads = {"s": s.grad}
    if v is not None:
        grads["v"] = v.grad
    for k, p in mod.named_parameters():
        if p.grad is not None:
            grads["w." + k] = p.grad
    ins = dict(s=s, edge_index=ei, frames=frames)
    if v is not None:
        ins["v"] = v
    meta = dict(node_inputs=int(rows_are_nodes), in_dims=in_dims, out_dims=out_dims)
    save(name, params=mod.state_dict(), inputs=ins, outputs=outs, grads=grads, meta=meta)


def fx_gcp2():
    relu_none = ("relu", None)
    run_gcp2("gcp2_edge_msg0", (160, 36), (64, 16), False, 10, nonlinearities=relu_none, bottleneck=4)
    run_gcp2("gcp2_edge_res", (64, 16), (64, 16), False, 11, nonlinearities=relu_none, bottleneck=4)
    run_gcp2("gcp2_node_ff0", (64, 16), (256, 32), True, 12, nonlinearities=relu_none, bottleneck=4)
    run_gcp2("gcp2_node_ff1", (256, 32), (64, 16), True, 13, nonlinearities=(None, None), bottleneck=4)
    run_gcp2("gcp2_node_scalar_only", (100, 16), (100, 0), True, 14, nonlinearities=relu_none)
    run_gcp2("gcp2_no_vector_in", (12, 0), (8, 3), False, 15, nonlinearities=relu_none)
    run_gcp2("gcp2_node_posupd", (64, 16), (64, 1), True, 16, nonlinearities=relu_none, bottleneck=4)
    run_gcp2("gcp2_silu_sigmoid", (24, 8), (24, 8), False, 17, nonlinearities=("silu", "sigmoid"), bottleneck=2)
    run_gcp2("gcp2_selfgate", (24, 8), (20, 6), False, 18, nonlinearities=("silu", "sigmoid"), vector_gate=False)
    run_gcp2("gcp2_vres_e3", (24, 8), (24, 8), True, 19, nonlinearities=("leakyrelu", None), bottleneck=4,
             vector_residual=True, enable_e3_equivariance=True)
    run_gcp2("gcp2_frame_gate", (24, 8), (16, 4), True, 20, nonlinearities=("silu", "silu"), bottleneck=2,
             frame_gate=True)
    run_gcp2("gcp2_frame_gate_edge", (24, 8), (16, 4), False, 21, nonlinearities=("relu", "sigmoid"), frame_gate=True)
    run_gcp2("gcp2_ablate_frames", (24, 8), (16, 4), False, 22, nonlinearities=("relu", None), bottleneck=4,
             ablate_frame_updates=True)
    # GCP3 (gcpnet.py:471-700) = GCP2 with silu defaults and an optional two-layer scalar_out (feedforward_out)
    run_gcp2("gcp3_edge_default", (40, 8), (24, 8), False, 23, cls="GCP3", bottleneck=4)
    run_gcp2("gcp3_node_default", (24, 8), (40, 12), True, 24, cls="GCP3", bottleneck=2)
    run_gcp2("gcp3_feedforward", (40, 8), (24, 8), False, 25, cls="GCP3", bottleneck=4, feedforward_out=True)
    run_gcp2("gcp3_feedforward_node", (64, 8), (16, 4), True, 26, cls="GCP3", bottleneck=2, feedforward_out=True,
             nonlinearities=(None, None))  # the last feed-forward GCP of GCPInteractions (gcpnet.py:1335-1343)
    run_gcp2("gcp3_feedforward_scalar", (12, 0), (8, 0), True, 27, cls="GCP3", feedforward_out=True,
             scalar_out_nonlinearity="relu")


def fx_gcp_original():
    """The original `GCP` block (gcpnet.py:30-249): GVP-like stage + frame stage."""
    run_gcp2("gcp_edge_default", (40, 8), (24, 8), False, 60, cls="GCP", bottleneck=4)
    run_gcp2("gcp_node_default", (24, 8), (40, 12), True, 61, cls="GCP", bottleneck=2, nonlinearities=("silu", "sigmoid"))
    run_gcp2("gcp_sigma_gate", (24, 8), (16, 8), False, 62, cls="GCP", nonlinearities=("relu", "sigmoid"), sigma_frame_gate=True,
             vector_residual=True)
    run_gcp2("gcp_frame_gate", (24, 8), (16, 4), True, 63, cls="GCP", nonlinearities=("silu", "silu"), bottleneck=2,
             frame_gate=True, vector_frame_residual=True)
    run_gcp2("gcp_selfgate_e3", (24, 8), (20, 6), False, 64, cls="GCP", nonlinearities=("silu", "sigmoid"), vector_gate=False,
             enable_e3_equivariance=True)
    run_gcp2("gcp_scalar_out", (32, 8), (16, 0), True, 65, cls="GCP", nonlinearities=("relu", None))
    run_gcp2("gcp_node_e3", (24, 8), (24, 8), True, 67, cls="GCP", nonlinearities=("silu", "sigmoid"), bottleneck=2,
             sigma_frame_gate=True, enable_e3_equivariance=True)
    run_gcp2("gcp_ablate_frames", (24, 8), (16, 4), False, 66, cls="GCP", nonlinearities=("relu", None), bottleneck=4,
             ablate_frame_updates=True)


def fx_layernorm():
    torch.manual_seed(30)
    ln = comp.GCPLayerNorm(SV(64, 16))
    with torch.no_grad():
        ln.scalar_norm.weight.uniform_(0.5, 1.5)
        ln.scalar_norm.bias.uniform_(-0.5, 0.5)
    s = randn(40, 64, seed=31).requires_grad_()
    v = randn(40, 16, 3, seed=32).requires_grad_()
    so, vo = ln(SV(s, v))
    sq_loss(so * randn(40, 64, seed=33), vo * randn(40, 16, 3, seed=34)).backward()
    save("layernorm", params=ln.state_dict(), inputs=dict(s=s, v=v), outputs=dict(s=so, v=vo),
         grads={"s": s.grad, "v": v.grad, "w.scalar_norm.weight": ln.scalar_norm.weight.grad,
                "w.scalar_norm.bias": ln.scalar_norm.bias.grad})


from gen_helpers import nms_like_batch  # noqa: E402  (shared with tests/test_full_size.py)


def fx_embedding():
    cfg = ref_stubs.make_cfg()
    for name, node_in, edge_in, hid_n, hid_e, int_h in (
        ("embedding_nms", (1, 3), (17, 1), (64, 16), (32, 4), False),
        ("embedding_lba", (9, 2), (16, 1), (100, 16), (32, 4), True),
    ):
        torch.manual_seed(40)
        emb = gn.GCPEmbedding(SV(*edge_in), SV(*node_in), SV(*hid_e), SV(*hid_n),
                              num_atom_types=9 if int_h else 0, cfg=cfg)
        b = nms_like_batch(3, 5, 41, h_dim=node_in[0], chi_dim=node_in[1], e_dim=edge_in[0], xi_dim=edge_in[1],
                           int_h=int_h)
        f = comp.localize(b["x"], b["edge_index"])
        bag = ref_stubs.Bag(**b, f_ij=f)
        (h, chi), (e, xi) = emb(bag)
        save(name, params=emb.state_dict(), inputs=dict(**b, frames=f), outputs=dict(h=h, chi=chi, e=e, xi=xi))


def fx_interactions():
    cfg, lc = ref_stubs.make_cfg(), ref_stubs.make_layer_cfg()
    nd, ed = SV(64, 16), SV(32, 4)
    ei, x = rand_graph(24, 96, 50)
    frames = comp.localize(x, ei)
    # message passing alone
    torch.manual_seed(51)
    mp = gn.GCPMessagePassing(nd, nd, ed, cfg=cfg, mp_cfg=lc.mp_cfg)
    h, chi = randn(24, 64, seed=52).requires_grad_(), randn(24, 16, 3, seed=53).requires_grad_()
    e, xi = randn(96, 32, seed=54).requires_grad_(), randn(96, 4, 3, seed=55).requires_grad_()
    msg = mp.message(SV(h, chi), SV(e, xi), ei, frames)
    out = mp(SV(h, chi), SV(e, xi), ei, frames)
    sq_loss(out[0], out[1]).backward()
    grads = dict(h=h.grad, chi=chi.grad, e=e.grad, xi=xi.grad)
    grads.update({"w." + k: p.grad for k, p in mp.named_parameters()})
    save("message_passing", params=mp.state_dict(), inputs=dict(h=h, chi=chi, e=e, xi=xi, edge_index=ei, frames=frames),
         outputs=dict(s=out[0], v=out[1], messages=msg), grads=grads)

    # full interaction layer, with and without position update
    for name, upd in (("interactions", False), ("interactions_posupd", True)):
        torch.manual_seed(56)
        layer = gn.GCPInteractions(nd, ed, cfg=cfg, layer_cfg=lc, dropout=0.0, updating_node_positions=upd)
        layer.eval()
        for t in (h, chi, e, xi):
            t.grad = None
        if upd:
            (ho, co), xo = layer((h, chi), (e, xi), ei, frames, node_pos=x)
            outs = dict(h=ho, chi=co, x=xo)
        else:
            ho, co = layer((h, chi), (e, xi), ei, frames)
            outs = dict(h=ho, chi=co)
        sq_loss(*outs.values()).backward()
        grads = dict(h=h.grad, chi=chi.grad, e=e.grad, xi=xi.grad)
        grads.update({"w." + k: p.grad for k, p in layer.named_parameters() if p.grad is not None})
        save(name, params=layer.state_dict(),
             inputs=dict(h=h, chi=chi, e=e, xi=xi, edge_index=ei, frames=frames, x=x), outputs=outs, grads=grads)

    # position update WITH the inter-node force term (ablate_x_force_update: false, gcpnet.py:1143-1153); the reference
    # initialises phi_force_ij with gain 0.001 -- re-drawn larger here so that the term is visible at fp32 tolerances
    cfgf = ref_stubs.make_cfg(ablate_x_force_update=False)
    torch.manual_seed(62)
    layer = gn.GCPInteractions(nd, ed, cfg=cfgf, layer_cfg=lc, dropout=0.0, updating_node_positions=True)
    layer.eval()
    with torch.no_grad():
        layer.phi_force_ij[1].weight.copy_(randn(3, nd[0], seed=63) * 0.2)
    for t in (h, chi, e, xi):
        t.grad = None
    (ho, co), xo = layer((h, chi), (e, xi), ei, frames, node_pos=x)
    outs = dict(h=ho, chi=co, x=xo)
    sq_loss(*outs.values()).backward()
    grads = dict(h=h.grad, chi=chi.grad, e=e.grad, xi=xi.grad)
    grads.update({"w." + k: p.grad for k, p in layer.named_parameters() if p.grad is not None})
    save("interactions_force", params=layer.state_dict(),
         inputs=dict(h=h, chi=chi, e=e, xi=xi, edge_index=ei, frames=frames, x=x), outputs=outs, grads=grads)

    # pre-norm ordering, 3 message layers / 3 FF layers, small dims
    lc2 = ref_stubs.make_layer_cfg(pre_norm=True, num_feedforward_layers=3, num_message_layers=3)
    cfg2 = ref_stubs.make_cfg(scalar_nonlinearity="silu", vector_nonlinearity="silu")
    torch.manual_seed(57)
    layer = gn.GCPInteractions(SV(16, 4), SV(8, 4), cfg=cfg2, layer_cfg=lc2, dropout=0.0)
    layer.eval()
    h2, chi2 = randn(24, 16, seed=58), randn(24, 4, 3, seed=59)
    e2, xi2 = randn(96, 8, seed=60), randn(96, 4, 3, seed=61)
    ho, co = layer((h2, chi2), (e2, xi2), ei, frames)
    save("interactions_prenorm_silu", params=layer.state_dict(),
         inputs=dict(h=h2, chi=chi2, e=e2, xi=xi2, edge_index=ei, frames=frames), outputs=dict(h=ho, chi=co))


def fx_ablations():
    """ablate_scalars / ablate_vectors through GCPMessagePassing (the flags zero a block's inputs and outputs, reference
    gcpnet.py:416-417,466-467); same graph, seeds and inputs as `message_passing`."""
    lc = ref_stubs.make_layer_cfg()
    nd, ed = SV(64, 16), SV(32, 4)
    ei, x = rand_graph(24, 96, 50)
    frames = comp.localize(x, ei)
    for name, over in (("message_passing_ablate_scalars", dict(ablate_scalars=True)),
                       ("message_passing_ablate_vectors", dict(ablate_vectors=True))):
        cfg = ref_stubs.make_cfg(**over)
        torch.manual_seed(51)
        mp = gn.GCPMessagePassing(nd, nd, ed, cfg=cfg, mp_cfg=lc.mp_cfg)
        h, chi = randn(24, 64, seed=52).requires_grad_(), randn(24, 16, 3, seed=53).requires_grad_()
        e, xi = randn(96, 32, seed=54).requires_grad_(), randn(96, 4, 3, seed=55).requires_grad_()
        msg = mp.message(SV(h, chi), SV(e, xi), ei, frames)
        out = mp(SV(h, chi), SV(e, xi), ei, frames)
        lw_s, lw_v = randn(*out[0].shape, seed=56), randn(*out[1].shape, seed=57)
        ((out[0] * lw_s).sum() + (out[1] * lw_v).sum() + sq_loss(out[0], out[1])).backward()
        zero = lambda t, like: t if t is not None else torch.zeros_like(like)
        grads = dict(h=zero(h.grad, h), chi=zero(chi.grad, chi), e=zero(e.grad, e), xi=zero(xi.grad, xi))
        grads.update({"w." + k: zero(p.grad, p) for k, p in mp.named_parameters()})
        save(name, params=mp.state_dict(), inputs=dict(h=h, chi=chi, e=e, xi=xi, edge_index=ei, frames=frames, lw_s=lw_s, lw_v=lw_v),
             outputs=dict(s=out[0], v=out[1], messages=msg), grads=grads)


def fx_masked():
    """Masked / autoregressive call paths: node_mask in centralize / localize / scalarize / vectorize
    (components/__init__.py:177-193,229-264,294-300,346-357), GCP2 with a mask, GCPInteractions with a mask (sub-graph
    feed-forward, gcpnet.py:1201-1251) and its autoregressive forward (:1066-1116), GCPMLPDecoder (:1454-1491)."""
    ei, x = rand_graph(24, 96, 70)
    g = torch.Generator().manual_seed(71)
    mask = torch.rand(24, generator=g) > 0.3
    mask[:3] = torch.tensor([True, False, True])
    bidx = torch.tensor([0] * 9 + [1] * 8 + [2] * 7)
    f = comp.localize(x, ei, norm_x_diff=True, node_mask=mask)
    vec_e, vec_n = randn(96, 3, 3, seed=72), randn(24, 3, 3, seed=73)
    gate_e, gate_n = randn(96, 9, seed=74), randn(24, 9, seed=75)
    cen, xc = comp.centralize(ref_stubs.Bag(x=x), "x", bidx, node_mask=mask)
    save("geometry_masked",
         inputs=dict(x=x, edge_index=ei, mask=mask, batch=bidx, vec_e=vec_e, vec_n=vec_n, gate_e=gate_e, gate_n=gate_n),
         outputs=dict(frames=f, centroid=cen, x_centered=xc,
                      scalarize_edge=comp.scalarize(vec_e, ei, f, False, False, 96, node_mask=mask),
                      scalarize_node=comp.scalarize(vec_n, ei, f, True, True, 24, node_mask=mask),
                      vectorize_edge=comp.vectorize(gate_e, ei, f, False, 96, node_mask=mask),
                      vectorize_node=comp.vectorize(gate_n, ei, f, True, 24, node_mask=mask)))

    # one GCP2 on node rows with a mask (frames of the masked edges are +inf, as localize produces them)
    torch.manual_seed(76)
    mod = gn.GCP2(SV(24, 8), SV(16, 4), nonlinearities=("silu", "sigmoid"), bottleneck=2)
    s, v = randn(24, 24, seed=77).requires_grad_(), randn(24, 8, 3, seed=78).requires_grad_()
    out = mod((s, v), ei, f, node_inputs=True, node_mask=mask)
    sq_loss(out[0], out[1]).backward()
    grads = dict(s=s.grad, v=v.grad)
    grads.update({"w." + k: p.grad for k, p in mod.named_parameters() if p.grad is not None})
    save("gcp2_masked_node", params=mod.state_dict(), inputs=dict(s=s, v=v, edge_index=ei, frames=f, mask=mask),
         outputs=dict(s=out[0], v=out[1]), grads=grads, meta=dict(node_inputs=1, in_dims=(24, 8), out_dims=(16, 4)))

    # GCPInteractions with a mask: with and without position update; all-true mask (no sub-graph)
    cfg, lc = ref_stubs.make_cfg(), ref_stubs.make_layer_cfg()
    nd, ed = SV(32, 8), SV(16, 4)
    h, chi = randn(24, 32, seed=79).requires_grad_(), randn(24, 8, 3, seed=80).requires_grad_()
    e, xi = randn(96, 16, seed=81).requires_grad_(), randn(96, 4, 3, seed=82).requires_grad_()
    all_true = torch.ones(24, dtype=torch.bool)
    for name, upd, m, pre_norm in (("interactions_masked", False, mask, False), ("interactions_masked_posupd", True, mask, False),
                                   ("interactions_masked_all", False, all_true, True)):
        lcm = ref_stubs.make_layer_cfg(pre_norm=pre_norm)
        fm = comp.localize(x, ei, node_mask=m)
        torch.manual_seed(83)
        layer = gn.GCPInteractions(nd, ed, cfg=cfg, layer_cfg=lcm, dropout=0.0, updating_node_positions=upd)
        layer.eval()
        for t in (h, chi, e, xi):
            t.grad = None
        # (the reference writes the new rows INTO its node input, :1250: a clone is fed so that the leaves survive)
        if upd:
            (ho, co), xo = layer((h.clone(), chi.clone()), (e, xi), ei, fm, node_mask=m, node_pos=x)
            outs = dict(h=ho, chi=co, x=xo)
            fin = torch.isfinite(xo).all(dim=1)  # (the un-masked force-free update only touches rows through finite frames)
            loss = sq_loss(ho, co, xo[fin])
        else:
            ho, co = layer((h.clone(), chi.clone()), (e, xi), ei, fm, node_mask=m)
            outs = dict(h=ho, chi=co)
            loss = sq_loss(ho, co)
        loss.backward()
        grads = dict(h=h.grad, chi=chi.grad, e=e.grad, xi=xi.grad)
        grads.update({"w." + k: p.grad for k, p in layer.named_parameters() if p.grad is not None})
        save(name, params=layer.state_dict(), inputs=dict(h=h, chi=chi, e=e, xi=xi, edge_index=ei, frames=fm, x=x, mask=m),
             outputs=outs, grads=grads, meta=dict(pre_norm=int(pre_norm)))

    # autoregressive forward (layers built with autoregressive=True sum their messages, :996)
    fr = comp.localize(x, ei)
    torch.manual_seed(84)
    layer = gn.GCPInteractions(nd, ed, cfg=cfg, layer_cfg=lc, dropout=0.0, autoregressive=True)
    layer.eval()
    hr, chir = randn(24, 32, seed=85).requires_grad_(), randn(24, 8, 3, seed=86).requires_grad_()
    for t in (h, chi, e, xi):
        t.grad = None
    ho, co = layer((h, chi), (e, xi), ei, fr, node_rep_regressive=(hr, chir))
    sq_loss(ho, co).backward()
    grads = dict(h=h.grad, chi=chi.grad, e=e.grad, xi=xi.grad, h_reg=hr.grad, chi_reg=chir.grad)
    grads.update({"w." + k: p.grad for k, p in layer.named_parameters() if p.grad is not None})
    save("interactions_autoregressive", params=layer.state_dict(),
         inputs=dict(h=h, chi=chi, e=e, xi=xi, edge_index=ei, frames=fr, h_reg=hr, chi_reg=chir), outputs=dict(h=ho, chi=co),
         grads=grads)

    # GCPMLPDecoder, both variants
    for name, res in (("mlp_decoder", False), ("mlp_decoder_residual", True)):
        torch.manual_seed(87)
        dec = gn.GCPMLPDecoder(32, vocab_size=20, num_layers=3, residual_updates=res)
        hh = randn(24, 32, seed=88).requires_grad_()
        logits, logp = dec(hh)
        (sq_loss(logits) + (logp * randn(24, 20, seed=89)).mean()).backward()
        grads = dict(h=hh.grad)
        grads.update({"w." + k: p.grad for k, p in dec.named_parameters()})
        save(name, params=dec.state_dict(), inputs=dict(h=hh, lw=randn(24, 20, seed=89)), outputs=dict(logits=logits, log_probs=logp),
             grads=grads)


def fx_models():
    """The LitModules cannot be imported (no Lightning here); their forwards (gcpnet_nms_module.py:127-151,
    gcpnet_lba_module.py:155-186) are driven step by step with the real reference components."""
    cfg, lc = ref_stubs.make_cfg(), ref_stubs.make_layer_cfg(num_message_layers=4)
    # --- NMS-style: (1,3)/(17,1) -> (32,8)/(16,4), 2 layers, 3 x 5-body
    torch.manual_seed(70)
    nd, ed = SV(32, 8), SV(16, 4)
    emb = gn.GCPEmbedding(SV(17, 1), SV(1, 3), ed, nd, num_atom_types=0, cfg=cfg)
    layers = torch.nn.ModuleList(
        gn.GCPInteractions(nd, ed, cfg=cfg, layer_cfg=lc, dropout=0.0, updating_node_positions=True)
        for _ in range(2))
    b = nms_like_batch(3, 5, 71)
    b["label"] = b["x"] + 0.3 * randn(15, 3, seed=77)  # step(): MSELoss(x_pred, label), gcpnet_nms_module.py:153-158
    for k in ("h", "chi", "e", "xi"):
        b[k] = b[k].requires_grad_()
    bag = ref_stubs.Bag(**b)
    cen, bag.x = comp.centralize(bag, "x", bag.batch)
    bag.f_ij = comp.localize(bag.x, bag.edge_index, norm_x_diff=True)
    (h, chi), (e, xi) = emb(bag)
    for layer in layers:
        (h, chi), bag.x = layer((h, chi), (e, xi), bag.edge_index, bag.f_ij, node_pos=bag.x)
    x_out = comp.decentralize(bag, "x", bag.batch, cen)
    params = {"gcp_embedding." + k: v for k, v in emb.state_dict().items()}
    params.update({"interaction_layers." + k: v for k, v in layers.state_dict().items()})
    loss = torch.nn.functional.mse_loss(x_out, b["label"])
    loss.backward()
    grads = {k: b[k].grad for k in ("h", "chi", "e", "xi")}
    grads.update({"w.gcp_embedding." + k: p.grad for k, p in emb.named_parameters() if p.grad is not None})
    grads.update({"w.interaction_layers." + k: p.grad for k, p in layers.named_parameters() if p.grad is not None})
    save("model_nms_small", params=params, inputs=b, outputs=dict(h=h, chi=chi, e=e, xi=xi, x=x_out, f_ij=bag.f_ij, loss=loss),
         grads=grads, meta=dict(num_layers=2, num_message_layers=4))

    # --- LBA-style: int atom types -> (20,4)/(8,4), 2 layers, readout
    torch.manual_seed(72)
    nd, ed = SV(20, 4), SV(8, 4)
    emb = gn.GCPEmbedding(SV(16, 1), SV(9, 2), ed, nd, num_atom_types=9, cfg=cfg)
    layers = torch.nn.ModuleList(gn.GCPInteractions(nd, ed, cfg=cfg, layer_cfg=lc, dropout=0.0) for _ in range(2))
    proj_norm = comp.GCPLayerNorm(nd)
    proj = gn.GCP2(nd, (nd.scalar, 0), nonlinearities=tuple(cfg.nonlinearities), scalar_gate=cfg.scalar_gate,
                   vector_gate=cfg.vector_gate, frame_gate=cfg.frame_gate, sigma_frame_gate=cfg.sigma_frame_gate,
                   vector_frame_residual=cfg.vector_frame_residual, ablate_frame_updates=cfg.ablate_frame_updates,
                   enable_e3_equivariance=cfg.enable_e3_equivariance, node_inputs=True)
    dense = torch.nn.Sequential(torch.nn.Linear(20, 40), torch.nn.ReLU(), torch.nn.Dropout(0.1),
                                torch.nn.Linear(40, 1)).eval()
    b = nms_like_batch(4, 6, 73, chi_dim=2, e_dim=16, xi_dim=1, int_h=True)
    b["label"] = randn(4, seed=78)  # step(): MSELoss(pred, label), gcpnet_lba_module.py:188-193
    for k in ("chi", "e", "xi"):
        b[k] = b[k].requires_grad_()
    bag = ref_stubs.Bag(**b)
    _, bag.x = comp.centralize(bag, "x", bag.batch)
    bag.f_ij = comp.localize(bag.x, bag.edge_index, norm_x_diff=True)
    (h, chi), (e, xi) = emb(bag)
    for layer in layers:
        (h, chi) = layer((h, chi), (e, xi), bag.edge_index, bag.f_ij)
    out = proj_norm(SV(h, chi))
    out = proj(out, bag.edge_index, bag.f_ij, node_inputs=True)
    from torch_scatter import scatter
    pred = dense(scatter(out, bag.batch, dim=0, reduce="mean")).squeeze()
    params = {"gcp_embedding." + k: v for k, v in emb.state_dict().items()}
    params.update({"interaction_layers." + k: v for k, v in layers.state_dict().items()})
    params.update({"invariant_node_projection.0." + k: v for k, v in proj_norm.state_dict().items()})
    params.update({"invariant_node_projection.1." + k: v for k, v in proj.state_dict().items()})
    params.update({"dense." + k: v for k, v in dense.state_dict().items()})
    loss = torch.nn.functional.mse_loss(pred, b["label"])
    loss.backward()
    grads = {k: b[k].grad for k in ("chi", "e", "xi")}
    for pre, mod in (("gcp_embedding.", emb), ("interaction_layers.", layers), ("invariant_node_projection.0.", proj_norm),
                     ("invariant_node_projection.1.", proj), ("dense.", dense)):
        grads.update({"w." + pre + k: p.grad for k, p in mod.named_parameters() if p.grad is not None})
    save("model_lba_small", params=params, inputs=b, outputs=dict(h=h, chi=chi, pred=pred, loss=loss), grads=grads,
         meta=dict(num_layers=2, num_message_layers=4))


def fx_interactions2():
    """GCPInteractions2 (gcpnet.py:1265-1451) as configs/model/gcpnet_eq.yaml builds it: GCP3 blocks, sum aggregation over
    `row`, learnable scalar message gate, one feed-forward GCP with a two-layer scalar_out; and a two-FF-layer variant with the
    position update."""
    import functools
    cfg = ref_stubs.make_cfg(selected_GCP=functools.partial(gn.GCP3))
    nd, ed = SV(64, 16), SV(32, 4)
    ei, x = rand_graph(24, 96, 70)
    frames = comp.localize(x, ei)
    h, chi = randn(24, 64, seed=72).requires_grad_(), randn(24, 16, 3, seed=73).requires_grad_()
    e, xi = randn(96, 32, seed=74).requires_grad_(), randn(96, 4, 3, seed=75).requires_grad_()
    cases = (("interactions2_eq", False, ref_stubs.make_layer_cfg(use_scalar_message_attention=True, aggregate_with_row=True,
                                                                  num_feedforward_layers=1)),
             ("interactions2_posupd", True, ref_stubs.make_layer_cfg(use_scalar_message_attention=True, num_message_layers=4,
                                                                     num_feedforward_layers=2)))
    for name, upd, lc in cases:
        torch.manual_seed(76)
        layer = gn.GCPInteractions2(nd, ed, cfg=cfg, layer_cfg=lc, dropout=0.0, updating_node_positions=upd)
        layer.eval()
        for t in (h, chi, e, xi):
            t.grad = None
        if upd:
            (ho, co), xo = layer((h, chi), (e, xi), ei, frames, node_pos=x)
            outs = dict(h=ho, chi=co, x=xo)
        else:
            ho, co = layer((h, chi), (e, xi), ei, frames)
            outs = dict(h=ho, chi=co)
        sq_loss(*outs.values()).backward()
        grads = dict(h=h.grad, chi=chi.grad, e=e.grad, xi=xi.grad)
        grads.update({"w." + k: p.grad for k, p in layer.named_parameters() if p.grad is not None})
        save(name, params=layer.state_dict(),
             inputs=dict(h=h, chi=chi, e=e, xi=xi, edge_index=ei, frames=frames, x=x), outputs=outs, grads=grads)


def fx_input_side():
    """Input side.  (1) NMS featuriser: the reference's real helper functions (`_rbf`, `_normalize`,
    `_orientations`, src/datamodules/components/helper.py) driven exactly as nms_dataset.py:23-61,180-206 drives them (that file
    itself needs PyG / atom3d to import), per graph, then collated.  (2) radius graph: the scipy cKDTree path of
    gcpnet_amd.synthetic (K nearest within r, per graph) -- the GPU builder must reproduce its edge list bit for bit."""
    from src.datamodules.components import helper as H
    from scipy.spatial import cKDTree

    g = torch.Generator().manual_seed(90)
    n_graphs, n_body = 3, 5
    xs, vels, attrs, eis, outs = [], [], [], [], dict(h=[], chi=[], e=[], xi=[])
    for k in range(n_graphs):
        loc, vel = torch.randn(n_body, 3, generator=g) * 2, torch.randn(n_body, 3, generator=g)
        if k == 1:
            loc[3] = loc[2]  # coincident bodies: 0 / 0 in _normalize -> nan_to_num -> 0
        charges = torch.randint(0, 2, (n_body,), generator=g).float() * 2 - 1
        idx = torch.arange(n_body)
        r, c = torch.meshgrid(idx, idx, indexing="ij")
        keep = r != c
        ei = torch.stack((r[keep], c[keep]))
        edge_attr = (charges[ei[0]] * charges[ei[1]]).unsqueeze(-1)
        # nms_dataset.py:33-45
        E_vectors = loc[ei[0]] - loc[ei[1]]
        rbf = H._rbf(E_vectors.norm(dim=-1), D_max=4.5, D_count=16)
        edge_s = torch.cat((edge_attr, rbf), dim=-1)
        edge_v = H._normalize(E_vectors).unsqueeze(-2)
        edge_s, edge_v = map(torch.nan_to_num, (edge_s, edge_v))
        # nms_dataset.py:56-59
        node_s = torch.sqrt(torch.sum(vel ** 2, dim=-1)).unsqueeze(-1)
        node_v = torch.cat((vel.unsqueeze(1), H._orientations(loc)), dim=1)
        xs.append(loc); vels.append(vel); attrs.append(edge_attr); eis.append(ei + k * n_body)
        for key, val in (("h", node_s), ("chi", node_v), ("e", edge_s), ("xi", edge_v)):
            outs[key].append(val)
    save("nms_features", inputs=dict(x=torch.cat(xs), vel=torch.cat(vels), edge_attr=torch.cat(attrs), edge_index=torch.cat(eis, 1),
                                     batch=torch.arange(n_graphs).repeat_interleave(n_body)),
         outputs={k: torch.cat(v) for k, v in outs.items()})

    rng = np.random.default_rng(91)
    pts, eis, bidx, off = [], [], [], 0
    for k, (n, side) in enumerate(((700, 22.0), (500, 30.0), (40, 3.0))):  # dense, sparse (some nodes below K), tiny (n - 1 < K)
        x = rng.uniform(0.0, side, size=(n, 3)).astype(np.float32)
        K = 16
        dist, nbr = cKDTree(x).query(x, k=min(K + 1, n), distance_upper_bound=4.5)
        dist, nbr = dist[:, 1:], nbr[:, 1:]
        ok = np.isfinite(dist)
        col = np.repeat(np.arange(n), dist.shape[1]).reshape(n, dist.shape[1])[ok]
        eis.append(np.stack((nbr[ok], col)).astype(np.int64) + off)
        pts.append(x); bidx.append(np.full(n, k)); off += n
    save("radius_graph", inputs=dict(x=torch.from_numpy(np.concatenate(pts)), batch=torch.from_numpy(np.concatenate(bidx))),
         outputs=dict(edge_index=torch.from_numpy(np.concatenate(eis, 1))), meta=dict(radius=4.5, max_neighbors=16))


def fx_cpd():
    """CPD task module: the reference's REAL `GCPNetCPDLitModule` (src/models/gcpnet_cpd_module.py,
    imported through ref_stubs.install_lightning_stubs) with the autoregressive decoder -- forward with teacher forcing (:153-218),
    the training-step loss and every gradient (:220-231), and `autoregressively_generate_samples` (:281-360) with
    `Categorical(...).sample()` replaced by argmax (the only way two implementations can agree on a sampled sequence); plus the
    direct-shot (MLP decoder) variant's forward."""
    import functools

    ref_stubs.install_lightning_stubs()
    import src.models.gcpnet_cpd_module as cpd

    class ArgmaxCategorical:
        def __init__(self, logits=None):
            self.logits = logits

        def sample(self):
            top2 = torch.topk(self.logits, 2, dim=-1).values
            ArgmaxCategorical.margin = min(ArgmaxCategorical.margin, float((top2[:, 0] - top2[:, 1]).min()))
            return self.logits.argmax(-1)

    ArgmaxCategorical.margin = float("inf")
    cpd.Categorical = ArgmaxCategorical
    n, k_nn = 12, 5
    g = torch.Generator().manual_seed(120)
    x = torch.randn(n, 3, generator=g) * 3
    d = torch.cdist(x, x)
    d.fill_diagonal_(float("inf"))
    nbr = d.topk(k_nn, largest=False).indices  # kNN graph (CATH recipe), edges source -> target
    ei = torch.stack((nbr.reshape(-1), torch.arange(n).repeat_interleave(k_nn)))
    e_cnt = ei.shape[1]
    mask = torch.ones(n, dtype=torch.bool)
    model_cfg = ref_stubs.DictConfig(h_hidden_dim=32, chi_hidden_dim=8, e_hidden_dim=16, xi_hidden_dim=4, output_dim=20,
                                     num_encoder_layers=2, num_decoder_layers=2, dropout=0.0, decoder_residual_updates=True)
    lc = ref_stubs.make_layer_cfg(num_message_layers=4)
    inputs = dict(h=randn(n, 6, seed=121), chi=randn(n, 3, 3, seed=122), e=randn(e_cnt, 32, seed=123), xi=randn(e_cnt, 1, 3, seed=124),
                  x=x, edge_index=ei, batch=torch.zeros(n, dtype=torch.long), mask=mask,
                  seq=torch.randint(0, 20, (n,), generator=g))
    for name, ar in (("model_cpd_small", True), ("model_cpd_direct", False)):
        torch.manual_seed(125)
        model = cpd.GCPNetCPDLitModule(layer_class=functools.partial(gn.GCPInteractions), optimizer=None, scheduler=None,
                                       node_input_dims=[6, 3], edge_input_dims=[32, 1], model_cfg=model_cfg, module_cfg=ref_stubs.make_cfg(),
                                       layer_cfg=lc, dropout=0.0, autoregressive_decoder=ar)
        model.eval()
        ins = {k: (v.clone().requires_grad_() if v.is_floating_point() and k != "x" else v.clone()) for k, v in inputs.items()}
        bag = ref_stubs.Bag(**ins)
        _, out = model.forward(bag)
        preds = out if ar else out[0]
        loss = model.criterion(preds[bag.mask], bag.seq[bag.mask])
        loss.backward()
        grads = {k: ins[k].grad for k in ("h", "chi", "e", "xi")}
        grads.update({"w." + k: p.grad for k, p in model.named_parameters() if p.grad is not None})
        outs = dict(preds=preds, loss=loss, h=bag.h, chi=bag.chi, f_ij=bag.f_ij)
        if ar:
            # the sampling loop takes RAW features and frames; positions centred as forward() does it
            with torch.no_grad():
                _, xc = comp.centralize(ref_stubs.Bag(x=x, batch=inputs["batch"]), "x", inputs["batch"], node_mask=mask)
                frames = comp.localize(xc, ei, norm_x_diff=True, node_mask=mask)
                seqs = model.autoregressively_generate_samples(SV(inputs["h"], inputs["chi"]), SV(inputs["e"], inputs["xi"]), ei, frames,
                                                               mask, num_samples=2, temperature=0.1)
            assert ArgmaxCategorical.margin > 1e-2, ArgmaxCategorical.margin  # (no near-ties: argmax is implementation-independent)
            outs.update(samples=seqs, sample_frames=frames)
            print("argmax margin of the sampled logits / temperature:", ArgmaxCategorical.margin)
        save(name, params=model.state_dict(), inputs=inputs, outputs=outs, grads=grads,
             meta=dict(num_encoder_layers=2, num_decoder_layers=2, num_message_layers=4, autoregressive=int(ar)))


def fx_lba_features():
    """ATOM3D / LBA input side: the reference's REAL `LBATransform` (atom3d_dataset.py:134-149 ->
    BaseTransform.__call__ :101-129 -> `_edge_features` :42-62, `_node_features` :65-84, helper.py) on two synthetic pocket +
    ligand structures (pandas frames with x, y, z, element), with `torch_cluster.radius_graph` = the brute-force stub (graphs kept
    below the neighbour cap, so the edge SET is independent of torch_cluster's selection order).  Then the collated batch, written
    out with PyG's `Batch.from_data_list` rules (cat along dim 0; edge_index along dim 1 with node offsets; scalars -> [G])."""
    import pandas as pd
    from src.datamodules.components import atom3d_dataset as A

    rng = np.random.default_rng(95)
    elements = ["C", "N", "O", "S", "H", "F", "Cl", "CL", "P", "Zn", "Fe"]
    tf = A.LBATransform()
    graphs = []
    for k, (n_pocket, n_lig, side) in enumerate(((70, 14, 16.0), (48, 9, 13.0))):
        def frame(n, lo):
            xyz = rng.uniform(lo, lo + side, size=(n, 3)).astype(np.float32)
            if k == 0 and n > 20:
                xyz[11] = xyz[10]  # coincident atoms: 0 / 0 in _normalize -> nan_to_num -> 0
            return pd.DataFrame(dict(x=xyz[:, 0], y=xyz[:, 1], z=xyz[:, 2], element=rng.choice(elements, size=n)))
        elem = dict(atoms_pocket=frame(n_pocket, 0.0), atoms_ligand=frame(n_lig, 2.0), scores=dict(neglog_aff=4.5 + k))
        d = tf(elem)
        graphs.append(d)
    ins, outs = {}, {}
    for k, (d, n_lig) in enumerate(zip(graphs, (14, 9))):
        ins[f"x{k}"] = d.x
        ins[f"n_ligand{k}"] = torch.tensor(n_lig)
        for key in ("h", "chi", "e", "xi", "edge_index", "lig_flag"):
            outs[f"{key}{k}"] = getattr(d, key)
        outs[f"label{k}"] = torch.tensor(d.label)
    offs = [0, graphs[0].x.shape[0]]
    for key in ("h", "chi", "e", "xi", "x", "lig_flag"):
        outs["batch_" + key] = torch.cat([getattr(d, key) for d in graphs], dim=0)
    outs["batch_edge_index"] = torch.cat([d.edge_index + o for d, o in zip(graphs, offs)], dim=1)
    outs["batch_label"] = torch.tensor([d.label for d in graphs])
    outs["batch_batch"] = torch.cat([torch.full((d.x.shape[0],), i) for i, d in enumerate(graphs)])
    save("lba_features", inputs=ins, outputs=outs, meta=dict(edge_cutoff=4.5, num_rbf=16, max_num_neighbors=32))


def fx_lba_features_capped():
    """The neighbour cap of the reference's graph recipe (atom3d_dataset.py:110-112, `radius_graph(..., max_num_neighbors=32)`):
    the reference's REAL `LBATransform` on a DENSE pocket + ligand structure -- most atoms have 40 - 70 atoms within 4.5 A -- with
    `torch_cluster.radius_graph` = the restatement of torch_cluster 1.6.0's index-order walk (ref_stubs._radius_graph_first).  Pins
    which neighbours survive the cap (the lowest ids, the self loop removed after the cap) and the features of those edges."""
    import pandas as pd
    from src.datamodules.components import atom3d_dataset as A

    rng = np.random.default_rng(123)
    elements = ["C", "N", "O", "S", "H"]
    ref_stubs.RADIUS_GRAPH_SELECT = "first"
    try:
        tf = A.LBATransform()

        def frame(n, lo, side):
            xyz = rng.uniform(lo, lo + side, size=(n, 3)).astype(np.float32)
            return pd.DataFrame(dict(x=xyz[:, 0], y=xyz[:, 1], z=xyz[:, 2], element=rng.choice(elements, size=n)))
        d = tf(dict(atoms_pocket=frame(110, 0.0, 7.0), atoms_ligand=frame(14, 2.0, 4.0), scores=dict(neglog_aff=6.25)))
    finally:
        ref_stubs.RADIUS_GRAPH_SELECT = "refuse"
    deg = torch.bincount(d.edge_index[1], minlength=d.x.shape[0])
    assert int(deg.max()) == 33 and int((deg >= 32).sum()) > 50, (int(deg.max()), int((deg >= 32).sum()))  # the cap binds, incl. the 33 case
    save("lba_features_capped", inputs=dict(x=d.x, n_ligand=torch.tensor(14)),
         outputs={k: getattr(d, k) for k in ("h", "chi", "e", "xi", "edge_index", "lig_flag")},
         meta=dict(edge_cutoff=4.5, num_rbf=16, max_num_neighbors=32))


if __name__ == "__main__":
    torch.set_num_threads(4)
    if len(sys.argv) > 1:  # only the named groups, e.g. `gen_fixtures.py fx_gcp_original`
        for name in sys.argv[1:]:
            globals()[name]()
        sys.exit(0)
    fx_geometry()
    fx_gcp2()
    fx_gcp_original()
    fx_layernorm()
    fx_embedding()
    fx_interactions()
    fx_ablations()
    fx_interactions2()
    fx_masked()
    fx_models()
    fx_input_side()
    fx_lba_features()
    fx_lba_features_capped()
    fx_cpd()
