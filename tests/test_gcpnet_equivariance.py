"""Re-enabled SE(3)-equivariance acceptance tests for the MI355X path.

The reference's own versions are commented out (tests/test_gcpnet_equivariance.py:123-1467 of the reference) and
only their helpers are live (:1470-1925).  These are this build's restatement of
  * GCP2EquivarianceTest (:726-1036): GCP2 blocks, GCP2 + dropout + layer norm, GCPMessagePassing, GCPInteractions,
  * GCP2NewtonianManyBodySystemEquivarianceTest (:1372-1467): NMS model, rotation + translation EQUIVARIANT positions,
  * GCP2LigandBindingAffinityEquivarianceTest (:1040-1140): LBA model, rotation + translation INVARIANT prediction,
with the reference's fixtures (:59-75: 300 nodes, 10 000 random edges, node dims (100,16), edge dims (32,4), seed 1),
its transformation convention (rotation applied on the left of each 3-vector, frames rotated row-wise, :1794-1800),
its tolerances (module level atol 1e-5 / rtol 1e-4, :1865-1866; model level atol 1e-4 / rtol 1e-4, :1615-1618,
:1747-1751), its permutation-sensitivity checks (:1867-1874) and its NaN check (:1854-1862).
"""
import pytest
import torch

from oracle.gcp_oracle import random_rotation

pytestmark = pytest.mark.gpu

SEED = 1
N_NODES, N_EDGES, BATCH = 300, 10000, 8
NODE_DIM, EDGE_DIM = (100, 16), (32, 4)
NODE_IN, EDGE_IN = (1, 2), (16, 1)


@pytest.fixture(scope="module")
def G():
    import gcpnet_amd

    return gcpnet_amd


@pytest.fixture()
def data():
    g = torch.Generator().manual_seed(SEED)
    d = dict(
        h=torch.randn(N_NODES, NODE_DIM[0], generator=g), chi=torch.randn(N_NODES, NODE_DIM[1], 3, generator=g),
        e=torch.randn(N_EDGES, EDGE_DIM[0], generator=g), xi=torch.randn(N_EDGES, EDGE_DIM[1], 3, generator=g),
        edge_index=torch.randint(0, N_NODES, (2, N_EDGES), generator=g),
        x=torch.randn(N_NODES, 3, generator=g) + torch.randint(1, 100, (1,), generator=g),
    )
    return {k: v.cuda() for k, v in d.items()}


def rot(Q, v):
    """rotation on the left of every 3-vector: (Q @ v^T)^T"""
    return (Q @ v.transpose(-1, -2)).transpose(-1, -2)


def check_rotation_and_permutation(model, d, frames, Q):
    """reference helper test_rotation_and_permutation_equivariance (:1773-1881)."""
    h, chi, e, xi, ei = d["h"], d["chi"], d["e"], d["xi"], d["edge_index"]
    out_h, out_chi = model((h, chi), (e, xi), ei, frames)
    out_h2, out_chi2 = model((h, rot(Q, chi)), (e, rot(Q, xi)), ei, rot(Q, frames))

    # swap two nodes (features and their out-edges), as the reference does
    gen = torch.Generator().manual_seed(SEED)
    a, b = torch.randperm(N_NODES, generator=gen)[:2].tolist()
    hp, cp = h.clone(), chi.clone()
    hp[a], cp[a], hp[b], cp[b] = h[b], chi[b], h[a], chi[a]
    eip, ep, xip, fp = ei.clone(), e.clone(), xi.clone(), frames.clone()
    ia, ib = torch.where(ei[0] == a)[0], torch.where(ei[0] == b)[0]
    k = min(len(ia), len(ib))
    ia, ib = ia[:k], ib[:k]
    eip[1, ia], eip[0, ia] = ei[1, ib], b
    eip[1, ib], eip[0, ib] = ei[1, ia], a
    ep[ia], ep[ib], xip[ia], xip[ib], fp[ia], fp[ib] = e[ib], e[ia], xi[ib], xi[ia], frames[ib], frames[ia]
    out_h3, out_chi3 = model((hp, cp), (ep, xip), eip, fp)

    for t in (out_h, out_chi, out_h2, out_chi2, out_h3, out_chi3):
        assert not t.isnan().any(), "No NaNs may be present"
    assert torch.allclose(out_h, out_h2, atol=1e-5, rtol=1e-4), \
        f"Scalar node features must be SO(3)-invariant (max err {(out_h - out_h2).abs().max():.2e})"
    assert torch.allclose(rot(Q, out_chi), out_chi2, atol=1e-5, rtol=1e-4), \
        f"Vector node features must be SO(3)-equivariant (max err {(rot(Q, out_chi) - out_chi2).abs().max():.2e})"
    for n in (a, b):
        assert not torch.allclose(out_h[n], out_h3[n], atol=1e-1, rtol=1e-4), "scalars must react to a node swap"
        assert not torch.allclose(out_chi[n], out_chi3[n], atol=1e-2, rtol=1e-4), "vectors must react to a node swap"


def _frames(G, d):
    with torch.no_grad():
        return G.localize(d["x"], d["edge_index"])


@pytest.mark.parametrize("kw", [
    dict(nonlinearities=("silu", "silu"), vector_gate=True, bottleneck=4),
    dict(nonlinearities=("relu", None), vector_gate=True, bottleneck=4),
    dict(nonlinearities=("silu", "sigmoid"), vector_gate=False, bottleneck=4),
    dict(nonlinearities=("relu", None), vector_gate=True, ablate_frame_updates=True),
    dict(nonlinearities=("silu", "silu"), frame_gate=True, bottleneck=4),
], ids=["vector_gate_silu", "vector_gate_relu", "self_gate", "baseline_no_frames", "frame_gate"])
@pytest.mark.parametrize("node_inputs", [True, False], ids=["node", "edge"])
def test_gcp2(G, data, kw, node_inputs):
    """GCP2EquivarianceTest.test_gcp2_* (:726-860): a single block on node rows or on edge rows."""
    torch.manual_seed(SEED)
    dims = NODE_DIM if node_inputs else EDGE_DIM
    block = G.GCP2(dims, dims, **kw).cuda()

    def model(nodes, edges, ei, fr):
        return block(nodes if node_inputs else edges, ei, fr, node_inputs=node_inputs)

    d = dict(data)
    if not node_inputs:  # swap-sensitivity is defined on node outputs; edge rows are checked for rotation only
        fr, Q = _frames(G, d), random_rotation(SEED).cuda()
        s0, v0 = model(None, (d["e"], d["xi"]), d["edge_index"], fr)
        s1, v1 = model(None, (d["e"], rot(Q, d["xi"])), d["edge_index"], rot(Q, fr))
        assert torch.allclose(s0, s1, atol=1e-5, rtol=1e-4)
        assert torch.allclose(rot(Q, v0), v1, atol=1e-5, rtol=1e-4)
        return
    with torch.no_grad():
        check_rotation_and_permutation(model, d, _frames(G, d), random_rotation(SEED).cuda())


def test_gcp2_sequence_with_dropout_and_layernorm(G, data):
    """GCP2EquivarianceTest.test_gcp2_sequence (:862-920): GCP2 -> GCPDropout -> GCPLayerNorm, eval mode."""
    torch.manual_seed(SEED)
    block = G.GCP2(NODE_DIM, NODE_DIM, nonlinearities=("silu", "silu"), bottleneck=4).cuda()
    drop, norm = G.GCPDropout(0.1).eval(), G.GCPLayerNorm(NODE_DIM).cuda()

    def model(nodes, edges, ei, fr):
        return norm(drop(block(nodes, ei, fr, node_inputs=True)))

    with torch.no_grad():
        check_rotation_and_permutation(model, data, _frames(G, data), random_rotation(SEED).cuda())


def test_gcp_message_passing(G, data):
    """GCP2EquivarianceTest.test_gcp_message_passing (:922-975)."""
    torch.manual_seed(SEED)
    mp = G.GCPMessagePassing(NODE_DIM, NODE_DIM, EDGE_DIM, cfg=G.default_module_cfg(),
                             mp_cfg=G.default_layer_cfg().mp_cfg).cuda()
    with torch.no_grad():
        check_rotation_and_permutation(lambda n, e, ei, fr: mp(n, e, ei, fr), data, _frames(G, data),
                                       random_rotation(SEED).cuda())


def test_gcp_interactions(G, data):
    """GCP2EquivarianceTest.test_gcp_interactions (:977-1036)."""
    torch.manual_seed(SEED)
    layer = G.GCPInteractions(NODE_DIM, EDGE_DIM, cfg=G.default_module_cfg(), layer_cfg=G.default_layer_cfg(),
                              dropout=0.1).cuda().eval()
    with torch.no_grad():
        check_rotation_and_permutation(lambda n, e, ei, fr: layer(n, e, ei, fr), data, _frames(G, data),
                                       random_rotation(SEED).cuda())


def test_gcp_interactions2_with_gcp3(G, data):
    """The AR / EQ layer (reference gcpnet.py:1265-1451 with GCP3 blocks, configs/model/gcpnet_eq.yaml): sum aggregation
    over `row`, scalar message gate, two-layer scalar_out; same checker as GCPInteractions."""
    import functools
    torch.manual_seed(SEED)
    cfg = G.default_module_cfg(selected_GCP=functools.partial(G.GCP3))
    lc = G.default_layer_cfg(use_scalar_message_attention=True, aggregate_with_row=True, num_feedforward_layers=1)
    layer = G.GCPInteractions2(NODE_DIM, EDGE_DIM, cfg=cfg, layer_cfg=lc, dropout=0.1).cuda().eval()
    with torch.no_grad():
        check_rotation_and_permutation(lambda n, e, ei, fr: layer(n, e, ei, fr), data, _frames(G, data),
                                       random_rotation(SEED).cuda())


def _random_batch(G, int_types, seed):
    """construct_batch_from_random_data_list (:1490-1507): 8 graphs of 37 nodes / 1250 random edges each
    (self-loops and duplicate edges possible), block-diagonal collation."""
    g = torch.Generator().manual_seed(seed)
    n, e = N_NODES // BATCH, N_EDGES // BATCH
    parts = dict(h=[], chi=[], e=[], xi=[], x=[], edge_index=[], batch=[])
    for k in range(BATCH):
        parts["h"].append(torch.randint(0, 9, (n,), generator=g) if int_types else torch.randn(n, NODE_IN[0], generator=g))
        parts["chi"].append(torch.randn(n, NODE_IN[1], 3, generator=g))
        parts["e"].append(torch.randn(e, EDGE_IN[0], generator=g))
        parts["xi"].append(torch.randn(e, EDGE_IN[1], 3, generator=g))
        parts["x"].append(torch.randn(n, 3, generator=g) + torch.randint(1, 100, (1,), generator=g))
        parts["edge_index"].append(torch.randint(0, n, (2, e), generator=g) + k * n)
        parts["batch"].append(torch.full((n,), k))
    cat = {k: torch.cat(v, dim=1 if k == "edge_index" else 0).cuda() for k, v in parts.items()}
    return cat


def _transformed(G, b, Q, t):
    out = dict(b)
    out["x"] = rot(Q, b["x"]) + t
    out["chi"], out["xi"] = rot(Q, b["chi"]), rot(Q, b["xi"])
    return G.Batch(**{k: v.clone() for k, v in out.items()})


def test_nms_model_rotation_and_translation_equivariance(G):
    """GCP2NewtonianManyBodySystemEquivarianceTest (:1372-1467) via helper :1639-1769."""
    torch.manual_seed(SEED)
    model_cfg = dict(h_input_dim=NODE_IN[0], chi_input_dim=NODE_IN[1], e_input_dim=EDGE_IN[0], xi_input_dim=EDGE_IN[1],
                     h_hidden_dim=64, chi_hidden_dim=16, e_hidden_dim=32, xi_hidden_dim=4, num_encoder_layers=4,
                     dropout=0.1)
    model = G.GCPNetNMS(model_cfg=model_cfg, module_cfg=G.default_module_cfg(), layer_cfg=G.default_layer_cfg()).cuda().eval()
    b = _random_batch(G, False, SEED)
    Q, t = random_rotation(SEED).cuda(), torch.randn(1, 3, generator=torch.Generator().manual_seed(2)).cuda()
    with torch.no_grad():
        b0, x0 = model(G.Batch(**{k: v.clone() for k, v in b.items()}))
        b1, x1 = model(_transformed(G, b, Q, t))
    tol = dict(atol=1e-4, rtol=1e-4)
    for v in (b0.h, b0.chi, x0, b1.h, b1.chi, x1):
        assert not v.isnan().any()
    assert torch.allclose(b0.h, b1.h, **tol), f"h: {(b0.h - b1.h).abs().max():.2e}"
    assert torch.allclose(b0.e, b1.e, **tol)
    assert torch.allclose(rot(Q, b0.chi), b1.chi, **tol), f"chi: {(rot(Q, b0.chi) - b1.chi).abs().max():.2e}"
    assert torch.allclose(rot(Q, b0.xi), b1.xi, **tol)
    assert torch.allclose(rot(Q, x0) + t, x1, **tol), f"x: {(rot(Q, x0) + t - x1).abs().max():.2e}"


def test_lba_model_rotation_and_translation_invariance(G):
    """GCP2LigandBindingAffinityEquivarianceTest (:1040-1140) via helper :1511-1635."""
    torch.manual_seed(SEED)
    model_cfg = dict(chi_input_dim=NODE_IN[1], e_input_dim=EDGE_IN[0], xi_input_dim=EDGE_IN[1], h_hidden_dim=100,
                     chi_hidden_dim=16, e_hidden_dim=32, xi_hidden_dim=4, output_dim=1, output_scale_factor=2,
                     num_encoder_layers=4, dropout=0.1, dense_dropout=0.1)
    model = G.GCPNetLBA(model_cfg=model_cfg, module_cfg=G.default_module_cfg(), layer_cfg=G.default_layer_cfg()).cuda().eval()
    b = _random_batch(G, True, SEED)
    Q, t = random_rotation(SEED).cuda(), torch.randn(1, 3, generator=torch.Generator().manual_seed(2)).cuda()
    with torch.no_grad():
        b0, p0 = model(G.Batch(**{k: v.clone() for k, v in b.items()}))
        b1, p1 = model(_transformed(G, b, Q, t))
    tol = dict(atol=1e-4, rtol=1e-4)
    assert not p0.isnan().any() and not p1.isnan().any()
    assert torch.allclose(b0.h, b1.h, **tol), f"h: {(b0.h - b1.h).abs().max():.2e}"
    assert torch.allclose(rot(Q, b0.chi), b1.chi, **tol), f"chi: {(rot(Q, b0.chi) - b1.chi).abs().max():.2e}"
    assert torch.allclose(p0, p1, **tol), f"pred: {(p0 - p1).abs().max():.2e}"
