"""GPU tests of the trainer-side pieces: HIP dropout (train-mode statistics, whole-vector masks, mask-consistent backward), a
train-mode layer with the shipped dropout 0.1, fused Adam against torch.optim.Adam."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import gcpnet_amd

    return gcpnet_amd


@pytest.mark.parametrize("p", [0.1, 0.5])
def test_dropout_statistics_and_backward(G, p):
    from gcpnet_amd import ops

    torch.manual_seed(3)
    n = 200000
    s = torch.ones(n, 8, device="cuda", requires_grad=True)
    v = torch.ones(n, 4, 3, device="cuda", requires_grad=True)
    ys, yv = ops.dropout(s, p, group=1), ops.dropout(v, p, group=3)
    keep = 1.0 - p
    # survivors are scaled by 1 / keep, everything else is exactly zero
    u = torch.unique(ys.detach()).tolist()
    assert len(u) == 2 and u[0] == 0.0 and abs(u[1] - 1.0 / keep) < 1e-6, u
    rate_s, rate_v = float((ys != 0).float().mean()), float((yv != 0).float().mean())
    tol = 5.0 * (keep * p / (n * 4)) ** 0.5  # five sigma of a Bernoulli mean over the fewer draws
    assert abs(rate_s - keep) < tol and abs(rate_v - keep) < tol, (rate_s, rate_v)
    assert abs(float(ys.mean()) - 1.0) < 5 * tol  # unbiased
    # whole 3-vectors are dropped together (components/__init__.py:113-114)
    alive = (yv != 0)
    assert bool((alive.all(-1) | (~alive).all(-1)).all())
    # rows are not dropped as a whole (independent draws per element / per vector)
    assert 0.0 < float(alive[:, 0, 0].float().mean()) < 1.0 and float((alive[:, 0, 0] ^ alive[:, 1, 0]).float().mean()) > 0.05
    # backward applies the very same mask and scale
    gs, gv = torch.autograd.grad([ys.sum(), yv.sum()], [s, v])
    assert torch.equal(gs, ys.detach()) and torch.equal(gv, yv.detach())
    # a different draw next time; the same draw for the same seed
    assert not torch.equal(ops.dropout(s, p, 1), ys)
    assert torch.equal(ops.dropout(s.detach(), p, 1, seed=11), ops.dropout(s.detach(), p, 1, seed=11))


def test_gcp_dropout_module_and_layer_in_train_mode(G):
    """GCPDropout is the identity in eval mode and active in train mode; a GCPInteractions layer with the shipped dropout 0.1
    runs fwd + bwd in train mode with finite results that differ from the eval-mode ones by O(p)."""
    from oracle import gcp_oracle as O
    from tests.helpers import rand_graph

    torch.manual_seed(4)
    d = G.GCPDropout(0.1).cuda()
    s, v = torch.randn(50, 16, device="cuda"), torch.randn(50, 4, 3, device="cuda")
    d.eval()
    out = d(G.ScalarVector(s, v))
    assert out[0] is s and out[1] is v
    d.train()
    out = d(G.ScalarVector(s, v))
    assert not torch.equal(out[0], s) and float((out[0] == 0).float().mean()) > 0.02
    n, e = 400, 5000
    layer = G.GCPInteractions((64, 16), (32, 4), cfg=G.default_module_cfg(), layer_cfg=G.default_layer_cfg(), dropout=0.1).cuda()
    ei, x = rand_graph(n, e, 5, sort_by_col=True)
    fr = O.localize(x, ei).cuda()
    g = torch.Generator().manual_seed(6)
    ins = [torch.randn(n, 64, generator=g).cuda().requires_grad_(), torch.randn(n, 16, 3, generator=g).cuda().requires_grad_(),
           torch.randn(e, 32, generator=g).cuda(), torch.randn(e, 4, 3, generator=g).cuda()]
    layer.eval()
    he, _ = layer((ins[0], ins[1]), (ins[2], ins[3]), ei.cuda(), fr)
    layer.train()
    ht, ct = layer((ins[0], ins[1]), (ins[2], ins[3]), ei.cuda(), fr)
    (ht.sum() + ct.sum()).backward()
    assert torch.isfinite(ht).all() and torch.isfinite(ins[0].grad).all() and torch.isfinite(ins[1].grad).all()
    rel = float((ht - he).norm() / he.norm())
    assert 1e-3 < rel < 1.0, rel


def test_fused_adam_matches_torch_adam(G):
    torch.manual_seed(7)
    shapes = [(64, 141), (64,), (4, 16), (3, 16), (16, 4), (16, 64), (16,), (1, 1), (300, 7)]
    ref = [torch.randn(s, device="cuda").requires_grad_() for s in shapes]
    mine = [t.detach().clone().requires_grad_() for t in ref]
    o1 = torch.optim.Adam(ref, lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    o2 = G.FusedAdam(mine, lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    for it in range(5):
        gs = [torch.randn_like(t) for t in ref]
        for a, b, g in zip(ref, mine, gs):
            a.grad, b.grad = g.clone(), g.clone()
        o1.step()
        o2.step()
        for a, b in zip(ref, mine):
            assert torch.allclose(a, b, atol=1e-6, rtol=1e-5), (it, a.shape, float((a - b).abs().max()))
    sd = o2.state_dict()
    assert sd["state"][0]["step"] == 5 and sd["state"][0]["exp_avg"].shape == shapes[0]


def test_nms_featurize_golden(G):
    """GPU featuriser against the fixture built with the reference's helper functions (incl. coincident bodies -> zeros)."""
    from tests.helpers import Fixture, close

    f = Fixture("nms_features")
    out = G.nms_featurize(f.i["x"].cuda(), f.i["vel"].cuda(), f.i["edge_attr"].cuda(), f.i["edge_index"].cuda(), f.i["batch"].cuda())
    for k in ("h", "chi", "e", "xi"):
        close(out[k].cpu(), f.o[k], atol=2e-6, rtol=1e-5)


def test_lba_featurize_and_collate_golden(G):
    """GPU LBA featuriser (radius graph + RBF / unit-difference edge features + orientations + ligand flag) and collate() against
    the fixture built by the reference's real LBATransform; edges compared in canonical (target, source) order."""
    from tests.helpers import Fixture, close

    def by_col_row(ei):
        return torch.argsort(ei[1] * (int(ei.max()) + 1) + ei[0])

    f = Fixture("lba_features")
    graphs = []
    for k in range(2):
        x = f.i[f"x{k}"].cuda()
        out = G.lba_featurize(x, f.o[f"h{k}"], n_ligand=int(f.i[f"n_ligand{k}"]), edge_cutoff=float(f.m["edge_cutoff"]),
                              num_rbf=int(f.m["num_rbf"]), max_num_neighbors=int(f.m["max_num_neighbors"]))
        want_ei = f.o[f"edge_index{k}"]
        pg, pw = by_col_row(out["edge_index"].cpu()), by_col_row(want_ei)
        assert torch.equal(out["edge_index"].cpu()[:, pg], want_ei[:, pw])
        assert bool((out["edge_index"][1, 1:] >= out["edge_index"][1, :-1]).all())  # col-sorted, as the kernels want it
        close(out["e"].cpu()[pg], f.o[f"e{k}"][pw], atol=2e-6, rtol=1e-5)
        close(out["xi"].cpu()[pg], f.o[f"xi{k}"][pw], atol=2e-6, rtol=1e-5)
        close(out["chi"].cpu(), f.o[f"chi{k}"], atol=2e-6, rtol=1e-5)
        assert torch.equal(out["h"].cpu(), f.o[f"h{k}"]) and torch.equal(out["lig_flag"].cpu(), f.o[f"lig_flag{k}"])
        out["label"] = f.o[f"label{k}"].cuda()
        graphs.append(out)
    b = G.collate(graphs)
    n0 = graphs[0]["x"].shape[0]
    assert torch.equal(b["batch"].cpu(), f.o["batch_batch"]) and b["ptr"].tolist() == [0, n0, n0 + graphs[1]["x"].shape[0]]
    assert torch.equal(b["label"].cpu(), f.o["batch_label"]) and torch.equal(b["h"].cpu(), f.o["batch_h"])
    assert torch.equal(b["lig_flag"].cpu(), f.o["batch_lig_flag"])
    close(b["x"].cpu(), f.o["batch_x"], atol=0, rtol=0)
    pg, pw = by_col_row(b["edge_index"].cpu()), by_col_row(f.o["batch_edge_index"])
    assert torch.equal(b["edge_index"].cpu()[:, pg], f.o["batch_edge_index"][:, pw])
    close(b["e"].cpu()[pg], f.o["batch_e"][pw], atol=2e-6, rtol=1e-5)
    # the collated batch featurised in ONE call (batch vector) gives the same tensors as per-structure calls + collate()
    one = G.lba_featurize(b["x"], b["h"], batch=b["batch"])
    assert torch.equal(one["edge_index"], b["edge_index"])
    for key in ("e", "xi", "chi"):
        close(one[key].cpu(), b[key].cpu(), atol=1e-6, rtol=1e-6)
    assert G.element_mapping(["C", "Cl", "CL", "Zn", "H"]).tolist() == [1, 6, 6, 8, 0]


def test_lba_featurize_neighbour_cap_first_found(G):
    """select="first": torch_cluster's choice among more than 32 in-range atoms (lowest ids, self loop removed after the cap), against
    the fixture of the reference's real LBATransform on a dense structure: identical edge list, features of those edges."""
    from tests.helpers import Fixture, close

    f = Fixture("lba_features_capped")
    x = f.i["x"].cuda()
    out = G.lba_featurize(x, f.o["h"], n_ligand=int(f.i["n_ligand"]), select="first")
    assert torch.equal(out["edge_index"].cpu(), f.o["edge_index"])
    close(out["e"].cpu(), f.o["e"], atol=2e-6, rtol=1e-5)
    close(out["xi"].cpu(), f.o["xi"], atol=2e-6, rtol=1e-5)
    close(out["chi"].cpu(), f.o["chi"], atol=2e-6, rtol=1e-5)
    assert torch.equal(out["lig_flag"].cpu(), f.o["lig_flag"])
    # the default keeps the NEAREST 32 instead: same targets, another (smaller or equal) edge list
    near = G.radius_graph(x, max_num_neighbors=32)
    assert int(torch.bincount(near[1]).max()) == 32 and not torch.equal(near.cpu(), f.o["edge_index"])
    # two copies of the structure as a batch: the walk stays inside each graph
    xb = torch.cat((x, x))
    bb = torch.cat((torch.zeros(x.shape[0], dtype=torch.long), torch.ones(x.shape[0], dtype=torch.long))).cuda()
    eb = G.radius_graph(xb, batch=bb, select="first").cpu()
    n = x.shape[0]
    assert torch.equal(eb, torch.cat((f.o["edge_index"], f.o["edge_index"] + n), dim=1))


@pytest.mark.parametrize("radius", [4.6, 0.3, 7.1])
def test_radius_graph_first_takes_the_cutoff_as_torch_cluster_forms_it(G, radius):
    """select="first" compares float d^2 with (float)(double r * double r), as torch_cluster does; the C ABI therefore
    takes the radius as a double (a float parameter squared in double differs by an ulp for radii that are not representable in
    float, and pairs at the boundary land on the other side).  Pairs placed EXACTLY at the two candidate thresholds decide."""
    import numpy as np
    from oracle import gcp_oracle as O

    thr = np.float32(radius * radius)                       # torch_cluster's threshold
    alt = np.float32(np.float64(np.float32(radius)) ** 2)   # the threshold a float-typed parameter would give
    pts = [[0.0, 0.0, 0.0]]
    for t in {float(thr), float(alt), float(np.nextafter(thr, np.float32(0))), float(np.nextafter(thr, np.float32(np.inf)))}:
        pts.append([float(np.sqrt(np.float64(t))), 0.0, 0.0])  # (x^2 in float is then within an ulp of t: both sides get covered)
    g = torch.Generator().manual_seed(3)
    x = torch.cat((torch.tensor(pts, dtype=torch.float32), torch.rand(300, 3, generator=g) * 2.5 * radius))
    want = O.radius_graph(x, torch.zeros(x.shape[0], dtype=torch.long), radius, 32, select="first")
    got = G.radius_graph(x.cuda(), r=radius, max_num_neighbors=32, select="first").cpu()
    assert torch.equal(got, want)


def test_radius_graph_matches_scipy_bit_for_bit(G):
    """GPU cell-list radius graph: the committed scipy fixture (3 graphs: dense, sparse, fewer nodes than K) and a fresh 20 000-node
    cloud against gcpnet_amd.synthetic.radius_graph -- identical edge_index arrays, col-sorted."""
    from tests.helpers import Fixture

    f = Fixture("radius_graph")
    ei = G.radius_graph(f.i["x"].cuda(), r=float(f.m["radius"]), max_num_neighbors=int(f.m["max_neighbors"]), batch=f.i["batch"].cuda())
    assert torch.equal(ei.cpu(), f.o["edge_index"])
    from gcpnet_amd.synthetic import radius_graph as scipy_graph

    x, want = scipy_graph(20000, 16, seed=5)
    got = G.radius_graph(x.cuda(), r=4.5, max_num_neighbors=16)
    assert torch.equal(got.cpu(), want)
    assert bool((got[1, 1:] >= got[1, :-1]).all())


@pytest.mark.parametrize("side_stream", [False, True], ids=["one-stream", "weight-grad-branch"])
def test_hip_graph_replay_matches_eager_step(G, side_stream):
    """A whole NMS `step()` + backward captured in a hipGraph (gcpnet_amd.graphs.GraphedStep): replays reproduce the eager step's
    loss and gradients bit for bit, and follow new data copied into the static input tensors.  side_stream: the weight-gradient
    stream stays inside the capture (forked by gcpnet_stream_wait_stream, joined by the end-of-backward callback) as a parallel
    branch of the graph."""
    from gcpnet_amd.graphs import GraphedStep
    from gcpnet_amd.synthetic import model_batch

    torch.manual_seed(2)
    batch, model_cfg, _, _ = model_batch("c1", seed=3)
    model = G.GCPNetNMS(model_cfg=model_cfg, module_cfg=G.default_module_cfg(), layer_cfg=G.default_layer_cfg()).cuda().eval()
    dev = {k: v.cuda() for k, v in batch.items()}
    params = list(model.parameters())

    def step():
        for p in params:
            p.grad = None
        loss, _, _ = model.step(G.Batch(**dev))
        loss.backward()
        return loss

    eager_loss = step().detach().clone()
    eager = [p.grad.clone() for p in params]
    graphed = GraphedStep(step, side_stream=side_stream)
    loss = graphed()
    torch.cuda.synchronize()
    assert torch.equal(loss.detach(), eager_loss)
    for p, g in zip(params, eager):
        assert torch.equal(p.grad, g)
    # new data through the static tensors
    batch2, _, _, _ = model_batch("c1", seed=4)
    for k in ("h", "chi", "e", "xi", "x", "label"):
        dev[k].copy_(batch2[k].cuda())
    loss2 = graphed().detach().clone()
    g2 = [p.grad.clone() for p in params]
    ref_loss = step().detach()
    assert torch.equal(loss2, ref_loss) and not torch.equal(loss2, eager_loss)
    for p, g in zip(params, g2):
        assert torch.equal(p.grad, g)


def test_captured_training_steps_with_adam_match_torch_adam_on_the_oracle(G):
    """forward + MSE loss + backward + Adam as ONE hipGraph (GraphedStep(step, optimizer=FusedAdam(capturable=True))): 2 eager
    warm-up steps + 8 replays must leave the parameters where 10 steps of torch.optim.Adam on the CPU oracle leave them
    (gcpnet_nms_module.py:153-178: the reference's step is forward + loss + backward + Adam, configs/model/gcpnet_nms.yaml:8-12).
    eps = 1e-4 for both optimizers: with the default 1e-8 Adam turns a gradient that is round-off (1e-12 in one implementation, 0 or
    -1e-12 in the other) into a full +-lr update, which compares the signs of noise, not the optimizers."""
    from gcpnet_amd.graphs import GraphedStep
    from oracle import gcp_oracle as O
    from tests.golden.gen_helpers import nms_like_batch

    torch.manual_seed(5)
    model_cfg = dict(h_input_dim=1, chi_input_dim=3, e_input_dim=17, xi_input_dim=1, h_hidden_dim=64, chi_hidden_dim=16,
                     e_hidden_dim=32, xi_hidden_dim=4, num_encoder_layers=2, dropout=0.0)
    mcfg = G.default_module_cfg()
    mcfg["nonlinearities"] = ["silu", None]
    model = G.GCPNetNMS(model_cfg=model_cfg, module_cfg=mcfg, layer_cfg=G.default_layer_cfg()).cuda().train()
    b = nms_like_batch(12, 5, 77)
    b["label"] = b["x"] + 0.3 * torch.randn(b["x"].shape, generator=torch.Generator().manual_seed(78))
    # ---- the oracle: 10 steps of torch.optim.Adam on the CPU
    P = {k: v.detach().cpu().clone().requires_grad_() for k, v in model.state_dict().items()}
    ocfg = O.default_module_cfg()
    ocfg["nonlinearities"] = ("silu", None)
    opt_ref = torch.optim.Adam(list(P.values()), lr=1e-3, eps=1e-4)
    ref_losses = []
    for _ in range(10):
        opt_ref.zero_grad(set_to_none=True)
        loss = torch.nn.functional.mse_loss(O.nms_forward(P, b, ocfg, O.default_layer_cfg(), 2)["x"], b["label"])
        loss.backward()
        opt_ref.step()
        ref_losses.append(float(loss))
    # ---- the product path: 2 eager steps (GraphedStep's warm-up) + 8 replays of the captured step
    dev = {k: v.cuda() for k, v in b.items()}
    params = list(model.parameters())
    opt = G.FusedAdam(params, lr=1e-3, eps=1e-4, capturable=True)
    losses = []

    def step():
        for p in params:
            p.grad = None
        loss, _, _ = model.step(G.Batch(**dev))
        loss.backward()
        return loss

    graphed = GraphedStep(step, warmup=2, optimizer=opt)
    for _ in range(8):
        losses.append(float(graphed()))
    torch.cuda.synchronize()
    assert int(opt.state[params[0]]["step"].item()) == 10
    for a, r in zip(losses, ref_losses[2:]):  # (replay k computes the loss of step k + 2)
        assert abs(a - r) <= 1e-5 * max(1.0, abs(r)), (losses, ref_losses)
    worst = 0.0
    for k, v in model.state_dict().items():
        worst = max(worst, float((v.cpu() - P[k].detach()).abs().max()))
    assert worst <= 1e-5, worst
    # an optimizer with a host-side step count still refuses capture
    opt2 = G.FusedAdam(params, lr=1e-3)
    with pytest.raises(RuntimeError):
        GraphedStep(step, warmup=1, optimizer=opt2)


def test_forward_without_backward_frees_saved_activations(G):
    """A training-mode forward whose backward never runs (validation without no_grad, an aborted step) must give its saved
    activations back once the outputs are dropped: nothing may hang off the autograd context in a cycle."""
    import gc

    from gcpnet_amd.synthetic import make_inputs

    ins = make_inputs(600, 10, node_dims=(128, 16), seed=5)
    ei, x = ins.pop("edge_index").cuda(), ins.pop("x").cuda()
    torch.manual_seed(0)
    layer = G.GCPInteractions((128, 16), (32, 4), cfg=G.default_module_cfg(), layer_cfg=G.default_layer_cfg(), dropout=0.0).cuda()
    fr = G.localize(x, ei)
    gi = {k: t.cuda().requires_grad_() for k, t in ins.items()}

    def fwd():
        return layer((gi["h"], gi["chi"]), (gi["e"], gi["xi"]), ei, fr)

    out = fwd()
    del out
    gc.collect()
    torch.cuda.synchronize()
    base = torch.cuda.memory_allocated()
    for _ in range(3):
        out = fwd()
        del out
    gc.collect()
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() <= base + (1 << 20), (torch.cuda.memory_allocated() - base)


def test_deepcopy_of_a_model_after_a_step_runs_and_agrees(G):
    """copy.deepcopy(model) after the pack caches have been filled (EMA / SWA copies, checkpointing the whole module): the copy carries
    no stale packed image and no handle of the original (ops._WgPackUser), and its step gives the same loss and gradients."""
    import copy

    from gcpnet_amd.synthetic import model_batch

    torch.manual_seed(6)
    batch, model_cfg, _, _ = model_batch("c1", seed=7)
    model = G.GCPNetNMS(model_cfg=model_cfg, module_cfg=G.default_module_cfg(), layer_cfg=G.default_layer_cfg()).cuda().eval()
    dev = {k: v.cuda() for k, v in batch.items()}

    def run(m):
        for p in m.parameters():
            p.grad = None
        loss, _, _ = m.step(G.Batch(**dev))
        loss.backward()
        torch.cuda.synchronize()
        return loss.detach().clone(), [p.grad.clone() for p in m.parameters()]

    l0, g0 = run(model)
    twin = copy.deepcopy(model)
    with torch.no_grad():  # (move the original's weights on: the twin must not see the original's images or weights)
        for p in model.parameters():
            p.mul_(1.5)
    l1, g1 = run(twin)
    assert torch.equal(l0, l1)
    for a, b in zip(g0, g1):
        assert torch.equal(a, b)
    l2, _ = run(model)
    assert not torch.equal(l2, l0)


def test_capturable_adam_state_dict_round_trip_through_the_cpu(G, tmp_path):
    """FusedAdam(capturable=True): save -> torch.load(map_location="cpu") -> load_state_dict -> step continues the run (the device step
    counter is rebuilt on the parameter device, never read through a host pointer), and matches torch.optim.Adam doing the same."""
    torch.manual_seed(0)
    w0 = [torch.randn(37, 5), torch.randn(11)]
    grads = [[torch.randn_like(w) for w in w0] for _ in range(5)]

    def run(make_opt, resume):
        ps = [torch.nn.Parameter(w.clone().cuda()) for w in w0]
        opt = make_opt(ps)
        for k, gs in enumerate(grads):
            if resume and k == 3:
                f = tmp_path / "opt.pt"
                torch.save(opt.state_dict(), f)
                opt = make_opt(ps)
                opt.load_state_dict(torch.load(f, map_location="cpu"))
            for p, g in zip(ps, gs):
                p.grad = g.cuda()
            opt.step()
        torch.cuda.synchronize()
        return [p.detach().cpu() for p in ps]

    want = run(lambda ps: torch.optim.Adam(ps, lr=1e-2), False)
    for resume in (False, True):
        got = run(lambda ps: G.FusedAdam(ps, lr=1e-2, capturable=True), resume)
        for a, b in zip(got, want):
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)
    assert "_step_dev" not in G.FusedAdam([torch.nn.Parameter(torch.zeros(1).cuda())], capturable=True).state_dict()["param_groups"][0]


def test_capturable_adam_refuses_a_changing_parameter_set(G):
    ps = [torch.nn.Parameter(torch.randn(4).cuda()) for _ in range(2)]
    opt = G.FusedAdam(ps, capturable=True)
    for p in ps:
        p.grad = torch.ones_like(p)
    opt.step()
    ps[1].grad = None
    with pytest.raises(RuntimeError, match="every parameter must take part"):
        opt.step()
