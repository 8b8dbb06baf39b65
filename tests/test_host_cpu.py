"""CPU-only checks of the host side: the C-ABI library loads and exports every declared symbol, the module tree
mirrors the reference's parameter names, config handling, graph plans, and loud failure without a GPU."""
import os
import re

import pytest
import torch

import gcpnet_amd as G
from gcpnet_amd import _lib, ops
from tests.helpers import Fixture

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "gcpnet_hip.h")).read()
    declared = set(re.findall(r"\b(gcpnet_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.gcpnet_abi_version() == 4


def test_host_only_entry_points():
    lib = _lib.load()
    # NMS message GCP0: (160,36)->(64,16), H=9: K=178 -> KP=184, NTG=2, NG=1, NUG=4, NGK=2
    n = lib.gcpnet_gcp2_pack_floats(160, 36, 64, 16, 9, 1)
    KK, NTG, NG, NUG, NGK, NS = 92, 2, 1, 4, 2, 32
    NTS = 5  # 32-wide tiles of the 160 scalar inputs (section F, register-resident chain kernel)
    # vector Linears as MFMA fragments: H + 3 = 12 rows: ceil(36/2) + 4*ceil(9/8) + 4*ceil(16/8) + ceil(36/32) * 4*ceil(12/8) steps
    SV = 18 + 8 + 8 + 2 * 8
    assert n == (NG * KK * 64 * NTG + NGK * NS * 64 * NUG + 1 * NS * 64 + 8 * NG * 64 * NTG + NTS * 16 * 64 * NTG
                 + SV * 64)
    # 128 row splits by default (csrc/tn_gemm.hip TN_TARGET_SPLITS): ceil(160000 / 128) rounded up to whole 32-row chunks = 1280 rows
    # = 125 splits, made even (a split's 16-row chunks keep their place inside the 32-row tiles of a tile-blocked operand: tn_pipe_kernel)
    assert lib.gcpnet_tn_splits(160000, 128, 142) == 126
    assert lib.gcpnet_debug_knobs_compiled() == 0  # the shipped build carries no result-changing measurement knob
    assert lib.gcpnet_tn_splits(0, 1, 1) == 2
    assert lib.gcpnet_tn_splits(160000, 0, 0) == 126 and lib.gcpnet_tn_splits(10000, 512, 145) == 106  # (by rows only: 96 rows per split)
    # packed image of a chainable block: the fp32 sections + B6 / F6 as TWO fp16 terms (csrc/gcp_f16x2.h) + the gate image C6 as three
    # bf16 terms (csrc/gcp_bf16x3.h): (128,16,H=4): NTG = 4, NKT = 5 -> 2*4*5*512 + 2*4*4*512 + 2*4*768 floats on top
    base = lambda si, vi, so, vo, H: lib.gcpnet_gcp2_pack_floats(si, vi, so, vo, H, 1)
    S = 128
    fp32_part = (1 * 72 * 64 * 4) + (2 * 64 * 64 * 4) + (1 * 64 * 64) + (8 * 1 * 64 * 4) + (4 * 16 * 64 * 4) + (8 + 4 + 8 + 4) * 64
    assert base(S, 16, S, 16, 4) == fp32_part + (2 * 4 * 5 + 2 * 4 * 4) * 512 + 2 * 4 * 768
    # which residual chains the register-resident forward kernel takes (ops asks before preferring it to the workgroup kernel)
    ok = lib.gcpnet_gcp2_chain_forward_registers_ok
    assert ok(128, 16, 128, 16, 4, 1) == 1 and ok(64, 16, 64, 16, 4, 1) == 1 and ok(100, 16, 100, 16, 4, 1) == 1
    assert ok(256, 32, 256, 32, 8, 1) == 0  # two output groups: the workgroup kernels
    assert ok(128, 16, 64, 16, 4, 1) == 0 and ok(32, 4, 32, 4, 1, 1) == 0  # not a residual shape / a single 32-wide tile


def test_bad_arguments_are_rejected_without_touching_the_gpu():
    lib = _lib.load()
    assert lib.gcpnet_segment_reduce(-1, None, None, None, 0, 4, 1, None, 0, 0, None) == -1
    assert lib.gcpnet_tn_gemm(0, None, None) == -1
    assert lib.gcpnet_localize(5, None, None, None, 1, None, None) == -1


@pytest.mark.parametrize("name,upd", [("interactions", False), ("interactions_posupd", True), ("interactions_force", True)])
def test_state_dict_names_match_reference(name, upd):
    f = Fixture(name)
    cfg = G.default_module_cfg(ablate_x_force_update=name != "interactions_force")
    layer = G.GCPInteractions((64, 16), (32, 4), cfg=cfg, layer_cfg=G.default_layer_cfg(),
                              dropout=0.0, updating_node_positions=upd)
    sd = layer.state_dict()
    assert list(sd) == list(f.p)
    for k in sd:
        assert tuple(sd[k].shape) == tuple(f.p[k].shape), k
    layer.load_state_dict(f.p)  # strict


def test_gcp3_feedforward_state_dict_names_match_reference():
    f = Fixture("gcp3_feedforward")  # scalar_out.0.* / scalar_out.2.* (reference gcpnet.py:529-533)
    mod = G.GCP3((40, 8), (24, 8), bottleneck=4, feedforward_out=True)
    assert list(mod.state_dict()) == list(f.p)
    mod.load_state_dict(f.p)


@pytest.mark.parametrize("name,seed,kw", [
    ("gcp_edge_default", 60, dict(bottleneck=4)),
    ("gcp_sigma_gate", 62, dict(nonlinearities=("relu", "sigmoid"), sigma_frame_gate=True, vector_residual=True)),
    ("gcp_frame_gate", 63, dict(nonlinearities=("silu", "silu"), bottleneck=2, frame_gate=True, vector_frame_residual=True)),
    ("gcp_scalar_out", 65, dict(nonlinearities=("relu", None))),
    ("gcp_ablate_frames", 66, dict(nonlinearities=("relu", None), bottleneck=4, ablate_frame_updates=True)),
])
def test_original_gcp_state_dict_and_seeded_init_match_reference(name, seed, kw):
    """The original GCP block (reference gcpnet.py:30-101): same keys in the same order, and -- parameters being created in
    the reference's order -- the same seed gives the reference's initial weights bit for bit."""
    import torch
    f = Fixture(name)
    torch.manual_seed(seed)
    mod = G.GCP(tuple(int(d) for d in f.m["in_dims"]), tuple(int(d) for d in f.m["out_dims"]), **kw)
    sd = mod.state_dict()
    assert list(sd) == list(f.p)
    for k in sd:
        assert torch.equal(sd[k], f.p[k]), k


def test_interactions2_state_dict_names_match_reference():
    import functools
    for name, kw, upd in (("interactions2_eq", dict(use_scalar_message_attention=True, aggregate_with_row=True,
                                                   num_feedforward_layers=1), False),
                          ("interactions2_posupd", dict(use_scalar_message_attention=True, num_message_layers=4,
                                                       num_feedforward_layers=2), True)):
        f = Fixture(name)
        cfg = G.default_module_cfg(selected_GCP=functools.partial(G.GCP3))
        layer = G.GCPInteractions2((64, 16), (32, 4), cfg=cfg, layer_cfg=G.default_layer_cfg(**kw), dropout=0.0,
                                   updating_node_positions=upd)
        assert list(layer.state_dict()) == list(f.p)
        layer.load_state_dict(f.p)


def test_model_state_dict_names_match_reference():
    f = Fixture("model_lba_small")
    model_cfg = dict(chi_input_dim=2, e_input_dim=16, xi_input_dim=1, h_hidden_dim=20, chi_hidden_dim=4, e_hidden_dim=8,
                     xi_hidden_dim=4, output_dim=1, output_scale_factor=2, num_encoder_layers=2, dropout=0.0,
                     dense_dropout=0.1)
    model = G.GCPNetLBA(model_cfg=model_cfg, module_cfg=G.default_module_cfg(),
                        layer_cfg=G.default_layer_cfg(num_message_layers=4))
    assert set(model.state_dict()) == set(f.p)
    model.load_state_dict(f.p)
    f = Fixture("model_nms_small")
    model_cfg = dict(h_input_dim=1, chi_input_dim=3, e_input_dim=17, xi_input_dim=1, h_hidden_dim=32, chi_hidden_dim=8,
                     e_hidden_dim=16, xi_hidden_dim=4, num_encoder_layers=2, dropout=0.0)
    model = G.GCPNetNMS(model_cfg=model_cfg, module_cfg=G.default_module_cfg(),
                        layer_cfg=G.default_layer_cfg(num_message_layers=4))
    assert set(model.state_dict()) == set(f.p)


def test_reference_errors_are_reproduced():
    with pytest.raises(AssertionError):  # gcpnet.py:294-296
        G.GCP2((8, 6), (8, 6), bottleneck=4)
    with pytest.raises(NotImplementedError):  # models/__init__.py:57
        G.GCP2((8, 4), (8, 4), nonlinearities=("gelu", None))
    with pytest.raises(NotImplementedError):
        G.get_nonlinearity("tanh")


def test_no_cpu_fallback():
    block = G.GCP2((8, 4), (8, 4), nonlinearities=("relu", None), bottleneck=4)
    s, v = torch.randn(5, 8), torch.randn(5, 4, 3)
    ei = torch.stack((torch.arange(5), torch.arange(5)))
    with pytest.raises(_lib.GcpnetHipError):
        block((s, v), ei, torch.randn(5, 3, 3))
    with pytest.raises(_lib.GcpnetHipError):
        G.GCPLayerNorm((8, 4))(G.ScalarVector(s, v))


def test_gather_plan():
    idx = torch.tensor([2, 0, 2, 1, 0, 2])
    p = ops.GatherPlan(idx, 4)
    assert p.perm is not None and idx[p.perm.long()].tolist() == [0, 0, 1, 2, 2, 2]
    assert p.seg_ptr.tolist() == [0, 2, 3, 6, 6]
    assert torch.allclose(p.inv_count, torch.tensor([0.5, 1.0, 1 / 3, 1.0]))
    p2 = ops.GatherPlan(torch.tensor([0, 0, 1, 3]), 4)
    assert p2.perm is None and p2.seg_ptr.tolist() == [0, 2, 3, 3, 4]
    ei = torch.tensor([[0, 1, 2, 0], [1, 2, 0, 2]])
    g1 = ops.GraphPlan.get(ei, 3)
    assert ops.GraphPlan.get(ei, 3) is g1 and g1.n_edges == 4


def test_scalar_vector_surface():
    s, v = torch.randn(4, 5), torch.randn(4, 2, 3)
    sv = G.ScalarVector(s, v)
    flat = sv.flatten()
    assert flat.shape == (4, 11)
    back = G.ScalarVector.recover(flat, 2)
    assert torch.equal(back.scalar, s) and torch.equal(back.vector, v)
    cs, cv = sv.concat((G.ScalarVector(s, v),))
    assert cs.shape == (4, 10) and cv.shape == (4, 4, 3)  # vectors join on the channel axis
    assert torch.equal((sv + sv).vector, 2 * v) and torch.equal((sv * 2).scalar, 2 * s)
    assert torch.equal(sv.idx(torch.tensor([1, 3])).scalar, s[[1, 3]])


def test_config_loader_accepts_reference_layout(tmp_path):
    """A configs/model tree with the reference's key names, `defaults:` composition, `${..x}` interpolation and
    `_target_` / `_partial_` entries (configs/model/gcpnet_nms.yaml, module_cfg/gcp_module_nms.yaml)."""
    m = tmp_path / "model"
    (m / "module_cfg").mkdir(parents=True)
    (m / "layer_cfg" / "mp_cfg").mkdir(parents=True)
    (m / "model_cfg").mkdir()
    (m / "gcpnet_nms.yaml").write_text(
        "_target_: src.models.gcpnet_nms_module.GCPNetNMSLitModule\n"
        "layer_class:\n  _target_: src.models.components.gcpnet.GCPInteractions\n  _partial_: true\n"
        "  updating_node_positions: true\n"
        "optimizer:\n  _target_: torch.optim.Adam\n  _partial_: true\n  lr: 1e-4\n"
        "defaults:\n  - model_cfg: gcp_model_nms.yaml\n  - module_cfg: gcp_module_nms.yaml\n"
        "  - layer_cfg: gcp_interaction_layer_nms.yaml\n")
    (m / "model_cfg" / "gcp_model_nms.yaml").write_text(
        "h_input_dim: 1\nchi_input_dim: 3\ne_input_dim: 17\nxi_input_dim: 1\nh_hidden_dim: 16\nchi_hidden_dim: 4\n"
        "e_hidden_dim: 8\nxi_hidden_dim: 4\nnum_encoder_layers: 2\ndropout: 0.1\n")
    (m / "module_cfg" / "gcp_module_nms.yaml").write_text(
        "selected_GCP:\n  _target_: src.models.components.gcpnet.GCP2\n  _partial_: true\n"
        "norm_x_diff: true\nscalar_gate: 0\nvector_gate: true\nvector_residual: false\nvector_frame_residual: false\n"
        "frame_gate: false\nsigma_frame_gate: false\nscalar_nonlinearity: relu\nvector_nonlinearity: \n"
        "nonlinearities:\n  - ${..scalar_nonlinearity}\n  - ${..vector_nonlinearity}\nbottleneck: 4\n"
        "vector_linear: true\nvector_identity: true\ndefault_vector_residual: false\ndefault_bottleneck: 4\n"
        "node_positions_weight: 1.0\nablate_frame_updates: false\nablate_scalars: false\nablate_vectors: false\n"
        "ablate_x_force_update: true\nenable_e3_equivariance: false\n")
    (m / "layer_cfg" / "gcp_interaction_layer_nms.yaml").write_text(
        "defaults:\n  - mp_cfg: gcp_mp_nms.yaml\npre_norm: false\nnum_feedforward_layers: 2\ndropout: 0.1\n"
        "nonlinearity_slope: 1e-2\n")
    (m / "layer_cfg" / "mp_cfg" / "gcp_mp_nms.yaml").write_text(
        "edge_encoder: false\nedge_gate: false\nnum_message_layers: 3\nmessage_residual: 0\n"
        "message_ff_multiplier: 1\nself_message: true\nuse_residual_message_gcp: true\n")
    cfg = G.load_model_config(str(m / "gcpnet_nms.yaml"))
    assert cfg.module_cfg.nonlinearities == ["relu", None]
    assert cfg.layer_cfg.mp_cfg.num_message_layers == 3
    model = G.instantiate(cfg)
    assert isinstance(model, G.GCPNetNMS)
    assert len(model.interaction_layers) == 2 and model.interaction_layers[0].updating_node_positions
    assert "interaction_layers.1.interaction.message_fusion.2.scalar_out.weight" in model.state_dict()


def test_compat_aliases_resolve_reference_dotted_names():
    """gcpnet_amd.compat.install_aliases(): the reference's `_target_` strings and LitModule imports resolve to this package."""
    import importlib
    import subprocess
    import sys

    code = (
        "import gcpnet_amd, gcpnet_amd.compat as c; c.install_aliases();"
        "from src.models.components.gcpnet import GCPInteractions, GCP2, GCPEmbedding;"
        "from src.models.components import ScalarVector, GCPLayerNorm, centralize, decentralize, localize;"
        "import importlib; m = importlib.import_module('src.models.components.gcpnet');"
        "assert getattr(m, 'GCPInteractions') is gcpnet_amd.GCPInteractions and GCP2 is gcpnet_amd.GCP2;"
        "assert ScalarVector is gcpnet_amd.ScalarVector and localize is gcpnet_amd.localize; print('ok')")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr[-2000:]


def test_pack_cache_handles_survive_deepcopy_and_pickle():
    """The batched-pack handle (ops._WgPackUser) sits in a module's pack cache and holds weak references: a deep copy or a pickle of the
    module must not trip over it -- the copy starts without a handle."""
    import copy
    import pickle

    from gcpnet_amd import ops

    u = ops._WgPackUser()
    cache = {"wg_user": u, "wg_key": (0, None)}
    u.cache = cache
    u.w_scalar, u.w_gate = None, None
    c2 = copy.deepcopy(cache)
    assert c2["wg_user"] is None and c2["wg_key"] == (0, None)
    c3 = pickle.loads(pickle.dumps({"wg_user": u, "wg_key": (0, None)}))
    assert c3["wg_user"] is None
