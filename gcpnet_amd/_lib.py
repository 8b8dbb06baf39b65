"""ctypes binding of libgcpnet_hip.so (the C ABI declared in include/gcpnet_hip.h).

There is deliberately no CPU fallback: if the library is missing the product path raises.  PyTorch is used for
device memory, the current HIP stream and autograd plumbing only.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (GCPNET_HIP_LIB: debugging knob -- load another build of the same ABI, e.g. an older kernel revision when bisecting a fault)
LIB_PATH = os.environ.get("GCPNET_HIP_LIB") or os.path.join(_HERE, "csrc", "libgcpnet_hip.so")

ACT = {None: 0, "relu": 1, "leakyrelu": 2, "selu": 3, "silu": 4, "sigmoid": 5}
VMODE_NONE, VMODE_SCALAR_GATE, VMODE_SELF_GATE = 0, 1, 2
MAX_SEG, TN_MAX_SEG, TN_MAX_PROBLEMS = 3, 4, 8

EXPORTS = [
    "gcpnet_abi_version", "gcpnet_debug_knobs_compiled", "gcpnet_gcp2_pack_floats", "gcpnet_pack_gcp2_weights", "gcpnet_pack_gcp2_weights_multi", "gcpnet_gcp2_forward",
    "gcpnet_gcp2_forward_lds_bytes",
    "gcpnet_gcp2_chain_forward", "gcpnet_gcp2_chain_forward_registers_ok", "gcpnet_gcp2_headchain_forward",
    "gcpnet_gcp2_backward", "gcpnet_gcp2_chain_backward", "gcpnet_gcp2_chain_backward_gathered", "gcpnet_gcp2_chain_backward_ok", "gcpnet_gcp2_chain_backward_flags", "gcpnet_debug_force_chain_split", "gcpnet_gcp2_chain_backward_split", "gcpnet_tb_floats", "gcpnet_tb_sign_words", "gcpnet_gcp2_bwd_tiles", "gcpnet_tn_gemm", "gcpnet_tn_splits", "gcpnet_reduce_partials",
    "gcpnet_reduce_partials_groups", "gcpnet_segment_reduce", "gcpnet_gather_rows",
    "gcpnet_localize", "gcpnet_layernorm_forward", "gcpnet_layernorm_backward", "gcpnet_layernorm_bwd_scratch_floats", "gcpnet_axpy_clamp", "gcpnet_rows_matmul_small", "gcpnet_edge_force_forward", "gcpnet_edge_force_backward",
    "gcpnet_edge_force_bwd_blocks", "gcpnet_row_gate_forward", "gcpnet_row_gate_backward", "gcpnet_row_gate_bwd_blocks",
    "gcpnet_debug_set_phase_timing", "gcpnet_debug_set_fp32_mfma", "gcpnet_debug_tn_occupancy",
    "gcpnet_wg_pack_floats", "gcpnet_wg_pack", "gcpnet_wg_pack_view", "gcpnet_wg_forward", "gcpnet_wg_backward_plan", "gcpnet_wg_backward",
    "gcpnet_wg_reduce", "gcpnet_wg_reduce_multi", "gcpnet_dropout", "gcpnet_adam_step", "gcpnet_adam_step_dev", "gcpnet_copy2d_multi", "gcpnet_axpy_clamp_backward", "gcpnet_nms_edge_features", "gcpnet_nms_node_features", "gcpnet_radius_graph", "gcpnet_radius_graph_first", "gcpnet_activation", "gcpnet_frame_gate_forward", "gcpnet_frame_gate_backward",
    "gcpnet_frame_gate_bwd_parts", "gcpnet_node_scalarize", "gcpnet_orientations",
    "gcpnet_gcp2_weight_grads", "gcpnet_gcp2_weight_grads_workspace", "gcpnet_stream_wait_stream", "gcpnet_wg_pack_multi",
]


class Concat(C.Structure):
    _fields_ = [("n", C.c_int), ("ptr", C.c_void_p * MAX_SEG), ("idx", C.c_void_p * MAX_SEG), ("dim", C.c_int * MAX_SEG)]


class Gcp2Weights(C.Structure):
    _fields_ = [
        ("si", C.c_int), ("vi", C.c_int), ("so", C.c_int), ("vo", C.c_int), ("hidden", C.c_int), ("use_frames", C.c_int),
        ("w_down", C.c_void_p), ("w_frames", C.c_void_p), ("w_up", C.c_void_p), ("w_scalar", C.c_void_p),
        ("b_scalar", C.c_void_p), ("w_gate", C.c_void_p), ("b_gate", C.c_void_p), ("pack", C.c_void_p),
    ]


class WgPackJob(C.Structure):
    _fields_ = [("w", Gcp2Weights), ("gated", C.c_int), ("W", C.c_void_p), ("ld", C.c_int), ("trans", C.c_int), ("nseg", C.c_int),
                ("start", C.c_int * 3), ("len", C.c_int * 3), ("out", C.c_void_p)]


class Gcp2Opts(C.Structure):
    _fields_ = [("act_s", C.c_int), ("act_v", C.c_int), ("slope", C.c_float), ("vmode", C.c_int),
                ("vector_residual", C.c_int), ("e3", C.c_int), ("fused_residual", C.c_int)]


class ChainItem(C.Structure):
    _fields_ = [("w", Gcp2Weights), ("o", Gcp2Opts), ("s_out", C.c_void_p), ("v_out", C.c_void_p), ("s_pre", C.c_void_p),
                ("gate", C.c_void_p), ("s_out_tb", C.c_int), ("s_pre_tb", C.c_int), ("s_sign", C.c_void_p)]


class Head(C.Structure):
    _fields_ = [("e_in", C.c_void_p), ("xi_in", C.c_void_p), ("s_add", Concat), ("v_add", Concat), ("w", Gcp2Weights),
                ("o", Gcp2Opts), ("s_out", C.c_void_p), ("v_out", C.c_void_p), ("s_pre", C.c_void_p), ("gate", C.c_void_p)]


class WgBlock(C.Structure):
    _fields_ = [("w", Gcp2Weights), ("o", Gcp2Opts), ("s_out", C.c_void_p), ("v_out", C.c_void_p), ("s_pre", C.c_void_p),
                ("gate", C.c_void_p), ("residual", C.c_int), ("s_out_tb", C.c_int), ("s_pre_tb", C.c_int)]


WG_MAX_BLOCKS = 9
MAX_CHAIN = 8  # GCP_MAX_CHAIN: blocks per launch of the wave-per-tile chain kernels


class Copy2dJob(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("rows", C.c_int), ("cols", C.c_int), ("src_rs", C.c_int64),
                ("src_cs", C.c_int64), ("dst_rs", C.c_int64), ("dst_cs", C.c_int64)]


class AdamTensor(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p), ("n", C.c_int64)]


class WgBwdPlan(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("nw", "kt", "fused", "split", "grid", "kw", "n_small", "ext_w", "dgate_w")]


class WgBwdArgs(C.Structure):
    _fields_ = [("w", Gcp2Weights), ("o", Gcp2Opts), ("residual", C.c_int), ("s_in", C.c_void_p), ("v_in", C.c_void_p),
                ("frames", C.c_void_p), ("v_add", C.POINTER(Concat)), ("s_pre", C.c_void_p), ("gate", C.c_void_p),
                ("d_s_out", C.c_void_p), ("d_v_out", C.c_void_p), ("d_s_in", C.c_void_p), ("d_v_in", C.c_void_p),
                ("ds_pre", C.c_void_p), ("dvhf", C.c_void_p), ("ext", C.c_void_p), ("dgate", C.c_void_p),
                ("dw_part", C.c_void_p), ("dwg_part", C.c_void_p), ("wsm_part", C.c_void_p), ("tb", C.c_int)]


class BwdScratch(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("ds_pre", "dgate", "ext", "w_part", "dvhf")]


class ChainBwdItem(C.Structure):
    _fields_ = [("w", Gcp2Weights), ("o", Gcp2Opts), ("v_in", C.c_void_p), ("s_pre", C.c_void_p), ("gate", C.c_void_p),
                ("sc", BwdScratch), ("tb", C.c_int), ("s_sign", C.c_void_p)]


class Operand(C.Structure):
    _fields_ = [("n", C.c_int), ("ptr", C.c_void_p * TN_MAX_SEG), ("idx", C.c_void_p * TN_MAX_SEG),
                ("dim", C.c_int * TN_MAX_SEG), ("ld", C.c_int * TN_MAX_SEG), ("act", C.c_int), ("slope", C.c_float),
                ("ones", C.c_int), ("tb", C.c_int * TN_MAX_SEG)]


class TnProblem(C.Structure):
    _fields_ = [("rows", C.c_int), ("a", Operand), ("b", Operand), ("out", C.c_void_p), ("out_sm", C.c_int64),
                ("out_sn", C.c_int64), ("out_m", C.c_int), ("out_n", C.c_int), ("out2", C.c_void_p), ("out2_n", C.c_int),
                ("partial", C.c_void_p), ("splits", C.c_int), ("m_split", C.c_int), ("out_b", C.c_void_p), ("out_b_sm", C.c_int64),
                ("out2_b", C.c_void_p)]


class ReduceJob(C.Structure):
    _fields_ = [("parts", C.c_void_p), ("n_parts", C.c_int), ("width", C.c_int), ("tmp", C.c_void_p), ("out", C.c_void_p)]


REDUCE_MAX_JOBS = 8


class WgradJob(C.Structure):
    _fields_ = [("rows", C.c_int), ("so", C.c_int), ("vo", C.c_int), ("vi", C.c_int), ("hidden", C.c_int), ("use_frames", C.c_int),
                ("gated", C.c_int), ("act_v", C.c_int), ("slope", C.c_float), ("s_in", Operand), ("ds_pre", C.c_void_p),
                ("ds_pre_tb", C.c_int), ("s_pre", C.c_void_p), ("s_pre_tb", C.c_int), ("ext", C.c_void_p), ("dgate", C.c_void_p),
                ("w_part", C.c_void_p), ("n_parts", C.c_int), ("w_width", C.c_int), ("d_w_scalar", C.c_void_p),
                ("d_b_scalar", C.c_void_p), ("d_w_small", C.c_void_p), ("d_w_gate", C.c_void_p), ("d_b_gate", C.c_void_p),
                ("gate_lin", C.c_int), ("w_scalar", C.c_void_p), ("b_scalar", C.c_void_p)]


class WgReduceJob(C.Structure):
    _fields_ = [("parts", C.c_void_p), ("n_parts", C.c_int), ("R", C.c_int), ("C", C.c_int), ("CW", C.c_int), ("out_w", C.c_void_p),
                ("out_b", C.c_void_p)]


class GcpnetHipError(RuntimeError):
    pass


_lib = None


def load():
    """Loads the HIP library; raises (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GcpnetHipError(
            f"{LIB_PATH} is missing: build it with `python -m gcpnet_amd.csrc.build` "
            "(hipcc --offload-arch=gfx950).  gcpnet_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_float
    P = C.POINTER
    lib.gcpnet_abi_version.restype = i32
    lib.gcpnet_gcp2_pack_floats.restype = i64
    lib.gcpnet_gcp2_pack_floats.argtypes = [i32] * 6
    lib.gcpnet_gcp2_forward_lds_bytes.restype = i64
    lib.gcpnet_gcp2_forward_lds_bytes.argtypes = [i32] * 6
    lib.gcpnet_pack_gcp2_weights.argtypes = [P(Gcp2Weights), vp, vp]
    lib.gcpnet_pack_gcp2_weights_multi.argtypes = [i32, P(Gcp2Weights), P(vp), vp]
    lib.gcpnet_gcp2_forward.argtypes = [i32, P(Concat), P(Concat), vp, P(Gcp2Weights), P(Gcp2Opts), P(Concat), P(Concat), vp,
                                        vp, vp, vp, vp, vp, vp]
    lib.gcpnet_gcp2_chain_forward.argtypes = [i32, vp, vp, vp, i32, P(ChainItem), vp]
    lib.gcpnet_gcp2_headchain_forward.argtypes = [i32, P(Head), vp, i32, P(ChainItem), vp]
    lib.gcpnet_gcp2_backward.argtypes = [i32, P(Concat), P(Concat), vp, P(Gcp2Weights), P(Gcp2Opts), P(Concat), vp, vp, vp,
                                         vp, vp, vp, P(BwdScratch), vp]
    lib.gcpnet_gcp2_chain_backward.argtypes = [i32, vp, i32, P(ChainBwdItem), vp, vp, vp, vp, vp]
    lib.gcpnet_gcp2_chain_backward_gathered.argtypes = [i32, vp, i32, P(ChainBwdItem), vp, vp, vp, vp, vp, vp, vp]
    lib.gcpnet_gcp2_chain_backward_split.argtypes = [i32, vp, i32, P(ChainBwdItem), vp, vp, vp, vp, vp, vp, vp, i32, vp]
    lib.gcpnet_gcp2_chain_backward_flags.argtypes = [i32] * 8
    lib.gcpnet_debug_force_chain_split.argtypes = [i32, i32, i32]
    lib.gcpnet_debug_force_chain_split.restype = None
    lib.gcpnet_tn_gemm.argtypes = [i32, P(TnProblem), vp]
    lib.gcpnet_tn_splits.argtypes = [i32, i32, i32]
    lib.gcpnet_gcp2_bwd_tiles.argtypes = [i32]
    lib.gcpnet_reduce_partials.argtypes = [i32, P(ReduceJob), vp]
    lib.gcpnet_reduce_partials_groups.argtypes = [i32]
    lib.gcpnet_gcp2_weight_grads.argtypes = [i32, P(WgradJob), vp, vp]
    lib.gcpnet_gcp2_weight_grads_workspace.argtypes = [i32, P(WgradJob)]
    lib.gcpnet_gcp2_weight_grads_workspace.restype = i64
    lib.gcpnet_stream_wait_stream.argtypes = [vp, vp]
    lib.gcpnet_segment_reduce.argtypes = [i32, vp, vp, vp, i64, i32, i32, vp, i64, i32, vp]
    lib.gcpnet_gather_rows.argtypes = [i32, vp, vp, i64, i32, vp, vp, i64, vp]
    lib.gcpnet_localize.argtypes = [i32, vp, vp, vp, i32, vp, vp]
    lib.gcpnet_layernorm_forward.argtypes = [i32, i32, i32] + [vp] * 12
    lib.gcpnet_layernorm_backward.argtypes = [i32, i32, i32] + [vp] * 11
    lib.gcpnet_layernorm_bwd_scratch_floats.argtypes = [i32, i32]
    lib.gcpnet_layernorm_bwd_scratch_floats.restype = i64
    lib.gcpnet_axpy_clamp.argtypes = [i64, vp, vp, f32, i32, f32, f32, vp, vp]
    lib.gcpnet_rows_matmul_small.argtypes = [i64, i32, i32, vp, i64, vp, vp, i64, vp]
    lib.gcpnet_edge_force_forward.argtypes = [i64, i32, vp, vp, vp, vp, vp, vp, i32, f32, vp, vp]
    lib.gcpnet_edge_force_backward.argtypes = [i64, i32, vp, vp, vp, vp, vp, vp, i32, f32, vp, vp, vp, vp]
    lib.gcpnet_edge_force_bwd_blocks.argtypes = [i64]
    lib.gcpnet_row_gate_forward.argtypes = [i64, i32, vp, vp, vp, vp, vp, vp]
    lib.gcpnet_row_gate_backward.argtypes = [i64, i32, vp, vp, vp, vp, vp, vp, vp]
    lib.gcpnet_row_gate_bwd_blocks.argtypes = [i64]
    lib.gcpnet_debug_set_phase_timing.argtypes = [vp, i64]
    lib.gcpnet_debug_set_fp32_mfma.argtypes = [i32]
    lib.gcpnet_debug_tn_occupancy.argtypes = [i32]
    lib.gcpnet_gcp2_chain_forward_registers_ok.argtypes = [i32] * 6
    lib.gcpnet_gcp2_chain_backward_ok.argtypes = [i32] * 6
    lib.gcpnet_tb_floats.argtypes = [i32, i32]
    lib.gcpnet_tb_floats.restype = i64
    lib.gcpnet_tb_sign_words.argtypes = [i32, i32]
    lib.gcpnet_tb_sign_words.restype = i64
    lib.gcpnet_wg_pack_floats.restype = i64
    lib.gcpnet_wg_pack_floats.argtypes = [i32] * 7
    lib.gcpnet_wg_pack.argtypes = [P(Gcp2Weights), i32, vp, vp]
    lib.gcpnet_wg_pack_multi.argtypes = [i32, P(WgPackJob), vp]
    lib.gcpnet_wg_pack_view.argtypes = [P(Gcp2Weights), i32, vp, i32, i32, i32, P(i32), P(i32), vp, vp]
    lib.gcpnet_wg_forward.argtypes = [i32, vp, vp, vp, P(Concat), P(Concat), i32, P(WgBlock), vp]
    lib.gcpnet_wg_backward_plan.argtypes = [i32, P(Gcp2Weights), P(Gcp2Opts), i32, P(WgBwdPlan)]
    lib.gcpnet_wg_backward.argtypes = [i32, P(WgBwdArgs), vp]
    lib.gcpnet_wg_reduce.argtypes = [vp, i32, i32, i32, i32, vp, vp, vp]
    lib.gcpnet_wg_reduce_multi.argtypes = [i32, vp, vp]
    lib.gcpnet_dropout.argtypes = [i64, i32, vp, f32, C.c_uint64, vp, vp]
    lib.gcpnet_adam_step.argtypes = [i32, P(AdamTensor), f32, f32, f32, f32, f32, i32, vp]
    lib.gcpnet_copy2d_multi.argtypes = [i32, P(Copy2dJob), vp]
    lib.gcpnet_axpy_clamp_backward.argtypes = [i64, vp, vp, f32, i32, f32, f32, vp, vp]
    lib.gcpnet_adam_step_dev.argtypes = [i32, P(AdamTensor), f32, f32, f32, f32, f32, vp, vp]
    lib.gcpnet_nms_edge_features.argtypes = [i64, vp, vp, vp, vp, i32, f32, i32, vp, vp, vp]
    lib.gcpnet_nms_node_features.argtypes = [i64, vp, vp, vp, vp, vp, vp]
    lib.gcpnet_orientations.argtypes = [i64, vp, vp, vp, vp]
    lib.gcpnet_activation.argtypes = [i64, vp, vp, i32, f32, vp, vp]
    lib.gcpnet_frame_gate_forward.argtypes = [i64, i32, vp, i32, vp, vp, vp, i32, f32, vp, vp]
    lib.gcpnet_frame_gate_backward.argtypes = [i64, i32, vp, i32, vp, vp, vp, i32, f32, vp, vp, vp, vp, vp]
    lib.gcpnet_frame_gate_bwd_parts.argtypes = [i64]
    lib.gcpnet_node_scalarize.argtypes = [i32, vp, vp, vp, i32, vp, i32, vp, vp, vp, vp]
    lib.gcpnet_radius_graph.argtypes = [i32, vp, vp, vp, vp, i32, i32, i32, C.c_double, i32, vp, vp, vp]
    lib.gcpnet_radius_graph_first.argtypes = [i32, vp, vp, vp, vp, i32, i32, i32, C.c_double, i32, vp, vp, vp]
    for name in EXPORTS:
        fn = getattr(lib, name)
        if name not in ("gcpnet_gcp2_pack_floats", "gcpnet_layernorm_bwd_scratch_floats", "gcpnet_gcp2_forward_lds_bytes",
                        "gcpnet_wg_pack_floats", "gcpnet_tb_floats", "gcpnet_tb_sign_words", "gcpnet_gcp2_weight_grads_workspace"):
            fn.restype = i32
    if lib.gcpnet_abi_version() != 4:
        raise GcpnetHipError("libgcpnet_hip.so ABI version mismatch")
    _lib = lib
    return lib


E_BADARG, E_UNSUPPORTED = -1, -2  # GCPNET_E_* of include/gcpnet_hip.h


def check(rc, what):
    if rc != 0:
        kind = {-1: "invalid argument", -2: "unsupported shape"}.get(rc, f"hipError {rc}")
        raise GcpnetHipError(f"{what} failed: {kind}")
