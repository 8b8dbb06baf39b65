"""Operators of the hot path: thin autograd wrappers that launch the HIP kernels through the C ABI.

Every function here requires CUDA(HIP) tensors and libgcpnet_hip.so; nothing falls back to PyTorch arithmetic.
PyTorch provides device allocations, the current stream and the autograd graph -- plumbing only.
"""
from __future__ import annotations

import os
import sys

import ctypes as C
import weakref
from dataclasses import dataclass, field, replace
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import (ACT, WgBwdArgs, WgBwdPlan, BwdScratch, ChainBwdItem, ChainItem, Head, Concat, Gcp2Opts, Gcp2Weights, Operand, ReduceJob, TnProblem, WgradJob, VMODE_NONE, VMODE_SCALAR_GATE,
                   VMODE_SELF_GATE, WgBlock, check)

Tensor = torch.Tensor


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """The caller's current HIP stream as a raw handle.  `torch.cuda.current_stream()` builds a Stream object through several Python
    layers (11 us per call, ~70 calls per layer and step: 5 % of a launch-bound NMS step, tools/pyprofile_step.py); the raw getter
    answers in well under a microsecond."""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t: Optional[Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _req(t: Tensor, name: str) -> Tensor:
    if not t.is_cuda:
        raise _lib.GcpnetHipError(f"{name} must live on the GPU: gcpnet_amd has no CPU path")
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


# ==============================================================================================================
# graph plans (host-side preprocessing; sorting/counting is data preparation)
# ==============================================================================================================
class GatherPlan:
    """Index structure for `out[r] = src[idx[r]]` and its adjoint (segmented sum over rows sharing idx)."""

    def __init__(self, idx: Tensor, n_src: int):
        idx64 = idx.long()
        self.n_src = int(n_src)
        self.rows = int(idx64.shape[0])
        self.idx = idx64.to(torch.int32).contiguous()
        if self.rows > 1 and not bool((idx64[1:] >= idx64[:-1]).all()):
            order = torch.argsort(idx64, stable=True)
            self.perm: Optional[Tensor] = order.to(torch.int32).contiguous()
        else:
            self.perm = None
        counts = torch.bincount(idx64, minlength=self.n_src)
        ptr = torch.zeros(self.n_src + 1, dtype=torch.int64, device=idx.device)
        ptr[1:] = torch.cumsum(counts, 0)
        self.seg_ptr = ptr.to(torch.int32).contiguous()
        self.inv_count = (1.0 / counts.clamp(min=1).to(torch.float32)).contiguous()

    _cache: dict = {}

    @classmethod
    def get(cls, idx: Tensor, n_src: Optional[int] = None) -> "GatherPlan":
        """Plan for an index tensor that is reused from step to step (the `batch` vector of a collated batch), cached on the
        tensor object: building one costs a sort, a bincount and two host synchronisations (`max`, the sortedness test) -- per
        call that is most of a small model's step, and it cannot happen inside a captured hipGraph.  n_src = idx.max() + 1 when
        not given (torch_scatter's dim_size default)."""
        import weakref

        key = (idx.data_ptr(), tuple(idx.shape), idx._version, n_src, str(idx.device))
        hit = cls._cache.get(key)
        if hit is not None and hit[0]() is idx:
            return hit[1]
        n = n_src if n_src is not None else (int(idx.max()) + 1 if idx.numel() else 0)
        plan = cls(idx, n)
        if len(cls._cache) > 64:
            cls._cache.clear()
        cls._cache[key] = (weakref.ref(idx), plan)
        return plan


class GraphPlan:
    """Per-batch preprocessing of `edge_index` ([2, E], row = source, col = target), cached on the caller's
    tensor: int32 copies, CSR by col (aggregation), CSR by row (node scalarize), and the per-node mean frame."""

    _cache: dict = {}

    def __init__(self, edge_index: Tensor, n_nodes: int):
        self.n_nodes, self.n_edges = int(n_nodes), int(edge_index.shape[1])
        self.row = GatherPlan(edge_index[0], n_nodes)
        self.col = GatherPlan(edge_index[1], n_nodes)
        self._fbar_key = None
        self._fbar = None

    @classmethod
    def get(cls, edge_index: Tensor, n_nodes: int) -> "GraphPlan":
        key = (edge_index.data_ptr(), tuple(edge_index.shape), int(n_nodes), edge_index._version, str(edge_index.device))
        hit = cls._cache.get(key)
        if hit is not None and hit[0]() is edge_index:
            return hit[1]
        import weakref

        plan = cls(edge_index, n_nodes)
        if len(cls._cache) > 64:
            cls._cache.clear()
        cls._cache[key] = (weakref.ref(edge_index), plan)
        return plan

    def node_frames(self, frames: Tensor) -> Tensor:
        """Mean frame over each node's out-edges (row == node); zero for nodes without out-edges.
        scalarize(node_inputs=True) (components/__init__.py:286,302,314-323) is linear in the frame, so the
        gather-by-row / project / scatter-mean-by-row sequence equals one projection onto this mean frame."""
        import weakref

        # keyed on the tensor OBJECT (weak reference) and its version: a frames tensor that was dropped and recomputed may
        # come back at the same address (caching allocator), and tensors filled through a raw pointer keep version 0
        hit = self._fbar_key
        if hit is None or hit[0]() is not frames or hit[1] != frames._version:
            with torch.no_grad():
                flat = _req(frames, "frames").reshape(self.n_edges, 9)
                self._fbar = segment_reduce(flat, self.row, mean=True).reshape(self.n_nodes, 3, 3)
            self._fbar_key = (weakref.ref(frames), frames._version)
        return self._fbar


# ==============================================================================================================
# segment reduce / gather
# ==============================================================================================================
def _segment_reduce_raw(x: Tensor, col0: int, D: int, ld: int, plan: GatherPlan, mean: bool) -> Tensor:
    lib = _lib.load()
    if plan.rows == 0:
        return torch.zeros((plan.n_src, D), dtype=torch.float32, device=x.device)
    out = torch.empty((plan.n_src, D), dtype=torch.float32, device=x.device)
    check(lib.gcpnet_segment_reduce(plan.n_src, _p(plan.seg_ptr), _p(plan.perm), C.c_void_p(x.data_ptr() + 4 * col0), ld,
                                    D, int(mean), _p(out), D, 0, _stream()), "segment_reduce")
    return out


def _gather_rows_raw(x: Tensor, plan: GatherPlan, scale: Optional[Tensor]) -> Tensor:
    lib = _lib.load()
    D = x.shape[1]
    out = torch.empty((plan.rows, D), dtype=torch.float32, device=x.device)
    if plan.rows == 0:
        return out
    check(lib.gcpnet_gather_rows(plan.rows, _p(plan.idx), _p(x), D, D, _p(scale), _p(out), D, _stream()), "gather_rows")
    return out


class _SegmentReduce(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, plan, mean):
        ctx.plan, ctx.mean = plan, mean
        return _segment_reduce_raw(x, 0, x.shape[1], x.shape[1], plan, mean)

    @staticmethod
    def backward(ctx, g):
        g = _req(g, "grad")
        return _gather_rows_raw(g, ctx.plan, ctx.plan.inv_count if ctx.mean else None), None, None


class _GatherRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, plan):
        ctx.plan = plan
        return _gather_rows_raw(x, plan, None)

    @staticmethod
    def backward(ctx, g):
        g = _req(g, "grad")
        return _segment_reduce_raw(g, 0, g.shape[1], g.shape[1], ctx.plan, False), None


def segment_reduce(x: Tensor, plan: GatherPlan, mean: bool) -> Tensor:
    """torch_scatter.scatter(x, plan.idx, dim=0, dim_size=plan.n_src, reduce='mean'|'sum') for 2-D x."""
    x = _req(x, "x")
    assert x.dim() == 2 and x.shape[0] == plan.rows
    return _SegmentReduce.apply(x, plan, bool(mean))


def gather_rows(x: Tensor, plan: GatherPlan) -> Tensor:
    x = _req(x, "x")
    assert x.dim() == 2 and x.shape[0] == plan.n_src
    return _GatherRows.apply(x, plan)


# ==============================================================================================================
# frames
# ==============================================================================================================
def localize(x: Tensor, plan: GraphPlan, norm_x_diff: bool = True) -> Tensor:
    lib = _lib.load()
    if x.requires_grad and torch.is_grad_enabled():
        # the reference's frames are differentiable w.r.t. the positions; every shipped pipeline builds them from data
        # (no gradient), and they are constants of the step here
        raise NotImplementedError("gcpnet_amd.localize: frames are constants of the step; positions that require grad are "
                                  "not differentiated through (call it under torch.no_grad() or on x.detach())")
    x = _req(x.detach(), "x")
    frames = torch.empty((plan.n_edges, 3, 3), dtype=torch.float32, device=x.device)
    check(lib.gcpnet_localize(plan.n_edges, _p(plan.row.idx), _p(plan.col.idx), _p(x), int(norm_x_diff), _p(frames),
                              _stream()), "localize")
    return frames


# ==============================================================================================================
# GCPLayerNorm (+ residual)
# ==============================================================================================================
class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, s_a, s_b, v_a, v_b, gamma, beta):
        lib = _lib.load()
        rows, sdim = s_a.shape
        vdim = 0 if v_a is None else v_a.shape[1]
        s_out, s_sum = torch.empty_like(s_a), torch.empty_like(s_a)
        v_out = torch.empty_like(v_a) if vdim else None
        v_sum = torch.empty_like(v_a) if vdim else None
        stats = torch.empty((rows, 3), dtype=torch.float32, device=s_a.device)
        check(lib.gcpnet_layernorm_forward(rows, sdim, vdim, _p(s_a), _p(s_b), _p(v_a), _p(v_b), _p(gamma), _p(beta),
                                           _p(s_out), _p(v_out), _p(stats), _p(s_sum), _p(v_sum), _stream()), "layernorm")
        ctx.save_for_backward(s_sum, v_sum, stats, gamma)
        ctx.has_b = (s_b is not None, v_b is not None)
        ctx.vdim = vdim
        if vdim:
            return s_out, v_out
        ctx.mark_non_differentiable()
        return s_out, None

    @staticmethod
    def backward(ctx, d_s, d_v):
        lib = _lib.load()
        s_sum, v_sum, stats, gamma = ctx.saved_tensors
        rows, sdim = s_sum.shape
        vdim = ctx.vdim
        d_s = _req(d_s, "grad") if d_s is not None else torch.zeros_like(s_sum)
        if vdim:
            d_v = _req(d_v, "grad") if d_v is not None else torch.zeros_like(v_sum)
        gs = torch.empty_like(s_sum)
        gv = torch.empty_like(v_sum) if vdim else None
        gb = torch.empty((2, sdim), dtype=torch.float32, device=gamma.device)  # d gamma, d beta
        scratch = torch.empty((int(lib.gcpnet_layernorm_bwd_scratch_floats(rows, sdim)),), dtype=torch.float32, device=gamma.device)
        check(lib.gcpnet_layernorm_backward(rows, sdim, vdim, _p(s_sum), _p(v_sum), _p(stats), _p(gamma), _p(d_s),
                                            _p(d_v) if vdim else None, _p(gs), _p(gv), _p(gb), _p(scratch), _stream()),
              "layernorm_backward")
        return gs, (gs if ctx.has_b[0] else None), gv, (gv if ctx.has_b[1] else None), gb[0], gb[1]


def layernorm(s_a: Tensor, v_a: Optional[Tensor], gamma: Tensor, beta: Tensor, s_b: Optional[Tensor] = None,
              v_b: Optional[Tensor] = None) -> Tuple[Tensor, Optional[Tensor]]:
    """GCPLayerNorm((s_a + s_b), (v_a + v_b)) -- components/__init__.py:138-167 with the residual add fused in."""
    s_a = _req(s_a, "s")
    s_b = _req(s_b, "s_b") if s_b is not None else None
    if v_a is not None and v_a.shape[1] == 0:
        v_a, v_b = None, None
    v_a = _req(v_a, "v") if v_a is not None else None
    v_b = _req(v_b, "v_b") if v_b is not None else None
    return _LayerNorm.apply(s_a, s_b, v_a, v_b, _req(gamma, "gamma"), _req(beta, "beta"))


# ==============================================================================================================
# position update:  y = a + clamp(alpha * b, lo, hi)   (components/gcpnet.py:1156-1158,1258)
# ==============================================================================================================
class _AxpyClamp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, alpha, clamp, lo, hi):
        lib = _lib.load()
        y = torch.empty_like(b)
        check(lib.gcpnet_axpy_clamp(b.numel(), _p(a), _p(b), float(alpha), int(clamp), float(lo), float(hi), _p(y),
                                    _stream()), "axpy_clamp")
        ctx.save_for_backward(b)
        ctx.cfg = (alpha, clamp, lo, hi)
        return y

    @staticmethod
    def backward(ctx, g):
        (b,) = ctx.saved_tensors
        alpha, clamp, lo, hi = ctx.cfg
        g = _req(g, "grad")
        gb = torch.empty_like(b)
        check(_lib.load().gcpnet_axpy_clamp_backward(b.numel(), _p(g), _p(b), float(alpha), int(clamp), float(lo), float(hi), _p(gb),
                                                     _stream()), "axpy_clamp_backward")
        return g, gb, None, None, None, None


def axpy_clamp(a: Tensor, b: Tensor, alpha: float, lo: float = -100.0, hi: float = 100.0) -> Tensor:
    """a + clamp(alpha * b, lo, hi)"""
    return _AxpyClamp.apply(_req(a, "a"), _req(b, "b"), alpha, True, lo, hi)


def axpy(a: Tensor, b: Tensor, alpha: float) -> Tensor:
    """a + alpha * b"""
    return _AxpyClamp.apply(_req(a, "a"), _req(b, "b"), alpha, False, 0.0, 0.0)


# ==============================================================================================================
# dropout (components/__init__.py:97-135)
# ==============================================================================================================
class _Dropout(torch.autograd.Function):
    """y = x * mask / keep with one Bernoulli(keep) draw per `group` consecutive floats (1: nn.Dropout, 3: VectorDropout); the mask
    is a counter-based hash of (seed, index), recomputed -- not stored -- in the backward."""

    @staticmethod
    def forward(ctx, x, keep: float, group: int, seed: int):
        lib = _lib.load()
        y = torch.empty_like(x)
        check(lib.gcpnet_dropout(x.numel() // group, group, _p(x), float(keep), C.c_uint64(seed), _p(y), _stream()), "dropout")
        ctx.cfg = (float(keep), int(group), int(seed))
        return y

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        keep, group, seed = ctx.cfg
        g = _req(g, "grad")
        dx = torch.empty_like(g)
        check(lib.gcpnet_dropout(g.numel() // group, group, _p(g), keep, C.c_uint64(seed), _p(dx), _stream()), "dropout")
        return dx, None, None, None


def dropout(x: Tensor, drop_rate: float, group: int = 1, seed: Optional[int] = None) -> Tensor:
    """Train-mode dropout of `x` (scaled by 1 / (1 - drop_rate)); group = 3 drops whole 3-vectors of a [..., 3] tensor.  The seed
    defaults to a draw from torch's CPU generator (reproducible under torch.manual_seed, no device synchronisation)."""
    x = _req(x, "x")
    if x.numel() == 0:
        return x
    assert x.numel() % group == 0
    if seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
    return _Dropout.apply(x, 1.0 - float(drop_rate), int(group), int(seed))


# ==============================================================================================================
# GCP2
# ==============================================================================================================
@dataclass
class Gcp2Spec:
    """Static description of one GCP2 application (dims, options, gather plans of the concatenated inputs)."""
    si: int
    vi: int
    so: int
    vo: int
    hidden: int
    use_frames: bool
    act_s: Optional[str]
    act_v: Optional[str]
    slope: float
    vmode: int
    vector_residual: bool
    e3: bool
    s_plans: List[Optional[GatherPlan]] = field(default_factory=list)
    v_plans: List[Optional[GatherPlan]] = field(default_factory=list)
    residual: bool = False  # out = x + GCP(x) with x the single ungathered source (ResGCP)
    pack_cache: Optional[dict] = None
    # pre-projected scalar inputs ("project, then gather"): one gather plan (or None) per [n_src, so] table passed to _Gcp2
    add_plans: List[Optional[GatherPlan]] = field(default_factory=list)
    # same for vector inputs: one plan per [n_src, 3, HF'] table ([vector_down ; vector_down_frames] applied at the source rows)
    vadd_plans: List[Optional[GatherPlan]] = field(default_factory=list)
    # some weight of this block is also an input of ANOTHER autograd Function in the same graph (GCP3 feedforward_out runs the
    # block as two launches sharing vector_down / vector_down_frames): autograd then sums two gradients inside the backward
    # pass, so this block's weight gradients must be complete on the caller's stream when its backward returns
    shared_weights: bool = False
    # scalar_out's weight as a VIEW of a wider stored matrix: (W_full [so, ld], [(first column, columns), ...] up to 3 ranges
    # adding up to K).  The workgroup kernels pack straight from it (gcpnet_wg_pack_view); weights[0] may then be None.
    w_view: Optional[tuple] = None

    @property
    def K(self):
        return self.si + ((self.hidden + (9 if self.use_frames else 0)) if self.vi > 0 else 0)


def _concat(tensors: Sequence[Tensor], plans: Sequence[Optional[GatherPlan]], vec: bool) -> Concat:
    c = Concat()
    c.n = len(tensors)
    for k, (t, pl) in enumerate(zip(tensors, plans)):
        c.ptr[k] = t.data_ptr()
        c.idx[k] = pl.idx.data_ptr() if pl is not None else None
        c.dim[k] = t.shape[1]
    return c


def _weights_struct(spec: Gcp2Spec, w, pack: Tensor) -> Gcp2Weights:
    w_scalar, b_scalar, w_down, w_frames, w_up, w_gate, b_gate = w
    ws = Gcp2Weights()
    ws.si, ws.vi, ws.so, ws.vo, ws.hidden, ws.use_frames = spec.si, spec.vi, spec.so, spec.vo, spec.hidden, int(spec.use_frames)
    ws.w_down, ws.w_frames, ws.w_up = (t.data_ptr() if t is not None else None for t in (w_down, w_frames, w_up))
    ws.w_scalar, ws.b_scalar = (w_scalar.data_ptr() if w_scalar is not None else None), b_scalar.data_ptr()
    ws.w_gate = w_gate.data_ptr() if w_gate is not None else None
    ws.b_gate = b_gate.data_ptr() if b_gate is not None else None
    ws.pack = pack.data_ptr() if pack is not None else None
    return ws


def _opts_struct(spec: Gcp2Spec, fused_residual: bool = False) -> Gcp2Opts:
    o = Gcp2Opts()
    o.act_s, o.act_v, o.slope = ACT[spec.act_s], ACT[spec.act_v], float(spec.slope)
    o.vmode, o.vector_residual, o.e3 = spec.vmode, int(spec.vector_residual), int(spec.e3)
    o.fused_residual = int(fused_residual)
    return o


_PACK_EPOCH = 0
# Which epoch the uses of the step before the current one fell into (the batched re-pack of _pack_wg lets only blocks that were
# used then ride along).  An epoch may advance more than once per optimizer step (FusedAdam plus an EMA swap): the first use after
# any number of invalidations closes the previous "use epoch".
_LAST_USE_EPOCH = 0
_PREV_USE_EPOCH = 0


def _note_pack_use() -> None:
    global _LAST_USE_EPOCH, _PREV_USE_EPOCH
    if _LAST_USE_EPOCH != _PACK_EPOCH:
        _PREV_USE_EPOCH, _LAST_USE_EPOCH = _LAST_USE_EPOCH, _PACK_EPOCH


def _dense_weights(spec: Gcp2Spec, w):
    """`w` with scalar_out's weight materialised when the spec only carries a view of it (the wave-per-tile kernels and the
    TN-GEMM helpers want a contiguous matrix)."""
    if w[0] is not None:
        return w
    W, segs = spec.w_view
    return (torch.cat([W[:, a:a + n] for a, n in segs], dim=1),) + tuple(w[1:])


def invalidate_packs() -> None:
    """Forces every packed-weight image to be rebuilt at its next use.  The caches are keyed on (data_ptr, _version) of the
    weights, which in-place updates through `p.data` (legacy optimizers, EMA / SWA weight swaps, `p.data.copy_`) do not
    change: code that updates weights that way calls this after the update."""
    global _PACK_EPOCH
    _PACK_EPOCH += 1


F16_WEIGHT_LIMIT = 65504.0 / 64.0  # |w| above this saturates in the two-term fp16 weight images (csrc/gcp_f16x2.h: 2^6 w must fit fp16)


def check_weight_range(module: torch.nn.Module) -> float:
    """The largest |w| over the `scalar_out` weights of a model's GCP blocks; raises GcpnetHipError if one is outside what the
    two-term fp16 images hold (|w| < 1023.5: the pack kernels SATURATE beyond it -- finite, but wrong -- because they cannot report).
    Not on any hot path (one reduction and one host read per call): for a trainer's sanity hooks, after loading a checkpoint."""
    worst = 0.0
    for name, p in module.named_parameters():
        if name.endswith("scalar_out.weight") or ".scalar_out." in name and name.endswith(".weight"):
            worst = max(worst, float(p.detach().abs().max()))
    if not worst < F16_WEIGHT_LIMIT:  # (also catches NaN)
        raise _lib.GcpnetHipError(f"scalar_out weight magnitude {worst:.4g} is outside the two-term fp16 weight images' range "
                                  f"(< {F16_WEIGHT_LIMIT:.1f}); build with -DGCP_ARITH_F16X2=0 for the bf16 form")
    return worst


def _pack(spec: Gcp2Spec, w) -> Tensor:
    lib = _lib.load()
    w = _dense_weights(spec, w)
    w_scalar, w_gate = w[0], w[5]
    key = (_PACK_EPOCH,) + tuple(None if t is None else (t.data_ptr(), t._version) for t in (w_scalar, w_gate, w[2], w[3], w[4]))
    cache = spec.pack_cache
    if cache is not None and cache.get("key") == key:
        return cache["pack"]
    n = lib.gcpnet_gcp2_pack_floats(spec.si, spec.vi, spec.so, spec.vo, spec.hidden, int(spec.use_frames))
    pack = torch.empty(int(n), dtype=torch.float32, device=w_scalar.device)
    ws = _weights_struct(spec, w, pack)
    check(lib.gcpnet_pack_gcp2_weights(C.byref(ws), _p(pack), _stream()), "pack_gcp2_weights")
    if cache is not None:
        cache["key"], cache["pack"] = key, pack
    return pack


def _pack_many(specs, ws) -> List[Tensor]:
    """_pack for the blocks of a chain (one shape): the images that are not cached are built by ONE launch."""
    lib = _lib.load()
    out: List[Optional[Tensor]] = [None] * len(specs)
    todo = []
    for k, (spec, w) in enumerate(zip(specs, ws)):
        w = _dense_weights(spec, w)
        key = (_PACK_EPOCH,) + tuple(None if t is None else (t.data_ptr(), t._version) for t in (w[0], w[5], w[2], w[3], w[4]))
        cache = spec.pack_cache
        if cache is not None and cache.get("key") == key:
            out[k] = cache["pack"]
        else:
            todo.append((k, spec, w, key))
    if len(todo) == 1:
        out[todo[0][0]] = _pack(todo[0][1], ws[todo[0][0]])
    elif todo:
        sp0 = todo[0][1]
        n = int(lib.gcpnet_gcp2_pack_floats(sp0.si, sp0.vi, sp0.so, sp0.vo, sp0.hidden, int(sp0.use_frames)))
        structs = (Gcp2Weights * len(todo))()
        ptrs = (C.c_void_p * len(todo))()
        for j, (k, spec, w, key) in enumerate(todo):
            assert (spec.si, spec.vi, spec.so, spec.vo, spec.hidden, spec.use_frames) == (sp0.si, sp0.vi, sp0.so, sp0.vo, sp0.hidden, sp0.use_frames)
            pack = torch.empty(n, dtype=torch.float32, device=w[0].device)
            structs[j] = _weights_struct(spec, w, pack)
            ptrs[j] = pack.data_ptr()
            out[k] = pack
            if spec.pack_cache is not None:
                spec.pack_cache["key"], spec.pack_cache["pack"] = key, pack
        check(lib.gcpnet_pack_gcp2_weights_multi(len(todo), structs, ptrs, _stream()), "pack_gcp2_weights_multi")
    return out


USE_WG_KERNELS = True  # module switch: multi-wave workgroup kernels (gcp_wg_*.hip) where the shape fits, else the wave-per-tile ones
USE_WG_BACKWARD = True  # (separately for the backward; both need USE_WG_KERNELS)
PREFER_WG_CHAIN_BACKWARD = True  # chains wider than 128 scalars: block by block through the workgroup backward kernel
# ResGCP chains with so <= 128: the register-resident wave-per-tile forward kernel (bf16 x 6 form; 0.62 ms per 7-block launch at
# (128,16) on 160 k rows against 0.73 ms for the workgroup forward) where its shape test passes; GCPNET_CHAIN_FWD=wg switches back
PREFER_WAVE_CHAIN_FORWARD = os.environ.get("GCPNET_CHAIN_FWD", "wave") == "wave"
# The message aggregation inside the ResGCP chain's autograd Function (forward: the same two reductions; backward: the chain backward
# kernel reads the node-level gradient tables through the edge -> node index, no [E, .] gradient copies).  GCPNET_FUSE_AGG=0: off.
FUSE_AGGREGATION = os.environ.get("GCPNET_FUSE_AGG", "1") != "0"
FORCE_WG_CHAIN_BACKWARD = False  # tests: the workgroup backward also for chains the wave-per-tile chain kernel covers
# Tensors that only the register-resident chain kernels and the weight-gradient GEMM exchange -- s_pre, ds_pre and the intermediate
# scalar states of a ResGCP chain -- in the tile-blocked layout (include/gcpnet_hip.h, gcp2_chain_item_t): every wave instruction on
# them moves 1 KB of whole lines, no LDS transposition.  GCPNET_CHAIN_TB=0: row-major as before (A/B measurements, tests).
CHAIN_TILE_BLOCKED = os.environ.get("GCPNET_CHAIN_TB", "1") != "0"
# (the same for the s_pre of a SINGLE block that runs in the workgroup kernels both ways: first message GCP, feed-forward GCPs)
BLOCK_TILE_BLOCKED = os.environ.get("GCPNET_BLOCK_TB", "1") != "0"
# Tail split of the wave-per-tile chain backward (include/gcpnet_hip.h, gcpnet_gcp2_chain_backward_split): the chains of some tiles
# run as two workgroups so that a partly filled last round of waves does not cost a whole one.  Bit-identical results
# (tests/test_chain_split.py), but OFF by default: measured (round 6, tools/chain_rows_sweep.py) the partial round of 902 tiles of
# configs[1] costs 0.15 - 0.17 ms against 0.125 ms for the same tiles at the full-round rate -- lone waves run 1.4 x faster than
# paired ones -- and the second workgroups' prologues eat what is left: 0.750 ms split against 0.744 ms.  GCPNET_CHAIN_SPLIT=1: on.
CHAIN_TAIL_SPLIT = os.environ.get("GCPNET_CHAIN_SPLIT", "0") == "1"
# Sign masks of s_pre (include/gcpnet_hip.h, gcp2_chain_item_t.s_sign): the register-resident chain forward writes, beside s_pre, one
# bit per element -- where it is positive -- and the chain backward kernel reads that instead of s_pre when every activation of the
# chain is piecewise linear (all its derivative depends on).  GCPNET_CHAIN_SIGN=0: s_pre itself as before (A/B; bit-identical results).
CHAIN_SIGN_MASKS = os.environ.get("GCPNET_CHAIN_SIGN", "1") != "0"
# The memory route (SURVEY.md section 7 step 6 / section 8d's byte model: "message chain recomputed in backward, no saved per-edge
# activations"): a ResGCP chain keeps only its inputs for the backward, which runs the chain's forward again to get s_pre / gates /
# the intermediate states back -- 1.4 GB instead of 20.8 GB per layer at configs[4] size for one more chain forward per layer and step.
# Same launches on the same inputs: gradients equal the plain route's bit for bit.  GCPNET_CHAIN_RECOMPUTE=1 or ops.CHAIN_RECOMPUTE.
CHAIN_RECOMPUTE = os.environ.get("GCPNET_CHAIN_RECOMPUTE", "0") == "1"
# d vector_out_scale.weight from the block's INPUTS instead of s_pre when the gate's activation is the identity (include/gcpnet_hip.h,
# gcp2_wgrad_job_t.gate_lin), and -- with the sign masks -- a chain forward that does not store s_pre at all (CHAIN_SKIP_S_PRE): 512 of
# the 1 280 bytes a block writes per row at width 128.  GCPNET_GATE_FROM_INPUTS=0 / GCPNET_CHAIN_SKIP_S_PRE=0: as before (A/B; the gate
# weight gradients then differ by fp32 round-off, everything else is bit-identical).
GATE_GRADS_FROM_INPUTS = os.environ.get("GCPNET_GATE_FROM_INPUTS", "1") != "0"
CHAIN_SKIP_S_PRE = os.environ.get("GCPNET_CHAIN_SKIP_S_PRE", "1") != "0"
_PWL_ACTS = (None, "relu", "leakyrelu")


def copy2d_multi(jobs) -> None:
    """jobs: (dst, src) pairs of <= 2-D fp32 views on the current device (any strides; src None = zero fill; shapes equal): ONE launch
    (gcpnet_copy2d_multi) instead of one ATen cat / pad / clone / copy_ each."""
    jobs = [(d, s_) for d, s_ in jobs if d.numel() > 0]
    if not jobs:
        return
    arr = (_lib.Copy2dJob * len(jobs))()
    for k, (d, s_) in enumerate(jobs):
        if d.dim() == 1:
            d = d.unsqueeze(0)
        if s_ is not None and s_.dim() == 1:
            s_ = s_.unsqueeze(0)
        assert d.dim() == 2 and d.dtype == torch.float32 and d.is_cuda and (s_ is None or (s_.shape == d.shape and s_.dtype == torch.float32))
        arr[k].dst, arr[k].rows, arr[k].cols = d.data_ptr(), d.shape[0], d.shape[1]
        arr[k].dst_rs, arr[k].dst_cs = d.stride(0), d.stride(1)
        if s_ is not None:
            arr[k].src, arr[k].src_rs, arr[k].src_cs = s_.data_ptr(), s_.stride(0), s_.stride(1)
    check(_lib.load().gcpnet_copy2d_multi(len(jobs), arr, _stream()), "copy2d_multi")


class TileBlocked:
    """A [rows, width] fp32 matrix in the tile-blocked layout: gcpnet_tb_floats(rows, width) floats, its own allocation or (`owner`,
    `offset` in floats) a region of a flat one."""
    __slots__ = ("rows", "width", "ptr", "_n", "_data", "_owner", "_off", "sign", "absent")

    def __init__(self, rows: int, width: int, device, owner: Optional[Tensor] = None, offset: int = 0, n: Optional[int] = None):
        self.rows, self.width = rows, width
        self.sign = None  # address of the sign mask written beside an s_pre (include/gcpnet_hip.h, gcp2_chain_item_t.s_sign)
        self.absent = False  # an s_pre the forward did not store (CHAIN_SKIP_S_PRE): only its sign mask exists, `ptr` is a placeholder
        self._n = int(n) if n is not None else int(_lib.load().gcpnet_tb_floats(rows, width))
        if owner is None:
            self._data = torch.empty((self._n,), dtype=torch.float32, device=device)
            self._owner, self._off, self.ptr = None, 0, self._data.data_ptr()
        else:
            self._data, self._owner, self._off, self.ptr = None, owner, offset, owner.data_ptr() + 4 * offset

    def data_ptr(self) -> int:
        return self.ptr

    @property
    def data(self) -> Tensor:
        if self._data is None:
            self._data = self._owner[self._off:self._off + self._n]
        return self._data

    @property
    def device(self):
        return (self._data if self._data is not None else self._owner).device

    @classmethod
    def from_rows(cls, x: Tensor) -> "TileBlocked":
        """Tile-blocked copy of a row-major [rows, width] matrix (tests, measurements); padding rows / columns are zero."""
        rows, width = x.shape
        wp, rp = (width + 31) // 32 * 32, (rows + 31) // 32 * 32
        t = cls(rows, width, x.device)
        pad = torch.zeros((rp, wp), dtype=torch.float32, device=x.device)
        pad[:rows, :width] = x
        t.data.copy_(pad.view(rp // 32, 32, wp // 32, 4, 2, 4).permute(0, 2, 3, 4, 1, 5).reshape(-1))  # [tile, e, t, q, hi, i] -> [tile, t, q, hi, e, i]
        return t

    def to_rows(self) -> Tensor:
        """Row-major copy (tests, debugging)."""
        wp = (self.width + 31) // 32 * 32
        t = self.data.view(-1, wp // 32, 4, 2, 32, 4)  # [tile, t, q, hi, e, i]
        return t.permute(0, 4, 1, 2, 3, 5).reshape(-1, wp)[:self.rows, :self.width].contiguous()


class _Region:
    """A row-major [rows, width] fp32 region of a flat allocation, for consumers that only need its address and shape (the backward
    kernels' scratch: one allocation per chain instead of four per block)."""
    __slots__ = ("owner", "ptr", "shape")

    def __init__(self, owner: Tensor, offset: int, rows: int, width: int):
        self.owner, self.ptr, self.shape = owner, owner.data_ptr() + 4 * offset, (rows, width)

    def data_ptr(self) -> int:
        return self.ptr

    @property
    def device(self):
        return self.owner.device


WG_STATS = {"fwd": 0, "fwd_chain": 0, "bwd": 0}  # launches that went through them (tests assert the path under test ran)


def _gated(spec: Gcp2Spec) -> bool:
    return spec.vmode == VMODE_SCALAR_GATE and spec.vo > 0 and spec.vi > 0


# Every pack cache (one per GCP module) that has asked for a workgroup-kernel image is known through a weakly held handle that lives
# IN that cache: when one image turns out stale -- the first block of a step after the optimizer has updated the weights -- the images
# of ALL known blocks whose weights have moved on are rebuilt by the same launch (gcpnet_wg_pack_multi; 34 single-image launches per
# NMS model step otherwise).  An image built ahead of its use is keyed on the weights' (address, version) like any other: a weight
# changed again before the block runs, or a block that asks with another column view, just misses again.
class _WgPackUser:
    __slots__ = ("cache", "dims", "gated", "segs", "w_scalar", "w_gate", "epoch", "__weakref__")

    # The handle lives in a module's pack cache, i.e. inside the module: copy.deepcopy(model) / torch.save(model) walk over it.  A copy
    # of the module starts without one (its first _pack_wg makes its own); weak references are neither copyable nor picklable.
    def __deepcopy__(self, memo):
        return None

    def __reduce__(self):
        return (type(None), ())


_WG_PACK_USERS = weakref.WeakSet()
BATCH_WG_PACKS = os.environ.get("GCPNET_BATCH_WG_PACKS", "1") != "0"


def _wg_pack_key(segs, w_scalar, w_gate):
    return (_PACK_EPOCH, segs) + tuple(None if t is None else (t.data_ptr(), t._version) for t in (w_scalar, w_gate))


def _pack_wg(spec: Gcp2Spec, w) -> Tensor:
    """Packed image of scalar_out / vector_out_scale for the workgroup kernels (gcpnet_wg_pack[_view]), cached per weight version."""
    view = spec.w_view
    w_scalar, w_gate = (view[0] if view is not None else w[0]), w[5]
    segs = None if view is None else tuple(view[1])
    key = _wg_pack_key(segs, w_scalar, w_gate)
    cache = spec.pack_cache
    _note_pack_use()
    if cache is not None and cache.get("wg_key") == key:
        me = cache.get("wg_user")
        if me is not None:
            me.epoch = _PACK_EPOCH  # "asked for in this epoch" -- also when the image was built ahead by another block's miss
        return cache["wg_pack"]
    dims = (spec.si, spec.vi, spec.so, spec.vo, spec.hidden, int(spec.use_frames))
    # (cache, dims, gated, column view, scalar weight matrix, gate weight, key) of every image this launch builds; the asked one first
    todo = [(cache, dims, int(_gated(spec)), segs, w_scalar, w_gate, key)]
    if cache is not None and BATCH_WG_PACKS:
        me = cache.get("wg_user")
        if me is None:
            me = cache["wg_user"] = _WgPackUser()
            me.cache = cache
            _WG_PACK_USERS.add(me)
        me.dims, me.gated, me.segs = dims, todo[0][2], segs
        me.w_scalar, me.w_gate = weakref.ref(w_scalar), (None if w_gate is None else weakref.ref(w_gate))
        me.epoch = _PACK_EPOCH
        # others ride along only if they took part in the step before this one: a frozen teacher, a second model or a block that is
        # no longer called would otherwise be re-packed at every optimizer step -- and, inside a GraphedStep capture, have their
        # images allocated in the graph's private pool and the extra work baked into every replay
        for u in list(_WG_PACK_USERS):
            if u is me or "wg_key" not in u.cache or getattr(u, "epoch", -1) < _PREV_USE_EPOCH:
                continue
            ws2, wg2 = u.w_scalar(), (None if u.w_gate is None else u.w_gate())
            if ws2 is None or (u.w_gate is not None and wg2 is None) or ws2.device != w_scalar.device:
                continue  # (weights that were temporaries of one call: that block packs on its own miss)
            key2 = _wg_pack_key(u.segs, ws2, wg2)
            if u.cache["wg_key"] != key2:
                todo.append((u.cache, u.dims, u.gated, u.segs, ws2, wg2, key2))
    lib = _lib.load()
    jobs = (_lib.WgPackJob * len(todo))()
    packs = []
    for j, (_, d, gated, sg, ws_, wg_, _) in zip(jobs, todo):
        pack = torch.empty(int(lib.gcpnet_wg_pack_floats(*d, gated)), dtype=torch.float32, device=ws_.device)
        packs.append(pack)
        jw = j.w
        jw.si, jw.vi, jw.so, jw.vo, jw.hidden, jw.use_frames = d
        jw.w_scalar, jw.w_gate = ws_.data_ptr(), (None if wg_ is None else wg_.data_ptr())
        j.gated, j.W, j.out = gated, ws_.data_ptr(), pack.data_ptr()
        if sg is None:
            K = d[0] + (d[4] + (9 if d[5] else 0) if d[1] > 0 else 0)
            j.ld, j.trans, j.nseg = K, 0, 1
            j.start[0], j.len[0] = 0, K
        else:
            j.ld, j.trans, j.nseg = ws_.stride(0), 0, len(sg)
            for k, (a, m) in enumerate(sg):
                j.start[k], j.len[k] = a, m
    check(lib.gcpnet_wg_pack_multi(len(todo), jobs, _stream()), "wg_pack_multi")
    for (c, _, _, _, _, _, key_), pack in zip(todo, packs):
        if c is not None:
            c["wg_key"], c["wg_pack"] = key_, pack
    return packs[0]


_ZERO_BIAS: dict = {}
_LINEAR_PACKS: dict = {}  # wg_linear: (epoch, weight pointer, version, view) -> packed image


def wg_linear(x: Tensor, W: Tensor, out_dim: int, in_dim: int, col0: int = 0, trans: bool = False,
              bias: Optional[Tensor] = None) -> Optional[Tensor]:
    """x [n, in_dim] @ W'^T -> [n, out_dim] through the workgroup forward kernel (a block without vectors is a plain Linear), with
    W' a view of the stored matrix W [*, ld]: W'[r, c] = W[r, col0 + c], or with `trans` W'[r, c] = W[c, col0 + r] (the input
    gradient of the former).  Returns None for shapes outside the kernel (the callers raise: there is no library GEMM on this path)."""
    lib = _lib.load()
    n = x.shape[0]
    if out_dim % 4 or out_dim < 4 or not x.is_contiguous() or x.shape[1] != in_dim:
        return None
    if n == 0:  # (a rank of the one-graph sharding without local rows: nothing to launch)
        return x.new_empty((0, out_dim))
    spec = Gcp2Spec(si=in_dim, vi=0, so=out_dim, vo=0, hidden=0, use_frames=False, act_s=None, act_v=None, slope=0.0,
                    vmode=VMODE_NONE, vector_residual=False, e3=False, s_plans=[None], v_plans=[])
    dev = x.device
    zb = bias
    if zb is None:
        zb = _ZERO_BIAS.get((dev, out_dim))
        if zb is None:
            zb = _ZERO_BIAS[(dev, out_dim)] = torch.zeros(out_dim, dtype=torch.float32, device=dev)
    w = (None, zb, None, None, None, None, None)
    # packed image of the view: cached per weight version like every other pack (a weight's column ranges / its transpose are
    # packed once per optimizer step, not once per call)
    key = (_PACK_EPOCH, W.data_ptr(), W._version, W.stride(0), col0, bool(trans), in_dim, out_dim, dev)
    hit = _LINEAR_PACKS.get(key)
    pack = hit[1] if hit is not None and hit[0]() is W else None  # (the very tensor object: an address can be recycled)
    if pack is None:
        if len(_LINEAR_PACKS) >= 512:  # (stale versions of updated weights)
            _LINEAR_PACKS.clear()
        npk = lib.gcpnet_wg_pack_floats(in_dim, 0, out_dim, 0, 0, 0, 0)
        pack = torch.empty(int(npk), dtype=torch.float32, device=dev)
        ws = _weights_struct(spec, w, pack)
        start, length = (C.c_int * 1)(0), (C.c_int * 1)(in_dim)
        base = C.c_void_p(W.data_ptr() + 4 * col0)
        check(lib.gcpnet_wg_pack_view(C.byref(ws), 0, base, W.stride(0), int(trans), 1, start, length, _p(pack), _stream()), "wg_pack_view")
        _LINEAR_PACKS[key] = (weakref.ref(W), pack)
    ws = _weights_struct(spec, w, pack)
    out = torch.empty((n, out_dim), dtype=torch.float32, device=dev)
    blk = WgBlock()
    blk.w, blk.o = ws, _opts_struct(spec)
    blk.s_out, blk.residual = _p(out), 0
    rc = lib.gcpnet_wg_forward(n, _p(x), None, None, None, None, 1, C.byref(blk), _stream())
    if rc == _lib.E_UNSUPPORTED:
        return None
    check(rc, "wg_forward(linear)")
    WG_STATS["linear"] = WG_STATS.get("linear", 0) + 1
    return out


def _wg_block(spec: Gcp2Spec, w, s_out, v_out, s_pre, gate, residual: bool, keep: list) -> WgBlock:
    """`keep` receives the packed-weight tensor: it must stay referenced until the launch is enqueued (specs without a pack
    cache get a fresh image per call; released earlier, the caching allocator may hand its memory to the next torch.empty)."""
    blk = WgBlock()
    pack = _pack_wg(spec, w)
    keep.append(pack)
    blk.w = _weights_struct(spec, w, pack)
    blk.o = _opts_struct(spec)
    blk.s_out, blk.v_out = _p(s_out), _p(v_out)
    blk.s_pre, blk.gate = _p(s_pre), _p(gate)
    blk.residual = int(residual)
    blk.s_out_tb, blk.s_pre_tb = int(isinstance(s_out, TileBlocked)), int(isinstance(s_pre, TileBlocked))
    return blk


class _Gcp2(torch.autograd.Function):
    """inputs: spec, frames, then n_s scalar sources, n_v vector sources, res_s, res_v, 7 weights, the scalar addend tables
    (len(spec.add_plans)) and the vector addend tables (len(spec.vadd_plans))."""

    @staticmethod
    def forward(ctx, spec: Gcp2Spec, frames, *tensors):
        lib = _lib.load()
        n_s, n_v = len(spec.s_plans), len(spec.v_plans)
        s_src = list(tensors[:n_s])
        v_src = list(tensors[n_s:n_s + n_v])
        res_s, res_v = tensors[n_s + n_v], tensors[n_s + n_v + 1]
        w = tuple(tensors[n_s + n_v + 2:n_s + n_v + 9])
        n_a = len(spec.add_plans)
        adds = list(tensors[n_s + n_v + 9:n_s + n_v + 9 + n_a])
        vadds = list(tensors[n_s + n_v + 9 + n_a:])
        assert len(vadds) == len(spec.vadd_plans)
        need_grad = any(ctx.needs_input_grad)
        rows, s_out, v_out, pack, s_pre, gate = _gcp2_forward_launch(spec, frames, s_src, v_src, res_s, res_v, w, adds, vadds,
                                                                     need_grad)
        if need_grad:
            ctx.spec, ctx.rows, ctx.n_s, ctx.n_v = spec, rows, n_s, n_v
            ctx.frames = frames
            ctx.has_res = (tensors[n_s + n_v] is not None, tensors[n_s + n_v + 1] is not None)
            ctx.w_leaf = not spec.shared_weights  # (checked again, against the live tensors, in the backward: _side_stream_ok)
            ctx.weights = w
            ctx.use_cells = _note_uses(w)
            ctx.s_pre_tb = isinstance(s_pre, TileBlocked)  # (saved as its flat tensor, wrapped again in the backward)
            ctx.save_for_backward(*s_src, *v_src, *[t for t in w], pack, s_pre.data if ctx.s_pre_tb else s_pre, gate, *vadds)
        if spec.vo:
            return s_out, v_out
        return s_out

    @staticmethod
    def backward(ctx, d_s_out, d_v_out=None):
        spec, rows, n_s, n_v = ctx.spec, ctx.rows, ctx.n_s, ctx.n_v
        saved = ctx.saved_tensors
        s_src, v_src = list(saved[:n_s]), list(saved[n_s:n_s + n_v])
        w = tuple(saved[n_s + n_v:n_s + n_v + 7])
        pack, s_pre, gate = saved[n_s + n_v + 7:n_s + n_v + 10]
        if getattr(ctx, "s_pre_tb", False):
            s_pre = TileBlocked(rows, spec.so, s_pre.device, owner=s_pre, offset=0, n=s_pre.numel())
        vadds = list(saved[n_s + n_v + 10:])
        f32 = dict(dtype=torch.float32, device=s_pre.device)
        d_s_out = _req(d_s_out, "grad") if d_s_out is not None else torch.zeros((rows, spec.so), **f32)
        if spec.vo:
            d_v_out = _req(d_v_out, "grad") if d_v_out is not None else torch.zeros((rows, spec.vo, 3), **f32)
        si, vi = spec.si, spec.vi
        need_w = ctx.needs_input_grad[2 + n_s + n_v + 2:2 + n_s + n_v + 9]
        side = ctx.w_leaf and _side_stream_ok(ctx.weights, _take_use_cells(ctx))
        d_s_in, d_v_in, scr = gcp2_backward_data(spec, rows, s_src, v_src, ctx.frames, w, pack, s_pre, gate, d_s_out,
                                                 d_v_out, need_w=any(need_w), vadds=vadds, side_reduce=side)
        wgrads = [None] * 7
        if any(need_w):
            wgrads = gcp2_weight_grads(spec, rows, s_src, s_pre, scr, in_backward_of_leaves=side)

        # ---- input gradients: un-concatenate, scatter-add the gathered sources back to their rows -----------------
        grads_s: List[Optional[Tensor]] = []
        off = 0
        for t, pl in zip(s_src, spec.s_plans):
            dim = t.shape[1]
            if pl is not None:
                grads_s.append(_segment_reduce_raw(d_s_in, off, dim, si, pl, False))
            else:
                grads_s.append(d_s_in if n_s == 1 else d_s_in[:, off:off + dim])
            off += dim
        grads_v: List[Optional[Tensor]] = []
        off = 0
        for t, pl in zip(v_src, spec.v_plans):
            ch = t.shape[1]
            if pl is not None:
                grads_v.append(_segment_reduce_raw(d_v_in, 3 * off, 3 * ch, 3 * vi, pl, False).reshape(pl.n_src, ch, 3))
            else:
                grads_v.append(d_v_in if n_v == 1 else d_v_in[:, off:off + ch, :])
            off += ch
        g_res_s = d_s_out if ctx.has_res[0] else None
        g_res_v = d_v_out if ctx.has_res[1] else None
        wgrads = [g if need else None for g, need in zip(wgrads, need_w)]
        # pre-projected inputs enter s_pre additively: their gradient is ds_pre, summed over the rows that gathered them
        ds_pre = scr["ds_pre"]
        grads_add = [ds_pre if pl is None else _segment_reduce_raw(ds_pre, 0, spec.so, spec.so, pl, False)
                     for pl in spec.add_plans]
        # pre-projected vector inputs enter [vh | vf] additively: their gradient is d[vh | vf], summed the same way
        grads_vadd = []
        for tb, pl in zip(vadds, spec.vadd_plans):
            dq = scr["dvhf"]
            width = dq.shape[1]
            g = dq if pl is None else _segment_reduce_raw(dq, 0, width, width, pl, False)
            grads_vadd.append(g.view(tb.shape))
        return (None, None, *grads_s, *grads_v, g_res_s, g_res_v, *wgrads, *grads_add, *grads_vadd)


def _gcp2_forward_launch(spec: Gcp2Spec, frames, s_src, v_src, res_s, res_v, w, adds, vadds, need_grad: bool):
    """One gcpnet_gcp2_forward launch.  Returns (rows, s_out, v_out, pack, s_pre, gate)."""
    lib = _lib.load()
    n_v = len(v_src)
    rows = spec.s_plans[0].rows if spec.s_plans[0] is not None else s_src[0].shape[0]
    dev = s_src[0].device
    pack = None  # (image for the wave-per-tile kernels: built only when one of them runs)
    opts = _opts_struct(spec)
    sc = _concat(s_src, spec.s_plans, False)
    vc = _concat(v_src, spec.v_plans, True) if n_v else Concat()
    ac = _concat(adds, spec.add_plans, False) if adds else None
    vac = _vadd_concat(vadds, spec.vadd_plans) if vadds else None
    if spec.residual:
        res_s, res_v = s_src[0], (v_src[0] if n_v else None)
    s_out = torch.empty((rows, spec.so), dtype=torch.float32, device=dev)
    v_out = torch.empty((rows, spec.vo, 3), dtype=torch.float32, device=dev) if spec.vo else None
    plain = len(s_src) == 1 and spec.s_plans[0] is None and n_v == 1 and spec.v_plans[0] is None
    wg_route = USE_WG_KERNELS and plain and (spec.residual or (res_s is None and res_v is None)) and spec.vi > 0
    # s_pre is read back only by this block's backward and its weight-gradient GEMM: tile-blocked (include/gcpnet_hip.h,
    # gcp2_chain_item_t) when the workgroup kernels will run both ways -- 16-byte pieces of whole lines instead of 32 lines per
    # wave instruction on the way in, no staging tile on the way out
    tbp = bool(CHAIN_TILE_BLOCKED and BLOCK_TILE_BLOCKED and need_grad and rows > 0 and wg_route and USE_WG_BACKWARD and spec.so % 32 == 0)
    if tbp:
        s_pre = TileBlocked(rows, spec.so, dev)
    else:
        s_pre = torch.empty((rows, spec.so), dtype=torch.float32, device=dev) if need_grad else None
    gated = spec.vmode == VMODE_SCALAR_GATE and spec.vo > 0 and spec.vi > 0
    gate = torch.empty((rows, spec.vo), dtype=torch.float32, device=dev) if (need_grad and gated) else None
    if rows == 0:  # (an empty edge set: one half of autoregressive_forward's edge split, a node without in-edges in the CPD
        return rows, s_out, v_out, pack, s_pre, gate  # sampling loop) -- nothing to launch, empty tensors have no address
    if wg_route:
        # one workgroup per 32-row tile, output columns split over its waves (gcp_wg_fwd.hip)
        keep: list = []
        blk = _wg_block(spec, w, s_out, v_out, s_pre, gate, spec.residual, keep)
        rc = lib.gcpnet_wg_forward(rows, _p(s_src[0]), _p(v_src[0]), _p(frames), C.byref(ac) if ac is not None else None,
                                   C.byref(vac) if vac is not None else None, 1, C.byref(blk), _stream())
        if rc != _lib.E_UNSUPPORTED:
            check(rc, "wg_forward")
            WG_STATS["fwd"] += 1
            return rows, s_out, v_out, pack, s_pre, gate
        if tbp:  # (the other kernels write rows)
            s_pre = torch.empty((rows, spec.so), dtype=torch.float32, device=dev)
    w = _dense_weights(spec, w)
    pack = _pack(spec, w)
    ws = _weights_struct(spec, w, pack)
    if ((adds or vadds) and len(s_src) == 1 and spec.s_plans[0] is None and n_v == 1 and spec.v_plans[0] is None
            and not spec.residual and res_s is None and res_v is None and spec.vo > 0 and USE_HEAD_KERNEL):
        # the first message GCP after project-then-gather: plain (e, xi) inputs + gathered addend tables -> the register-
        # resident kernel of the chain, run for this one block (gcp2_chain_fwd.hip, HEAD)
        hd = Head()
        hd.e_in, hd.xi_in = s_src[0].data_ptr(), v_src[0].data_ptr()
        hd.s_add = ac if ac is not None else Concat()
        hd.v_add = vac if vac is not None else Concat()
        hd.w, hd.o = ws, opts
        hd.s_out, hd.v_out = s_out.data_ptr(), v_out.data_ptr()
        hd.s_pre = s_pre.data_ptr() if s_pre is not None else None
        hd.gate = gate.data_ptr() if gate is not None else None
        rc = lib.gcpnet_gcp2_headchain_forward(rows, C.byref(hd), _p(frames), 0, None, _stream())
        if rc != _lib.E_UNSUPPORTED:
            check(rc, "gcp2_headchain_forward")
            return rows, s_out, v_out, pack, s_pre, gate
    check(lib.gcpnet_gcp2_forward(rows, C.byref(sc), C.byref(vc), _p(frames), C.byref(ws), C.byref(opts),
                                  C.byref(ac) if ac is not None else None, C.byref(vac) if vac is not None else None,
                                  _p(res_s), _p(res_v), _p(s_out), _p(v_out), _p(s_pre), _p(gate), _stream()),
          "gcp2_forward")
    return rows, s_out, v_out, pack, s_pre, gate


USE_HEAD_KERNEL = True  # module switch


def _vadd_concat(tables: Sequence[Tensor], plans: Sequence[Optional[GatherPlan]]) -> Concat:
    c = Concat()
    c.n = len(tables)
    for k, (t, pl) in enumerate(zip(tables, plans)):
        c.ptr[k] = t.data_ptr()
        c.idx[k] = pl.idx.data_ptr() if pl is not None else None
        c.dim[k] = t.shape[2]  # HF' (tables are [n_src, 3, HF'])
    return c


def gcp2_backward_data(spec: Gcp2Spec, rows: int, s_src, v_src, frames, w, pack, s_pre, gate, d_s_out, d_v_out,
                       need_w: bool = True, vadds: Sequence[Tensor] = (), side_reduce: bool = False, tb_out: bool = False):
    """Launches the backward data-path kernel.  Returns (d_s_in, d_v_in, scratch dict for the weight-gradient GEMMs).
    `side_reduce`: the weights are leaves whose gradients nothing reads before the end of the backward pass (_side_stream_ok): the
    fused kernel's partial-sum reductions may then run on the weight-gradient stream."""
    lib = _lib.load()
    f32 = dict(dtype=torch.float32, device=s_pre.device)
    si, vi, vo = spec.si, spec.vi, spec.vo
    if rows == 0:  # nothing to launch: empty input gradients, zero weight gradients (assembled like the fused form's)
        H, K = spec.hidden, spec.K
        nf = 9 if (spec.use_frames and vi > 0) else 0
        g: List[Optional[Tensor]] = [torch.zeros((spec.so, K), **f32), torch.zeros((spec.so,), **f32), None, None, None, None, None]
        if vi > 0:
            g[2] = torch.zeros((H, vi), **f32)
            if nf:
                g[3] = torch.zeros((3, vi), **f32)
            if vo > 0:
                g[4] = torch.zeros((vo, H), **f32)
                if spec.vmode == VMODE_SCALAR_GATE:
                    g[5], g[6] = torch.zeros((vo, spec.so), **f32), torch.zeros((vo,), **f32)
        t = dict(ds_pre=torch.zeros((0, spec.so), **f32), fused=g)
        if len(vadds):
            t["dvhf"] = torch.zeros((0, 3 * vadds[0].shape[2]), **f32)
        return torch.zeros((0, si), **f32), (torch.zeros((0, vi, 3), **f32) if vi > 0 else None), t
    if (USE_WG_KERNELS and USE_WG_BACKWARD and len(s_src) == 1 and spec.s_plans[0] is None and len(v_src) == 1
            and spec.v_plans[0] is None and vi > 0 and rows > 0):
        res = _wg_backward(spec, rows, s_src[0], v_src[0], frames, w, s_pre, gate, d_s_out, d_v_out, need_w, vadds, side_reduce,
                           tb_out=tb_out)
        if res is not None:
            return res
    if tb_out or any(isinstance(t_, TileBlocked) for t_ in (d_s_out, *s_src)):
        raise _lib.GcpnetHipError("tile-blocked tensors reached a block the workgroup backward kernel refuses")
    if isinstance(s_pre, TileBlocked):  # (a single block whose forward ran in the workgroup kernel and whose backward cannot: rows again)
        s_pre = s_pre.to_rows()
    d_s_in = torch.empty((rows, si), **f32)
    d_v_in = torch.empty((rows, vi, 3), **f32) if vi > 0 else None
    scr, t = _alloc_bwd_scratch(spec, rows, need_w, s_pre.device)
    w = _dense_weights(spec, w)
    if pack is None:
        pack = _pack(spec, w)
    ws = _weights_struct(spec, w, pack)
    opts = _opts_struct(spec, fused_residual=spec.residual)
    sc = _concat(s_src, spec.s_plans, False)
    vc = _concat(v_src, spec.v_plans, True) if len(v_src) else Concat()
    vac = None
    if len(vadds):
        vac = _vadd_concat(vadds, spec.vadd_plans)
        t["dvhf"] = torch.empty((rows, 3 * vadds[0].shape[2]), **f32)
        scr.dvhf = t["dvhf"].data_ptr()
    check(lib.gcpnet_gcp2_backward(rows, C.byref(sc), C.byref(vc), _p(frames), C.byref(ws), C.byref(opts),
                                   C.byref(vac) if vac is not None else None, _p(s_pre), _p(gate), _p(d_s_out),
                                   _p(d_v_out) if vo else None, _p(d_s_in), _p(d_v_in), C.byref(scr), _stream()),
          "gcp2_backward")
    return d_s_in, d_v_in, t


def _wg_backward_supported(spec: Gcp2Spec, rows: int, w) -> bool:
    lib = _lib.load()
    if spec.vi <= 0:
        return False
    ws = _weights_struct(spec, w, None)  # (the plan depends on dims and options only)
    opts = _opts_struct(spec)
    plan = WgBwdPlan()
    return lib.gcpnet_wg_backward_plan(rows, C.byref(ws), C.byref(opts), 1, C.byref(plan)) == 0


def _wg_backward(spec: Gcp2Spec, rows: int, s_in, v_in, frames, w, s_pre, gate, d_s_out, d_v_out, need_w: bool, vadds,
                 side_reduce: bool = False, tb_out: bool = False):
    """Backward of one block through the workgroup kernel (gcp_wg_bwd.hip).  Returns (d_s_in, d_v_in, scratch dict) like
    gcp2_backward_data, or None when the shape is outside that kernel.  In fused mode the scratch dict carries the finished
    weight gradients under "fused" (summed over the persistent workgroups' partials in a fixed order); otherwise the per-row
    operands of the TN GEMMs, as the wave-per-tile kernel leaves them."""
    lib = _lib.load()
    f32 = dict(dtype=torch.float32, device=s_pre.device)
    si, vi, so, vo, H = spec.si, spec.vi, spec.so, spec.vo, spec.hidden
    wg_pack = _pack_wg(spec, w)  # (stays referenced until the launch below is enqueued)
    ws = _weights_struct(spec, w, wg_pack)
    opts = _opts_struct(spec)
    plan = WgBwdPlan()
    rc = lib.gcpnet_wg_backward_plan(rows, C.byref(ws), C.byref(opts), int(need_w), C.byref(plan))
    if rc == _lib.E_UNSUPPORTED:
        return None
    check(rc, "wg_backward_plan")
    gated = _gated(spec)
    nf = 9 if spec.use_frames else 0
    a = WgBwdArgs()
    a.w, a.o, a.residual = ws, opts, int(spec.residual)
    a.s_in, a.v_in, a.frames = _p(s_in), _p(v_in), _p(frames)
    vac = _vadd_concat(vadds, spec.vadd_plans) if len(vadds) else None
    a.v_add = C.pointer(vac) if vac is not None else None
    a.s_pre, a.gate, a.d_s_out, a.d_v_out = _p(s_pre), _p(gate), _p(d_s_out), _p(d_v_out) if vo else None
    # tile-blocked tensors (a ResGCP chain walked block by block: s_pre from the workgroup forward, the state gradient between
    # the blocks, ds_pre for the weight-gradient GEMM -- nothing else reads them)
    tb = int(isinstance(s_pre, TileBlocked)) | (int(isinstance(d_s_out, TileBlocked)) << 1) | (int(tb_out) << 2)
    fused = bool(plan.fused)
    if (tb & ~3 or isinstance(s_in, TileBlocked)) and fused:  # (the fused form reads s_in as rows; its ds_pre output stays row-major)
        return None
    d_s_in = TileBlocked(rows, si, s_pre.device) if tb_out else torch.empty((rows, si), **f32)
    d_v_in = torch.empty((rows, vi, 3), **f32)
    a.d_s_in, a.d_v_in = _p(d_s_in), _p(d_v_in)
    t = {}
    if (not fused and need_w) or spec.add_plans:
        if tb and not fused and not spec.add_plans and so % 32 == 0:
            t["ds_pre"] = TileBlocked(rows, so, s_pre.device)
            tb |= 8
        else:
            t["ds_pre"] = torch.empty((rows, so), **f32)
        a.ds_pre = _p(t["ds_pre"])
    else:
        t["ds_pre"] = None
    a.tb = tb
    if len(vadds):
        t["dvhf"] = torch.empty((rows, 3 * vadds[0].shape[2]), **f32)
        a.dvhf = _p(t["dvhf"])
    grid = plan.grid
    if need_w:
        t["w_part"] = torch.empty((grid, plan.n_small), **f32)
        a.wsm_part = _p(t["w_part"])
    if fused:
        dw_part = torch.empty((grid, so * plan.kw), **f32)
        a.dw_part = _p(dw_part)
        dwg_part = torch.empty((grid, vo * (so + 1)), **f32) if gated else None
        a.dwg_part = _p(dwg_part)
    elif need_w:
        t["ext"] = torch.empty((rows, plan.ext_w), **f32)
        a.ext = _p(t["ext"])
        if gated:
            t["dgate"] = torch.empty((rows, plan.dgate_w), **f32)
            a.dgate = _p(t["dgate"])
    rc = lib.gcpnet_wg_backward(rows, C.byref(a), _stream())
    if rc == _lib.E_UNSUPPORTED:
        return None
    check(rc, "wg_backward")
    del wg_pack
    WG_STATS["bwd"] += 1
    if fused:
        K = spec.K
        g: List[Optional[Tensor]] = [None] * 7
        g[0], g[1] = torch.empty((so, K), **f32), torch.empty((so,), **f32)
        if gated:
            g[5], g[6] = torch.empty((vo, so), **f32), torch.empty((vo,), **f32)
        wv = torch.empty((plan.n_small,), **f32)
        w_part = t["w_part"]
        n_small, kw = plan.n_small, plan.kw

        def reduces():  # per-workgroup partial sums -> gradients, fixed order (one launch for the two or three buffers)
            jobs = [(dw_part, so, kw, K, g[0], g[1])]
            if gated:
                jobs.append((dwg_part, vo, so + 1, so, g[5], g[6]))
            jobs.append((w_part, 1, n_small, n_small, wv, None))
            arr = (_lib.WgReduceJob * len(jobs))()
            for j, (parts, R, Cc, CW, ow, ob) in zip(arr, jobs):
                j.parts, j.n_parts, j.R, j.C, j.CW, j.out_w, j.out_b = _p(parts), grid, R, Cc, CW, _p(ow), _p(ob)
            check(lib.gcpnet_wg_reduce_multi(len(jobs), arr, _stream()), "wg_reduce")

        if side_reduce and WEIGHT_GRADS_ON_SIDE_STREAM:
            # off the critical path, like the TN GEMMs of the unfused route: they run under the next block's kernel.  Only the
            # PARTIAL buffers are kept alive until the join -- never the gradients themselves (`g`, `wv`): a tensor that is
            # still referenced from Python when AccumulateGrad receives it is CLONED (on the caller's stream, which has no
            # dependency on the weight-gradient stream: .grad would be a copy of memory the reduction has not written yet)
            # instead of adopted (tests/test_side_stream.py)
            _side_submit(reduces, (dw_part, dwg_part, w_part))
        else:
            reduces()
        if _DEBUG_GRAD_PTRS is not None:
            _DEBUG_GRAD_PTRS.extend(t_.data_ptr() for t_ in (g[0], g[1], g[5], g[6], wv) if t_ is not None)
        del reduces  # (the closure references g / wv)
        o1, o2 = vo * H, vo * H + H * vi
        if vo:
            g[4] = wv[:o1].view(vo, H)
        g[2] = wv[o1:o2].view(H, vi)
        if nf:
            g[3] = wv[o2:].view(3, vi)
        t["fused"] = g  # (popped by _WeightGradJob: the scratch dict outlives the backward call on the side-stream keep lists)
        t["keep"] = (dw_part, dwg_part)
    return d_s_in, d_v_in, t


def _alloc_bwd_scratch(spec: Gcp2Spec, rows: int, need_w: bool, device, tb: bool = False):
    """The backward kernels' per-row / per-tile outputs for the weight gradients (include/gcpnet_hip.h, gcp2_bwd_scratch_t)."""
    lib = _lib.load()
    f32 = dict(dtype=torch.float32, device=device)
    H, vi, vo, so = spec.hidden, spec.vi, spec.vo, spec.so
    nf = 9 if (spec.use_frames and vi > 0) else 0
    has_vec, has_vout = vi > 0, vi > 0 and vo > 0
    gated = spec.vmode == VMODE_SCALAR_GATE and has_vout
    scr = BwdScratch()
    t = dict(ds_pre=TileBlocked(rows, so, device) if tb else torch.empty((rows, so), **f32))
    scr.ds_pre = t["ds_pre"].data_ptr()
    r4 = lambda x: (x + 3) // 4 * 4
    if has_vec:
        t["ext"] = torch.empty((rows, r4(H + nf)), **f32)
        scr.ext = t["ext"].data_ptr()
        if need_w:  # per-tile shares of the small vector weight gradients (summed over tiles in _WeightGradJob)
            t["w_part"] = torch.empty((lib.gcpnet_gcp2_bwd_tiles(rows), vo * H + vi * (H + 3)), **f32)
            scr.w_part = t["w_part"].data_ptr()
        if gated:
            t["dgate"] = torch.empty((rows, r4(vo)), **f32)
            scr.dgate = t["dgate"].data_ptr()
    return scr, t


class _WeightGradJob:
    """TN-GEMM problems for the weight gradients of one GCP2 block, fed by the backward kernel's scratch `t`."""

    def __init__(self, spec: Gcp2Spec, rows: int, s_src, s_pre, t, w=None):
        lib = _lib.load()
        f32 = dict(dtype=torch.float32, device=s_pre.device)  # (s_pre: a Tensor or a TileBlocked)
        H, vi, vo, so = spec.hidden, spec.vi, spec.vo, spec.so
        nf = 9 if (spec.use_frames and vi > 0) else 0
        self.spec, self.nf, self.device = spec, nf, s_pre.device
        if "fused" in t:  # the workgroup backward kernel has produced the gradients itself
            self.job, self.keep, self.g = None, [t], t.pop("fused")
            return
        has_vec, has_vout = vi > 0, vi > 0 and vo > 0
        gated = spec.vmode == VMODE_SCALAR_GATE and has_vout
        self.keep = [t, s_pre, list(s_src)]
        # one record per block (include/gcpnet_hip.h, gcp2_wgrad_job_t): gcpnet_gcp2_weight_grads derives the two GEMM problems
        #     d scalar_out.weight | bias = ds_pre^T [s sources | norms | frame scalars | 1],  d gate.weight | bias = dgate^T [act_v(s_pre) | 1]
        # and the column sums of the per-tile partials from it, and carves their scratch out of ONE workspace per call
        j = WgradJob()
        j.rows, j.so, j.vo, j.vi, j.hidden, j.use_frames = rows, so, vo, vi, H, int(bool(spec.use_frames))
        j.gated, j.act_v, j.slope = int(gated), ACT[spec.act_v], float(spec.slope)
        op = j.s_in
        op.n = len(s_src)
        for k, (x, pl) in enumerate(zip(s_src, spec.s_plans)):
            tb = isinstance(x, TileBlocked)
            width = x.width if tb else x.shape[1]
            op.ptr[k], op.dim[k], op.ld[k], op.tb[k] = x.data_ptr(), width, width, int(tb)
            op.idx[k] = pl.idx.data_ptr() if pl is not None else None
            assert not (tb and pl is not None)
        ds = t["ds_pre"]
        j.ds_pre, j.ds_pre_tb = ds.data_ptr(), int(isinstance(ds, TileBlocked))
        g: List[Optional[Tensor]] = [None] * 7
        self.g = g
        g[0], g[1] = torch.empty((so, spec.K), **f32), torch.empty((so,), **f32)
        j.d_w_scalar, j.d_b_scalar = g[0].data_ptr(), g[1].data_ptr()
        if gated:  # d vector_out_scale.weight / bias
            g[5], g[6] = torch.empty((vo, so), **f32), torch.empty((vo,), **f32)
            j.dgate, j.s_pre, j.s_pre_tb = t["dgate"].data_ptr(), s_pre.data_ptr(), int(isinstance(s_pre, TileBlocked))
            j.d_w_gate, j.d_b_gate = g[5].data_ptr(), g[6].data_ptr()
            # act_v = identity (every shipped configuration): the gate Linear reads s_pre = [s | norms | frame scalars] W^T + b itself,
            # so its weight gradient follows from dgate^T [s | ...] -- the operand the scalar_out gradient streams anyway -- and s_pre
            # is not needed (gcp2_wgrad_job_t.gate_lin); REQUIRED when the forward did not store it (TileBlocked.absent)
            absent = bool(getattr(s_pre, "absent", False))
            if GATE_GRADS_FROM_INPUTS and spec.act_v is None and w is not None and w[0] is not None and w[1] is not None and w[0].is_contiguous():
                j.gate_lin, j.w_scalar, j.b_scalar = 1, w[0].data_ptr(), w[1].data_ptr()
                self.keep.append((w[0], w[1]))
            elif absent:
                raise _lib.GcpnetHipError("s_pre was not stored by the forward and the gate gradients cannot be formed from the block's inputs")
        if has_vec:  # d vector_up / vector_down(.frames): the backward kernel left one partial sum per tile
            j.ext = t["ext"].data_ptr()
            part = t["w_part"]
            wv = torch.empty((part.shape[1],), **f32)
            j.w_part, j.n_parts, j.w_width, j.d_w_small = part.data_ptr(), part.shape[0], part.shape[1], wv.data_ptr()
            o1, o2 = vo * H, vo * H + H * vi
            if has_vout:
                g[4] = wv[:o1].view(vo, H)
            g[2] = wv[o1:o2].view(H, vi)
            if nf:
                g[3] = wv[o2:].view(3, vi)
        self.job = j

    def grads(self) -> List[Optional[Tensor]]:
        """(scalar_out.weight, scalar_out.bias, vector_down, vector_down_frames, vector_up, gate.weight, gate.bias).
        Hands the tensors over: the job keeps no reference, so that autograd can take them as .grad without a copy (and
        without reading them on its own stream before the weight-gradient stream has written them)."""
        g, self.g = self.g, None
        return g


WEIGHT_GRADS_ON_SIDE_STREAM = os.environ.get("GCPNET_SIDE_STREAM", "1") != "0"  # module switch, see set_weight_grad_stream()
_DEBUG_GRAD_PTRS: Optional[list] = None  # tests: addresses of the gradient buffers the fused backward produced (adoption check)


def set_weight_grad_stream(enabled: bool) -> None:
    """Weight gradients on a second HIP stream (default on).  They are complete when `backward()` returns (the caller's stream
    joins in an end-of-backward callback), but NOT while it is still running: code that reads a parameter's gradient from
    inside the backward pass -- `torch.nn.parallel.DistributedDataParallel`'s bucket hooks, `register_hook` /
    `register_post_accumulate_grad_hook` on parameters -- must switch this off (or use `gcpnet_amd.parallel.GradAllReducer`,
    which runs after the backward pass)."""
    global WEIGHT_GRADS_ON_SIDE_STREAM
    WEIGHT_GRADS_ON_SIDE_STREAM = bool(enabled)
_side_streams: dict = {}
_main_streams: dict = {}
_side_pending: list = []


class _UseCell:
    """[the weight, number of autograd Functions of live graphs that take it]; list-like for its readers (c[0], c[1])."""
    __slots__ = ("t", "n", "__weakref__")

    def __init__(self, t):
        self.t, self.n = t, 0

    def __getitem__(self, i):
        return self.t if i == 0 else self.n


# id(weight) -> its cell, held WEAKLY: the autograd contexts that counted a use own the cell, so it lives exactly as long as a graph
# that uses the weight does and nothing ever has to clear the table (a clear that landed between two uses of one weight in one graph
# -- an auxiliary backward in the middle of a forward -- made both Functions see a count of 1: ADVICE round 3)
# (a plain dict of weakref.ref objects without callbacks: WeakValueDictionary's Python-level __setitem__ / removal callback cost 0.5 ms
# of a host-bound configs[1] step for 280 weights; dead references are simply overwritten -- the table is bounded by the number of
# distinct weight addresses, and `cell.t is t` rejects a recycled id)
_weight_uses: dict = {}


def _note_uses(weights) -> list:
    """Called by the forward of every Function that may put weight gradients on the side stream: counts, per weight tensor, the
    Functions of the graph under construction that take it as an input.  A weight used twice (autoregressive_forward applies
    `interaction` twice; any module called twice per step) receives two gradients that the autograd engine SUMS on the caller's
    stream as soon as the second arrives -- inside the backward pass, before the end-of-backward join -- so both must be complete
    on the caller's stream.  Returns the shared counter cells; the backward reads them through _side_stream_ok.
    A cell dies with the last context that holds it (the graph is freed); a graph that is kept alive makes later answers about its
    weights conservative (count > 1: caller's stream), never wrong."""
    cells = []
    for t in weights:
        if t is None:
            continue
        r = _weight_uses.get(id(t))
        cell = r() if r is not None else None
        if cell is None or cell.t is not t:
            cell = _UseCell(t)
            _weight_uses[id(t)] = weakref.ref(cell)
        cell.n += 1
        cells.append(cell)
    return cells


def _side_stream_ok(weights, use_cells=None) -> bool:
    """The weight-gradient stream may be used only when nothing can READ the returned gradients before the end-of-backward join:
    every weight is a leaf that autograd will simply adopt as .grad -- no existing .grad to accumulate into (gradient
    accumulation, zero_grad(set_to_none=False)), no tensor hooks, no second Function of the same graph using it (`use_cells`,
    see _note_uses) -- and no distributed wrapper is reducing gradients from inside the backward pass
    (DistributedDataParallel's bucket hooks; gcpnet_amd.parallel.GradAllReducer runs after it and opts in)."""
    _ensure_end_of_backward_callback()
    if not WEIGHT_GRADS_ON_SIDE_STREAM:
        return False
    if torch.distributed.is_available() and torch.distributed.is_initialized() and not SIDE_STREAM_UNDER_DISTRIBUTED:
        return False
    for t in weights:
        if t is None:
            continue
        if not t.is_leaf or t.grad is not None or t._backward_hooks or getattr(t, "_post_accumulate_grad_hooks", None):
            return False
    if use_cells is False:  # (a context whose cells were released by an earlier backward over the same graph: stay on the caller's stream)
        return False
    if use_cells is not None and any(c[1] > 1 for c in use_cells):
        return False
    return True


def _take_use_cells(ctx):
    """The context's use cells, released: the cells must die with the backward pass, not with the graph object (a `loss` tensor that
    the caller still holds while the next step's forward runs keeps every context of the old graph alive -- its cells would make the
    next step count every weight twice).  A second backward over a retained graph finds False: no side stream."""
    cells = getattr(ctx, "use_cells", None)
    ctx.use_cells = False
    return cells


SIDE_STREAM_UNDER_DISTRIBUTED = False  # set by gcpnet_amd.parallel.GradAllReducer (it reduces after the backward pass)


_end_callback_task = -1  # id of the backward pass (graph task) that has _end_of_backward queued


def _ensure_end_of_backward_callback() -> None:
    """Queues _end_of_backward once per backward pass.  Keyed on the engine's graph-task id, not on a flag: a pass that dies
    with an exception never runs its callbacks, and a flag would then keep every later pass from queueing its join.  Outside a
    backward pass (id -1: a test calling a backward helper directly) nothing is queued."""
    global _end_callback_task
    task = torch._C._current_graph_task_id()
    if task < 0 or task == _end_callback_task:
        return
    if _side_pending:  # left over from a pass that raised: its launches are long enqueued, join them now
        _join_side_stream()
    torch.autograd.Variable._execution_engine.queue_callback(_end_of_backward)
    _end_callback_task = task


DEFER_CHAIN_WEIGHT_GRADS = os.environ.get("GCPNET_DEFER_TN", "0") == "1"
_deferred_jobs: list = []


def _flush_deferred_weight_grads() -> None:
    while _deferred_jobs:
        run_weight_grad_jobs(_deferred_jobs.pop(0), in_backward_of_leaves=True)


def _end_of_backward():
    """End-of-backward callback: the caller's stream waits for the weight-gradient stream; scratch is released; the table of
    weight uses of the graph that was just differentiated is dropped."""
    global _end_callback_task
    _end_callback_task = -1
    _flush_deferred_weight_grads()
    _join_side_stream()


def _join_side_stream():
    for main, side in {(m, s) for m, s, _ in _side_pending}:
        main.wait_stream(side)
    _side_pending.clear()


def _make_side_stream(dev: int):
    """The weight-gradient stream.  GCPNET_SIDE_CU_MASK=<n> (tuning knob) confines it to the first n compute units of every XCD-
    interleaved group through hipExtStreamCreateWithCUMask, so that the TN GEMMs cannot occupy the whole chip while the caller's
    stream runs its small launches; default: an ordinary stream."""
    n = int(os.environ.get("GCPNET_SIDE_CU_MASK", "0") or 0)
    prio = os.environ.get("GCPNET_SIDE_PRIORITY", "")
    if n <= 0 and prio == "":
        return torch.cuda.Stream(device=dev)
    hip = C.CDLL("libamdhip64.so")
    if n <= 0:  # GCPNET_SIDE_PRIORITY=<p> (tuning knob): a HIP stream priority for the weight-gradient stream (1 = low on this runtime,
        #          0 = normal, -1 = high; clamped to the device's range), so that the caller's stream's launches get the CUs first
        lo, hi_ = C.c_int(), C.c_int()
        raw = C.c_void_p()
        with torch.cuda.device(dev):
            if hip.hipDeviceGetStreamPriorityRange(C.byref(lo), C.byref(hi_)) != 0:
                raise _lib.GcpnetHipError("hipDeviceGetStreamPriorityRange failed")
            p_ = max(min(int(prio), lo.value), hi_.value)  # (least priority = the larger number)
            err = hip.hipStreamCreateWithPriority(C.byref(raw), C.c_uint(1), C.c_int(p_))  # (hipStreamNonBlocking)
        if err != 0:
            raise _lib.GcpnetHipError(f"hipStreamCreateWithPriority failed: {err}")
        if os.environ.get("GCPNET_SIDE_PRIORITY_VERBOSE"):
            sys.stderr.write(f"[gcpnet_amd] weight-gradient stream priority {p_} (device range: least {lo.value}, greatest {hi_.value})\n")
        return torch.cuda.ExternalStream(raw.value, device=dev)
    total = torch.cuda.get_device_properties(dev).multi_processor_count
    words = (total + 31) // 32
    mask = (C.c_uint32 * words)()
    for cu in range(min(n, total)):
        mask[cu // 32] |= 1 << (cu % 32)
    raw = C.c_void_p()
    with torch.cuda.device(dev):
        err = hip.hipExtStreamCreateWithCUMask(C.byref(raw), C.c_uint32(words), mask)
    if err != 0:
        raise _lib.GcpnetHipError(f"hipExtStreamCreateWithCUMask failed: {err}")
    return torch.cuda.ExternalStream(raw.value, device=dev)


def _side_submit(fn, keep) -> None:
    """Runs fn() with the weight-gradient stream current (ordered after everything enqueued on the caller's stream so far);
    `keep` stays referenced until the caller's stream has joined at the end of the backward pass."""
    if torch._C._current_graph_task_id() < 0:  # not inside a backward pass: nothing would join the side stream
        fn()
        return
    main, side = _fork_side_stream(keep)
    with torch.cuda.stream(side):
        fn()


def _fork_side_stream(keep):
    """(caller's stream, weight-gradient stream) with the latter ordered behind everything enqueued on the former so far; `keep` is
    registered for the end-of-backward join.  The fork is one library call on the raw handles (gcpnet_stream_wait_stream: event
    record + stream wait, ~3 us) instead of torch's Event / Stream objects (~20 us, 16 forks per configs[1] step)."""
    dev = torch.cuda.current_device()
    raw = _raw_stream(dev) if _raw_stream is not None else torch.cuda.current_stream().cuda_stream
    main = _main_streams.get((dev, raw))
    if main is None:  # (the Stream object of the caller's stream, for the join: built once per stream, not per fork)
        main = _main_streams[(dev, raw)] = torch.cuda.current_stream()
    side = _side_streams.get(dev)
    if side is None:
        side = _side_streams[dev] = _make_side_stream(dev)
    check(_lib.load().gcpnet_stream_wait_stream(C.c_void_p(side.cuda_stream), C.c_void_p(raw)), "stream_wait_stream")
    _ensure_end_of_backward_callback()
    _side_pending.append((main, side, keep))
    return main, side


def run_weight_grad_jobs(jobs: Sequence[_WeightGradJob], in_backward_of_leaves: bool = False) -> None:
    """Launches the TN GEMMs of several blocks, up to 8 problems per launch.

    Weight gradients are off the critical path of the backward pass (nothing downstream reads them when the weights are leaf
    parameters), so from inside autograd they are enqueued on a second HIP stream: they then run concurrently with the data-path
    kernels of the following blocks, which on their own leave most CUs idle for node-row launches.  The caller's stream
    joins that stream in a callback at the end of the backward pass (before any optimizer / all-reduce can touch .grad)."""
    jobs = [j for j in jobs if j.job is not None]  # (fused blocks have produced their gradients already)
    if not jobs:
        return
    lib = _lib.load()
    arr = (WgradJob * len(jobs))(*[j.job for j in jobs])
    ws = torch.empty((max(int(lib.gcpnet_gcp2_weight_grads_workspace(len(jobs), arr)), 1),), dtype=torch.float32, device=jobs[0].device)
    stream = _stream()
    if in_backward_of_leaves and WEIGHT_GRADS_ON_SIDE_STREAM and torch._C._current_graph_task_id() >= 0:
        # operands, partials and the workspace (allocated on the caller's stream, used on the other one) stay referenced until the
        # end-of-backward join; the launches take the weight-gradient stream's raw handle -- no stream context to enter and leave
        _, side = _fork_side_stream([j.keep for j in jobs] + [ws])
        stream = C.c_void_p(side.cuda_stream)
    check(lib.gcpnet_gcp2_weight_grads(len(jobs), arr, _p(ws), stream), "gcp2_weight_grads")


def gcp2_weight_grads(spec: Gcp2Spec, rows: int, s_src, s_pre, t, in_backward_of_leaves: bool = False) -> List[Optional[Tensor]]:
    job = _WeightGradJob(spec, rows, s_src, s_pre, t)
    run_weight_grad_jobs([job], in_backward_of_leaves)
    return job.grads()


class _Gcp2Chain(torch.autograd.Function):
    """x_k = x_{k-1} + GCP_k(x_{k-1}), k = 1..n, in ONE forward launch with the tile state kept on chip.
    inputs: specs (one per block), agg, frames, s0, v0, then 7 weights per block.
    `agg` = None, or (GatherPlan, mean): the Function then returns the segment sum / mean of the chain's output over the plan's
    segments (the message aggregation, components/gcpnet.py:939-947) instead of the per-row output, and its backward hands the
    segment-level gradient tables to the chain backward kernel, which reads row r's incoming gradient from table row plan.idx[r]
    (gcpnet_gcp2_chain_backward_gathered) -- the [rows, s] and [rows, V, 3] copies a separate aggregation's backward would write,
    and the chain backward would read back, never exist."""

    @staticmethod
    def forward(ctx, specs, agg, frames, s0, v0, *weights):
        lib = _lib.load()
        n = len(specs)
        rows, dev = s0.shape[0], s0.device
        want_grad = any(ctx.needs_input_grad)
        # CHAIN_RECOMPUTE (the memory route): the forward keeps NOTHING per block -- no s_pre, gates, intermediate states -- only the
        # chain's inputs; the backward first runs this forward again, in saving mode (`_recompute_pass`), and then proceeds as usual
        light = CHAIN_RECOMPUTE and want_grad and rows > 0 and not getattr(ctx, "_recompute_pass", False)
        need_grad = want_grad and not light
        f32 = dict(dtype=torch.float32, device=dev)
        items = (ChainItem * n)()
        ws, packs, outs = [], [], []
        sp0 = specs[0]
        wave_first = (PREFER_WAVE_CHAIN_FORWARD and sp0.so <= 128 and n <= _lib.MAX_CHAIN and
                      lib.gcpnet_gcp2_chain_forward_registers_ok(sp0.si, sp0.vi, sp0.so, sp0.vo, sp0.hidden, int(sp0.use_frames)) == 1)
        # what only the chain kernels and the weight-gradient GEMM read -- s_pre, the intermediate scalar states -- is saved
        # tile-blocked when both the forward and the backward will run in the register-resident chain kernels (decided HERE: the
        # backward of this graph then takes that route whatever the module switches say by then)
        tb = (CHAIN_TILE_BLOCKED and need_grad and wave_first and rows > 0 and _wave_chain_backward(sp0) and
              lib.gcpnet_gcp2_chain_backward_ok(sp0.si, sp0.vi, sp0.so, sp0.vo, sp0.hidden, int(sp0.use_frames)) == 1)
        all_w = [tuple(weights[7 * k:7 * k + 7]) for k in range(n)]
        # the same for chains that run in the workgroup kernels both ways (wider than 128: (256,32)): s_pre and the intermediate
        # states tile-blocked when the backward will walk the chain block by block through gcpnet_wg_backward's plain form
        wtb = False
        if (CHAIN_TILE_BLOCKED and need_grad and not tb and not wave_first and rows > 0 and USE_WG_KERNELS and USE_WG_BACKWARD
                and n <= _lib.WG_MAX_BLOCKS and sp0.so % 32 == 0 and sp0.si == sp0.so and sp0.vi > 0 and not _wave_chain_backward(sp0)
                and all((sp.si, sp.vi, sp.so, sp.vo, sp.hidden) == (sp0.si, sp0.vi, sp0.so, sp0.vo, sp0.hidden) and not sp.add_plans
                        for sp in specs)):
            plan = WgBwdPlan()
            wtb = (lib.gcpnet_wg_backward_plan(rows, C.byref(_weights_struct(sp0, all_w[0], None)), C.byref(_opts_struct(sp0)), 1,
                                               C.byref(plan)) == 0 and not plan.fused)
        all_packs = _pack_many(specs, all_w) if n <= _lib.MAX_CHAIN else [_pack(sp, w_) for sp, w_ in zip(specs, all_w)]
        if tb:
            # everything this route saves for its backward -- intermediate states, s_pre, gates: read back only by address -- comes
            # out of ONE allocation (the blocks of a chain share their dimensions)
            so, vo = sp0.so, sp0.vo
            r64 = lambda x: (x + 63) // 64 * 64
            n_tb = int(lib.gcpnet_tb_floats(rows, so))
            gated0 = sp0.vmode == VMODE_SCALAR_GATE
            o_gate = r64(n_tb)
            o_sout = o_gate + (r64(rows * vo) if gated0 else 0)
            o_vout = o_sout + r64(n_tb)
            o_sign = o_vout + r64(rows * 3 * vo)
            # sign masks of s_pre: all the chain backward kernel needs of it when the activations are piecewise linear
            n_sign = (int(lib.gcpnet_tb_sign_words(rows, so))
                      if CHAIN_SIGN_MASKS and all(sp.act_s in _PWL_ACTS and sp.act_v in _PWL_ACTS for sp in specs) else 0)
            # ... and then s_pre itself need not be stored when nothing else reads it: the gate gradients come from the block's inputs
            skip_pre = bool(n_sign and CHAIN_SKIP_S_PRE and GATE_GRADS_FROM_INPUTS and gated0 and not light and
                            all(sp.act_v is None and w_[0] is not None and w_[1] is not None and w_[0].is_contiguous() for sp, w_ in zip(specs, all_w)))
            if skip_pre:  # (its region goes: everything behind it moves up)
                o_gate, o_sout, o_vout, o_sign = (o - r64(n_tb) for o in (o_gate, o_sout, o_vout, o_sign))
            per = o_sign + r64(n_sign)
            flat = torch.empty((per * n,), **f32)
        for k, spec in enumerate(specs):
            w = all_w[k]
            pack = all_packs[k]
            last = k == n - 1  # intermediate states are only materialised when the backward will need them
            gated = spec.vmode == VMODE_SCALAR_GATE
            if tb:
                assert (spec.so, spec.vo, gated) == (so, vo, gated0)
                base = per * k
                s_pre = TileBlocked(rows, so, dev, owner=flat, offset=base, n=0 if skip_pre else n_tb)
                s_pre.absent = skip_pre
                if n_sign:
                    s_pre.sign = flat.data_ptr() + 4 * (base + o_sign)
                gate = _Region(flat, base + o_gate, rows, vo) if gated else None
                if last:
                    s_out, v_out = torch.empty((rows, so), **f32), torch.empty((rows, vo, 3), **f32)
                else:
                    s_out = TileBlocked(rows, so, dev, owner=flat, offset=base + o_sout, n=n_tb)
                    v_out = _Region(flat, base + o_vout, rows, 3 * vo)
            else:
                if wtb and not last:
                    s_out = TileBlocked(rows, spec.so, dev)
                else:
                    s_out = torch.empty((rows, spec.so), **f32) if (need_grad or last) else None
                v_out = torch.empty((rows, spec.vo, 3), **f32) if (need_grad or last) else None
                if wtb:
                    s_pre = TileBlocked(rows, spec.so, dev)
                else:
                    s_pre = torch.empty((rows, spec.so), **f32) if need_grad else None
                gate = torch.empty((rows, spec.vo), **f32) if (need_grad and gated) else None
            items[k].w = _weights_struct(spec, w, pack)
            items[k].o = _opts_struct(spec)
            items[k].s_out = s_out.data_ptr() if s_out is not None else None
            items[k].v_out = v_out.data_ptr() if v_out is not None else None
            items[k].s_pre = s_pre.data_ptr() if (s_pre is not None and not getattr(s_pre, "absent", False)) else None
            items[k].gate = gate.data_ptr() if gate is not None else None
            items[k].s_out_tb, items[k].s_pre_tb = int(isinstance(s_out, TileBlocked)), int(isinstance(s_pre, TileBlocked))
            items[k].s_sign = s_pre.sign if isinstance(s_pre, TileBlocked) else None
            ws.append(w); packs.append(pack); outs.append((s_out, v_out, s_pre, gate))
        if rows == 0:  # (an empty edge set: nothing to launch)
            if need_grad:
                ctx.rows, ctx.n_weights, ctx.agg, ctx.weights = 0, len(weights), agg, weights
            if agg is not None:
                return (torch.zeros((agg[0].n_src, specs[-1].so), **f32), torch.zeros((agg[0].n_src, specs[-1].vo, 3), **f32))
            return outs[-1][0], outs[-1][1]
        rc = _lib.E_UNSUPPORTED
        if USE_WG_KERNELS and n <= _lib.WG_MAX_BLOCKS and not wave_first:
            keep: list = []
            blks = (WgBlock * n)(*[_wg_block(spec, w, *outs[k], True, keep) for k, (spec, w) in enumerate(zip(specs, ws))])
            rc = lib.gcpnet_wg_forward(rows, _p(s0), _p(v0), _p(frames), None, None, n, blks, _stream())
            if rc != _lib.E_UNSUPPORTED:
                check(rc, "wg_forward")
                WG_STATS["fwd_chain"] += 1
            elif wtb:
                raise _lib.GcpnetHipError("gcpnet_wg_forward refused a chain whose backward plan it accepted (tile-blocked activations)")
        if rc == _lib.E_UNSUPPORTED:
            check(lib.gcpnet_gcp2_chain_forward(rows, _p(s0), _p(v0), _p(frames), n, items, _stream()), "gcp2_chain_forward")
        if light:
            ctx.specs, ctx.frames, ctx.rows = specs, frames, rows
            ctx.state = (s0, v0, None, None, None)
            ctx.in_versions = (s0._version, v0._version)
            ctx.w_leaf = not any(sp.shared_weights for sp in specs)
            ctx.weights = weights
            ctx.use_cells = _note_uses(weights)
            ctx.agg, ctx.tb, ctx.wtb, ctx.fwd_items, ctx.recompute = agg, False, False, None, True
        if need_grad:
            ctx.specs, ctx.frames, ctx.rows = specs, frames, rows
            # (the chain's own outputs must not hang off ctx: output -> grad_fn -> ctx -> output is a cycle through C++ that the
            # garbage collector cannot see, i.e. every forward whose backward never runs would leak its saved activations)
            saved = outs[:-1] + [(None, None, outs[-1][2], outs[-1][3])]
            ctx.state = (s0, v0, ws, packs, saved)
            ctx.in_versions = (s0._version, v0._version)  # (plain attributes bypass autograd's saved-tensor check: do it by hand)
            ctx.w_leaf = not any(sp.shared_weights for sp in specs)
            ctx.weights = weights
            ctx.use_cells = None if getattr(ctx, "_recompute_pass", False) else _note_uses(weights)
            ctx.agg = agg
            ctx.tb = tb
            ctx.wtb = wtb
            ctx.fwd_items = items  # (the backward's records start from these: same weights, packs, options)
            ctx.recompute = False
        if agg is not None:
            plan, mean = agg
            m_s, m_v = outs[-1][0], outs[-1][1]
            vo = specs[-1].vo
            return (_segment_reduce_raw(m_s, 0, m_s.shape[1], m_s.shape[1], plan, mean),
                    _segment_reduce_raw(m_v.view(rows, 3 * vo), 0, 3 * vo, 3 * vo, plan, mean).view(plan.n_src, vo, 3))
        return outs[-1][0], outs[-1][1]

    @staticmethod
    def backward(ctx, d_s, d_v):
        agg = ctx.agg
        if ctx.rows == 0:  # no rows: the input gradients are empty, the weight gradients zero (as the single-block path returns them)
            wz = [torch.zeros_like(w) if (w is not None and need) else None for w, need in zip(ctx.weights, ctx.needs_input_grad[5:])]
            ctx.weights = None
            if agg is not None:  # (no row for the segment-level gradients to reach)
                return (None, None, None, None, None, *wz)
            return (None, None, None, d_s, d_v, *wz)
        specs, frames, rows = ctx.specs, ctx.frames, ctx.rows
        s0, v0, ws, packs, outs = ctx.state
        ctx_tb, ctx_wtb, ctx_items = ctx.tb, getattr(ctx, "wtb", False), getattr(ctx, "fwd_items", None)
        if (s0._version, v0._version) != ctx.in_versions:  # (e.g. a masked layer's in-place row update, gcpnet.py:1248-1251, on a tensor
            raise RuntimeError("one of the variables needed for gradient computation has been modified by an inplace operation "  # this
                               "(the input state of a ResGCP chain)")                                                             # chain read)
        if getattr(ctx, "recompute", False):  # the memory route: the saving forward runs now (same launches, same bits as the plain route)
            import types

            again = types.SimpleNamespace(needs_input_grad=(False, False, False, True, True) + (True,) * len(ctx.weights), _recompute_pass=True)
            with torch.no_grad():
                _Gcp2Chain.forward(again, specs, None, frames, s0, v0, *ctx.weights)
            _, _, ws, packs, outs = again.state
            ctx_tb, ctx_wtb, ctx_items = again.tb, again.wtb, again.fwd_items
        n = len(specs)
        f32 = dict(dtype=torch.float32, device=s0.device)
        out_rows = agg[0].n_src if agg is not None else rows
        d_s = _req(d_s, "grad") if d_s is not None else torch.zeros((out_rows, specs[0].so), **f32)
        d_v = _req(d_v, "grad") if d_v is not None else torch.zeros((out_rows, specs[0].vo, 3), **f32)
        need_w = ctx.needs_input_grad[5:]
        jobs: List[Optional[_WeightGradJob]] = [None] * n
        ins = [(s0, v0) if k == 0 else (outs[k - 1][0], outs[k - 1][1]) for k in range(n)]
        nws = [any(need_w[7 * k:7 * k + 7]) for k in range(n)]
        # workgroup kernels: block by block with the weight gradients fused; else one launch of the wave-per-tile chain kernel
        # (gradient state on chip); shapes outside both go block by block through the generic kernel
        # The wave-per-tile chain kernel (one launch, gradient state on chip, so <= 128) is the faster one where it applies
        # (measured at (128,16): 1.2 ms + 0.76 ms of weight-gradient GEMMs per 7 blocks against 7 x 0.33 ms); wider chains --
        # (256,32) -- go block by block through the workgroup kernel; shapes outside both through the generic kernel.
        res = None
        side_ok = ctx.w_leaf and _side_stream_ok(ctx.weights, _take_use_cells(ctx))  # (asked once: the cells are released by the question)
        if getattr(ctx, "recompute", False):
            # (the weight-gradient stream keeps its operands -- here the recomputed s_pre / states, 20 GB per layer at configs[4] size --
            # alive until the end-of-backward join: on the memory route the GEMMs run on the caller's stream and the buffers go back
            # to the allocator as soon as this Function returns)
            side_ok = False
        wave_chain = ctx_tb or (not ctx_wtb and _wave_chain_backward(specs[0]))
        if wave_chain:  # (with `agg` the kernel reads the segment-level tables itself)
            res = gcp2_chain_backward_data(specs, rows, ins, outs, frames, ws, packs, d_s, d_v, nws, out_agg=agg, fwd_items=ctx_items)
            assert res is not None or not ctx_tb, "tile-blocked activations were saved for a chain the backward kernel refuses"
        if res is None and agg is not None:
            # block-by-block routes take per-row gradients: the adjoint of the aggregation as its own launches
            plan, mean = agg
            scale = plan.inv_count if mean else None
            d_s = _gather_rows_raw(d_s, plan, scale)
            d_v = _gather_rows_raw(d_v.reshape(plan.n_src, -1), plan, scale).view(rows, specs[0].vo, 3)
        if res is not None:
            d_s, d_v, scrs = res
            for k in range(n):
                if nws[k]:
                    jobs[k] = _WeightGradJob(specs[k], rows, [ins[k][0]], outs[k][2], scrs[k], w=ws[k])
        else:
            side = side_ok
            for k in range(n - 1, -1, -1):
                s_in, v_in = ins[k]
                _, _, s_pre, gate = outs[k]
                d_s, d_v, scr = gcp2_backward_data(specs[k], rows, [s_in], [v_in], frames, ws[k], packs[k], s_pre, gate, d_s,
                                                   d_v, need_w=nws[k], side_reduce=side, tb_out=bool(ctx_wtb) and k > 0)
                if nws[k]:
                    jobs[k] = _WeightGradJob(specs[k], rows, [s_in], s_pre, scr)
        live = [j for j in jobs if j is not None]
        if live:
            side = side_ok
            if side and DEFER_CHAIN_WEIGHT_GRADS:
                # (experiment, DESIGN.md 6b: the chain's TN GEMMs are handed to the side stream only after the first message GCP's
                # backward and its HBM-bound input-gradient reductions have been enqueued -- _flush_deferred_weight_grads)
                _deferred_jobs.append(live)
            else:
                run_weight_grad_jobs(live, in_backward_of_leaves=side)
        wgrads: List[Optional[Tensor]] = []
        for k in range(n):
            g = jobs[k].grads() if jobs[k] is not None else [None] * 7
            wgrads += [gi if need else None for gi, need in zip(g, need_w[7 * k:7 * k + 7])]
        ctx.state = None
        return (None, None, None, d_s, d_v, *wgrads)


def _wave_chain_backward(sp0: Gcp2Spec) -> bool:
    """The module switches' choice for the backward of a ResGCP chain: one launch of the wave-per-tile chain kernel (True) or
    block by block through the workgroup backward kernel."""
    return (not (USE_WG_KERNELS and USE_WG_BACKWARD and PREFER_WG_CHAIN_BACKWARD)) or (sp0.so <= 128 and not FORCE_WG_CHAIN_BACKWARD)


def gcp2_chain_backward_data(specs, rows: int, ins, outs, frames, ws, packs, d_s: Tensor, d_v: Tensor, need_w: Sequence[bool],
                             out_agg=None, fwd_items=None):
    """Backward data path of a whole ResGCP chain in one launch (gcpnet_gcp2_chain_backward).  ins[k] = (s, V) input of block
    k, outs[k] = (s_out, v_out, s_pre, gate) saved by the forward.  Returns (d_s_in, d_v_in, per-block scratch dicts), or None
    when the shape is outside that kernel (the caller then goes block by block).  `out_agg` = (GatherPlan, mean): d_s / d_v are
    the gradients of the segment sum / mean of the chain's output, [n_seg, so] and [n_seg, vo, 3] (gcpnet_gcp2_chain_backward_gathered)."""
    lib = _lib.load()
    n = len(specs)
    items = (ChainBwdItem * n)()
    scrs = []
    # the blocks of a chain share their dimensions: the four scratch regions of every block come out of ONE allocation
    sp0 = specs[0]
    H, vi, vo, so = sp0.hidden, sp0.vi, sp0.vo, sp0.so
    nf = 9 if (sp0.use_frames and vi > 0) else 0
    has_vec = vi > 0
    gated = sp0.vmode == VMODE_SCALAR_GATE and vi > 0 and vo > 0
    r4 = lambda x: (x + 3) // 4 * 4
    r64 = lambda x: (x + 63) // 64 * 64
    tb_all = isinstance(outs[0][2], TileBlocked)
    n_ds = int(lib.gcpnet_tb_floats(rows, so)) if tb_all else rows * so
    EP, VOP = r4(H + nf), r4(vo)
    tiles = int(lib.gcpnet_gcp2_bwd_tiles(rows)) if has_vec else 0
    w_width = vo * H + vi * (H + 3)
    o_ext = r64(n_ds)
    o_dg = o_ext + (r64(rows * EP) if has_vec else 0)
    o_wp = o_dg + (r64(rows * VOP) if gated else 0)
    per = o_wp + (r64(tiles * w_width) if has_vec else 0)
    # flag words of the launch's tail split (gcpnet_gcp2_chain_backward_split), behind the blocks' regions of the same allocation
    n_flags = (int(lib.gcpnet_gcp2_chain_backward_flags(rows, n, sp0.si, vi, so, vo, H, int(sp0.use_frames)))
               if CHAIN_TAIL_SPLIT and has_vec else 0)
    flat = torch.empty((per * n + r64(n_flags),), dtype=torch.float32, device=d_s.device)
    for k in range(n):
        assert isinstance(outs[k][2], TileBlocked) == tb_all
        base = per * k
        scr = BwdScratch()
        ds = TileBlocked(rows, so, d_s.device, owner=flat, offset=base, n=n_ds) if tb_all else _Region(flat, base, rows, so)
        t = dict(ds_pre=ds)
        scr.ds_pre = ds.ptr
        if has_vec:
            t["ext"] = _Region(flat, base + o_ext, rows, EP)
            scr.ext = t["ext"].ptr
            if need_w[k]:
                t["w_part"] = _Region(flat, base + o_wp, tiles, w_width)
                scr.w_part = t["w_part"].ptr
            if gated:
                t["dgate"] = _Region(flat, base + o_dg, rows, VOP)
                scr.dgate = t["dgate"].ptr
        scrs.append(t)
        if fwd_items is not None:  # (the forward's records of the same weights, packs and options)
            items[k].w, items[k].o = fwd_items[k].w, fwd_items[k].o
            items[k].o.fused_residual = 1
        else:
            items[k].w = _weights_struct(specs[k], ws[k], packs[k])
            items[k].o = _opts_struct(specs[k], fused_residual=True)
        items[k].v_in = ins[k][1].data_ptr()
        items[k].tb = int(tb_all)
        if getattr(outs[k][2], "absent", False) and not (CHAIN_SIGN_MASKS and outs[k][2].sign):
            raise _lib.GcpnetHipError("this chain's forward stored sign masks instead of s_pre: the backward needs ops.CHAIN_SIGN_MASKS")
        items[k].s_pre = None if getattr(outs[k][2], "absent", False) else outs[k][2].data_ptr()  # (absent: only the sign-mask kernels can run it -- the library refuses any other form)
        items[k].s_sign = outs[k][2].sign if (CHAIN_SIGN_MASKS and isinstance(outs[k][2], TileBlocked)) else None
        items[k].gate = outs[k][3].data_ptr() if outs[k][3] is not None else None
        items[k].sc = scr
    d_s_in = torch.empty((rows, specs[0].si), dtype=torch.float32, device=d_s.device)
    d_v_in = torch.empty((rows, specs[0].vi, 3), dtype=torch.float32, device=d_s.device)
    plan, mean = out_agg if out_agg is not None else (None, False)
    if n_flags > 0:
        rc = lib.gcpnet_gcp2_chain_backward_split(rows, _p(frames), n, items, _p(d_s), _p(d_v), None if plan is None else _p(plan.idx),
                                                  _p(plan.inv_count) if mean else None, _p(d_s_in), _p(d_v_in),
                                                  C.c_void_p(flat.data_ptr() + 4 * per * n), n_flags, _stream())
    elif plan is not None:
        rc = lib.gcpnet_gcp2_chain_backward_gathered(rows, _p(frames), n, items, _p(d_s), _p(d_v), _p(plan.idx),
                                                     _p(plan.inv_count) if mean else None, _p(d_s_in), _p(d_v_in), _stream())
    else:
        rc = lib.gcpnet_gcp2_chain_backward(rows, _p(frames), n, items, _p(d_s), _p(d_v), _p(d_s_in), _p(d_v_in), _stream())
    if rc == _lib.E_UNSUPPORTED:
        return None
    check(rc, "gcp2_chain_backward")
    return d_s_in, d_v_in, scrs


def gcp2_chain(specs: Sequence[Gcp2Spec], s0: Tensor, v0: Tensor, frames: Optional[Tensor], weights: Sequence[tuple],
               agg: Optional[Tuple[GatherPlan, bool]] = None):
    """Chain of residual GCP2 blocks with identical dims (ResGCP).  `weights[k]` as for gcp2().  `agg` = (plan, mean): returns the
    segment sum / mean of the chain's output over plan's segments ([n_seg, so], [n_seg, vo, 3]) instead of the per-row output."""
    s0, v0 = _req(s0, "scalar input"), _req(v0, "vector input")
    if frames is not None:
        frames = _req(frames.detach(), "frames")
    flat = [None if t is None else _req(t, "weight") for w in weights for t in w]
    if agg is not None:
        assert agg[0].rows == s0.shape[0]
    return _Gcp2Chain.apply(list(specs), agg, frames, s0, v0, *flat)


def gcp2(spec: Gcp2Spec, s_sources: Sequence[Tensor], v_sources: Sequence[Tensor], frames: Optional[Tensor], weights,
         res_s: Optional[Tensor] = None, res_v: Optional[Tensor] = None):
    """Applies one GCP2 block.  `weights` = (scalar_out.weight, scalar_out.bias, vector_down.weight,
    vector_down_frames.weight, vector_up.weight, vector_out_scale.weight, vector_out_scale.bias), None where absent."""
    assert len(s_sources) == len(spec.s_plans) and len(v_sources) == len(spec.v_plans)
    assert sum(t.shape[1] for t in s_sources) == spec.si and sum(t.shape[1] for t in v_sources) == spec.vi
    s_sources = [_req(t, "scalar input") for t in s_sources]
    v_sources = [_req(t, "vector input") for t in v_sources]
    if frames is not None:
        frames = _req(frames.detach(), "frames")
    weights = tuple(None if t is None else _req(t, "weight") for t in weights)
    if res_s is not None:
        res_s = _req(res_s, "residual")
    if res_v is not None:
        res_v = _req(res_v, "residual")
    proj = _projectable(spec, s_sources) if PROJECT_GATHERED_SCALARS else None
    if proj is None:
        rows = spec.s_plans[0].rows if spec.s_plans[0] is not None else s_sources[0].shape[0]
        wide = _too_wide(spec, s_sources, rows)
        if wide is not None:
            # scalar_out is linear, so leading columns of the widest source can go through the workgroup kernel's plain-Linear form
            # and enter as an addend; the block kernel reduces over the remaining columns.  Done when the 32 x (si + H + 9) merged
            # tile of a wave does not fit in LDS (the second feed-forward GCP at (256,32): 1024 scalar inputs).  (The same split for
            # launches of few rows -- measured at 10 000 node rows in round 3: no gain -- is gone.)
            k, cut = wide
            w_scalar = weights[0]
            dims = [t.shape[1] for t in s_sources]
            off = sum(dims[:k])
            src = s_sources[k]
            left, right = _SplitCols.apply(src, cut)
            add = _Project.apply(left, w_scalar[:, off:off + cut])
            w_rest = torch.cat([w_scalar[:, :off], w_scalar[:, off + cut:]], dim=1)
            s_sources = list(s_sources[:k]) + [right] + list(s_sources[k + 1:])
            spec = replace(spec, si=spec.si - cut, pack_cache=None, add_plans=[spec.s_plans[k]])
            weights = (w_rest,) + tuple(weights[1:])
            return _Gcp2.apply(spec, frames, *s_sources, *v_sources, res_s, res_v, *weights, add)
    if proj is not None:
        # "Project, then gather" (see _Gcp2Projected): the shares of GATHERED sources (h[row], h[col] / chi[row], chi[col] in a
        # message GCP, reference gcpnet.py:907-917) in scalar_out / vector_down(.frames) are computed once per source row.
        vproj = _v_projectable(spec, v_sources)
        return _Gcp2Projected.apply(spec, frames, list(proj[0]), list(vproj[0]) if vproj is not None else [], *s_sources,
                                    *v_sources, res_s, res_v, *weights)
    return _Gcp2.apply(spec, frames, *s_sources, *v_sources, res_s, res_v, *weights)


PROJECT_GATHERED_SCALARS = True  # module switches (tests compare both paths)


def _tn_weight_grad(a2d: Tensor, b2d: Tensor) -> Tensor:
    """a2d^T b2d for a2d [rows, M], b2d [rows, N] (unit column stride, any widths / row strides) through gcpnet_tn_gemm."""
    lib = _lib.load()
    rows, M = a2d.shape
    N = b2d.shape[1]
    a, b = Operand(), Operand()
    a.n, b.n = 1, 1
    a.ptr[0], a.dim[0], a.ld[0] = a2d.data_ptr(), M, a2d.stride(0)
    b.ptr[0], b.dim[0], b.ld[0] = b2d.data_ptr(), N, b2d.stride(0)
    out = torch.empty((M, N), dtype=torch.float32, device=a2d.device)
    pr = TnProblem()
    pr.rows, pr.a, pr.b = rows, a, b
    pr.out, pr.out_sm, pr.out_sn, pr.out_m, pr.out_n = out.data_ptr(), N, 1, M, N
    pr.out2, pr.out2_n = None, 0
    pr.splits = lib.gcpnet_tn_splits(rows, M, N)
    part = torch.empty((pr.splits, M, N), dtype=torch.float32, device=a2d.device)
    pr.partial = part.data_ptr()
    check(lib.gcpnet_tn_gemm(1, C.byref(pr), _stream()), "tn_gemm")
    return out


def _rows_matmul_small(x2d: Tensor, w: Tensor) -> Tensor:
    """x2d [rows, K] @ w [K, J]: gcpnet_rows_matmul_small for a tiny w, the workgroup kernel's plain-Linear form otherwise."""
    K, J = w.shape
    if K * J > 4096:
        pad = (-J) % 4  # (the workgroup kernel's output width is a multiple of 4: zero columns for the launch, sliced back)
        wp = w if not pad else torch.nn.functional.pad(w, (0, pad))
        out = wg_linear(x2d.contiguous(), wp, J + pad, K, trans=True)
        if out is None:
            raise _lib.GcpnetHipError(f"rows x [{K}, {J}] product outside the HIP kernels' shapes (no library fallback)")
        return out if not pad else out[:, :J]
    lib = _lib.load()
    x2d = _req(x2d, "rows")
    out = torch.empty((x2d.shape[0], J), dtype=torch.float32, device=x2d.device)
    check(lib.gcpnet_rows_matmul_small(x2d.shape[0], K, J, _p(x2d), x2d.stride(0), _p(w), _p(out), J, _stream()),
          "rows_matmul_small")
    return out


def _tn_weight_grads_into(items) -> None:
    """For every (a2d, b2d, out) of `items`: out[M, N] (any row stride, unit column stride) = a2d^T b2d, all in ONE gcpnet_tn_gemm
    launch (+ one reduction) per GCP_TN_MAX_PROBLEMS problems."""
    lib = _lib.load()
    probs, parts = [], []
    for a2d, b2d, out in items:
        rows, M = a2d.shape
        N = b2d.shape[1]
        assert out.shape == (M, N) and out.stride(1) == 1
        a, b = Operand(), Operand()
        a.n, b.n = 1, 1
        a.ptr[0], a.dim[0], a.ld[0] = a2d.data_ptr(), M, a2d.stride(0)
        b.ptr[0], b.dim[0], b.ld[0] = b2d.data_ptr(), N, b2d.stride(0)
        pr = TnProblem()
        pr.rows, pr.a, pr.b = rows, a, b
        pr.out, pr.out_sm, pr.out_sn, pr.out_m, pr.out_n = out.data_ptr(), out.stride(0), 1, M, N
        pr.out2, pr.out2_n = None, 0
        pr.splits = lib.gcpnet_tn_splits(rows, M, N)
        part = torch.empty((pr.splits, M, N), dtype=torch.float32, device=a2d.device)
        pr.partial = part.data_ptr()
        probs.append(pr)
        parts.append(part)  # (alive until the launches are enqueued; the caching allocator keeps them stream-ordered afterwards)
    # (a launch takes the DMA kernels only if ALL its problems can -- widths / strides multiples of 4 floats, 16-byte aligned --: the
    # others go in launches of their own)
    def dma_ok(it):
        return all(t.shape[1] % 4 == 0 and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0 for t in it[:2])

    for group in ([pr for pr, it in zip(probs, items) if dma_ok(it)], [pr for pr, it in zip(probs, items) if not dma_ok(it)]):
        for i in range(0, len(group), _lib.TN_MAX_PROBLEMS):
            chunk = group[i:i + _lib.TN_MAX_PROBLEMS]
            arr = (TnProblem * len(chunk))(*chunk)
            check(lib.gcpnet_tn_gemm(len(chunk), arr, _stream()), "tn_gemm")


def _tn_weight_grad_into(a2d: Tensor, b2d: Tensor, out: Tensor) -> None:
    """out[M, N] (any row stride, unit column stride) = a2d^T b2d through gcpnet_tn_gemm."""
    _tn_weight_grads_into([(a2d, b2d, out)])


class _Gcp2Projected(torch.autograd.Function):
    """One GCP2 block whose GATHERED sources are projected at their source rows ("project, then gather").

    scalar_out, vector_down and vector_down_frames are linear in their concatenated inputs, so the share of a gathered source
    (h[row], h[col], chi[row], chi[col] in the first message GCP, reference gcpnet.py:907-917) is computed once per source
    row -- [n_src, dim] x [dim, so] and [3 n_src, V] x [V, H + 3] products on 16x fewer rows than edges -- and the kernels
    add the gathered result rows (gcpnet_gcp2_forward's s_add / v_add).  The per-edge reductions shrink from
    K = 2 s + 32 + H + 9 to 32 + H + 9 and from V_in = 2 V + 4 to 4.
    inputs: spec (all sources), frames, indices of the projected scalar / vector sources, then the tensors of _Gcp2
    (sources, res_s, res_v, the 7 LEAF weights).  The backward assembles the full weight gradients itself -- kernel results
    for the un-gathered columns, row-split TN GEMMs for the projected ones -- on the weight-gradient stream."""

    @staticmethod
    def forward(ctx, spec: Gcp2Spec, frames, sg, vg, *tensors):
        n_s, n_v = len(spec.s_plans), len(spec.v_plans)
        s_src, v_src = list(tensors[:n_s]), list(tensors[n_s:n_s + n_v])
        res_s, res_v = tensors[n_s + n_v], tensors[n_s + n_v + 1]
        w = tuple(tensors[n_s + n_v + 2:n_s + n_v + 9])
        w_scalar, b_scalar, w_down, w_frames, w_up, w_gate, b_gate = w
        dims = [t.shape[1] for t in s_src]
        offs = [sum(dims[:k]) for k in range(n_s)]
        rest = [k for k in range(n_s) if k not in sg]
        # node-level projections h W_k^T through the workgroup kernel, packed straight from W's column range (no slice copies,
        # no library GEMM); the columns the edge kernel keeps -- [un-gathered sources | norms | frame scalars] -- as a view too
        adds = []
        for k in sg:
            a = wg_linear(s_src[k], w_scalar, spec.so, dims[k], col0=offs[k])  # (whatever USE_WG_KERNELS says: there is no other Linear)
            if a is None:
                raise _lib.GcpnetHipError("node-level projection outside the workgroup kernel's shapes (no library fallback)")
            adds.append(a)
        segs = [(offs[k], dims[k]) for k in rest] + ([(spec.si, spec.K - spec.si)] if spec.K > spec.si else [])
        use_view = USE_WG_KERNELS and len(segs) <= 3 and w_scalar.stride(1) == 1
        w_rest = None if use_view else torch.cat([w_scalar[:, a:a + m] for a, m in segs], dim=1)
        chans = [t.shape[1] for t in v_src]
        voffs = [sum(chans[:k]) for k in range(n_v)]
        vr = [k for k in range(n_v) if k not in vg]
        H = spec.hidden
        hfp = (H + 3 + 3) // 4 * 4
        wvs, vts, vadds = [], [], []
        wd_rest, wf_rest = w_down, w_frames
        if vg:
            # column slices of the small vector weights in the forms the launches want -- [HF', V_k] (zero rows past H + 3), its
            # transpose, the un-gathered rest -- derived once per weight version, not once per call (a dozen tiny launches)
            vkey = (_PACK_EPOCH, tuple(vg), tuple(chans), w_down.data_ptr(), w_down._version, w_frames.data_ptr(), w_frames._version)
            vc = spec.pack_cache.get("vproj") if spec.pack_cache is not None else None
            if vc is None or vc["key"] != vkey:
                vc = dict(key=vkey, wk=[], wkt=[])
                f32v = dict(dtype=torch.float32, device=w_down.device)
                wd_, wf_ = w_down.detach(), w_frames.detach()
                jobs = []
                for k in vg:  # [HF', V] = [vector_down ; vector_down_frames ; zero rows] of the source's channels, and its transpose
                    wk, wkt = torch.empty((hfp, chans[k]), **f32v), torch.empty((chans[k], hfp), **f32v)
                    for dst in (wk, wkt.t()):
                        jobs += [(dst[:H], wd_[:, voffs[k]:voffs[k] + chans[k]]), (dst[H:H + 3], wf_[:, voffs[k]:voffs[k] + chans[k]]),
                                 (dst[H + 3:], None)]
                    vc["wk"].append(wk); vc["wkt"].append(wkt)
                nrest = sum(chans[k] for k in vr)
                vc["wd_rest"], vc["wf_rest"] = torch.empty((H, nrest), **f32v), torch.empty((3, nrest), **f32v)
                c = 0
                for k in vr:
                    jobs += [(vc["wd_rest"][:, c:c + chans[k]], wd_[:, voffs[k]:voffs[k] + chans[k]]),
                             (vc["wf_rest"][:, c:c + chans[k]], wf_[:, voffs[k]:voffs[k] + chans[k]])]
                    c += chans[k]
                copy2d_multi(jobs)  # (one launch; as ATen cat / pad / clone: a dozen)
                if spec.pack_cache is not None:
                    spec.pack_cache["vproj"] = vc
            for j, k in enumerate(vg):
                same = [i for i in range(j) if v_src[vg[i]] is v_src[k]]  # (chi gathered by row and by col: one transposition)
                vt = vts[same[0]] if same else v_src[k].transpose(1, 2).contiguous()  # [n, 3, V]
                wvs.append(vc["wk"][j]); vts.append(vt)
                vadds.append(_rows_matmul_small(vt.view(-1, chans[k]), vc["wkt"][j]).view(vt.shape[0], 3, hfp))
            wd_rest, wf_rest = vc["wd_rest"], vc["wf_rest"]
        # (its packed image -- a view over scalar_out.weight -- is cached next to the block's own, per weight version: the
        # forward and the backward of a step share it, and so do the steps between two optimizer updates)
        sub_cache = spec.pack_cache.setdefault("projected", {}) if spec.pack_cache is not None and use_view else None
        spec2 = replace(spec, si=sum(dims[k] for k in rest), s_plans=[spec.s_plans[k] for k in rest], pack_cache=sub_cache,
                        add_plans=[spec.s_plans[k] for k in sg], vi=sum(chans[k] for k in vr),
                        v_plans=[spec.v_plans[k] for k in vr], vadd_plans=[spec.v_plans[k] for k in vg],
                        w_view=(w_scalar, segs) if use_view else None)
        w2 = (w_rest, b_scalar, wd_rest, wf_rest, w_up, w_gate, b_gate)
        need_grad = any(ctx.needs_input_grad)
        rows, s_out, v_out, pack, s_pre, gate = _gcp2_forward_launch(spec2, frames, [s_src[k] for k in rest],
                                                                     [v_src[k] for k in vr], res_s, res_v, w2, adds, vadds,
                                                                     need_grad)
        if need_grad:
            ctx.spec, ctx.spec2, ctx.rows, ctx.frames, ctx.sg, ctx.vg = spec, spec2, rows, frames, list(sg), list(vg)
            ctx.has_res = (res_s is not None, res_v is not None)
            ctx.w_leaf = not spec.shared_weights
            ctx.weights = w
            ctx.use_cells = _note_uses(w)
            ctx.n_w2 = [t is not None for t in w2]
            ctx.s_pre_tb = isinstance(s_pre, TileBlocked)
            ctx.save_for_backward(*s_src, *v_src, *[t for t in w2 if t is not None], pack, s_pre.data if ctx.s_pre_tb else s_pre, gate,
                                  *wvs, *vts, *vadds)
        if spec.vo:
            return s_out, v_out
        return s_out

    @staticmethod
    def backward(ctx, d_s_out, d_v_out=None):
        spec, spec2, rows, sg, vg = ctx.spec, ctx.spec2, ctx.rows, ctx.sg, ctx.vg
        n_s, n_v = len(spec.s_plans), len(spec.v_plans)
        saved = list(ctx.saved_tensors)
        s_src, v_src = saved[:n_s], saved[n_s:n_s + n_v]
        pos = n_s + n_v
        w2 = []
        for present in ctx.n_w2:
            w2.append(saved[pos] if present else None)
            pos += 1 if present else 0
        w2 = tuple(w2)
        pack, s_pre, gate = saved[pos:pos + 3]
        if getattr(ctx, "s_pre_tb", False):
            s_pre = TileBlocked(rows, spec.so, s_pre.device, owner=s_pre, offset=0, n=s_pre.numel())
        pos += 3
        wvs = saved[pos:pos + len(vg)]; pos += len(vg)
        vts = saved[pos:pos + len(vg)]; pos += len(vg)
        vadds = saved[pos:pos + len(vg)]
        f32 = dict(dtype=torch.float32, device=s_pre.device)
        d_s_out = _req(d_s_out, "grad") if d_s_out is not None else torch.zeros((rows, spec.so), **f32)
        if spec.vo:
            d_v_out = _req(d_v_out, "grad") if d_v_out is not None else torch.zeros((rows, spec.vo, 3), **f32)
        rest = [k for k in range(n_s) if k not in sg]
        vr = [k for k in range(n_v) if k not in vg]
        dims = [t.shape[1] for t in s_src]
        offs = [sum(dims[:k]) for k in range(n_s)]
        chans = [t.shape[1] for t in v_src]
        voffs = [sum(chans[:k]) for k in range(n_v)]
        base = 4  # index of the first tensor input in needs_input_grad
        need_w = ctx.needs_input_grad[base + n_s + n_v + 2:base + n_s + n_v + 9]
        s_rest, v_rest = [s_src[k] for k in rest], [v_src[k] for k in vr]
        d_s_in, d_v_in, scr = gcp2_backward_data(spec2, rows, s_rest, v_rest, ctx.frames, w2, pack, s_pre, gate, d_s_out,
                                                 d_v_out, need_w=any(need_w), vadds=vadds)
        ds_pre = scr["ds_pre"]
        # ---- input gradients ---------------------------------------------------------------------------------------------
        grads_s: List[Optional[Tensor]] = [None] * n_s
        off2 = 0
        for k in rest:
            pl = spec.s_plans[k]
            if pl is not None:
                grads_s[k] = _segment_reduce_raw(d_s_in, off2, dims[k], spec2.si, pl, False)
            else:
                grads_s[k] = d_s_in if len(rest) == 1 else d_s_in[:, off2:off2 + dims[k]]
            off2 += dims[k]
        dP = []
        w_scalar = ctx.weights[0]
        for k in sg:  # projected sources: d(table) = ds_pre summed over the gathering rows, then the Linear's adjoint (W_k^T view)
            pl = spec.s_plans[k]
            g = ds_pre if pl is None else _segment_reduce_raw(ds_pre, 0, spec.so, spec.so, pl, False)
            dP.append(g)
            if ctx.needs_input_grad[base + k]:
                dx = wg_linear(g, w_scalar, dims[k], spec.so, col0=offs[k], trans=True)
                if dx is None:
                    raise _lib.GcpnetHipError("node-level projection adjoint outside the workgroup kernel's shapes (no library fallback)")
                grads_s[k] = dx
        grads_v: List[Optional[Tensor]] = [None] * n_v
        off2 = 0
        for k in vr:
            pl = spec.v_plans[k]
            if pl is not None:
                grads_v[k] = _segment_reduce_raw(d_v_in, 3 * off2, 3 * chans[k], 3 * spec2.vi, pl, False).reshape(pl.n_src, chans[k], 3)
            else:
                grads_v[k] = d_v_in if len(vr) == 1 else d_v_in[:, off2:off2 + chans[k], :]
            off2 += chans[k]
        dQ = []
        for k, wk in zip(vg, wvs):
            pl = spec.v_plans[k]
            dq = scr["dvhf"]
            g = dq if pl is None else _segment_reduce_raw(dq, 0, dq.shape[1], dq.shape[1], pl, False)  # [n, 3 HF']
            dQ.append(g)
            if ctx.needs_input_grad[base + n_s + k]:
                hfp = wk.shape[0]
                grads_v[k] = _rows_matmul_small(g.view(-1, hfp), wk).view(g.shape[0], 3, chans[k]).transpose(1, 2)
        # ---- weight gradients: un-gathered columns from the kernel's operands, projected columns by row-split TN GEMMs, all
        #      assembled into the full-size tensors on the weight-gradient stream ------------------------------------------
        wgrads: List[Optional[Tensor]] = [None] * 7
        if any(need_w):
            H = spec.hidden
            job = _WeightGradJob(spec2, rows, s_rest, s_pre, scr)
            gj = job.grads()
            g0 = torch.empty((spec.so, spec.K), **f32)
            gd = torch.empty((H, spec.vi), **f32) if vg else gj[2]
            gf = torch.empty((3, spec.vi), **f32) if vg else gj[3]

            def assemble():
                run_weight_grad_jobs([job])
                c = 0
                cjobs = []
                for k in rest:
                    cjobs.append((g0[:, offs[k]:offs[k] + dims[k]], gj[0][:, c:c + dims[k]]))
                    c += dims[k]
                cjobs.append((g0[:, spec.si:], gj[0][:, c:]))
                # the projected sources' weight gradients (row-split products over the SOURCE rows): one launch for all of them
                items = [(g, s_src[k], g0[:, offs[k]:offs[k] + dims[k]]) for k, g in zip(sg, dP)]
                tmps = []
                if vg:
                    c = 0
                    for k in vr:
                        cjobs += [(gd[:, voffs[k]:voffs[k] + chans[k]], gj[2][:, c:c + chans[k]]),
                                  (gf[:, voffs[k]:voffs[k] + chans[k]], gj[3][:, c:c + chans[k]])]
                        c += chans[k]
                    for k, g, vt in zip(vg, dQ, vts):
                        hfp = g.shape[1] // 3
                        tmp = torch.empty((hfp, chans[k]), **f32)  # [HF', V]
                        tmps.append((k, tmp))
                        items.append((g.view(-1, hfp), vt.view(-1, chans[k]), tmp))
                _tn_weight_grads_into(items)
                for k, tmp in tmps:
                    cjobs += [(gd[:, voffs[k]:voffs[k] + chans[k]], tmp[:H]), (gf[:, voffs[k]:voffs[k] + chans[k]], tmp[H:H + 3])]
                copy2d_multi(cjobs)  # (every piece of the assembled gradients in one launch)

            keep = [job.keep, dP, dQ, list(s_src), list(vts), scr]
            if ctx.w_leaf and _side_stream_ok(ctx.weights, _take_use_cells(ctx)):
                _side_submit(assemble, keep)
            else:
                assemble()
            full = [g0, gj[1], gd, gf, gj[4], gj[5], gj[6]]
            wgrads = [g if need else None for g, need in zip(full, need_w)]
        g_res_s = d_s_out if ctx.has_res[0] else None
        g_res_v = d_v_out if ctx.has_res[1] else None
        _flush_deferred_weight_grads()
        return (None, None, None, None, *grads_s, *grads_v, g_res_s, g_res_v, *wgrads)


class _Linear(torch.autograd.Function):
    """nn.Linear on node rows (phi_force_i / phi_force_j of the position update, gcpnet.py:1052-1056): forward and input gradient
    through the workgroup kernel (wg_linear), weight / bias gradient through the row-split TN GEMM."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        out = wg_linear(x, weight, weight.shape[0], weight.shape[1], bias=bias)
        if out is None:
            raise _lib.GcpnetHipError("Linear outside the workgroup kernel's shapes (no library fallback)")
        ctx.save_for_backward(x, weight)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, weight = ctx.saved_tensors
        g = _req(g, "grad")
        so, dim = weight.shape
        dx = None
        if ctx.needs_input_grad[0]:
            dx = wg_linear(g, weight, dim, so, trans=True)
            if dx is None:
                raise _lib.GcpnetHipError("Linear adjoint outside the workgroup kernel's shapes (no library fallback)")
        dw = db = None
        if (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]) and x.shape[0] == 0:
            dw = torch.zeros((so, dim), dtype=torch.float32, device=g.device)
            db = torch.zeros((so,), dtype=torch.float32, device=g.device)
        elif ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            n = x.shape[0]
            a, b = Operand(), Operand()
            a.n, b.n = 1, 1
            a.ptr[0], a.dim[0], a.ld[0] = g.data_ptr(), so, so
            b.ptr[0], b.dim[0], b.ld[0] = x.data_ptr(), dim, x.stride(0)
            b.ones = 1
            dw = torch.empty((so, dim), dtype=torch.float32, device=g.device)
            db = torch.empty((so,), dtype=torch.float32, device=g.device)
            pr = TnProblem()
            pr.rows, pr.a, pr.b = n, a, b
            pr.out, pr.out_sm, pr.out_sn, pr.out_m, pr.out_n = dw.data_ptr(), dim, 1, so, dim
            pr.out2, pr.out2_n = db.data_ptr(), dim
            pr.splits = lib.gcpnet_tn_splits(n, so, dim + 1)
            part = torch.empty((pr.splits, so, dim + 1), dtype=torch.float32, device=g.device)
            pr.partial = part.data_ptr()
            check(lib.gcpnet_tn_gemm(1, C.byref(pr), _stream()), "tn_gemm")
        return dx, dw, db


def linear(x: Tensor, weight: Tensor, bias: Tensor) -> Tensor:
    return _Linear.apply(_req(x, "x"), _req(weight, "weight"), _req(bias, "bias"))


def linear_padded(x: Tensor, weight: Tensor, bias: Tensor) -> Tensor:
    """`linear` for any output width: widths that are not multiples of 4 (a 1-wide regression head, a 20-wide vocabulary is
    fine) are padded with zero rows for the launch and sliced back."""
    out = weight.shape[0]
    pad = (-out) % 4
    if not pad:
        return linear(x, weight, bias)
    w = torch.nn.functional.pad(weight, (0, 0, 0, pad))
    return linear(x, w, torch.nn.functional.pad(bias, (0, pad)))[:, :out]


class _Activation(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act, slope):
        lib = _lib.load()
        y = torch.empty_like(x)
        check(lib.gcpnet_activation(x.numel(), _p(x), None, ACT[act], float(slope), _p(y), _stream()), "activation")
        ctx.save_for_backward(x)
        ctx.cfg = (act, float(slope))
        return y

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        (x,) = ctx.saved_tensors
        g = _req(g, "grad")
        dx = torch.empty_like(x)
        check(lib.gcpnet_activation(x.numel(), _p(x), _p(g), ACT[ctx.cfg[0]], ctx.cfg[1], _p(dx), _stream()), "activation")
        return dx, None, None


def activation(x: Tensor, act: Optional[str], slope: float = 1e-2) -> Tensor:
    """get_nonlinearity(act)(x) (models/__init__.py:42-57); identity for None."""
    return x if act is None else _Activation.apply(_req(x, "x"), act, slope)


class _FrameGate(torch.autograd.Function):
    """v_out = vu * act_v(safe_norm(vector_up_frames(vectorize(g)))) per row (components/gcpnet.py:369-384), see gcpnet_frame_gate_*."""

    @staticmethod
    def forward(ctx, g, frames, w_up_frames, vu, act, slope):
        lib = _lib.load()
        rows, vo = vu.shape[0], vu.shape[1]
        out = torch.empty_like(vu)
        check(lib.gcpnet_frame_gate_forward(rows, vo, _p(g), g.shape[1], _p(frames), _p(w_up_frames), _p(vu), ACT[act], float(slope),
                                            _p(out), _stream()), "frame_gate_forward")
        ctx.save_for_backward(g, frames, w_up_frames, vu)
        ctx.cfg = (act, float(slope))
        return out

    @staticmethod
    def backward(ctx, d_out):
        lib = _lib.load()
        g, frames, wf, vu = ctx.saved_tensors
        rows, vo = vu.shape[0], vu.shape[1]
        f32 = dict(dtype=torch.float32, device=vu.device)
        d_out = _req(d_out, "grad")
        d_vu, d_g = torch.empty_like(vu), torch.empty_like(g)
        n_parts = max(int(lib.gcpnet_frame_gate_bwd_parts(rows)), 1)
        part = torch.zeros((n_parts, vo * 3), **f32)
        check(lib.gcpnet_frame_gate_backward(rows, vo, _p(g), g.shape[1], _p(frames), _p(wf), _p(vu), ACT[ctx.cfg[0]], ctx.cfg[1],
                                             _p(d_out), _p(d_vu), _p(d_g), _p(part), _stream()), "frame_gate_backward")
        d_w = torch.zeros((vo * 3,), **f32)
        if rows:
            tmp = torch.empty((lib.gcpnet_reduce_partials_groups(n_parts), vo * 3), **f32)
            job = ReduceJob()
            job.parts, job.n_parts, job.width, job.tmp, job.out = part.data_ptr(), n_parts, vo * 3, tmp.data_ptr(), d_w.data_ptr()
            check(lib.gcpnet_reduce_partials(1, C.byref(job), _stream()), "reduce_partials")
        return d_g, None, d_w.view(vo, 3), d_vu, None, None


def frame_gate(g: Tensor, frames: Tensor, w_up_frames: Tensor, vu: Tensor, act: Optional[str], slope: float = 1e-2) -> Tensor:
    return _FrameGate.apply(_req(g, "gate scalars"), _req(frames.detach(), "frames"), _req(w_up_frames, "vector_up_frames.weight"),
                            _req(vu, "vector_up output"), act, slope)


class _NodeScalarize(torch.autograd.Function):
    """scalarize(vf, node_inputs=True, enable_e3_equivariance) (components/__init__.py:283-321): per out-edge projection of the
    node's three frame channels, |.| on the x_cross axis, mean over the out-edges.  vf: [N, 3 (xyz), ldk >= 3]; frames [E, 3, 3]
    are constants of the step."""

    @staticmethod
    def forward(ctx, vf, frames, plan: GatherPlan, e3: bool):
        lib = _lib.load()
        n, ldk = vf.shape[0], vf.shape[2]
        out = torch.empty((n, 9), dtype=torch.float32, device=vf.device)
        check(lib.gcpnet_node_scalarize(n, _p(plan.seg_ptr), _p(plan.perm), _p(vf), ldk, _p(frames), int(e3), _p(out), None, None,
                                        _stream()), "node_scalarize")
        ctx.save_for_backward(vf, frames)
        ctx.plan, ctx.e3 = plan, bool(e3)
        return out

    @staticmethod
    def backward(ctx, d_out):
        lib = _lib.load()
        vf, frames = ctx.saved_tensors
        d_out = _req(d_out, "grad")
        d_vf = torch.zeros_like(vf)
        check(lib.gcpnet_node_scalarize(vf.shape[0], _p(ctx.plan.seg_ptr), _p(ctx.plan.perm), _p(vf), vf.shape[2], _p(frames),
                                        int(ctx.e3), None, _p(d_out), _p(d_vf), _stream()), "node_scalarize backward")
        return d_vf, None, None, None


def node_scalarize(vf: Tensor, frames: Tensor, plan: GatherPlan, e3: bool = True) -> Tensor:
    """[N, 9] frame scalars of node rows from per-node frame channels vf [N, 3 (xyz), >= 3 (channel)] and the edges' frames;
    `plan` = the CSR of the edges by SOURCE node (GraphPlan.row)."""
    assert vf.dim() == 3 and vf.shape[1] == 3 and vf.shape[2] >= 3 and plan.n_src == vf.shape[0]
    return _NodeScalarize.apply(_req(vf, "frame channels"), _req(frames.detach(), "frames"), plan, e3)


class _EdgeForce(torch.autograd.Function):
    """force[e] = sum_k coef[e, k] f_ij[e, k, :] with coef = W3 act(A[row] + B[col]) (reference gcpnet.py:1143-1150).
    A, B: per-node tables [N, s]; W3 [3, s]; frames [E, 3, 3] (constants of the step)."""

    @staticmethod
    def forward(ctx, A, B, W3, frames, plan: GraphPlan, act, slope: float):
        lib = _lib.load()
        E, s = plan.n_edges, A.shape[1]
        force = torch.empty((E, 3), dtype=torch.float32, device=A.device)
        check(lib.gcpnet_edge_force_forward(E, s, _p(A), _p(B), _p(plan.row.idx), _p(plan.col.idx), _p(W3), _p(frames),
                                            ACT[act], float(slope), _p(force), _stream()), "edge_force_forward")
        ctx.save_for_backward(A, B, W3, frames)
        ctx.plan, ctx.act, ctx.slope = plan, act, float(slope)
        return force

    @staticmethod
    def backward(ctx, d_force):
        lib = _lib.load()
        A, B, W3, frames = ctx.saved_tensors
        plan = ctx.plan
        E, s = plan.n_edges, A.shape[1]
        f32 = dict(dtype=torch.float32, device=A.device)
        d_force = _req(d_force, "grad")
        d_pre = torch.empty((E, s), **f32)
        nb = lib.gcpnet_edge_force_bwd_blocks(E)
        part = torch.empty((nb, 3 * s), **f32)
        check(lib.gcpnet_edge_force_backward(E, s, _p(A), _p(B), _p(plan.row.idx), _p(plan.col.idx), _p(W3), _p(frames),
                                             ACT[ctx.act], ctx.slope, _p(d_force), _p(d_pre), _p(part), _stream()),
              "edge_force_backward")
        dA = _segment_reduce_raw(d_pre, 0, s, s, plan.row, False)
        dB = _segment_reduce_raw(d_pre, 0, s, s, plan.col, False)
        dW3 = torch.empty((3, s), **f32)
        tmp = torch.empty((lib.gcpnet_reduce_partials_groups(nb), 3 * s), **f32)
        job = ReduceJob()
        job.parts, job.n_parts, job.width, job.tmp, job.out = part.data_ptr(), nb, 3 * s, tmp.data_ptr(), dW3.data_ptr()
        check(lib.gcpnet_reduce_partials(1, C.byref(job), _stream()), "reduce_partials")
        return dA, dB, dW3, None, None, None, None


def edge_force(A: Tensor, B: Tensor, W3: Tensor, frames: Tensor, plan: GraphPlan, act, slope: float) -> Tensor:
    return _EdgeForce.apply(_req(A, "A"), _req(B, "B"), _req(W3, "W3"), _req(frames.detach(), "frames"), plan, act, slope)


class _RowGate(torch.autograd.Function):
    """out = x * sigmoid(x w^T + b) per row (reference gcpnet.py:932-934); x [rows, s], w [1, s], b [1]."""

    @staticmethod
    def forward(ctx, x, w, b):
        lib = _lib.load()
        rows, s = x.shape
        out = torch.empty_like(x)
        att = torch.empty((rows,), dtype=torch.float32, device=x.device)
        check(lib.gcpnet_row_gate_forward(rows, s, _p(x), _p(w), _p(b), _p(out), _p(att), _stream()), "row_gate_forward")
        ctx.save_for_backward(x, w, att)
        return out

    @staticmethod
    def backward(ctx, d_out):
        lib = _lib.load()
        x, w, att = ctx.saved_tensors
        rows, s = x.shape
        f32 = dict(dtype=torch.float32, device=x.device)
        d_out = _req(d_out, "grad")
        d_x = torch.empty_like(x)
        nb = lib.gcpnet_row_gate_bwd_blocks(rows)
        part = torch.empty((nb, s + 4), **f32)
        check(lib.gcpnet_row_gate_backward(rows, s, _p(x), _p(w), _p(att), _p(d_out), _p(d_x), _p(part), _stream()),
              "row_gate_backward")
        dwb = torch.zeros((s + 4,), **f32)
        if rows:
            tmp = torch.empty((lib.gcpnet_reduce_partials_groups(nb), s + 4), **f32)
            job = ReduceJob()
            job.parts, job.n_parts, job.width, job.tmp, job.out = part.data_ptr(), nb, s + 4, tmp.data_ptr(), dwb.data_ptr()
            check(lib.gcpnet_reduce_partials(1, C.byref(job), _stream()), "reduce_partials")
        return d_x, dwb[:s].view(1, s), dwb[s:s + 1]


def row_gate(x: Tensor, w: Tensor, b: Tensor) -> Tensor:
    return _RowGate.apply(_req(x, "x"), _req(w, "w"), _req(b, "b"))


class _SplitCols(torch.autograd.Function):
    """x [n, w] -> (x[:, :cut] as a view, x[:, cut:] as a contiguous copy).  The backward writes the two gradient pieces into ONE
    fresh [n, w] tensor.  Plain slicing leaves this to autograd, which builds a zero-filled [n, w] tensor per slice and adds them:
    for the second feed-forward GCP of configs[4] ([100 000, 1 024], cut = 896) ~2.8 GB of traffic per layer instead of 0.8."""

    @staticmethod
    def forward(ctx, x, cut):
        ctx.cut, ctx.shape = int(cut), tuple(x.shape)
        return x[:, :cut], x[:, cut:].contiguous()

    @staticmethod
    def backward(ctx, g_left, g_right):
        cut = ctx.cut
        ref = g_left if g_left is not None else g_right
        g = torch.empty(ctx.shape, dtype=ref.dtype, device=ref.device)
        if g_left is not None:
            g[:, :cut].copy_(g_left)
        else:
            g[:, :cut].zero_()
        if g_right is not None:
            g[:, cut:].copy_(g_right)
        else:
            g[:, cut:].zero_()
        return g, None


class _Project(torch.autograd.Function):
    """P = x @ w^T for x [n, dim] (contiguous) and w [so, dim] (a column slice of scalar_out.weight is fine): the leading columns
    of a block input too wide for one workgroup-kernel launch (the (1049 -> 256) feed-forward block of BASELINE configs[4]).
    Forward and input gradient: the workgroup kernel's plain-Linear form (wg_linear; a library GEMM until round 5); the weight
    gradient dP^T x reduces over the n rows and goes through gcpnet_tn_gemm (row-split, deterministic)."""

    @staticmethod
    def forward(ctx, x, w):
        x = x.contiguous()  # (the leading columns of a wider tensor: the workgroup kernel reads rows of exactly `dim` floats)
        ctx.save_for_backward(x, w)
        base, col0 = _Project._stored(w)
        out = wg_linear(x, base, w.shape[0], w.shape[1], col0=col0)
        if out is None:
            raise _lib.GcpnetHipError("projection outside the workgroup kernel's shapes (no library fallback)")
        return out

    @staticmethod
    def _stored(w):
        """(stored matrix, first column) of a column-slice view: wg_linear caches the packed image per stored tensor object."""
        base = w._base
        if base is not None and base.dim() == 2 and base.stride(1) == 1 and w.stride() == base.stride():
            col0 = w.storage_offset() - base.storage_offset()
            if 0 <= col0 and col0 + w.shape[1] <= base.shape[1] and w.shape[0] == base.shape[0]:
                return base, col0
        return w, 0

    @staticmethod
    def backward(ctx, dP):
        x, w = ctx.saved_tensors
        dx = None
        if ctx.needs_input_grad[0]:
            base, col0 = _Project._stored(w)
            dx = wg_linear(_req(dP, "grad"), base, w.shape[1], w.shape[0], col0=col0, trans=True)
            if dx is None:
                raise _lib.GcpnetHipError("projection adjoint outside the workgroup kernel's shapes (no library fallback)")
        dw = None
        if ctx.needs_input_grad[1]:
            dP = _req(dP, "grad")
            n, so = dP.shape
            dim = x.shape[1]
            if n == 0:
                dw = torch.zeros((so, dim), dtype=torch.float32, device=dP.device)
            else:
                lib = _lib.load()
                a, b = Operand(), Operand()
                a.n, b.n = 1, 1
                a.ptr[0], a.dim[0], a.ld[0] = dP.data_ptr(), so, so
                b.ptr[0], b.dim[0], b.ld[0] = x.data_ptr(), dim, x.stride(0)
                dw = torch.empty((so, dim), dtype=torch.float32, device=dP.device)
                pr = TnProblem()
                pr.rows, pr.a, pr.b = n, a, b
                pr.out, pr.out_sm, pr.out_sn, pr.out_m, pr.out_n = dw.data_ptr(), dim, 1, so, dim
                pr.out2, pr.out2_n = None, 0
                pr.splits = lib.gcpnet_tn_splits(n, so, dim)
                part = torch.empty((pr.splits, so, dim), dtype=torch.float32, device=dP.device)
                pr.partial = part.data_ptr()
                check(lib.gcpnet_tn_gemm(1, C.byref(pr), _stream()), "tn_gemm")
        return dx, dw


LDS_LIMIT = 160 * 1024




def _too_wide(spec: Gcp2Spec, s_sources, rows: int):
    """(source index, leading columns to project), see gcp2(); None to leave the block as it is."""
    lib = _lib.load()
    need = lambda si: lib.gcpnet_gcp2_forward_lds_bytes(si, spec.vi, spec.so, spec.vo, spec.hidden, int(spec.use_frames))
    if spec.residual or spec.add_plans:
        return None
    dims = [t.shape[1] for t in s_sources]
    k = max(range(len(dims)), key=lambda i: dims[i])
    if need(spec.si) <= LDS_LIMIT:
        return None
    cut = 0
    while cut + 32 < dims[k] and need(spec.si - cut) > LDS_LIMIT // 2:  # leave room for two waves per CU
        cut += 32
    return (k, cut) if cut > 0 and need(spec.si - cut) <= LDS_LIMIT else None


PROJECT_GATHERED_VECTORS = True


def _v_projectable(spec: Gcp2Spec, v_sources):
    """Gathered vector sources worth projecting at their source rows (see gcp2()); the kernels' MFMA form of the vector
    stage must apply (H + 3 <= 32, vo <= 32), frames in use, and at least one un-gathered source must remain."""
    if not PROJECT_GATHERED_VECTORS or not spec.use_frames or spec.vi == 0 or spec.vo == 0 or spec.vo > 32:
        return None
    if spec.hidden + 3 > 32 or spec.vector_residual or spec.residual:
        return None
    gath = [k for k, pl in enumerate(spec.v_plans) if pl is not None and pl.rows >= 2 * v_sources[k].shape[0]]
    rest = [k for k in range(len(v_sources)) if k not in gath]
    if not gath or not rest:
        return None
    return gath, rest


def _projectable(spec: Gcp2Spec, s_sources):
    """Which scalar sources are worth projecting at their source rows: gathered ones read by >= 2x more rows than they
    have; needs a single output group (so <= 128) and at least one source left for the kernel's own reduction."""
    if spec.residual or spec.add_plans:
        return None
    gath = [k for k, pl in enumerate(spec.s_plans) if pl is not None and pl.rows >= 2 * s_sources[k].shape[0]]
    rest = [k for k in range(len(s_sources)) if k not in gath]
    if not gath or not rest:
        return None
    return gath, rest
