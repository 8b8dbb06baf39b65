"""Host-side mirror of the reference's `src/models/components/gcpnet.py` hot path:
GCP2, GCPEmbedding, get_GCP_with_custom_cfg, GCPMessagePassing, GCPInteractions.

Same constructor / forward signatures, same parameter (state_dict) names, same config keys, same Python error
behaviour; the arithmetic is launched on the MI355X through gcpnet_amd.ops.  Citations are file:line in
/root/reference/src/models/components/gcpnet.py.
"""
from __future__ import annotations

from copy import copy
from dataclasses import replace
from functools import partial
from typing import Any, List, Optional, Sequence, Tuple, Union

import torch
from torch import nn

from . import ops
from ._lib import VMODE_NONE, VMODE_SCALAR_GATE, VMODE_SELF_GATE
from .components import GCPDropout, GCPLayerNorm, ScalarVector, canonical_act, edge_mask_of, mask_frames
from .config import to_container
from .ops import GatherPlan, Gcp2Spec, GraphPlan

NUM_ATOM_TYPES = 9  # src/datamodules/components/atom3d_dataset.py:20-33


def _sv_add(a, b) -> "ScalarVector":
    return ScalarVector(ops.axpy(a[0], b[0], 1.0), ops.axpy(a[1], b[1], 1.0))


def _unsupported(what: str):
    raise NotImplementedError(f"{what} is not on the MI355X path yet (SURVEY.md section 8, rows f2/f3)")


def _node_e3_frame_scalars(v: torch.Tensor, w_frames: torch.Tensor, edge_index, frames) -> torch.Tensor:
    """scalarize(vector_down_frames(v), node_inputs=True, enable_e3_equivariance=True) (components/__init__.py:283-321) -> [N, 9].
    The |.| on the x_cross projections is taken per out-edge BEFORE the mean, so the mean out-edge frame of the node-row kernels
    does not apply: the three frame channels per node come from the Linear kernel, the per-edge projections / |.| / mean from
    gcpnet_node_scalarize."""
    n, vi = v.shape[0], v.shape[1]
    x2 = v.transpose(1, 2).reshape(n * 3, vi)
    w4 = torch.nn.functional.pad(w_frames, (0, 0, 0, 1))  # 3 -> 4 output channels (16-byte rows)
    vf = ops.linear(x2, w4, w_frames.new_zeros(4)).view(n, 3, 4)
    return ops.node_scalarize(vf, frames, GraphPlan.get(edge_index, n).row, e3=True)


class GCP2(nn.Module):
    """Geometry-complete perceptron, :252-468.  Parameters are created in the reference's order so that the same
    seed yields the same initial weights."""

    def __init__(
        self,
        input_dims,
        output_dims,
        nonlinearities: Tuple[Optional[str]] = ("relu", "sigmoid"),
        scalar_gate: int = 0,
        vector_gate: bool = True,
        frame_gate: bool = False,
        sigma_frame_gate: bool = False,
        bottleneck: int = 1,
        vector_residual: bool = False,
        vector_frame_residual: bool = False,
        ablate_frame_updates: bool = False,
        ablate_scalars: bool = False,
        ablate_vectors: bool = False,
        enable_e3_equivariance: bool = False,
        scalarization_vectorization_output_dim: int = 3,
        **kwargs,
    ):
        super().__init__()
        if nonlinearities is None:
            nonlinearities = (None, None)
        self.scalar_input_dim, self.vector_input_dim = input_dims
        self.scalar_output_dim, self.vector_output_dim = output_dims
        self.act_s, self.act_v = canonical_act(nonlinearities[0]), canonical_act(nonlinearities[1])
        self.scalar_gate, self.vector_gate, self.frame_gate, self.sigma_frame_gate = (
            scalar_gate, vector_gate, frame_gate, sigma_frame_gate)
        self.vector_residual, self.vector_frame_residual = vector_residual, vector_frame_residual
        self.ablate_frame_updates = ablate_frame_updates
        self.ablate_scalars, self.ablate_vectors = ablate_scalars, ablate_vectors
        self.enable_e3_equivariance = enable_e3_equivariance
        self.slope = 1e-2
        if scalarization_vectorization_output_dim != 3:
            _unsupported("scalarization_vectorization_output_dim != 3")

        if self.scalar_gate > 0:  # created but never used by the reference's GCP2.forward (:290-291)
            self.norm = nn.LayerNorm(self.scalar_output_dim)

        self.hidden_dim = 0
        if self.vector_input_dim:
            assert self.vector_input_dim % bottleneck == 0, (
                f"Input channel of vector ({self.vector_input_dim}) must be divisible with bottleneck factor ({bottleneck})")
            self.hidden_dim = (self.vector_input_dim // bottleneck if bottleneck > 1
                               else max(self.vector_input_dim, self.vector_output_dim))
            frame_dim = 9 if not ablate_frame_updates else 0
            self.vector_down = nn.Linear(self.vector_input_dim, self.hidden_dim, bias=False)
            self.scalar_out = self._make_scalar_out(self.hidden_dim + self.scalar_input_dim + frame_dim, self.scalar_output_dim)
            if not ablate_frame_updates:
                self.vector_down_frames = nn.Linear(self.vector_input_dim, 3, bias=False)
            if self.vector_output_dim:
                self.vector_up = nn.Linear(self.hidden_dim, self.vector_output_dim, bias=False)
                if not ablate_frame_updates and self.frame_gate:
                    self.vector_out_scale_frames = nn.Linear(self.scalar_output_dim, 9)
                    self.vector_up_frames = nn.Linear(3, self.vector_output_dim, bias=False)
                elif self.vector_gate:
                    self.vector_out_scale = nn.Linear(self.scalar_output_dim, self.vector_output_dim)
        else:
            self.scalar_out = self._make_scalar_out(self.scalar_input_dim, self.scalar_output_dim)
        self._pack_cache: dict = {}

    def _make_scalar_out(self, in_dim: int, out_dim: int) -> nn.Module:
        return nn.Linear(in_dim, out_dim)

    # ---- kernel dispatch ------------------------------------------------------------------------------------
    def _vmode(self) -> int:
        if not (self.vector_input_dim and self.vector_output_dim):
            return VMODE_NONE
        if self.frame_gate and not self.ablate_frame_updates:
            return VMODE_NONE  # (the frame gate runs as its own kernel behind the block: _apply_rows)
        if self.vector_gate:
            return VMODE_SCALAR_GATE
        return VMODE_SELF_GATE if self.act_v is not None else VMODE_NONE

    def _weights(self):
        # (straight from the registries: nn.Module.__getattr__ costs ~0.3 us per hop and this runs for every block of every step --
        # 40 calls x 12 lookups per configs[1] step, which is host-bound)
        mods = self._modules
        so = mods["scalar_out"]._parameters
        if "weight" in so:
            def w(name):
                m = mods.get(name)
                return None if m is None else m._parameters.get("weight")
            gate = mods.get("vector_out_scale")
            gp = None if gate is None else gate._parameters
            return (so["weight"], so.get("bias"), w("vector_down"), w("vector_down_frames"), w("vector_up"),
                    None if gp is None else gp.get("weight"), None if gp is None else gp.get("bias"))
        g = lambda name: getattr(self, name).weight if hasattr(self, name) else None
        gate = getattr(self, "vector_out_scale", None)
        return (self.scalar_out.weight, self.scalar_out.bias, g("vector_down"), g("vector_down_frames"), g("vector_up"),
                None if gate is None else gate.weight, None if gate is None else gate.bias)

    def make_spec(self, s_plans, v_plans, residual: bool = False) -> Gcp2Spec:
        use_frames = bool(self.vector_input_dim) and not self.ablate_frame_updates
        return Gcp2Spec(
            si=self.scalar_input_dim, vi=self.vector_input_dim, so=self.scalar_output_dim, vo=self.vector_output_dim,
            hidden=self.hidden_dim, use_frames=use_frames, act_s=self.act_s, act_v=self.act_v, slope=self.slope,
            vmode=self._vmode(), vector_residual=bool(self.vector_residual) and bool(self.vector_input_dim),
            e3=bool(self.enable_e3_equivariance), s_plans=list(s_plans), v_plans=list(v_plans), residual=residual,
            pack_cache=self._pack_cache)

    def apply_rows(self, s_sources: Sequence[torch.Tensor], s_plans: Sequence[Optional[GatherPlan]],
                   v_sources: Sequence[torch.Tensor], v_plans: Sequence[Optional[GatherPlan]],
                   row_frames: Optional[torch.Tensor], residual: bool = False):
        """Runs the block on rows whose inputs are concatenations of (optionally gathered) sources.  `row_frames`
        holds one frame per row.  With `residual` the result is x + GCP(x) for the single source x (ResGCP)."""
        ablating = self.ablate_scalars or self.ablate_vectors
        if not ablating:
            return self._apply_rows(s_sources, s_plans, v_sources, v_plans, row_frames, residual)
        # ablations (:416-417, :466-467): the block sees zero inputs and returns zero outputs for the ablated kind; a ResGCP
        # (:921-924) still adds the block's result to the UN-ablated message, so the residual add is done outside the kernel
        zin = bool(self.vector_input_dim)  # (a block without vector input passes its scalars through un-ablated, :439-441)
        zs = [torch.zeros_like(t) for t in s_sources] if (self.ablate_scalars and zin) else s_sources
        zv = [torch.zeros_like(t) for t in v_sources] if (self.ablate_vectors and zin) else v_sources
        out = self._ablate_outputs(self._apply_rows(zs, s_plans, zv, v_plans, row_frames, False))
        if residual:
            if isinstance(out, tuple):
                return ops.axpy(s_sources[0], out[0], 1.0), ops.axpy(v_sources[0], out[1], 1.0)
            return ops.axpy(s_sources[0], out, 1.0)
        return out

    def _apply_rows(self, s_sources, s_plans, v_sources, v_plans, row_frames, residual: bool):
        if (self.frame_gate and not self.ablate_frame_updates and self.vector_input_dim and self.vector_output_dim
                and hasattr(self, "vector_up_frames")):
            return self._apply_rows_frame_gate(s_sources, s_plans, v_sources, v_plans, row_frames, residual)
        spec = self.make_spec(s_plans, v_plans, residual)
        return ops.gcp2(spec, s_sources, v_sources, row_frames if spec.use_frames else None, self._weights())

    def _apply_rows_frame_gate(self, s_sources, s_plans, v_sources, v_plans, row_frames, residual: bool):
        """`frame_gate: true` (:369-384; no shipped config sets it): the block runs un-gated and without scalar activation -- its
        outputs are then s_pre and vector_up(vh) (+ v) --, the two activations, the Linear(so -> 9) and the frame gate itself
        (vectorize -> vector_up_frames -> norm -> act_v -> scale) follow as their own HIP launches."""
        spec = replace(self.make_spec(s_plans, v_plans, False), act_s=None, act_v=None, vmode=VMODE_NONE)
        w = self._weights()
        s_pre, vu = ops.gcp2(spec, s_sources, v_sources, row_frames, (w[0], w[1], w[2], w[3], w[4], None, None))
        lin = self.vector_out_scale_frames
        w12 = torch.nn.functional.pad(lin.weight, (0, 0, 0, 3))  # 9 -> 12 output rows (16-byte rows of the gate scalars)
        b12 = torch.nn.functional.pad(lin.bias, (0, 3))
        g = ops.linear(ops.activation(s_pre, self.act_v, self.slope), w12, b12)
        v_out = ops.frame_gate(g, row_frames, self.vector_up_frames.weight, vu, self.act_v, self.slope)
        s_out = ops.activation(s_pre, self.act_s, self.slope)
        if residual:
            return ops.axpy(s_sources[0], s_out, 1.0), ops.axpy(v_sources[0], v_out, 1.0)
        return s_out, v_out

    def _ablate_outputs(self, out):
        """:443-446, :466-467 -- ablated outputs are zeros (of the right shape)."""
        if not (self.ablate_scalars or self.ablate_vectors):
            return out
        if not isinstance(out, tuple):  # no vector output: the nonlinearity is applied to the zeroed pre-activation (:443-446)
            if not self.ablate_scalars:
                return out
            return torch.full_like(out, 0.5) if self.act_s == "sigmoid" else torch.zeros_like(out)
        s_out, v_out = out
        if self.ablate_scalars:
            s_out = torch.zeros_like(s_out)
        if self.ablate_vectors:
            v_out = torch.zeros_like(v_out)
        return s_out, v_out

    def forward(self, s_maybe_v, edge_index, frames, node_inputs: bool = False, node_mask=None):
        """:394-468.  Returns ScalarVector, or a Tensor when the block has no vector output.  `node_mask` (scalarize / vectorize,
        components/__init__.py:295-302, 346-357): edges with a masked end point act through zero frames (components.mask_frames)."""
        frames = mask_frames(frames, edge_index, node_mask)
        if self.vector_input_dim:
            s, v = s_maybe_v
            row_frames = None
            if not self.ablate_frame_updates:
                if node_inputs:
                    if self.enable_e3_equivariance:
                        out = self._forward_node_e3(s, v, edge_index, frames)
                        return out if not self.vector_output_dim else ScalarVector(*out)
                    row_frames = GraphPlan.get(edge_index, s.shape[0]).node_frames(frames)
                else:
                    row_frames = frames
            out = self.apply_rows([s], [None], [v], [None], row_frames)
        else:
            out = self.apply_rows([s_maybe_v], [None], [], [], None)
        if not self.vector_output_dim:
            return out
        return ScalarVector(*out)

    def _forward_node_e3(self, s, v, edge_index, frames):
        """Node rows with `enable_e3_equivariance` (:421-432 -> components/__init__.py:305-309): the nine frame scalars are
        computed per out-edge (|.| before the mean) by `_node_e3_frame_scalars` and enter the block as a second scalar source; the
        block itself then runs without frames, with scalar_out's columns reordered to [s | frame scalars | norms]."""
        if getattr(self, "feedforward_out", False) or (self.frame_gate and self.vector_output_dim):
            _unsupported("enable_e3_equivariance on node rows together with feedforward_out / frame_gate")
        if self.ablate_scalars:
            s = torch.zeros_like(s)
        if self.ablate_vectors:
            v = torch.zeros_like(v)
        fs = _node_e3_frame_scalars(v, self.vector_down_frames.weight, edge_index, frames)
        si, H = self.scalar_input_dim, self.hidden_dim
        w = self.scalar_out.weight
        w_re = torch.cat((w[:, :si], w[:, si + H:], w[:, si:si + H]), dim=1)
        spec = replace(self.make_spec([None, None], [None]), si=si + 9, use_frames=False, e3=False, pack_cache=None,
                       shared_weights=True)
        ws = self._weights()
        out = ops.gcp2(spec, [s, fs], [v], None, (w_re, ws[1], ws[2], None, ws[4], ws[5], ws[6]))
        return self._ablate_outputs(out)


class GCP3(GCP2):
    """`GCP3` (:471-700) is `GCP2` with silu defaults and an optional two-layer `scalar_out` (`feedforward_out`, :529-533,
    :552-556).  Without `feedforward_out` the two are the same computation with the same parameter names.  With it,
    `scalar_out = Linear -> act -> Linear` runs as two launches of the same kernels: a scalar-only block (first Linear over
    [s | norms | frame projections], `scalar_out_nonlinearity`) and a block whose scalar input is that result (second Linear;
    its norm / frame columns carry zero weights) and which produces the gated vectors."""

    def __init__(self, input_dims, output_dims, nonlinearities: Tuple[Optional[str]] = ("silu", "silu"),
                 scalar_out_nonlinearity: Optional[str] = "silu", scalar_gate: int = 0, vector_gate: bool = True,
                 frame_gate: bool = False, sigma_frame_gate: bool = False, feedforward_out: bool = False, bottleneck: int = 1,
                 vector_residual: bool = False, vector_frame_residual: bool = False, ablate_frame_updates: bool = False,
                 ablate_scalars: bool = False, ablate_vectors: bool = False, enable_e3_equivariance: bool = False,
                 scalarization_vectorization_output_dim: int = 3, **kwargs):
        object.__setattr__(self, "_ff_out", bool(feedforward_out))  # (read by _make_scalar_out during the base constructor)
        super().__init__(input_dims, output_dims, nonlinearities=nonlinearities, scalar_gate=scalar_gate, vector_gate=vector_gate,
                         frame_gate=frame_gate, sigma_frame_gate=sigma_frame_gate, bottleneck=bottleneck,
                         vector_residual=vector_residual, vector_frame_residual=vector_frame_residual,
                         ablate_frame_updates=ablate_frame_updates, ablate_scalars=ablate_scalars, ablate_vectors=ablate_vectors,
                         enable_e3_equivariance=enable_e3_equivariance,
                         scalarization_vectorization_output_dim=scalarization_vectorization_output_dim, **kwargs)
        self.scalar_out_nonlinearity = scalar_out_nonlinearity
        self.feedforward_out = bool(feedforward_out)
        if self.feedforward_out:
            self.act_mid = canonical_act(scalar_out_nonlinearity)
            self._pack_cache_first: dict = {}

    def _make_scalar_out(self, in_dim: int, out_dim: int) -> nn.Module:
        """:529-533 / :552-556 -- both Linears are created here, at the reference's position in the construction order (same
        state_dict keys AND the same RNG consumption, so the same seed gives the same initial weights)."""
        if not self._ff_out:
            return nn.Linear(in_dim, out_dim)
        return nn.Sequential(nn.Linear(in_dim, out_dim), nn.Identity(),  # (the activation itself runs inside the kernel)
                             nn.Linear(out_dim, out_dim))

    def _apply_rows(self, s_sources, s_plans, v_sources, v_plans, row_frames, residual: bool = False):
        if not self.feedforward_out:
            return super()._apply_rows(s_sources, s_plans, v_sources, v_plans, row_frames, residual)
        first, second = self.scalar_out[0], self.scalar_out[2]
        g = lambda name: getattr(self, name).weight if hasattr(self, name) else None
        spec = self.make_spec(s_plans, v_plans, False)
        frames = row_frames if spec.use_frames else None
        spec_a = replace(spec, vo=0, act_s=self.act_mid, vmode=VMODE_NONE, vector_residual=False,
                         pack_cache=self._pack_cache_first, shared_weights=True)
        s_mid = ops.gcp2(spec_a, s_sources, v_sources, frames,
                         (first.weight, first.bias, g("vector_down"), g("vector_down_frames"), None, None, None))
        extra = spec.K - spec.si  # the norm / frame-projection columns, already consumed by the first Linear
        w_second = torch.cat((second.weight, second.weight.new_zeros(self.scalar_output_dim, extra)), dim=1) if extra \
            else second.weight
        gate = getattr(self, "vector_out_scale", None)
        spec_b = replace(spec, si=self.scalar_output_dim, s_plans=[None], pack_cache=None, shared_weights=True)
        out = ops.gcp2(spec_b, [s_mid], v_sources, frames,
                       (w_second, second.bias, g("vector_down"), g("vector_down_frames"), g("vector_up"),
                        None if gate is None else gate.weight, None if gate is None else gate.bias))
        if residual:  # ResGCP: x + GCP(x) for the single ungathered source
            if isinstance(out, tuple):
                return out[0] + s_sources[0], out[1] + v_sources[0]
            return out + s_sources[0]
        return out


class GCP(nn.Module):
    """The original geometry-complete perceptron (:30-249): a GVP-like stage -- `scalar_out` over [s | |vector_down v|], gated
    `vector_up` (:204-224, process_vector :103-119) -- followed by the frame stage -- `vector_down_frames` of the NEW vectors,
    scalarize, `scalar_out_frames`, process_vector_frames (:129-161, :226-249).  Both stages run on the GCP2 kernels:

      * stage 1 is a GCP2 block without frame scalars (`use_frames=False`);
      * stage 2 is a GCP2 block on (s1, v1) whose merged input is [s1 | one norm column | 9 frame scalars]: a single hidden channel
        with zero `vector_down` / `vector_up` weights (its norm column gets a zero weight column), `vector_residual` on, so that
        the "vector_up" result is v1 itself, which the kernel epilogue then gates (sigma gate = scalar-gate mode with
        `vector_out_scale_sigma_frames`; self gate; none).  `frame_gate` uses the frame-gate kernels behind an un-gated stage 2.

    Parameters are created in the reference's order (same state_dict keys, same seeded initial weights)."""

    feedforward_out = False

    def __init__(self, input_dims, output_dims, nonlinearities: Tuple[Optional[str]] = ("relu", "sigmoid"), scalar_gate: int = 0,
                 vector_gate: bool = True, frame_gate: bool = False, sigma_frame_gate: bool = False, bottleneck: int = 1,
                 vector_residual: bool = False, vector_frame_residual: bool = False, ablate_frame_updates: bool = False,
                 ablate_scalars: bool = False, ablate_vectors: bool = False, enable_e3_equivariance: bool = False,
                 scalarization_vectorization_output_dim: int = 3, **kwargs):
        super().__init__()
        if nonlinearities is None:
            nonlinearities = (None, None)
        self.scalar_input_dim, self.vector_input_dim = input_dims
        self.scalar_output_dim, self.vector_output_dim = output_dims
        self.act_s, self.act_v = canonical_act(nonlinearities[0]), canonical_act(nonlinearities[1])
        self.scalar_gate, self.vector_gate, self.frame_gate, self.sigma_frame_gate = (
            scalar_gate, vector_gate, frame_gate, sigma_frame_gate)
        self.vector_residual, self.vector_frame_residual = vector_residual, vector_frame_residual
        self.ablate_frame_updates = ablate_frame_updates
        self.ablate_scalars, self.ablate_vectors = ablate_scalars, ablate_vectors
        self.enable_e3_equivariance = enable_e3_equivariance
        self.slope = 1e-2
        if scalarization_vectorization_output_dim != 3:
            _unsupported("scalarization_vectorization_output_dim != 3")
        if self.scalar_gate > 0:  # created, never used by GCP.forward (:69-70)
            self.norm = nn.LayerNorm(self.scalar_output_dim)
        self.hidden_dim = 0
        so, vi, vo = self.scalar_output_dim, self.vector_input_dim, self.vector_output_dim
        if vi:
            assert vi % bottleneck == 0, f"Input channel of vector ({vi}) must be divisible with bottleneck factor ({bottleneck})"
            self.hidden_dim = vi // bottleneck if bottleneck > 1 else max(vi, vo)
            self.vector_down = nn.Linear(vi, self.hidden_dim, bias=False)
            self.scalar_out = nn.Linear(self.hidden_dim + self.scalar_input_dim, so)
            if vo:
                self.vector_up = nn.Linear(self.hidden_dim, vo, bias=False)
                if self.vector_gate:
                    self.vector_out_scale = nn.Linear(so, vo)
            if not ablate_frame_updates:
                self.vector_down_frames = nn.Linear(self.hidden_dim if not vo else vo, 3, bias=False)
                self.scalar_out_frames = nn.Linear(so + 9, so)
                if vo and self.sigma_frame_gate:
                    self.vector_out_scale_sigma_frames = nn.Linear(so, vo)
                elif vo and self.frame_gate:
                    self.vector_out_scale_frames = nn.Linear(so, 9)
                    self.vector_up_frames = nn.Linear(3, vo, bias=False)
        else:
            self.scalar_out = nn.Linear(self.scalar_input_dim, so)
        self._pack_cache: dict = {}

    def _vmode(self) -> int:  # (of stage 1)
        if not (self.vector_input_dim and self.vector_output_dim):
            return VMODE_NONE
        if self.vector_gate:
            return VMODE_SCALAR_GATE
        return VMODE_SELF_GATE if self.act_v is not None else VMODE_NONE

    def apply_rows(self, s_sources, s_plans, v_sources, v_plans, row_frames, residual: bool = False, node_e3=None):
        """Same contract as GCP2.apply_rows.  `node_e3` = (edge_index, frames): node rows with enable_e3_equivariance -- the frame
        scalars of stage 2 then come from `_node_e3_frame_scalars` instead of a per-row frame."""
        if not self.vector_input_dim:
            if not self.ablate_frame_updates:  # (the reference's forward reads vector_down_frames, which does not exist then)
                raise AttributeError("GCP without vector input has no vector_down_frames: only ablate_frame_updates=True runs "
                                     "(reference gcpnet.py:93-94, 229)")
            assert len(s_sources) == 1 and s_plans[0] is None
            x = torch.zeros_like(s_sources[0]) if self.ablate_scalars else s_sources[0]
            s_out = ops.activation(ops.linear(x, self.scalar_out.weight, self.scalar_out.bias), self.act_s, self.slope)
            if not self.vector_output_dim:
                return s_out
            return s_out, s_out.new_zeros(s_out.shape[0], self.vector_output_dim, 3)
        if self.ablate_scalars:
            s_sources = [torch.zeros_like(t) for t in s_sources]
        if self.ablate_vectors:
            v_sources = [torch.zeros_like(t) for t in v_sources]
        vo, so = self.vector_output_dim, self.scalar_output_dim
        gate = getattr(self, "vector_out_scale", None)
        spec1 = Gcp2Spec(si=self.scalar_input_dim, vi=self.vector_input_dim, so=so, vo=vo, hidden=self.hidden_dim,
                         use_frames=False, act_s=self.act_s, act_v=self.act_v, slope=self.slope, vmode=self._vmode(),
                         vector_residual=bool(self.vector_residual) and bool(vo), e3=False, s_plans=list(s_plans),
                         v_plans=list(v_plans), residual=False, pack_cache=self._pack_cache)
        out1 = ops.gcp2(spec1, s_sources, v_sources, None,
                        (self.scalar_out.weight, self.scalar_out.bias, self.vector_down.weight, None,
                         self.vector_up.weight if vo else None,
                         None if gate is None or not vo else gate.weight, None if gate is None or not vo else gate.bias))
        if self.ablate_frame_updates:
            out = out1
        else:
            s1, v1 = (out1 if vo else (out1, None))
            v_src, v_pl = ([v1], [None]) if vo else (list(v_sources), list(v_plans))
            vi2 = vo if vo else self.vector_input_dim
            wf = self.scalar_out_frames.weight
            w2 = torch.cat((wf[:, :so], wf.new_zeros(so, 1), wf[:, so:]), dim=1)  # [s1 | the unused norm column | frame scalars]
            w_down = wf.new_zeros(1, vi2)
            sigma = vo and self.sigma_frame_gate
            fgate = vo and self.frame_gate and not sigma
            if not vo or fgate:
                vmode2 = VMODE_NONE
            elif sigma:
                vmode2 = VMODE_SCALAR_GATE
            else:
                vmode2 = VMODE_SELF_GATE if self.act_v is not None else VMODE_NONE
            sg = self.vector_out_scale_sigma_frames if sigma else None
            if node_e3 is not None:
                if fgate:
                    _unsupported("enable_e3_equivariance on node rows together with frame_gate")
                assert len(v_src) == 1 and v_pl[0] is None
                fs = _node_e3_frame_scalars(v_src[0], self.vector_down_frames.weight, *node_e3)
                spec2 = Gcp2Spec(si=so + 9, vi=vi2, so=so, vo=vo, hidden=1, use_frames=False, act_s=self.act_s, act_v=self.act_v,
                                 slope=self.slope, vmode=vmode2, vector_residual=bool(vo), e3=False, s_plans=[None, None],
                                 v_plans=v_pl, residual=False, pack_cache=None, shared_weights=True)
                out = ops.gcp2(spec2, [s1, fs], v_src, None,
                               (torch.cat((wf, wf.new_zeros(so, 1)), dim=1), self.scalar_out_frames.bias, w_down, None,
                                wf.new_zeros(vo, 1) if vo else None, None if sg is None else sg.weight,
                                None if sg is None else sg.bias))
            else:
                spec2 = Gcp2Spec(si=so, vi=vi2, so=so, vo=vo, hidden=1, use_frames=True, act_s=None if fgate else self.act_s,
                                 act_v=self.act_v, slope=self.slope, vmode=vmode2, vector_residual=bool(vo),
                                 e3=bool(self.enable_e3_equivariance), s_plans=[None], v_plans=v_pl, residual=False,
                                 pack_cache=None, shared_weights=True)
                out = ops.gcp2(spec2, [s1], v_src, row_frames,
                               (w2, self.scalar_out_frames.bias, w_down, self.vector_down_frames.weight,
                                wf.new_zeros(vo, 1) if vo else None, None if sg is None else sg.weight,
                                None if sg is None else sg.bias))
            if fgate:  # :139-155
                s_pre2, v_pass = out
                lin = self.vector_out_scale_frames
                w12 = torch.nn.functional.pad(lin.weight, (0, 0, 0, 3))
                b12 = torch.nn.functional.pad(lin.bias, (0, 3))
                g = ops.linear(ops.activation(s_pre2, self.act_v, self.slope), w12, b12)
                v_out = ops.frame_gate(g, row_frames, self.vector_up_frames.weight, v_pass, self.act_v, self.slope)
                if self.vector_frame_residual:
                    v_out = ops.axpy(v_pass, v_out, 1.0)
                out = (ops.activation(s_pre2, self.act_s, self.slope), v_out)
            if isinstance(out, tuple):  # :246-248 (the early exits above return un-ablated outputs, as the reference does)
                s_o, v_o = out
                out = (torch.zeros_like(s_o) if self.ablate_scalars else s_o, torch.zeros_like(v_o) if self.ablate_vectors else v_o)
            elif self.ablate_scalars:  # :243-245: the nonlinearity of a zeroed pre-activation
                out = torch.full_like(out, 0.5) if self.act_s == "sigmoid" else torch.zeros_like(out)
        if residual:
            if isinstance(out, tuple):
                return ops.axpy(s_sources[0], out[0], 1.0), ops.axpy(v_sources[0], out[1], 1.0)
            return ops.axpy(s_sources[0], out, 1.0)
        return out

    def forward(self, s_maybe_v, edge_index, frames, node_inputs: bool = False, node_mask=None):
        """:163-249.  Returns ScalarVector, or a Tensor when the block has no vector output.  `node_mask`: as in GCP2.forward."""
        frames = mask_frames(frames, edge_index, node_mask)
        if self.vector_input_dim:
            s, v = s_maybe_v
            row_frames = None
            if not self.ablate_frame_updates:
                if node_inputs and self.enable_e3_equivariance:
                    out = self.apply_rows([s], [None], [v], [None], None, node_e3=(edge_index, frames))
                    return out if not self.vector_output_dim else ScalarVector(*out)
                row_frames = GraphPlan.get(edge_index, s.shape[0]).node_frames(frames) if node_inputs else frames
            out = self.apply_rows([s], [None], [v], [None], row_frames)
        else:
            out = self.apply_rows([s_maybe_v], [None], [], [], None)
        if not self.vector_output_dim:
            return out
        return ScalarVector(*out)


def get_GCP_with_custom_cfg(input_dims, output_dims, cfg, **kwargs):
    """:826-835 -- the whole module_cfg is forwarded; unknown keys are swallowed by GCP2's **kwargs."""
    cfg_dict = copy(to_container(cfg))
    cfg_dict["nonlinearities"] = cfg.nonlinearities
    del cfg_dict["scalar_nonlinearity"]
    del cfg_dict["vector_nonlinearity"]
    for key in kwargs:
        cfg_dict[key] = kwargs[key]
    return cfg.selected_GCP(input_dims, output_dims, **cfg_dict)


class GCPEmbedding(nn.Module):
    """:703-823"""

    def __init__(self, edge_input_dims, node_input_dims, edge_hidden_dims, node_hidden_dims,
                 num_atom_types: int = NUM_ATOM_TYPES, nonlinearities: Tuple[Optional[str]] = (None, None),
                 num_lig_flags: int = 2, cfg=None, pre_norm: bool = True):
        super().__init__()
        edge_input_dims, node_input_dims = ScalarVector(*edge_input_dims), ScalarVector(*node_input_dims)
        self.atom_embedding = nn.Embedding(num_atom_types, num_atom_types) if num_atom_types > 0 else None
        self.concatenate_lig_flag = getattr(cfg, "concatenate_lig_flag", None)
        if self.concatenate_lig_flag:
            node_input_dims = ScalarVector(node_input_dims[0] + num_lig_flags, node_input_dims[1])
            self.lig_flag_embedding = nn.Embedding(num_lig_flags, num_lig_flags)
        self.pre_norm = pre_norm
        if pre_norm:
            self.edge_normalization = GCPLayerNorm(edge_input_dims)
            self.node_normalization = GCPLayerNorm(node_input_dims)
        else:
            self.edge_normalization = GCPLayerNorm(edge_hidden_dims)
            self.node_normalization = GCPLayerNorm(node_hidden_dims)
        common = dict(
            scalar_gate=cfg.scalar_gate, vector_gate=cfg.vector_gate, frame_gate=cfg.frame_gate,
            sigma_frame_gate=cfg.sigma_frame_gate, vector_frame_residual=cfg.vector_frame_residual,
            ablate_frame_updates=cfg.ablate_frame_updates, ablate_scalars=cfg.ablate_scalars,
            ablate_vectors=cfg.ablate_vectors, enable_e3_equivariance=cfg.enable_e3_equivariance)
        self.edge_embedding = cfg.selected_GCP(edge_input_dims, edge_hidden_dims, nonlinearities=nonlinearities, **common)
        self.node_embedding = cfg.selected_GCP(node_input_dims, node_hidden_dims, nonlinearities=(None, None), **common)

    def forward(self, batch):
        h = self.atom_embedding(batch.h) if self.atom_embedding is not None else batch.h
        if self.concatenate_lig_flag:
            h = torch.cat((h, self.lig_flag_embedding(batch.lig_flag.long())), dim=-1)
        node_rep = ScalarVector(h, batch.chi)
        edge_rep = ScalarVector(batch.e, batch.xi)
        edge_rep = edge_rep.scalar if not self.edge_embedding.vector_input_dim else edge_rep
        node_rep = node_rep.scalar if not self.node_embedding.vector_input_dim else node_rep
        mask = getattr(batch, "mask", None)
        if self.pre_norm:
            edge_rep = self.edge_normalization(edge_rep)
            node_rep = self.node_normalization(node_rep)
        edge_rep = self.edge_embedding(edge_rep, batch.edge_index, batch.f_ij, node_inputs=False, node_mask=mask)
        node_rep = self.node_embedding(node_rep, batch.edge_index, batch.f_ij, node_inputs=True, node_mask=mask)
        if not self.pre_norm:
            edge_rep = self.edge_normalization(edge_rep)
            node_rep = self.node_normalization(node_rep)
        return node_rep, edge_rep


class GCPMessagePassing(nn.Module):
    """:838-960"""

    def __init__(self, input_dims, output_dims, edge_dims, cfg, mp_cfg, reduce_function: str = "mean",
                 use_scalar_message_attention: bool = False, aggregate_with_row: bool = False):
        super().__init__()
        input_dims, output_dims, edge_dims = ScalarVector(*input_dims), ScalarVector(*output_dims), ScalarVector(*edge_dims)
        self.scalar_input_dim, self.vector_input_dim = input_dims
        self.scalar_output_dim, self.vector_output_dim = output_dims
        self.edge_scalar_dim, self.edge_vector_dim = edge_dims
        self.conv_cfg = mp_cfg
        self.self_message = self.conv_cfg.self_message
        self.use_residual_message_gcp = self.conv_cfg.use_residual_message_gcp
        self.reduce_function = reduce_function
        self.use_scalar_message_attention = use_scalar_message_attention
        self.aggregate_with_row = aggregate_with_row
        if reduce_function not in ("mean", "sum", "add"):
            raise NotImplementedError(reduce_function)

        scalars_in_dim = 2 * self.scalar_input_dim + self.edge_scalar_dim
        vectors_in_dim = 2 * self.vector_input_dim + self.edge_vector_dim

        soft_cfg = copy(cfg)  # :867-868
        soft_cfg.bottleneck, soft_cfg.vector_residual = cfg.default_bottleneck, cfg.default_vector_residual
        primary = partial(get_GCP_with_custom_cfg, cfg=soft_cfg)
        secondary = partial(get_GCP_with_custom_cfg, cfg=cfg)
        n_layers = self.conv_cfg.num_message_layers
        modules = [primary((scalars_in_dim, vectors_in_dim), output_dims,
                           nonlinearities=cfg.nonlinearities if n_layers > 1 else None,
                           enable_e3_equivariance=cfg.enable_e3_equivariance)]
        for _ in range(n_layers - 2):
            modules.append(secondary(output_dims, output_dims, enable_e3_equivariance=cfg.enable_e3_equivariance))
        if n_layers > 1:
            modules.append(primary(output_dims, output_dims, nonlinearities=(None, None),
                                   enable_e3_equivariance=cfg.enable_e3_equivariance))
        self.message_fusion = nn.ModuleList(modules)
        if use_scalar_message_attention:  # :892-896
            self.scalar_message_attention = nn.Sequential(nn.Linear(output_dims.scalar, 1), nn.Sigmoid())

    def _messages(self, node_rep, edge_rep, edge_index, frames) -> ScalarVector:
        m, _ = self._fused_messages(node_rep, edge_rep, edge_index, frames)
        if self.use_scalar_message_attention:  # :932-934, one fused kernel (dot product, sigmoid, scale)
            lin = self.scalar_message_attention[0]
            m = ScalarVector(ops.row_gate(m[0], lin.weight, lin.bias), m[1])
        return m

    def _fused_messages(self, node_rep, edge_rep, edge_index, frames, agg=None):
        """-> (messages, False), or with `agg` = (GatherPlan, mean) and a chain that takes it (aggregated messages, True): the
        ResGCP chain Function then returns the segment sum / mean itself (ops._Gcp2Chain)."""
        h, chi = node_rep
        e, xi = edge_rep
        plan = GraphPlan.get(edge_index, h.shape[0])
        first = self.message_fusion[0]
        # message = [h_row | e | h_col], [chi_row | xi | chi_col] (:907-917): gathered inside the kernel's tile loader
        m = first.apply_rows([h, e, h], [plan.row, None, plan.col], [chi, xi, chi], [plan.row, None, plan.col], frames)
        m = ScalarVector(*m)
        rest = list(self.message_fusion[1:])
        if rest and self._chainable(rest):
            # ResGCP chain (:921-924) in one launch: the (s, V) state of each 32-edge tile never leaves the chip
            specs = [mod.make_spec([None], [None], residual=True) for mod in rest]
            if agg is not None and not self.use_scalar_message_attention:
                return ScalarVector(*ops.gcp2_chain(specs, m[0], m[1], frames, [mod._weights() for mod in rest], agg=agg)), True
            return ScalarVector(*ops.gcp2_chain(specs, m[0], m[1], frames, [mod._weights() for mod in rest])), False
        for module in rest:
            same = (module.scalar_input_dim == module.scalar_output_dim
                    and module.vector_input_dim == module.vector_output_dim)
            if self.use_residual_message_gcp and same:  # ResGCP (:921-924), residual add fused into the kernel
                m = ScalarVector(*module.apply_rows([m[0]], [None], [m[1]], [None], frames, residual=True))
            elif self.use_residual_message_gcp:
                m = _sv_add(m, module.apply_rows([m[0]], [None], [m[1]], [None], frames))
            else:
                m = ScalarVector(*module.apply_rows([m[0]], [None], [m[1]], [None], frames))
        return m, False

    def _chainable(self, mods) -> bool:
        if not self.use_residual_message_gcp or len(mods) > 8 or any(getattr(m, "feedforward_out", False) for m in mods):
            return False
        if any(isinstance(m, GCP) for m in mods):  # (the original two-stage block: two launches per block)
            return False
        a = mods[0]
        if a.scalar_output_dim > (512 if ops.USE_WG_KERNELS else 128) or not a.vector_input_dim or not a.vector_output_dim or a.vector_output_dim > 64:
            return False
        key = lambda m: (m.scalar_input_dim, m.vector_input_dim, m.scalar_output_dim, m.vector_output_dim, m.hidden_dim,
                         m.ablate_frame_updates, m._vmode(), bool(m.vector_residual), bool(m.enable_e3_equivariance),
                         m.ablate_scalars, m.ablate_vectors)
        return (a.scalar_input_dim == a.scalar_output_dim and a.vector_input_dim == a.vector_output_dim
                and not a.ablate_scalars and not a.ablate_vectors and all(key(m) == key(a) for m in mods))

    def message(self, node_rep, edge_rep, edge_index, frames, node_mask=None):
        frames = mask_frames(frames, edge_index, node_mask)  # (:920-929: the mask only reaches the blocks' scalarize / vectorize)
        return self._messages(ScalarVector(*node_rep), ScalarVector(*edge_rep), edge_index, frames).flatten()

    def aggregate(self, message, edge_index, dim_size: int):
        plan = GraphPlan.get(edge_index, dim_size)
        side = plan.row if self.aggregate_with_row else plan.col
        return ops.segment_reduce(message, side, mean=self.reduce_function == "mean")

    def forward(self, node_rep, edge_rep, edge_index, frames, node_mask=None) -> ScalarVector:
        frames = mask_frames(frames, edge_index, node_mask)
        node_rep, edge_rep = ScalarVector(*node_rep), ScalarVector(*edge_rep)
        n = node_rep[0].shape[0]
        plan = GraphPlan.get(edge_index, n)
        side = plan.row if self.aggregate_with_row else plan.col
        mean = self.reduce_function == "mean"
        if ops.FUSE_AGGREGATION and not self.use_scalar_message_attention:
            m, done = self._fused_messages(node_rep, edge_rep, edge_index, frames, agg=(side, mean))
            if done:
                return m
        else:
            m = self._messages(node_rep, edge_rep, edge_index, frames)
        # scatter(message, col, reduce) (:939-947) as wavefront-segmented reductions over the CSR segments
        agg_s = ops.segment_reduce(m[0], side, mean)
        agg_v = ops.segment_reduce(m[1].reshape(m[1].shape[0], 3 * self.vector_output_dim), side, mean).reshape(n, self.vector_output_dim, 3)
        return ScalarVector(agg_s, agg_v)


class GCPInteractions(nn.Module):
    """:963-1262: the plain forward, the masked forward (`node_mask`: sub-graph feed-forward, in-place write-back of the unmasked
    rows, :1201-1251) and `autoregressive_forward` (:1066-1116)."""

    def __init__(self, node_dims, edge_dims, cfg, layer_cfg, dropout: float = 0.1, autoregressive: bool = False,
                 nonlinearities: Optional[Tuple[Any, Any]] = None, updating_node_positions: bool = False):
        super().__init__()
        node_dims, edge_dims = ScalarVector(*node_dims), ScalarVector(*edge_dims)
        if nonlinearities is None:
            nonlinearities = cfg.nonlinearities
        self.pre_norm = layer_cfg.pre_norm
        self.updating_node_positions = updating_node_positions
        self.ablate_x_force_update = getattr(cfg, "ablate_x_force_update", True)
        self.node_positions_weight = getattr(cfg, "node_positions_weight", 1.0)
        reduce_function = "add" if autoregressive else "mean"

        self.interaction = GCPMessagePassing(node_dims, node_dims, edge_dims, reduce_function=reduce_function, cfg=cfg,
                                             mp_cfg=layer_cfg.mp_cfg)

        ff_cfg = copy(cfg)  # :1001-1004
        ff_cfg.nonlinearities = nonlinearities
        ff_without_res_cfg = copy(cfg)
        ff_without_res_cfg.vector_residual = False
        ff_GCP = partial(get_GCP_with_custom_cfg, cfg=ff_cfg)
        ff_without_res_GCP = partial(get_GCP_with_custom_cfg, cfg=ff_without_res_cfg)

        self.gcp_norm = nn.ModuleList([GCPLayerNorm(node_dims) for _ in range(2)])
        self.gcp_dropout = nn.ModuleList([GCPDropout(dropout) for _ in range(2)])

        n_ff = layer_cfg.num_feedforward_layers
        # the reference's un-parenthesised conditional (:1014) evaluates to this tuple
        hidden_dims = (node_dims if n_ff == 1 else 4 * node_dims.scalar), 2 * node_dims.vector
        ff = [ff_without_res_GCP(node_dims, hidden_dims, nonlinearities=None if n_ff == 1 else cfg.nonlinearities,
                                 enable_e3_equivariance=cfg.enable_e3_equivariance)]
        ff.extend(ff_GCP(hidden_dims, hidden_dims, enable_e3_equivariance=cfg.enable_e3_equivariance)
                  for _ in range(n_ff - 2))
        if n_ff > 1:
            ff.append(ff_without_res_GCP(hidden_dims, node_dims, nonlinearities=(None, None),
                                         enable_e3_equivariance=cfg.enable_e3_equivariance))
        self.feedforward_network = nn.ModuleList(ff)

        if updating_node_positions:
            self.node_position_update_network = nn.ModuleList([
                ff_without_res_GCP(node_dims, (node_dims.scalar, 1), nonlinearities=cfg.nonlinearities,
                                   enable_e3_equivariance=cfg.enable_e3_equivariance)])
            # node position force-update layers (:1052-1063), created in the reference's order
            s_dim = node_dims.scalar
            self.force_act = canonical_act(cfg.nonlinearities[0])
            self.force_slope = float(getattr(layer_cfg, "nonlinearity_slope", 1e-2))
            if self.ablate_x_force_update:
                self.phi_force_i = self.phi_force_j = self.phi_force_ij = None
            else:
                self.phi_force_i = nn.Linear(s_dim, s_dim)
                self.phi_force_j = nn.Linear(s_dim, s_dim)
                last = nn.Linear(s_dim, 3, bias=False)
                torch.nn.init.xavier_uniform_(last.weight, gain=0.001)
                # (index 0 of the reference's Sequential is the parameter-free nonlinearity; it is applied inside the kernel)
                self.phi_force_ij = nn.Sequential(nn.Identity(), last)

    def derive_x_update(self, node_rep, edge_index, f_ij, node_mask=None):
        """:1119-1158.  Returns the un-weighted update (vector channel + inter-node force term); the weight and clamp are
        applied together with the position add."""
        h_v, chi_v = node_rep
        for gcp in self.node_position_update_network:
            h_v, chi_v = gcp((h_v, chi_v), edge_index, f_ij, node_inputs=True, node_mask=node_mask)
        upd = chi_v.reshape(chi_v.shape[0], 3)
        if not self.ablate_x_force_update:  # :1143-1153: per-node Linears (ops.linear: workgroup kernel), per-edge force, mean over in-edges
            plan = GraphPlan.get(edge_index, h_v.shape[0])
            a = ops.linear(h_v, self.phi_force_i.weight, self.phi_force_i.bias)
            b = ops.linear(h_v, self.phi_force_j.weight, self.phi_force_j.bias)
            force = ops.edge_force(a, b, self.phi_force_ij[1].weight, f_ij, plan, self.force_act, self.force_slope)
            upd = ops.axpy(upd, ops.segment_reduce(force, plan.col, mean=True), 1.0)
        return upd

    def autoregressive_forward(self, node_rep, edge_rep, edge_index, frames, autoregressive_node_rep, node_mask=None):
        """:1066-1116 -- edges row < col carry messages of the current representation, the others of the autoregressive one; the
        two sums (layers built with autoregressive=True aggregate with "add") are divided by the in-degree over ALL edges."""
        fwd = edge_index[0] < edge_index[1]
        i_f, i_b = torch.nonzero(fwd).squeeze(1), torch.nonzero(~fwd).squeeze(1)
        pick = lambda sv, i: ScalarVector(sv[0].index_select(0, i), sv[1].index_select(0, i))
        a = self.interaction(node_rep, pick(edge_rep, i_f), edge_index[:, i_f], frames.index_select(0, i_f), node_mask=node_mask)
        b = self.interaction(ScalarVector(*autoregressive_node_rep), pick(edge_rep, i_b), edge_index[:, i_b],
                             frames.index_select(0, i_b), node_mask=node_mask)
        inv = GraphPlan.get(edge_index, node_rep[0].shape[0]).col.inv_count  # 1 / max(in-degree, 1)
        m = _sv_add(a, b)
        return ScalarVector(m[0] * inv[:, None], m[1] * inv[:, None, None])

    def _forward_masked(self, node_rep, edge_rep, edge_index, frames, node_rep_regressive, node_mask, node_pos):
        """:1186-1262 with `node_mask`: the message passing sees the mask through its frames; everything after it runs on the
        unmasked rows only -- the feed-forward GCPs on the sub-graph those nodes induce (relabelled edge ids; the reference hands
        them the FULL-size mask, indexed by the relabelled ids, :1238) -- and the result replaces the unmasked rows of the
        (pre-normed) input."""
        if self.pre_norm:
            node_rep = self.gcp_norm[0](node_rep)
        # The reference writes the new rows INTO the tensors it was given (:1248-1251 `node_rep_residual[0][node_mask] = ...`;
        # with pre_norm, into the normed copies) and returns those very tensors -- callers alias them: the CPD module's
        # `encoder_embedding = (h, chi)` is overwritten by its first decoder layer (gcpnet_cpd_module.py:183-201), the sampling
        # loop's `node_rep_cache[j]` by every layer call (:341-349).  Mirrored here: the computation runs on a private copy (the
        # autograd Functions of this package save their inputs; an in-place write to a saved tensor would invalidate them), the
        # result rows are then index_copy_'d into the caller's tensors, which are returned.
        target = node_rep
        node_rep = ScalarVector(target[0].clone(), target[1].clone())
        if node_rep_regressive is not None:
            reg = tuple(node_rep[k] if node_rep_regressive[k] is target[k] else node_rep_regressive[k] for k in (0, 1))
            hidden = self.autoregressive_forward(node_rep, edge_rep, edge_index, frames, reg, node_mask=node_mask)
        else:
            hidden = self.interaction(node_rep, edge_rep, edge_index, frames, node_mask=node_mask)
        idx = torch.nonzero(node_mask).squeeze(1)
        sel = lambda sv: ScalarVector(sv[0].index_select(0, idx), sv[1].index_select(0, idx))
        node_rep, hidden = sel(node_rep), sel(hidden)
        ff_index, ff_frames = edge_index, frames
        if not bool(node_mask.all()) and edge_index.shape[1] > 0:  # torch_geometric.utils.subgraph(..., relabel_nodes=True)
            keep = torch.nonzero(edge_mask_of(edge_index, node_mask)).squeeze(1)
            relabel = torch.cumsum(node_mask.long(), 0) - 1
            ff_index, ff_frames = relabel[edge_index[:, keep]], frames.index_select(0, keep)
        if self.gcp_dropout[0].active:
            hidden = self.gcp_dropout[0](hidden)
        node_rep = self.gcp_norm[1 if self.pre_norm else 0](node_rep, residual=hidden)
        hidden = node_rep
        for module in self.feedforward_network:
            hidden = module(hidden, ff_index, ff_frames, node_inputs=True, node_mask=node_mask)
        if self.gcp_dropout[1].active:
            hidden = self.gcp_dropout[1](hidden)
        node_rep = _sv_add(node_rep, hidden) if self.pre_norm else self.gcp_norm[1](node_rep, residual=hidden)
        target[0].index_copy_(0, idx, node_rep[0])  # :1248-1251, in place (raises for a leaf that requires grad, as the
        target[1].index_copy_(0, idx, node_rep[1])  # reference's indexed assignment does: feed such a leaf as `t.clone()`)
        node_rep = ScalarVector(target[0], target[1])
        if not self.updating_node_positions:
            return node_rep
        upd = self.derive_x_update(node_rep, edge_index, frames, node_mask=node_mask)
        return node_rep, ops.axpy_clamp(node_pos, upd, float(self.node_positions_weight), -100.0, 100.0)

    def forward(self, node_rep, edge_rep, edge_index, frames, node_rep_regressive=None, node_mask=None, node_pos=None):
        node_rep = ScalarVector(node_rep[0], node_rep[1])
        edge_rep = ScalarVector(edge_rep[0], edge_rep[1])
        if node_mask is not None:
            return self._forward_masked(node_rep, edge_rep, edge_index, frames, node_rep_regressive, node_mask, node_pos)

        if self.pre_norm:
            node_rep = self.gcp_norm[0](node_rep)
        if node_rep_regressive is not None:
            hidden = self.autoregressive_forward(node_rep, edge_rep, edge_index, frames, node_rep_regressive)
        else:
            hidden = self.interaction(node_rep, edge_rep, edge_index, frames)
        if self.gcp_dropout[0].active:
            hidden = self.gcp_dropout[0](hidden)
        if self.pre_norm:  # :1220-1226
            node_rep = self.gcp_norm[1](node_rep, residual=hidden)
        else:
            node_rep = self.gcp_norm[0](node_rep, residual=hidden)

        hidden = node_rep
        for module in self.feedforward_network:
            hidden = module(hidden, edge_index, frames, node_inputs=True)
        if self.gcp_dropout[1].active:
            hidden = self.gcp_dropout[1](hidden)
        if self.pre_norm:  # :1242-1246
            node_rep = _sv_add(node_rep, hidden)
        else:
            node_rep = self.gcp_norm[1](node_rep, residual=hidden)

        if not self.updating_node_positions:
            return node_rep
        upd = self.derive_x_update(node_rep, edge_index, frames)
        node_pos = ops.axpy_clamp(node_pos, upd, float(self.node_positions_weight), -100.0, 100.0)  # :1156-1158,1258
        return node_rep, node_pos


class GCPInteractions2(nn.Module):
    """:1265-1451 (unmasked call path): the AR / EQ layer.  Sum-aggregated messages (optionally over `row`, with the learnable
    scalar message gate), a feed-forward network on [aggregate | node] whose last GCP has a two-layer `scalar_out`
    (`feedforward_out`, GCP3), one residual + one GCPLayerNorm, and an un-clamped position update without force term."""

    def __init__(self, node_dims, edge_dims, cfg, layer_cfg, dropout: float = 0.1,
                 nonlinearities: Optional[Tuple[Any, Any]] = None, updating_node_positions: bool = False):
        super().__init__()
        node_dims, edge_dims = ScalarVector(*node_dims), ScalarVector(*edge_dims)
        if nonlinearities is None:
            nonlinearities = cfg.nonlinearities
        self.pre_norm = layer_cfg.pre_norm
        self.updating_node_positions = updating_node_positions
        self.node_positions_weight = getattr(cfg, "node_positions_weight", 1.0)

        self.interaction = GCPMessagePassing(
            node_dims, node_dims, edge_dims, cfg=cfg, mp_cfg=layer_cfg.mp_cfg, reduce_function="sum",
            use_scalar_message_attention=getattr(layer_cfg, "use_scalar_message_attention", False),
            aggregate_with_row=getattr(layer_cfg, "aggregate_with_row", False))

        ff_cfg = copy(cfg)  # :1303-1309
        ff_cfg.nonlinearities = nonlinearities
        ff_without_res_cfg = copy(cfg)
        ff_without_res_cfg.vector_residual = False
        ff_GCP = partial(get_GCP_with_custom_cfg, cfg=ff_cfg)
        ff_without_res_GCP = partial(get_GCP_with_custom_cfg, cfg=ff_without_res_cfg)

        self.gcp_norm = nn.ModuleList([GCPLayerNorm(node_dims)])
        self.gcp_dropout = nn.ModuleList([GCPDropout(dropout)])

        n_ff = layer_cfg.num_feedforward_layers
        hidden_dims = ((node_dims.scalar, node_dims.vector) if n_ff == 1
                       else (4 * node_dims.scalar, 2 * node_dims.vector))  # :1315-1319
        ff = [ff_without_res_GCP((node_dims.scalar * 2, node_dims.vector * 2), hidden_dims,
                                 nonlinearities=(None, None) if n_ff == 1 else cfg.nonlinearities,
                                 feedforward_out=n_ff == 1, enable_e3_equivariance=cfg.enable_e3_equivariance)]
        ff.extend(ff_GCP(hidden_dims, hidden_dims, enable_e3_equivariance=cfg.enable_e3_equivariance)
                  for _ in range(n_ff - 2))
        if n_ff > 1:
            ff.append(ff_without_res_GCP(hidden_dims, node_dims, nonlinearities=(None, None), feedforward_out=True,
                                         enable_e3_equivariance=cfg.enable_e3_equivariance))
        self.feedforward_network = nn.ModuleList(ff)

        if updating_node_positions:  # :1346-1353
            self.node_position_update_gcp = ff_without_res_GCP(node_dims, (node_dims.scalar, 1),
                                                               nonlinearities=cfg.nonlinearities,
                                                               enable_e3_equivariance=cfg.enable_e3_equivariance)

    def derive_x_update(self, node_rep, edge_index, f_ij, node_mask=None):
        """:1356-1378.  Returns the un-weighted update."""
        _, chi_v = self.node_position_update_gcp(node_rep, edge_index, f_ij, node_inputs=True, node_mask=node_mask)
        return chi_v.reshape(chi_v.shape[0], 3)

    def forward(self, node_rep, edge_rep, edge_index, frames, node_mask=None, node_pos=None):
        """:1380-1451.  `node_mask`: the blocks see it through their frames (components.mask_frames); the outputs of masked nodes
        are zeroed (:1435-1436, :1448-1449)."""
        node_rep = ScalarVector(node_rep[0], node_rep[1])
        edge_rep = ScalarVector(edge_rep[0], edge_rep[1])
        frames = mask_frames(frames, edge_index, node_mask)
        if self.pre_norm:
            node_rep = self.gcp_norm[0](node_rep)
        hidden = self.interaction(node_rep, edge_rep, edge_index, frames)
        hidden = ScalarVector(*hidden.concat((node_rep,)))  # [aggregate | node] (:1414)
        for module in self.feedforward_network:
            hidden = module(hidden, edge_index, frames, node_inputs=True)
        if self.gcp_dropout[0].active:
            hidden = self.gcp_dropout[0](hidden)
        if self.pre_norm:  # :1427-1431
            node_rep = _sv_add(node_rep, hidden)
        else:
            node_rep = self.gcp_norm[0](node_rep, residual=hidden)
        if node_mask is not None:
            node_rep = node_rep.mask(node_mask.float())
        if not self.updating_node_positions:
            return node_rep
        upd = self.derive_x_update(node_rep, edge_index, frames)
        node_pos = ops.axpy(node_pos, upd, float(self.node_positions_weight))  # :1442-1444
        if node_mask is not None:
            node_pos = node_pos * node_mask.float().unsqueeze(-1)
        return node_rep, node_pos


class GCPMLPDecoder(nn.Module):
    """:1454-1491 -- Linear readout stack over the node scalars (optionally with residual updates on all but the last layer),
    returning (logits, log_softmax(logits)).  The Linears run on the workgroup GEMM kernel (ops.linear; output widths that are
    not multiples of 4 -- the default vocabulary of 20 is -- are padded with zero rows for the launch)."""

    def __init__(self, hidden_dim: int, vocab_size: int = 20, num_layers: int = 1, residual_updates: bool = False):
        super().__init__()
        self.residual_updates = residual_updates
        layers = [nn.Linear(hidden_dim, hidden_dim) for _ in range(num_layers - 1)] + [nn.Linear(hidden_dim, vocab_size)]
        self.readout = nn.ModuleList(layers) if residual_updates else nn.Sequential(*layers)

    @staticmethod
    def _linear(layer: nn.Linear, x: torch.Tensor) -> torch.Tensor:
        return ops.linear_padded(x, layer.weight, layer.bias)

    def residual_forward(self, h: torch.Tensor) -> torch.Tensor:
        x = h
        for layer in list(self.readout)[:-1]:
            x = ops.axpy(x, self._linear(layer, x), 1.0)
        return self._linear(self.readout[-1], x)

    def forward(self, h: torch.Tensor):
        if self.residual_updates:
            logits = self.residual_forward(h)
        else:
            logits = h
            for layer in self.readout:
                logits = self._linear(layer, logits)
        return logits, torch.log_softmax(logits, dim=-1)
