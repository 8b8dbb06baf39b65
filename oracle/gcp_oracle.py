"""CPU oracle for GCPNet's geometry-complete message-passing hot path.

TEST INFRASTRUCTURE ONLY.  This file is a plain-PyTorch (CPU, fp32) restatement of the reference algorithm for
the path named by BASELINE.json.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it, and only as the checker / reported baseline -- never as a product path.

Pinning: the reference holds no golden vectors for this path, so the oracle is pinned
against outputs of the reference itself, run in the authoring container through ``tests/golden/ref_stubs.py``
and committed as ``tests/golden/*.npz`` by ``tests/golden/gen_fixtures.py``;
``tests/test_oracle_golden.py`` checks every function here against those fixtures.

Style: functional.  Every block takes a flat ``params`` mapping with the reference's state_dict key names
(e.g. ``interaction.message_fusion.0.scalar_out.weight``) plus a ``prefix``; shapes are inferred from the
weights.  Citations are ``file:line`` relative to ``/root/reference/src/models``.
"""
from __future__ import annotations

import math
from typing import Dict, Mapping, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Mapping[str, Tensor]
# Tests may set this to a list: gcp2() then appends (prefix, pre-activation of scalar_out) of every block it evaluates.
TRACE_PRE: Optional[list] = None


# ----------------------------------------------------------------------------------------------------------
# third-party arithmetic restated: torch_scatter.scatter (pytorch-scatter 2.0.9; call sites
# components/__init__.py:179,197,316,369 and components/gcpnet.py:946,1105,1153)
# ----------------------------------------------------------------------------------------------------------
def scatter(src: Tensor, index: Tensor, dim_size: Optional[int] = None, reduce: str = "sum") -> Tensor:
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() else 0
    out = torch.zeros((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype).index_add(0, index, src)
    if reduce in ("sum", "add"):
        return out
    if reduce != "mean":
        raise NotImplementedError(reduce)
    cnt = torch.zeros(dim_size, dtype=src.dtype).index_add(0, index, torch.ones(index.shape[0], dtype=src.dtype))
    return out / cnt.clamp(min=1).reshape((-1,) + (1,) * (src.dim() - 1))


# ----------------------------------------------------------------------------------------------------------
# small pieces
# ----------------------------------------------------------------------------------------------------------
def nonlinearity(name: Optional[str], x: Tensor, slope: float = 1e-2) -> Tensor:
    """models/__init__.py:42-57 (get_nonlinearity)."""
    if name is None:
        return x
    key = name.lower().strip()
    if key == "relu":
        return F.relu(x)
    if key == "leakyrelu":
        return F.leaky_relu(x, negative_slope=slope)
    if key == "selu":
        return F.selu(x)
    if key == "silu":
        return F.silu(x)
    if key == "sigmoid":
        return torch.sigmoid(x)
    raise NotImplementedError(f"The nonlinearity {name} is currently not implemented.")


def safe_norm(x: Tensor, dim: int = -1, eps: float = 1e-8, keepdim: bool = False, sqrt: bool = True) -> Tensor:
    """components/__init__.py:382-392."""
    n = (x * x).sum(dim=dim, keepdim=keepdim)
    if sqrt:
        n = torch.sqrt(n + eps)
    return n + eps


def flatten_sv(s: Tensor, v: Tensor) -> Tensor:
    """ScalarVector.flatten, components/__init__.py:61-63: [s | V x 3 row-major]."""
    return torch.cat((s, v.reshape(v.shape[0], v.shape[1] * v.shape[2])), dim=-1)  # (explicit width: rows may be 0)


def recover_sv(x: Tensor, vdim: int) -> Tuple[Tensor, Tensor]:
    """ScalarVector.recover, components/__init__.py:65-69."""
    return x[..., : x.shape[-1] - 3 * vdim], x[..., x.shape[-1] - 3 * vdim:].reshape(x.shape[0], vdim, 3)


def centralize(x: Tensor, batch_index: Tensor, node_mask: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """components/__init__.py:170-200.  Masked: the centroid of the unmasked nodes; masked rows of the centred copy are +inf."""
    if node_mask is not None:
        c = scatter(x[node_mask], batch_index[node_mask], reduce="mean")
        out = torch.full_like(x, float("inf"))
        out[node_mask] = x[node_mask] - c[batch_index][node_mask]
        return c, out
    c = scatter(x, batch_index, reduce="mean")
    return c, x - c[batch_index]


def decentralize(x: Tensor, batch_index: Tensor, centroid: Tensor, node_mask: Optional[Tensor] = None) -> Tensor:
    """components/__init__.py:203-217."""
    if node_mask is not None:
        out = torch.full_like(x, float("inf"))
        out[node_mask] = x[node_mask] + centroid[batch_index]  # (as written, :212: shapes agree when `batch_index` lists the
        return out                                              #  unmasked nodes only, or no node is masked)
    return x + centroid[batch_index]


def edge_mask_of(edge_index: Tensor, node_mask: Tensor) -> Tensor:
    """components/__init__.py:230, 296, 347: an edge is kept when both end points are unmasked."""
    return node_mask[edge_index[0]] & node_mask[edge_index[1]]


def localize(x: Tensor, edge_index: Tensor, norm_x_diff: bool = True, node_mask: Optional[Tensor] = None) -> Tensor:
    """components/__init__.py:221-269: frame rows [x_diff; x_cross; x_vertical]; with a node mask the frames of the masked edges
    are +inf (:232-236, :262-264)."""
    xi, xj = x[edge_index[0]], x[edge_index[1]]
    d = xi - xj
    c = torch.cross(xi, xj, dim=-1)
    if norm_x_diff:
        d = d / (d.pow(2).sum(1, keepdim=True).sqrt() + 1)
        c = c / (c.pow(2).sum(1, keepdim=True).sqrt() + 1)
    vert = torch.cross(d, c, dim=-1)
    f = torch.stack((d, c, vert), dim=1)
    if node_mask is not None:
        f = torch.where(edge_mask_of(edge_index, node_mask)[:, None, None], f, torch.full_like(f, float("inf")))
    return f


def _masked_frames(frames: Tensor, edge_index: Tensor, node_mask: Optional[Tensor]) -> Tensor:
    """What the masked branches of scalarize (:295-302) and vectorize (:346-357) amount to: masked edges contribute zeros (and
    still count in the node-row means), i.e. their frames act as zero frames."""
    if node_mask is None:
        return frames
    return torch.where(edge_mask_of(edge_index, node_mask)[:, None, None], frames, torch.zeros_like(frames))


def scalarize(vec: Tensor, edge_index: Tensor, frames: Tensor, node_inputs: bool, e3: bool, dim_size: int,
              node_mask: Optional[Tensor] = None) -> Tensor:
    """components/__init__.py:273-325.  ``vec`` is [rows, 3 channels, 3 xyz]; the result is
    [rows, 9] with entry 3*channel + frame_row = <frame_row, channel vector>."""
    row = edge_index[0]
    src = vec[row] if node_inputs else vec
    local = torch.einsum("ead,ekd->eka", _masked_frames(frames, edge_index, node_mask), src)
    if e3:
        local = torch.cat((local[..., :1], local[..., 1:2].abs(), local[..., 2:]), dim=-1)
    flat = local.reshape(local.shape[0], 9)
    return scatter(flat, row, dim_size=dim_size, reduce="mean") if node_inputs else flat


def vectorize(gate: Tensor, edge_index: Tensor, frames: Tensor, node_inputs: bool, dim_size: int,
              node_mask: Optional[Tensor] = None) -> Tensor:
    """components/__init__.py:329-378: 9 gate scalars -> 3 vectors sum_a g[3c+a] * frame_row_a."""
    row = edge_index[0]
    g = gate[row] if node_inputs else gate
    out = torch.einsum("eca,ead->ecd", g.reshape(-1, 3, 3), _masked_frames(frames, edge_index, node_mask))
    return scatter(out, row, dim_size=dim_size, reduce="mean") if node_inputs else out


def subgraph(node_mask: Tensor, edge_index: Tensor, edge_attr: Tensor) -> Tuple[Tensor, Tensor]:
    """torch_geometric.utils.subgraph(subset, edge_index, edge_attr, relabel_nodes=True) (PyG 2.x; call site
    components/gcpnet.py:1212-1217), documented semantics restated: the edges with both end points in the subset, in their
    original order, node ids relabelled to the position inside the (sorted) subset."""
    keep = edge_mask_of(edge_index, node_mask)
    relabel = torch.cumsum(node_mask.long(), 0) - 1
    return relabel[edge_index[:, keep]], edge_attr[keep]


def gcp_layer_norm(P: Params, pre: str, s: Tensor, v: Optional[Tensor], eps: float = 1e-8):
    """GCPLayerNorm, components/__init__.py:138-167.  nn.LayerNorm (eps 1e-5, affine) on scalars; vectors divided
    by sqrt(mean_channels(clamp(|v|^2, eps)))."""
    if s.shape[0] == 0:
        return s, v
    d = s.shape[-1]
    s_out = F.layer_norm(s, (d,), P[pre + "scalar_norm.weight"], P[pre + "scalar_norm.bias"], 1e-5)
    if v is None or v.shape[1] == 0:
        return s_out, v
    vn = torch.clamp((v * v).sum(-1, keepdim=True), min=eps)
    vn = torch.sqrt(vn.mean(dim=-2, keepdim=True))
    return s_out, v / vn


# ----------------------------------------------------------------------------------------------------------
# GCP2 -- components/gcpnet.py:252-468
# ----------------------------------------------------------------------------------------------------------
def gcp2(
    P: Params,
    pre: str,
    s: Tensor,
    v: Optional[Tensor],
    edge_index: Tensor,
    frames: Tensor,
    node_inputs: bool = False,
    nonlinearities: Sequence[Optional[str]] = ("relu", "sigmoid"),
    vector_gate: bool = True,
    frame_gate: bool = False,
    vector_residual: bool = False,
    ablate_frame_updates: bool = False,
    ablate_scalars: bool = False,
    ablate_vectors: bool = False,
    enable_e3_equivariance: bool = False,
    vector_output_dim: Optional[int] = None,
    slope: float = 1e-2,
    scalar_out_nonlinearity: Optional[str] = "silu",
    node_mask: Optional[Tensor] = None,
):
    """One geometry-complete perceptron.  Returns (s_out, v_out) or just s_out when there is no vector output.

    Dims are read off the weights: ``vector_down`` [H, V_in], ``scalar_out`` [s_out, s_in + H + 9],
    ``vector_down_frames`` [3, V_in], ``vector_up`` [V_out, H], ``vector_out_scale`` [V_out, s_out]
    (gcpnet.py:298-324)."""
    if nonlinearities is None:
        nonlinearities = (None, None)
    act_s, act_v = nonlinearities
    has_vin = (pre + "vector_down.weight") in P
    has_vout = (pre + "vector_up.weight") in P
    two_layer = (pre + "scalar_out.0.weight") in P  # GCP3(feedforward_out=True), gcpnet.py:529-533
    W_s, b_s = (P[pre + "scalar_out.0.weight"], P[pre + "scalar_out.0.bias"]) if two_layer \
        else (P[pre + "scalar_out.weight"], P[pre + "scalar_out.bias"])

    vh = None
    if has_vin:  # gcpnet.py:414-436
        if ablate_scalars:
            s = torch.zeros_like(s)
        if ablate_vectors:
            v = torch.zeros_like(v)
        vh = torch.einsum("rcd,hc->rdh", v, P[pre + "vector_down.weight"])  # [rows, xyz, H]
        merged = torch.cat((s, safe_norm(vh, dim=-2)), dim=-1)
        if not ablate_frame_updates:
            vf = torch.einsum("rcd,kc->rkd", v, P[pre + "vector_down_frames.weight"])  # [rows, 3 ch, xyz]
            sh = scalarize(vf, edge_index, frames, node_inputs, enable_e3_equivariance, vf.shape[0], node_mask)
            merged = torch.cat((merged, sh), dim=-1)
    else:
        merged = s

    s_pre = merged @ W_s.t() + b_s  # gcpnet.py:441
    if TRACE_PRE is not None:  # (tests: the ReLU census of tests/test_full_size.py)
        TRACE_PRE.append((pre, s_pre.detach()))
    if two_layer:  # Linear -> act -> Linear
        s_pre = nonlinearity(scalar_out_nonlinearity, s_pre, slope) @ P[pre + "scalar_out.2.weight"].t() \
            + P[pre + "scalar_out.2.bias"]

    if not has_vout and not vector_output_dim:  # gcpnet.py:443-446
        if ablate_scalars:
            s_pre = torch.zeros_like(s_pre)
        return nonlinearity(act_s, s_pre, slope)
    if not has_vin:  # gcpnet.py:447-449
        v_out = torch.zeros(s_pre.shape[0], vector_output_dim, 3, dtype=s_pre.dtype)
    else:  # gcpnet.py:334-391
        vu = torch.einsum("rdh,oh->rdo", vh, P[pre + "vector_up.weight"])
        if vector_residual:
            vu = vu + v.transpose(-1, -2)
        v_out = vu.transpose(-1, -2)  # [rows, V_out, xyz]
        if frame_gate and not ablate_frame_updates:
            g = nonlinearity(act_v, s_pre, slope) @ P[pre + "vector_out_scale_frames.weight"].t() \
                + P[pre + "vector_out_scale_frames.bias"]
            gv = vectorize(g, edge_index, frames, node_inputs, s_pre.shape[0], node_mask)  # [rows, 3, xyz]
            gvr = torch.einsum("rkd,ok->rod", gv, P[pre + "vector_up_frames.weight"])
            v_out = v_out * nonlinearity(act_v, safe_norm(gvr, dim=-1, keepdim=True), slope)
        elif vector_gate:
            g = nonlinearity(act_v, s_pre, slope) @ P[pre + "vector_out_scale.weight"].t() \
                + P[pre + "vector_out_scale.bias"]
            v_out = v_out * torch.sigmoid(g).unsqueeze(-1)
        elif act_v is not None:
            v_out = v_out * nonlinearity(act_v, safe_norm(v_out, dim=-1, keepdim=True), slope)

    s_out = nonlinearity(act_s, s_pre, slope)  # gcpnet.py:465-468
    if ablate_scalars:
        s_out = torch.zeros_like(s_out)
    if ablate_vectors:
        v_out = torch.zeros_like(v_out)
    return s_out, v_out


# ----------------------------------------------------------------------------------------------------------
# GCP (the original block) -- components/gcpnet.py:30-249
# ----------------------------------------------------------------------------------------------------------
def gcp(
    P: Params,
    pre: str,
    s: Tensor,
    v: Optional[Tensor],
    edge_index: Tensor,
    frames: Tensor,
    node_inputs: bool = False,
    nonlinearities: Sequence[Optional[str]] = ("relu", "sigmoid"),
    vector_gate: bool = True,
    frame_gate: bool = False,
    sigma_frame_gate: bool = False,
    vector_residual: bool = False,
    vector_frame_residual: bool = False,
    ablate_frame_updates: bool = False,
    ablate_scalars: bool = False,
    ablate_vectors: bool = False,
    enable_e3_equivariance: bool = False,
    slope: float = 1e-2,
    node_mask: Optional[Tensor] = None,
):
    """The original two-stage perceptron: a GVP-like stage (scalar_out over [s | |vector_down v|], gated vector_up,
    gcpnet.py:204-224 + process_vector :103-119) followed by the frame stage (vector_down_frames -> scalarize -> scalar_out_frames,
    process_vector_frames :129-161; gcpnet.py:226-249).  Only blocks with vector input (the reference's forward needs
    `vector_down_frames`, which exists only then)."""
    if nonlinearities is None:
        nonlinearities = (None, None)
    act_s, act_v = nonlinearities
    has_vout = (pre + "vector_up.weight") in P
    if ablate_scalars:
        s = torch.zeros_like(s)
    if ablate_vectors:
        v = torch.zeros_like(v)
    vh = torch.einsum("rcd,hc->rdh", v, P[pre + "vector_down.weight"])  # [rows, xyz, H]            :208-209
    merged = torch.cat((s, safe_norm(vh, dim=-2)), dim=-1)
    s_pre = merged @ P[pre + "scalar_out.weight"].t() + P[pre + "scalar_out.bias"]  # :215
    v_cur = v
    if has_vout:  # process_vector :103-119
        vu = torch.einsum("rdh,oh->rdo", vh, P[pre + "vector_up.weight"])
        if vector_residual:
            vu = vu + v.transpose(-1, -2)
        v_cur = vu.transpose(-1, -2)
        if vector_gate:
            g = nonlinearity(act_v, s_pre, slope) @ P[pre + "vector_out_scale.weight"].t() + P[pre + "vector_out_scale.bias"]
            v_cur = v_cur * torch.sigmoid(g).unsqueeze(-1)
        elif act_v is not None:
            v_cur = v_cur * nonlinearity(act_v, safe_norm(v_cur, dim=-1, keepdim=True), slope)
    s_cur = nonlinearity(act_s, s_pre, slope)  # :220
    if ablate_frame_updates:  # :225-226 (no output ablation on this exit)
        return (s_cur, v_cur) if has_vout else s_cur
    # frame stage :228-249
    vf = torch.einsum("rcd,kc->rkd", v_cur, P[pre + "vector_down_frames.weight"])  # [rows, 3 ch, xyz]
    sh = scalarize(vf, edge_index, frames, node_inputs, enable_e3_equivariance, vf.shape[0], node_mask)
    s_pre2 = torch.cat((s_cur, sh), dim=-1) @ P[pre + "scalar_out_frames.weight"].t() + P[pre + "scalar_out_frames.bias"]
    if not has_vout:  # :243-246
        if ablate_scalars:
            s_pre2 = torch.zeros_like(s_pre2)
        return nonlinearity(act_s, s_pre2, slope)
    v_out = v_cur  # process_vector_frames :129-161
    if sigma_frame_gate:
        g = nonlinearity(act_v, s_pre2, slope) @ P[pre + "vector_out_scale_sigma_frames.weight"].t() \
            + P[pre + "vector_out_scale_sigma_frames.bias"]
        v_out = v_out * torch.sigmoid(g).unsqueeze(-1)
    elif frame_gate:
        g = nonlinearity(act_v, s_pre2, slope) @ P[pre + "vector_out_scale_frames.weight"].t() \
            + P[pre + "vector_out_scale_frames.bias"]
        gv = vectorize(g, edge_index, frames, node_inputs, s_pre2.shape[0], node_mask)
        gvr = torch.einsum("rkd,ok->rod", gv, P[pre + "vector_up_frames.weight"])
        v_out = v_out * nonlinearity(act_v, safe_norm(gvr, dim=-1, keepdim=True), slope)
        if vector_frame_residual:
            v_out = v_out + v_cur
    elif act_v is not None:
        v_out = v_out * nonlinearity(act_v, safe_norm(v_out, dim=-1, keepdim=True), slope)
    s_out = nonlinearity(act_s, s_pre2, slope)
    if ablate_scalars:
        s_out = torch.zeros_like(s_out)
    if ablate_vectors:
        v_out = torch.zeros_like(v_out)
    return s_out, v_out


def _gcp_kwargs(cfg: Mapping, **over) -> Dict:
    """What get_GCP_with_custom_cfg (gcpnet.py:826-835) forwards that GCP2 actually reads."""
    kw = dict(
        nonlinearities=tuple(cfg["nonlinearities"]),
        vector_gate=cfg["vector_gate"],
        frame_gate=cfg["frame_gate"],
        vector_residual=cfg["vector_residual"],
        ablate_frame_updates=cfg["ablate_frame_updates"],
        ablate_scalars=cfg["ablate_scalars"],
        ablate_vectors=cfg["ablate_vectors"],
        enable_e3_equivariance=cfg["enable_e3_equivariance"],
    )
    kw.update(over)
    if kw["nonlinearities"] is None:
        kw["nonlinearities"] = (None, None)
    return kw


# ----------------------------------------------------------------------------------------------------------
# GCPEmbedding -- components/gcpnet.py:703-823
# ----------------------------------------------------------------------------------------------------------
def gcp_embedding(
    P: Params,
    pre: str,
    h: Tensor,
    chi: Tensor,
    e: Tensor,
    xi: Tensor,
    edge_index: Tensor,
    frames: Tensor,
    cfg: Mapping,
    nonlinearities: Sequence[Optional[str]] = (None, None),
    pre_norm: bool = True,
):
    if (pre + "atom_embedding.weight") in P:  # gcpnet.py:785-788
        h = P[pre + "atom_embedding.weight"][h]
    kw = dict(
        vector_gate=cfg["vector_gate"],
        frame_gate=cfg["frame_gate"],
        ablate_frame_updates=cfg["ablate_frame_updates"],
        ablate_scalars=cfg["ablate_scalars"],
        ablate_vectors=cfg["ablate_vectors"],
        enable_e3_equivariance=cfg["enable_e3_equivariance"],
    )
    if pre_norm:  # gcpnet.py:800-802
        e, xi = gcp_layer_norm(P, pre + "edge_normalization.", e, xi)
        h, chi = gcp_layer_norm(P, pre + "node_normalization.", h, chi)
    edge_rep = gcp2(P, pre + "edge_embedding.", e, xi, edge_index, frames, node_inputs=False,
                    nonlinearities=nonlinearities, **kw)
    node_rep = gcp2(P, pre + "node_embedding.", h, chi, edge_index, frames, node_inputs=True,
                    nonlinearities=(None, None), **kw)
    if not pre_norm:  # gcpnet.py:819-821
        edge_rep = gcp_layer_norm(P, pre + "edge_normalization.", *edge_rep)
        node_rep = gcp_layer_norm(P, pre + "node_normalization.", *node_rep)
    return node_rep, edge_rep


# ----------------------------------------------------------------------------------------------------------
# GCPMessagePassing -- components/gcpnet.py:838-960
# ----------------------------------------------------------------------------------------------------------
def _soft_cfg(cfg: Mapping) -> Dict:
    c = dict(cfg)  # gcpnet.py:867-868
    c["bottleneck"], c["vector_residual"] = cfg["default_bottleneck"], cfg["default_vector_residual"]
    return c


def message_passing(
    P: Params,
    pre: str,
    h: Tensor,
    chi: Tensor,
    e: Tensor,
    xi: Tensor,
    edge_index: Tensor,
    frames: Tensor,
    cfg: Mapping,
    mp_cfg: Mapping,
    reduce_function: str = "mean",
    return_messages: bool = False,
    aggregate_with_row: bool = False,
    node_mask: Optional[Tensor] = None,
):
    row, col = edge_index[0], edge_index[1]
    n_msg = mp_cfg["num_message_layers"]
    soft = _soft_cfg(cfg)
    ms = torch.cat((h[row], e, h[col]), dim=-1)  # gcpnet.py:917
    mv = torch.cat((chi[row], xi, chi[col]), dim=1)

    first = _gcp_kwargs(soft, nonlinearities=tuple(cfg["nonlinearities"]) if n_msg > 1 else None)
    mid = _gcp_kwargs(cfg)
    last = _gcp_kwargs(soft, nonlinearities=(None, None))
    kws = [first] + [mid] * (n_msg - 2) + ([last] if n_msg > 1 else [])

    if mp_cfg["use_residual_message_gcp"]:  # gcpnet.py:919-924
        ms, mv = gcp2(P, f"{pre}message_fusion.0.", ms, mv, edge_index, frames, node_mask=node_mask, **kws[0])
        for k in range(1, len(kws)):
            ds, dv = gcp2(P, f"{pre}message_fusion.{k}.", ms, mv, edge_index, frames, node_mask=node_mask, **kws[k])
            ms, mv = ms + ds, mv + dv
    else:
        for k in range(len(kws)):
            ms, mv = gcp2(P, f"{pre}message_fusion.{k}.", ms, mv, edge_index, frames, node_mask=node_mask, **kws[k])

    if (pre + "scalar_message_attention.0.weight") in P:  # gcpnet.py:932-934
        att = torch.sigmoid(ms @ P[pre + "scalar_message_attention.0.weight"].t()
                            + P[pre + "scalar_message_attention.0.bias"])
        ms = ms * att
    msg = flatten_sv(ms, mv)
    agg = scatter(msg, row if aggregate_with_row else col, dim_size=h.shape[0], reduce=reduce_function)  # gcpnet.py:939-947
    out = recover_sv(agg, mv.shape[1])
    return (out, msg) if return_messages else out


# ----------------------------------------------------------------------------------------------------------
# GCPInteractions -- components/gcpnet.py:963-1262 (non-autoregressive, unmasked)
# ----------------------------------------------------------------------------------------------------------
def gcp_interactions(
    P: Params,
    pre: str,
    h: Tensor,
    chi: Tensor,
    e: Tensor,
    xi: Tensor,
    edge_index: Tensor,
    frames: Tensor,
    cfg: Mapping,
    layer_cfg: Mapping,
    node_pos: Optional[Tensor] = None,
    nonlinearities: Optional[Sequence[Optional[str]]] = None,
    node_mask: Optional[Tensor] = None,
    regressive: Optional[Tuple[Tensor, Tensor]] = None,
):
    """Eval-mode (dropout = identity) forward.  Returns (h, chi) or ((h, chi), node_pos).  `node_mask`: the masked call path
    (:1201-1217, :1248-1251); `regressive` = (h, chi) of the autoregressive node representation (:1066-1116, layers built
    with autoregressive=True aggregate with "add")."""
    if nonlinearities is None:
        nonlinearities = cfg["nonlinearities"]
    pre_norm = layer_cfg["pre_norm"]
    n_ff = layer_cfg["num_feedforward_layers"]
    updating = (pre + "node_position_update_network.0.scalar_out.weight") in P

    if pre_norm:  # gcpnet.py:1188-1189
        h, chi = gcp_layer_norm(P, pre + "gcp_norm.0.", h, chi)
    mp = layer_cfg["mp_cfg"]
    if regressive is not None:  # autoregressive_forward, gcpnet.py:1066-1116
        fwd = edge_index[0] < edge_index[1]
        a = message_passing(P, pre + "interaction.", h, chi, e[fwd], xi[fwd], edge_index[:, fwd], frames[fwd], cfg, mp,
                            reduce_function="add", node_mask=node_mask)
        b = message_passing(P, pre + "interaction.", regressive[0], regressive[1], e[~fwd], xi[~fwd], edge_index[:, ~fwd],
                            frames[~fwd], cfg, mp, reduce_function="add", node_mask=node_mask)
        cnt = scatter(torch.ones(edge_index.shape[1], dtype=h.dtype), edge_index[1], dim_size=h.shape[0]).clamp(min=1)
        rs, rv = (a[0] + b[0]) / cnt[:, None], (a[1] + b[1]) / cnt[:, None, None]
    else:
        rs, rv = message_passing(P, pre + "interaction.", h, chi, e, xi, edge_index, frames, cfg, mp, node_mask=node_mask)
    ff_index, ff_frames = edge_index, frames
    if node_mask is not None:  # :1201-1217: the rest of the layer runs on the unmasked nodes (and, if any node is masked, on
        h_full, chi_full = h, chi  # the sub-graph they induce -- with the FULL-size mask indexed by the relabelled ids, :1238)
        h, chi, rs, rv = h[node_mask], chi[node_mask], rs[node_mask], rv[node_mask]
        if not bool(node_mask.all()) and edge_index.shape[1] > 0:
            ff_index, ff_frames = subgraph(node_mask, edge_index, frames)
    h, chi = h + rs, chi + rv  # gcpnet.py:1220
    h, chi = gcp_layer_norm(P, pre + ("gcp_norm.1." if pre_norm else "gcp_norm.0."), h, chi)

    no_res = dict(cfg)
    no_res["vector_residual"] = False  # gcpnet.py:1003-1004
    ff_cfg = dict(cfg)
    ff_cfg["nonlinearities"] = nonlinearities  # gcpnet.py:1001-1002
    kws = [_gcp_kwargs(no_res, nonlinearities=None if n_ff == 1 else tuple(cfg["nonlinearities"]))]
    kws += [_gcp_kwargs(ff_cfg)] * (n_ff - 2)
    if n_ff > 1:
        kws.append(_gcp_kwargs(no_res, nonlinearities=(None, None)))
    fs, fv = h, chi
    for k, kw in enumerate(kws):  # gcpnet.py:1229-1239
        fs, fv = gcp2(P, f"{pre}feedforward_network.{k}.", fs, fv, ff_index, ff_frames, node_inputs=True, node_mask=node_mask, **kw)
    h, chi = h + fs, chi + fv  # gcpnet.py:1242
    if not pre_norm:
        h, chi = gcp_layer_norm(P, pre + "gcp_norm.1.", h, chi)
    if node_mask is not None:  # :1248-1251: only the unmasked rows are replaced
        h_full, chi_full = h_full.clone(), chi_full.clone()
        h_full[node_mask], chi_full[node_mask] = h, chi
        h, chi = h_full, chi_full

    if not updating:
        return h, chi
    # derive_x_update, gcpnet.py:1119-1158 (force term ablated in every shipped config)
    hv, xv = gcp2(P, pre + "node_position_update_network.0.", h, chi, edge_index, frames, node_inputs=True, node_mask=node_mask,
                  **_gcp_kwargs(no_res, nonlinearities=tuple(cfg["nonlinearities"])))
    upd = xv.squeeze(1)
    if (pre + "phi_force_i.weight") in P:  # force term (:1143-1153): uses the scalar OUTPUT of the position-update GCP
        row, col = edge_index[0], edge_index[1]
        hi = hv[row] @ P[pre + "phi_force_i.weight"].t() + P[pre + "phi_force_i.bias"]
        hj = hv[col] @ P[pre + "phi_force_j.weight"].t() + P[pre + "phi_force_j.bias"]
        coef = nonlinearity(cfg["nonlinearities"][0], hi + hj, layer_cfg["nonlinearity_slope"]) \
            @ P[pre + "phi_force_ij.1.weight"].t()
        force = torch.einsum("ea,ead->ed", coef, frames)
        upd = upd + scatter(force, col, reduce="mean")
    upd = (upd * cfg.get("node_positions_weight", 1.0)).clamp(min=-100, max=100)
    return (h, chi), node_pos + upd


# ----------------------------------------------------------------------------------------------------------
# GCPInteractions2 -- components/gcpnet.py:1265-1451 (unmasked): the AR / EQ layer
# ----------------------------------------------------------------------------------------------------------
def gcp_interactions2(
    P: Params,
    pre: str,
    h: Tensor,
    chi: Tensor,
    e: Tensor,
    xi: Tensor,
    edge_index: Tensor,
    frames: Tensor,
    cfg: Mapping,
    layer_cfg: Mapping,
    node_pos: Optional[Tensor] = None,
    nonlinearities: Optional[Sequence[Optional[str]]] = None,
    node_mask: Optional[Tensor] = None,
):
    """Eval-mode forward.  Returns (h, chi) or ((h, chi), node_pos).  The two-layer `scalar_out` of the GCP3 blocks and the
    scalar message gate are picked up from the parameter names."""
    if nonlinearities is None:
        nonlinearities = cfg["nonlinearities"]
    pre_norm = layer_cfg["pre_norm"]
    n_ff = layer_cfg["num_feedforward_layers"]
    if pre_norm:  # gcpnet.py:1402-1403
        h, chi = gcp_layer_norm(P, pre + "gcp_norm.0.", h, chi)
    rs, rv = message_passing(P, pre + "interaction.", h, chi, e, xi, edge_index, frames, cfg, layer_cfg["mp_cfg"],
                             reduce_function="sum", aggregate_with_row=layer_cfg.get("aggregate_with_row", False),
                             node_mask=node_mask)
    fs, fv = torch.cat((rs, h), dim=-1), torch.cat((rv, chi), dim=1)  # gcpnet.py:1414
    no_res = dict(cfg)
    no_res["vector_residual"] = False
    ff_cfg = dict(cfg)
    ff_cfg["nonlinearities"] = nonlinearities
    kws = [_gcp_kwargs(no_res, nonlinearities=(None, None) if n_ff == 1 else tuple(cfg["nonlinearities"]))]
    kws += [_gcp_kwargs(ff_cfg)] * (n_ff - 2)
    if n_ff > 1:
        kws.append(_gcp_kwargs(no_res, nonlinearities=(None, None)))
    for k, kw in enumerate(kws):  # gcpnet.py:1417-1424
        fs, fv = gcp2(P, f"{pre}feedforward_network.{k}.", fs, fv, edge_index, frames, node_inputs=True, node_mask=node_mask, **kw)
    h, chi = h + fs, chi + fv  # gcpnet.py:1427
    if not pre_norm:
        h, chi = gcp_layer_norm(P, pre + "gcp_norm.0.", h, chi)
    if node_mask is not None:  # :1435-1436
        h, chi = h * node_mask.to(h.dtype)[:, None], chi * node_mask.to(h.dtype)[:, None, None]
    if (pre + "node_position_update_gcp.scalar_out.weight") not in P:
        return h, chi
    _, xv = gcp2(P, pre + "node_position_update_gcp.", h, chi, edge_index, frames, node_inputs=True, node_mask=node_mask,
                 **_gcp_kwargs(no_res, nonlinearities=tuple(cfg["nonlinearities"])))
    node_pos = node_pos + xv.squeeze(1) * cfg.get("node_positions_weight", 1.0)  # gcpnet.py:1356-1378,1442
    if node_mask is not None:  # :1448-1449
        node_pos = node_pos * node_mask.to(h.dtype)[:, None]
    return (h, chi), node_pos


# ----------------------------------------------------------------------------------------------------------
# GCPMLPDecoder -- components/gcpnet.py:1454-1491
# ----------------------------------------------------------------------------------------------------------
def mlp_decoder(P: Params, pre: str, h: Tensor, residual_updates: bool = False) -> Tuple[Tensor, Tensor]:
    """Linear readout stack (`readout.k`), optionally with residual updates on all but the last layer; (logits, log_softmax)."""
    n = 0
    while f"{pre}readout.{n}.weight" in P:
        n += 1
    x = h
    for k in range(n):
        y = x @ P[f"{pre}readout.{k}.weight"].t() + P[f"{pre}readout.{k}.bias"]
        x = x + y if (residual_updates and k < n - 1) else y
    return x, F.log_softmax(x, dim=-1)


# ----------------------------------------------------------------------------------------------------------
# task-level forwards: gcpnet_nms_module.py:127-151, gcpnet_lba_module.py:155-186
# ----------------------------------------------------------------------------------------------------------
def nms_forward(P: Params, batch: Mapping[str, Tensor], cfg: Mapping, layer_cfg: Mapping, num_layers: int):
    bidx, ei = batch["batch"], batch["edge_index"]
    centroid, x = centralize(batch["x"], bidx)
    f_ij = localize(x, ei, norm_x_diff=cfg["norm_x_diff"])
    (h, chi), (e, xi) = gcp_embedding(P, "gcp_embedding.", batch["h"], batch["chi"], batch["e"], batch["xi"], ei,
                                      f_ij, cfg)
    for i in range(num_layers):
        (h, chi), x = gcp_interactions(P, f"interaction_layers.{i}.", h, chi, e, xi, ei, f_ij, cfg, layer_cfg,
                                       node_pos=x)
    return dict(h=h, chi=chi, e=e, xi=xi, f_ij=f_ij, x=decentralize(x, bidx, centroid))


def lba_forward(P: Params, batch: Mapping[str, Tensor], cfg: Mapping, layer_cfg: Mapping, num_layers: int):
    bidx, ei = batch["batch"], batch["edge_index"]
    _, x = centralize(batch["x"], bidx)
    f_ij = localize(x, ei, norm_x_diff=cfg["norm_x_diff"])
    (h, chi), (e, xi) = gcp_embedding(P, "gcp_embedding.", batch["h"], batch["chi"], batch["e"], batch["xi"], ei,
                                      f_ij, cfg)
    for i in range(num_layers):
        h, chi = gcp_interactions(P, f"interaction_layers.{i}.", h, chi, e, xi, ei, f_ij, cfg, layer_cfg)
    s, v = gcp_layer_norm(P, "invariant_node_projection.0.", h, chi)  # gcpnet_lba_module.py:176-184
    out = gcp2(P, "invariant_node_projection.1.", s, v, ei, f_ij, node_inputs=True,
               nonlinearities=tuple(cfg["nonlinearities"]), vector_gate=cfg["vector_gate"],
               frame_gate=cfg["frame_gate"], ablate_frame_updates=cfg["ablate_frame_updates"],
               enable_e3_equivariance=cfg["enable_e3_equivariance"])
    out = scatter(out, bidx, reduce="mean")
    out = F.relu(out @ P["dense.0.weight"].t() + P["dense.0.bias"])
    out = out @ P["dense.3.weight"].t() + P["dense.3.bias"]
    return dict(h=h, chi=chi, pred=out.squeeze())


# ----------------------------------------------------------------------------------------------------------
# CPD task module -- src/models/gcpnet_cpd_module.py: forward :153-218, sampling loop :281-360
# (pinned by tests/golden/model_cpd_small.npz, generated by the reference's real LitModule)
# ----------------------------------------------------------------------------------------------------------
def _cpd_decoder_cfg(cfg: Mapping) -> Dict:
    c = dict(cfg)  # gcpnet_cpd_module.py:94-97
    c["vector_gate"], c["frame_gate"], c["ablate_frame_updates"] = cfg["frame_gate"], False, True
    return c


def _cpd_embed(P: Params, h, chi, e, xi, edge_index, frames, cfg, mask):
    kw = dict(vector_gate=cfg["vector_gate"], frame_gate=cfg["frame_gate"], ablate_frame_updates=cfg["ablate_frame_updates"],
              ablate_scalars=cfg["ablate_scalars"], ablate_vectors=cfg["ablate_vectors"],
              enable_e3_equivariance=cfg["enable_e3_equivariance"], nonlinearities=(None, None), node_mask=mask)
    er = gcp2(P, "gcp_embedding.edge_embedding.", e, xi, edge_index, frames, node_inputs=False, **kw)
    nr = gcp2(P, "gcp_embedding.node_embedding.", h, chi, edge_index, frames, node_inputs=True, **kw)
    return gcp_layer_norm(P, "gcp_embedding.node_normalization.", *nr), gcp_layer_norm(P, "gcp_embedding.edge_normalization.", *er)


def _cpd_project(P: Params, h, chi, edge_index, frames, cfg, mask):
    return gcp2(P, "invariant_node_projection.", h, chi, edge_index, frames, node_inputs=True, nonlinearities=(None, None),
                vector_gate=cfg["vector_gate"], frame_gate=cfg["frame_gate"], ablate_frame_updates=cfg["ablate_frame_updates"],
                ablate_scalars=cfg["ablate_scalars"], ablate_vectors=cfg["ablate_vectors"],
                enable_e3_equivariance=cfg["enable_e3_equivariance"], node_mask=mask)


def cpd_forward(P: Params, batch: Mapping[str, Tensor], cfg: Mapping, layer_cfg: Mapping, num_encoder_layers: int,
                num_decoder_layers: int, autoregressive_decoder: bool, decoder_residual_updates: bool = True):
    """Returns dict(out=...) with `out` the [N, vocab] projection (autoregressive decoder, teacher forcing on batch["seq"]) or
    (logits, log_probs) of the MLP decoder; plus h, chi."""
    mask, ei = batch["mask"], batch["edge_index"]
    _, x = centralize(batch["x"], batch["batch"], node_mask=mask)
    frames = localize(x, ei, norm_x_diff=cfg["norm_x_diff"], node_mask=mask)
    (h, chi), (e, xi) = _cpd_embed(P, batch["h"], batch["chi"], batch["e"], batch["xi"], ei, frames, cfg, mask)
    for k in range(num_encoder_layers):
        h, chi = gcp_interactions(P, f"encoder_layers.{k}.", h, chi, e, xi, ei, frames, cfg, layer_cfg, node_mask=mask)
    pcfg = cfg
    if autoregressive_decoder:
        pcfg = _cpd_decoder_cfg(cfg)
        seq_emb = P["atom_embedding.weight"][batch["seq"]][ei[0]]  # :186-188
        seq_emb = torch.where((ei[0] >= ei[1])[:, None], torch.zeros_like(seq_emb), seq_emb)
        e = torch.cat((e, seq_emb), dim=-1)
        for k in range(num_decoder_layers):
            # `encoder_embedding = (h, chi)` (:183) names the very tensors the masked layers update IN PLACE and return
            # (gcpnet.py:1248-1251): from the second decoder layer on, "the encoder's representation" is the previous decoder
            # layer's output -- node_rep_regressive is always the layer's own input
            h, chi = gcp_interactions(P, f"decoder_layers.{k}.", h, chi, e, xi, ei, frames, pcfg, layer_cfg, node_mask=mask,
                                      regressive=(h, chi))
    out = _cpd_project(P, h, chi, ei, frames, pcfg, mask)
    if not autoregressive_decoder:
        out = mlp_decoder(P, "decoder.", out, residual_updates=decoder_residual_updates)
    return dict(out=out, h=h, chi=chi, f_ij=frames)


def cpd_sample(P: Params, h, chi, e, xi, edge_index, frames, encoder_node_mask, cfg: Mapping, layer_cfg: Mapping,
               num_encoder_layers: int, num_decoder_layers: int, num_samples: int, temperature: float = 0.1, sampler=None) -> Tensor:
    """`autoregressively_generate_samples` (:281-360), step by step as written there; `sampler(logits) -> indices` stands in for
    `Categorical(logits=...).sample()` (tests pass argmax)."""
    n = h.shape[0]
    (h, chi), (e, xi) = _cpd_embed(P, h, chi, e, xi, edge_index, frames, cfg, encoder_node_mask)
    for k in range(num_encoder_layers):
        h, chi = gcp_interactions(P, f"encoder_layers.{k}.", h, chi, e, xi, edge_index, frames, cfg, layer_cfg, node_mask=encoder_node_mask)
    dcfg = _cpd_decoder_cfg(cfg)
    h, chi = h.repeat(num_samples, 1), chi.repeat(num_samples, 1, 1)  # :309-310
    e, xi = e.repeat(num_samples, 1), xi.repeat(num_samples, 1, 1)
    ei = torch.cat([edge_index + k * n for k in range(num_samples)], dim=-1)  # :312-316
    fr = frames.repeat(num_samples, 1, 1)
    seq = torch.zeros(num_samples * n, dtype=torch.long)
    seq_emb = torch.zeros(num_samples * n, P["atom_embedding.weight"].shape[0], dtype=h.dtype)
    cache = [(h.clone(), chi.clone()) for _ in range(num_decoder_layers)]
    mask_all = encoder_node_mask.repeat(num_samples)
    for i in range(n):
        se = seq_emb[ei[0]]  # :326-328
        se = torch.where((ei[0] >= ei[1])[:, None], torch.zeros_like(se), se)
        em = (ei[1] % n) == i  # :330-333
        ei_i, e_i, xi_i, fr_i = ei[:, em], torch.cat((e, se), dim=-1)[em], xi[em], fr[em]
        nm = torch.zeros(num_samples * n, dtype=torch.bool)
        nm[i::n] = True
        nm = nm & mask_all
        for j in range(num_decoder_layers):  # :341-355
            full = gcp_interactions(P, f"decoder_layers.{j}.", cache[j][0], cache[j][1], e_i, xi_i, ei_i, fr_i, dcfg, layer_cfg,
                                    node_mask=nm, regressive=cache[0])
            cache[j] = (full[0], full[1])  # (the layer writes its result into the tensors it was given: gcpnet.py:1248-1251)
            oh, oc = full[0][nm], full[1][nm]
            if j < num_decoder_layers - 1:
                cache[j + 1][0][i::n] = oh
                cache[j + 1][1][i::n] = oc
        logits = _cpd_project(P, oh, oc, ei_i, fr_i, dcfg, nm)  # :357-363
        scaled = logits / temperature
        seq[i::n] = sampler(scaled) if sampler is not None else torch.distributions.Categorical(logits=scaled).sample()
        seq_emb[i::n] = P["atom_embedding.weight"][seq[i::n]]
    return seq.reshape(num_samples, n)


# ----------------------------------------------------------------------------------------------------------
# reference-default configs (values of configs/model/module_cfg/gcp_module_nms.yaml etc.), as plain dicts
# ----------------------------------------------------------------------------------------------------------
def default_module_cfg(**over) -> Dict:
    cfg = dict(
        norm_x_diff=True, scalar_gate=0, vector_gate=True, vector_residual=False, vector_frame_residual=False,
        frame_gate=False, sigma_frame_gate=False, scalar_nonlinearity="relu", vector_nonlinearity=None,
        nonlinearities=("relu", None), bottleneck=4, default_vector_residual=False, default_bottleneck=4,
        node_positions_weight=1.0, ablate_frame_updates=False, ablate_scalars=False, ablate_vectors=False,
        ablate_x_force_update=True, enable_e3_equivariance=False,
    )
    cfg.update(over)
    return cfg


def default_layer_cfg(**over) -> Dict:
    mp = dict(num_message_layers=8, self_message=True, use_residual_message_gcp=True)
    cfg = dict(pre_norm=False, num_feedforward_layers=2, dropout=0.1, nonlinearity_slope=1e-2, mp_cfg=mp)
    for k, v in over.items():
        (mp if k in mp else cfg)[k] = v
    return cfg


def random_rotation(seed: int = 0) -> Tensor:
    """Proper rotation from a seeded QR (tests/test_gcpnet_equivariance.py uses scipy's Rotation.random)."""
    g = torch.Generator().manual_seed(seed)
    q, r = torch.linalg.qr(torch.randn(3, 3, generator=g, dtype=torch.float64))
    q = q * torch.sign(torch.diagonal(r))
    if torch.det(q) < 0:
        q[:, 0] = -q[:, 0]
    return q.to(torch.float32)


__all__ = [n for n in dir() if not n.startswith("_") and n not in ("math", "torch", "F", "annotations")]


# ----------------------------------------------------------------------------------------------------------
# input side: NMS featuriser, src/datamodules/components/nms_dataset.py:23-61 with
# helper.py:16-59 (_normalize, _rbf, _orientations); pinned by tests/golden/nms_features.npz (generated with the
# reference's real helper functions) and, for the radius graph, tests/golden/radius_graph.npz (scipy cKDTree)
# ----------------------------------------------------------------------------------------------------------
def nms_features(x: Tensor, vel: Tensor, edge_attr: Tensor, edge_index: Tensor, batch: Tensor, d_max: float = 4.5,
                 num_rbf: int = 16) -> Dict[str, Tensor]:
    row, col = edge_index[0], edge_index[1]
    ev = x[row] - x[col]  # nms_dataset.py:33
    d = ev.norm(dim=-1)
    mu = torch.linspace(0.0, d_max, num_rbf).view(1, -1)  # helper.py:40-45
    rbf = torch.exp(-(((d.unsqueeze(-1) - mu) / (d_max / num_rbf)) ** 2))
    e = torch.nan_to_num(torch.cat((edge_attr, rbf), dim=-1))
    xi = torch.nan_to_num(torch.nan_to_num(ev / d.unsqueeze(-1)).unsqueeze(-2))  # helper.py:23-25, nms_dataset.py:41-43
    h = torch.sqrt((vel ** 2).sum(-1)).unsqueeze(-1)  # nms_dataset.py:58
    fwd, bwd = torch.zeros_like(x), torch.zeros_like(x)  # helper.py:52-59, per graph
    same_next = batch[1:] == batch[:-1]
    f = torch.nan_to_num((x[1:] - x[:-1]) / (x[1:] - x[:-1]).norm(dim=-1, keepdim=True))
    fwd[:-1][same_next] = f[same_next]
    bwd[1:][same_next] = -f[same_next]
    chi = torch.stack((vel, fwd, bwd), dim=1)
    return dict(h=h, chi=chi, e=e, xi=xi)


def lba_features(x: Tensor, edge_index: Tensor, batch: Optional[Tensor] = None, d_max: float = 4.5, num_rbf: int = 16) -> Dict[str, Tensor]:
    """ATOM3D / LBA featuriser, src/datamodules/components/atom3d_dataset.py:42-84: e = RBF(|x_row - x_col|) [E, num_rbf] with
    D_max = the edge cutoff, xi = unit(x_row - x_col) [E, 1, 3] (both nan_to_num'ed), chi = `_orientations(x)` [N, 2, 3] along the
    atom order of each graph (helper.py:52-59).  Pinned by tests/golden/lba_features.npz (the reference's real LBATransform)."""
    row, col = edge_index[0], edge_index[1]
    ev = x[row] - x[col]  # atom3d_dataset.py:53
    d = ev.norm(dim=-1)
    mu = torch.linspace(0.0, d_max, num_rbf).view(1, -1)  # helper.py:40-45
    e = torch.nan_to_num(torch.exp(-(((d.unsqueeze(-1) - mu) / (d_max / num_rbf)) ** 2)))
    xi = torch.nan_to_num(torch.nan_to_num(ev / d.unsqueeze(-1)).unsqueeze(-2))
    fwd, bwd = torch.zeros_like(x), torch.zeros_like(x)
    b = batch if batch is not None else torch.zeros(x.shape[0], dtype=torch.long)
    same_next = b[1:] == b[:-1]
    f = torch.nan_to_num((x[1:] - x[:-1]) / (x[1:] - x[:-1]).norm(dim=-1, keepdim=True))
    fwd[:-1][same_next] = f[same_next]
    bwd[1:][same_next] = -f[same_next]
    return dict(e=e, xi=xi, chi=torch.stack((fwd, bwd), dim=1))


def collate(graphs: Sequence[Mapping[str, Tensor]]) -> Dict[str, Tensor]:
    """torch_geometric 2.1 `Batch.from_data_list` for dict samples: tensors cat along dim 0, `*index*` attributes along the last
    dim with node offsets, scalars -> [G]; plus `batch` and `ptr` (documented PyG semantics restated; PyG is not in the image)."""
    counts = [int((g["x"] if "x" in g else g["h"]).shape[0]) for g in graphs]
    offs = [0]
    for c in counts:
        offs.append(offs[-1] + c)
    out: Dict[str, Tensor] = {}
    for k in graphs[0]:
        vals = [torch.as_tensor(g[k]) for g in graphs]
        if vals[0].dim() == 0:
            out[k] = torch.stack(vals)
        elif "index" in k:
            out[k] = torch.cat([v + o for v, o in zip(vals, offs[:-1])], dim=-1)
        else:
            out[k] = torch.cat(vals, dim=0)
    out["batch"] = torch.repeat_interleave(torch.arange(len(graphs)), torch.tensor(counts))
    out["ptr"] = torch.tensor(offs)
    return out


def radius_graph(x: Tensor, batch: Tensor, radius: float = 4.5, max_neighbors: int = 32, select: str = "nearest") -> Tensor:
    """K nearest other nodes of the same graph within `radius`, ascending by distance; edges (row = neighbour, col = node),
    col-sorted.  Brute force in float64 (small inputs only).
    select="first": torch_cluster 1.6.0's selection as its CUDA kernel makes it (torch_cluster/radius.py `radius_graph` ->
    csrc/cuda/radius_cuda.cu `radius_kernel`, the call of src/datamodules/components/atom3d_dataset.py:110-112): per target the
    nodes of its graph in ascending index order, taken when dist^2 < r^2, at most max_neighbors + 1 with the node itself among the
    candidates, the self loop removed afterwards."""
    xd = x.double()
    if select == "first":
        # that kernel decides in the coordinates' own precision: float differences, `dist += d * d` over x, y, z contracted to fused
        # multiply-adds (one rounding per step: the exact product + the running sum, rounded to float -- restated through float64,
        # which holds a float product exactly), compared with float(r * r)
        xf = x.float()
        df = (xf.unsqueeze(0) - xf.unsqueeze(1)).double()  # df[target, candidate] = x[candidate] - x[target], a float each
        dist = (df[..., 0] * df[..., 0]).float().double()
        dist = (df[..., 1] * df[..., 1] + dist).float().double()
        dist = (df[..., 2] * df[..., 2] + dist).float()
        ok = (batch.unsqueeze(1) == batch.unsqueeze(0)) & (dist < torch.tensor(radius * radius, dtype=torch.float64).float())
        rank = torch.cumsum(ok.long(), dim=1)  # position of every in-range node in the target's index-order walk (1-based)
        keep = ok & (rank <= max_neighbors + 1)
        keep.fill_diagonal_(False)
        col, row = torch.nonzero(keep, as_tuple=True)
        return torch.stack((row, col))
    d2 = ((xd.unsqueeze(1) - xd.unsqueeze(0)) ** 2).sum(-1)
    ok = (batch.unsqueeze(1) == batch.unsqueeze(0)) & (d2 <= radius * radius)
    ok.fill_diagonal_(False)
    d2 = torch.where(ok, d2, torch.full_like(d2, float("inf")))
    k = min(max_neighbors, x.shape[0])
    val, idx = torch.topk(d2, k, dim=1, largest=False, sorted=True)
    keep = torch.isfinite(val)
    col = torch.arange(x.shape[0]).unsqueeze(1).expand_as(idx)[keep]
    return torch.stack((idx[keep], col))
