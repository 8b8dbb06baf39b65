"""Task-level callers of the hot path: the `forward` / `step` of the reference's NMS and LBA LightningModules
(src/models/gcpnet_nms_module.py:56-83,127-158; src/models/gcpnet_lba_module.py:59-110,155-193) as plain
nn.Modules with identical sub-module / parameter names, so reference checkpoints' state_dicts load by key.
Lightning, torchmetrics and logging hooks are out of scope.
"""
from __future__ import annotations

from typing import Any, Tuple

import torch
from torch import nn

from . import ops
from .components import GCPLayerNorm, ScalarVector, centralize, decentralize, localize
from .config import as_cfg
from .gcpnet import GCPEmbedding, GCPInteractions, NUM_ATOM_TYPES
from .ops import GatherPlan


class Batch:
    """Attribute bag with the fields of a torch_geometric Batch that the forwards touch:
    h, chi, e, xi, x, edge_index, batch, label (and f_ij once computed)."""

    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)

    def __getitem__(self, k):
        return getattr(self, k)

    def __setitem__(self, k, v):
        setattr(self, k, v)

    def to(self, device):
        for k, v in list(vars(self).items()):
            if torch.is_tensor(v):
                setattr(self, k, v.to(device))
        return self


class GCPNetNMS(nn.Module):
    def __init__(self, layer_class=None, model_cfg=None, module_cfg=None, layer_cfg=None, **kwargs):
        super().__init__()
        model_cfg, module_cfg, layer_cfg = as_cfg(model_cfg), as_cfg(module_cfg), as_cfg(layer_cfg)
        self.module_cfg = module_cfg
        if layer_class is None:
            from functools import partial

            layer_class = partial(GCPInteractions, updating_node_positions=True)
        edge_input_dims = ScalarVector(model_cfg.e_input_dim, model_cfg.xi_input_dim)
        node_input_dims = ScalarVector(model_cfg.h_input_dim, model_cfg.chi_input_dim)
        self.edge_dims = ScalarVector(model_cfg.e_hidden_dim, model_cfg.xi_hidden_dim)
        self.node_dims = ScalarVector(model_cfg.h_hidden_dim, model_cfg.chi_hidden_dim)
        self.gcp_embedding = GCPEmbedding(edge_input_dims, node_input_dims, self.edge_dims, self.node_dims,
                                          num_atom_types=0, cfg=module_cfg)
        self.interaction_layers = nn.ModuleList(
            layer_class(self.node_dims, self.edge_dims, cfg=module_cfg, layer_cfg=layer_cfg, dropout=model_cfg.dropout)
            for _ in range(model_cfg.num_encoder_layers))
        self.criterion = nn.MSELoss()

    def forward(self, batch: Any) -> Tuple[Any, torch.Tensor]:
        x_centroid, batch.x = centralize(batch, key="x", batch_index=batch.batch)
        batch.f_ij = localize(batch.x, batch.edge_index, norm_x_diff=self.module_cfg.norm_x_diff)
        (h, chi), (e, xi) = self.gcp_embedding(batch)
        for layer in self.interaction_layers:
            (h, chi), batch.x = layer((h, chi), (e, xi), batch.edge_index, batch.f_ij, node_pos=batch.x)
        batch.h, batch.chi, batch.e, batch.xi = h, chi, e, xi
        batch.x = decentralize(batch, key="x", batch_index=batch.batch, entities_centroid=x_centroid)
        return batch, batch.x

    def step(self, batch: Any):
        labels = batch.label
        _, preds = self.forward(batch)
        return self.criterion(preds, labels), preds, labels


class GCPNetLBA(nn.Module):
    def __init__(self, layer_class=None, model_cfg=None, module_cfg=None, layer_cfg=None,
                 num_atom_types: int = NUM_ATOM_TYPES, **kwargs):
        super().__init__()
        model_cfg, module_cfg, layer_cfg = as_cfg(model_cfg), as_cfg(module_cfg), as_cfg(layer_cfg)
        self.module_cfg = module_cfg
        if layer_class is None:
            layer_class = GCPInteractions
        edge_input_dims = ScalarVector(model_cfg.e_input_dim, model_cfg.xi_input_dim)
        node_input_dims = ScalarVector(num_atom_types, model_cfg.chi_input_dim)
        self.edge_dims = ScalarVector(model_cfg.e_hidden_dim, model_cfg.xi_hidden_dim)
        self.node_dims = ScalarVector(model_cfg.h_hidden_dim, model_cfg.chi_hidden_dim)
        self.gcp_embedding = GCPEmbedding(edge_input_dims, node_input_dims, self.edge_dims, self.node_dims,
                                          num_atom_types=num_atom_types, cfg=module_cfg)
        self.interaction_layers = nn.ModuleList(
            layer_class(self.node_dims, self.edge_dims, cfg=module_cfg, layer_cfg=layer_cfg, dropout=model_cfg.dropout)
            for _ in range(model_cfg.num_encoder_layers))
        self.invariant_node_projection = nn.ModuleList([
            GCPLayerNorm(self.node_dims),
            module_cfg.selected_GCP(
                self.node_dims, (self.node_dims.scalar, 0), nonlinearities=tuple(module_cfg.nonlinearities),
                scalar_gate=module_cfg.scalar_gate, vector_gate=module_cfg.vector_gate, frame_gate=module_cfg.frame_gate,
                sigma_frame_gate=module_cfg.sigma_frame_gate, vector_frame_residual=module_cfg.vector_frame_residual,
                ablate_frame_updates=module_cfg.ablate_frame_updates,
                enable_e3_equivariance=module_cfg.enable_e3_equivariance, node_inputs=True)])
        # readout head, [num_graphs, s] -> [num_graphs, output_dim] (gcpnet_lba_module.py:104-109): the modules hold the
        # parameters (reference state_dict keys dense.0.* / dense.3.*); `_readout_head` runs them on the HIP kernels
        self.dense = nn.Sequential(
            nn.Linear(self.node_dims.scalar, self.node_dims.scalar * model_cfg.output_scale_factor), nn.ReLU(inplace=True),
            nn.Dropout(model_cfg.dense_dropout),
            nn.Linear(self.node_dims.scalar * model_cfg.output_scale_factor, model_cfg.output_dim))
        self.criterion = nn.MSELoss()

    def forward(self, batch: Any) -> Tuple[Any, torch.Tensor]:
        _, batch.x = centralize(batch, key="x", batch_index=batch.batch)
        batch.f_ij = localize(batch.x, batch.edge_index, norm_x_diff=self.module_cfg.norm_x_diff)
        (h, chi), (e, xi) = self.gcp_embedding(batch)
        for layer in self.interaction_layers:
            (h, chi) = layer((h, chi), (e, xi), batch.edge_index, batch.f_ij)
        batch.h, batch.chi, batch.e, batch.xi = h, chi, e, xi
        out = self.invariant_node_projection[0]((h, chi))
        out = self.invariant_node_projection[1](out, batch.edge_index, batch.f_ij, node_inputs=True)
        out = ops.segment_reduce(out, GatherPlan.get(batch.batch), mean=True)  # scatter(..., reduce="mean"), dim_size = max + 1
        out = self._readout_head(out).squeeze()
        return batch, out

    def _readout_head(self, x: torch.Tensor) -> torch.Tensor:
        """dense = Linear -> ReLU -> Dropout -> Linear (gcpnet_lba_module.py:104-109, applied at :184) behind the graph mean
        (segment_reduce above): both Linears on the workgroup GEMM kernel, ReLU and (train mode) dropout as HIP launches -- no
        vendor-library GEMM or ATen kernel in the readout."""
        l0, drop, l1 = self.dense[0], self.dense[2], self.dense[3]
        y = ops.activation(ops.linear_padded(x, l0.weight, l0.bias), "relu", 0.0)
        if self.training and drop.p > 0:
            y = ops.dropout(y, drop.p, group=1)
        return ops.linear_padded(y, l1.weight, l1.bias)

    def step(self, batch: Any):
        labels = batch.label
        _, preds = self.forward(batch)
        return self.criterion(preds, labels), preds, labels


class GCPNetCPD(nn.Module):
    """Computational protein design (CPD) task module: `forward` / `step` / `autoregressively_generate_samples` of the reference's
    `GCPNetCPDLitModule` (src/models/gcpnet_cpd_module.py:44-147 constructor, :153-218 forward, :220-231 training step, :281-360
    sampling loop) as a plain nn.Module with the same sub-module / parameter names.  Masked encoder layers, the autoregressive
    decoder layers (`GCPInteractions(autoregressive=True)` fed the encoder's representation as `node_rep_regressive`) and the
    invariant projection all run on the HIP kernels of this package; Lightning, torchmetrics, the test-split bookkeeping and the
    sequence-recovery metrics of the LitModule are out of scope."""

    def __init__(self, layer_class=None, node_input_dims=(6, 3), edge_input_dims=(32, 1), model_cfg=None, module_cfg=None,
                 layer_cfg=None, dropout: float = 0.1, autoregressive_decoder: bool = False, **kwargs):
        super().__init__()
        import copy

        model_cfg, module_cfg, layer_cfg = as_cfg(model_cfg), copy.copy(as_cfg(module_cfg)), as_cfg(layer_cfg)
        self.module_cfg, self.model_cfg = module_cfg, model_cfg
        self.norm_x_diff = module_cfg.norm_x_diff
        self.autoregressive_decoder = bool(autoregressive_decoder)
        if layer_class is None:
            layer_class = GCPInteractions
        self.node_dims = ScalarVector(model_cfg.h_hidden_dim, model_cfg.chi_hidden_dim)
        self.edge_dims = ScalarVector(model_cfg.e_hidden_dim, model_cfg.xi_hidden_dim)
        edge_hidden_dims = (self.edge_dims[0] + 20, self.edge_dims[1])  # (:68; the literal 20 is the reference's)
        self.gcp_embedding = GCPEmbedding(edge_input_dims, node_input_dims, self.edge_dims, self.node_dims, num_atom_types=0,
                                          cfg=module_cfg, pre_norm=False)
        self.encoder_layers = nn.ModuleList(
            layer_class(self.node_dims, self.edge_dims, cfg=module_cfg, layer_cfg=layer_cfg, dropout=dropout)
            for _ in range(model_cfg.num_encoder_layers))
        if self.autoregressive_decoder:
            # (:94-97: the decoder's blocks run without frame updates; `vector_gate` takes the value `frame_gate` had)
            module_cfg.vector_gate = module_cfg.frame_gate
            module_cfg.frame_gate = False
            module_cfg.ablate_frame_updates = True
            self.atom_embedding = nn.Embedding(model_cfg.output_dim, model_cfg.output_dim)
            self.decoder_layers = nn.ModuleList(
                layer_class(self.node_dims, edge_hidden_dims, cfg=module_cfg, layer_cfg=layer_cfg, dropout=dropout, autoregressive=True)
                for _ in range(model_cfg.num_decoder_layers))
        proj_dim = model_cfg.output_dim if self.autoregressive_decoder else self.node_dims[0]
        self.invariant_node_projection = module_cfg.selected_GCP(
            self.node_dims, (proj_dim, 0), nonlinearities=(None, None), scalar_gate=module_cfg.scalar_gate,
            vector_gate=module_cfg.vector_gate, frame_gate=module_cfg.frame_gate, sigma_frame_gate=module_cfg.sigma_frame_gate,
            vector_frame_residual=module_cfg.vector_frame_residual, ablate_frame_updates=module_cfg.ablate_frame_updates,
            ablate_scalars=module_cfg.ablate_scalars, ablate_vectors=module_cfg.ablate_vectors,
            enable_e3_equivariance=module_cfg.enable_e3_equivariance)
        if not self.autoregressive_decoder:
            from .gcpnet import GCPMLPDecoder

            self.decoder = GCPMLPDecoder(proj_dim, vocab_size=model_cfg.output_dim, num_layers=model_cfg.num_decoder_layers,
                                         residual_updates=model_cfg.decoder_residual_updates)
        self.criterion = nn.CrossEntropyLoss()

    def forward(self, batch: Any):
        """:153-218.  Returns (batch, out): `out` = (logits, log_probs) of the MLP decoder, or -- autoregressive decoder, teacher
        forcing with `batch.seq` -- the [N, vocab] scalars of the invariant projection."""
        _, batch.x = centralize(batch, key="x", batch_index=batch.batch, node_mask=batch.mask)
        batch.f_ij = localize(batch.x, batch.edge_index, norm_x_diff=self.norm_x_diff, node_mask=batch.mask)
        (h, chi), (e, xi) = self.gcp_embedding(batch)
        for layer in self.encoder_layers:
            (h, chi) = layer((h, chi), (e, xi), batch.edge_index, batch.f_ij, node_mask=batch.mask)
        if self.autoregressive_decoder:
            # (`encoder_embedding` names the tensors that the masked layers update in place and return, gcpnet.py:1248-1251:
            # from the second decoder layer on it holds the previous decoder layer's output -- kept, trained checkpoints rely on it)
            encoder_embedding = (h, chi)
            row, col = batch.edge_index[0], batch.edge_index[1]
            seq_emb = self.atom_embedding(batch.seq)[row]
            seq_emb = seq_emb * (row < col).to(seq_emb.dtype)[:, None]  # (:188: zero where source >= target)
            e = torch.cat((e, seq_emb), dim=-1)
            for layer in self.decoder_layers:
                (h, chi) = layer((h, chi), (e, xi), batch.edge_index, batch.f_ij, node_rep_regressive=encoder_embedding,
                                 node_mask=batch.mask)
        batch.h, batch.chi, batch.e, batch.xi = h, chi, e, xi
        out = self.invariant_node_projection((h, chi), batch.edge_index, batch.f_ij, node_inputs=True, node_mask=batch.mask)
        if not self.autoregressive_decoder:
            out = self.decoder(out)
        return batch, out

    def step(self, batch: Any):
        """:220-231 (`training_step`): cross entropy of the unmasked nodes' predictions against their residue types."""
        _, out = self.forward(batch)
        preds = out[0] if isinstance(out, tuple) else out
        preds, labels = preds[batch.mask], batch.seq[batch.mask]
        return self.criterion(preds, labels), preds, labels

    @torch.no_grad()
    def autoregressively_generate_samples(self, node_rep, edge_rep, edge_index, frames, encoder_node_mask, num_samples: int,
                                          temperature: float = 0.1, sampler=None):
        """:281-360 -- residues are drawn one node at a time: node i's decoder pass sees the residues sampled for the nodes before
        it through the sequence embedding on its in-edges (source < target), every decoder layer reads the cached output of the
        layer below for the other nodes, and only node i's row (one per sample) is recomputed.  `node_rep` / `edge_rep` are the RAW
        input features; `sampler(logits / temperature) -> indices` defaults to `Categorical(logits=...).sample()` (torch's CPU
        generator seeds it; tests pass argmax).  Returns [num_samples, num_nodes] residue indices.

        Index bookkeeping that the reference rebuilds with boolean masks at every step (edges whose target is node i in any
        sample) is built ONCE here: the tiled edge list is grouped by `target % num_nodes`, so step i takes a contiguous slice."""
        from torch.distributions import Categorical

        assert self.autoregressive_decoder, "the sampling loop belongs to the autoregressive decoder"
        node_rep, edge_rep = ScalarVector(*node_rep), ScalarVector(*edge_rep)
        num_nodes = node_rep[0].shape[0]
        dev = node_rep[0].device
        emb = self.gcp_embedding
        edge_rep = emb.edge_normalization(emb.edge_embedding(edge_rep, edge_index, frames, node_inputs=False, node_mask=encoder_node_mask))
        node_rep = emb.node_normalization(emb.node_embedding(node_rep, edge_index, frames, node_inputs=True, node_mask=encoder_node_mask))
        for layer in self.encoder_layers:
            node_rep = layer(node_rep, edge_rep, edge_index, frames, node_mask=encoder_node_mask)
        node_rep = ScalarVector(node_rep[0].repeat(num_samples, 1), node_rep[1].repeat(num_samples, 1, 1))
        e_s, e_v = edge_rep[0].repeat(num_samples, 1), edge_rep[1].repeat(num_samples, 1, 1)
        n_edges = edge_index.shape[1]
        offset = num_nodes * torch.arange(num_samples, device=dev).repeat_interleave(n_edges)
        ei = edge_index.repeat(1, num_samples) + offset
        fr = frames.repeat(num_samples, 1, 1)
        # edges grouped by (target node within its sample); inside a group the reference's order (sample-major) is kept
        tgt = edge_index[1].repeat(num_samples)
        order = torch.argsort(tgt, stable=True)
        ei, fr, e_s, e_v = ei[:, order], fr[order].contiguous(), e_s[order].contiguous(), e_v[order].contiguous()
        ptr = torch.zeros(num_nodes + 1, dtype=torch.long, device=dev)
        ptr[1:] = torch.cumsum(torch.bincount(tgt, minlength=num_nodes), 0)
        ptr = ptr.tolist()
        causal = (ei[0] < ei[1]).to(e_s.dtype)[:, None]  # (:326-327: sequence information flows from lower to higher node ids)
        vocab = self.atom_embedding.weight.shape[0]
        residue_sequence = torch.zeros(num_samples * num_nodes, dtype=torch.long, device=dev)
        sequence_embedding = torch.zeros(num_samples * num_nodes, vocab, device=dev)
        cache = [node_rep.clone() for _ in self.decoder_layers]
        mask_all = encoder_node_mask.repeat(num_samples)
        rows_of = torch.arange(num_samples, device=dev) * num_nodes
        for i in range(num_nodes):
            a, b = ptr[i], ptr[i + 1]
            ei_i, fr_i = ei[:, a:b], fr[a:b]
            seq_i = sequence_embedding[ei_i[0]] * causal[a:b]
            edge_i = ScalarVector(torch.cat((e_s[a:b], seq_i), dim=-1), e_v[a:b])
            node_mask = torch.zeros(num_samples * num_nodes, dtype=torch.bool, device=dev)
            node_mask[rows_of + i] = True
            node_mask &= mask_all  # (:338: nodes with missing coordinates stay masked)
            if not bool(node_mask.any()):
                continue  # (the reference would draw from an empty logits tensor here: nothing to assign)
            for j, layer in enumerate(self.decoder_layers):
                out = layer(cache[j], edge_i, ei_i, fr_i, node_rep_regressive=cache[0], node_mask=node_mask)  # (updates cache[j] in place)
                out = ScalarVector(out[0][node_mask], out[1][node_mask])
                if j < len(self.decoder_layers) - 1:
                    cache[j + 1][0][node_mask] = out[0]
                    cache[j + 1][1][node_mask] = out[1]
            logits = self.invariant_node_projection(out, ei_i, fr_i, node_inputs=True, node_mask=node_mask)
            scaled = logits / temperature
            drawn = sampler(scaled) if sampler is not None else Categorical(logits=scaled.cpu()).sample().to(dev)
            residue_sequence[node_mask] = drawn.to(torch.long)
            sequence_embedding[node_mask] = self.atom_embedding(residue_sequence[node_mask])
        return residue_sequence.reshape(num_samples, num_nodes)
