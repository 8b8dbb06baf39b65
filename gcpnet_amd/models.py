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
