"""gcpnet_amd: MI355X-native implementation of GCPNet's geometry-complete message-passing hot path behind the
reference's own Python class API."""
from .components import (GCPDropout, GCPLayerNorm, ScalarVector, VectorDropout, centralize, decentralize,
                         get_nonlinearity, is_identity, localize)
from .config import AttrDict, default_layer_cfg, default_module_cfg, instantiate, load_model_config
from .gcpnet import GCP, GCP2, GCP3, GCPEmbedding, GCPInteractions, GCPInteractions2, GCPMLPDecoder, GCPMessagePassing, get_GCP_with_custom_cfg
from .models import Batch, GCPNetCPD, GCPNetLBA, GCPNetNMS
from .ops import check_weight_range, invalidate_packs, set_weight_grad_stream
from .optim import FusedAdam
from .data import collate, element_mapping, lba_featurize, nms_featurize, radius_graph

__all__ = [
    "GCP", "GCP2", "GCP3", "GCPEmbedding", "GCPInteractions", "GCPInteractions2", "GCPMLPDecoder", "GCPMessagePassing", "get_GCP_with_custom_cfg", "ScalarVector",
    "GCPLayerNorm", "GCPDropout", "VectorDropout", "centralize", "decentralize", "localize", "get_nonlinearity",
    "is_identity", "AttrDict", "default_module_cfg", "default_layer_cfg", "instantiate", "load_model_config", "Batch",
    "GCPNetNMS", "GCPNetLBA", "GCPNetCPD", "set_weight_grad_stream", "invalidate_packs", "check_weight_range", "FusedAdam", "nms_featurize", "radius_graph", "lba_featurize", "element_mapping", "collate",
]
