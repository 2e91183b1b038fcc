"""neighbour_feature_pooling_amd — MI355X (gfx950) Neighbourhood Feature Pooling.

Drop-in for the reference's NFP hot path:
    NFPPooling, EnhancedNFPPooling   <- models/pooling/nfp.py::NFPPooling
    nfp_pooling                      <- models/NFP_Pooling.py::nfp_pooling
    MultiRadiusNFPPooling            <- the per-radius layers + concatenation of models/nfp_heads.py::MultiRadiusNFPHead
    nfp_op, nfp_pool, NfpConfig      functional forms (autograd ops over libnfp_hip.so)
    nfp_pooled                       <- F.adaptive_avg_pool2d(NFPPooling(feat), 1) of models/texture_pooling.py:251-252, 320-321
"""
from .functional import NfpConfig, nfp_multi_radius, nfp_pool, nfp_pooled
from .functional import nfp as nfp_op  # (`nfp` itself is the submodule holding NFPPooling)
from .nfp import EnhancedNFPPooling, MultiRadiusNFPPooling, NFPPooling
from .pooling import nfp_pooling
from . import _ops  # registers torch.ops.nfp_amd.* (torch.compile support)

__all__ = ["NFPPooling", "EnhancedNFPPooling", "MultiRadiusNFPPooling", "nfp_pooling", "nfp_op", "nfp_pool", "nfp_pooled",
           "nfp_multi_radius", "NfpConfig"]
__version__ = "0.4.0"
