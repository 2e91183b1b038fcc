"""neighbour_feature_pooling_amd — MI355X (gfx950) Neighbourhood Feature Pooling.

Drop-in for the reference's NFP hot path:
    NFPPooling, EnhancedNFPPooling   <- models/pooling/nfp.py::NFPPooling
    nfp_pooling                      <- models/NFP_Pooling.py::nfp_pooling
    nfp, NfpConfig                   functional form (autograd op over libnfp_hip.so)
"""
from .functional import NfpConfig, nfp
from .nfp import EnhancedNFPPooling, NFPPooling
from .pooling import nfp_pooling

__all__ = ["NFPPooling", "EnhancedNFPPooling", "nfp_pooling", "nfp", "NfpConfig"]
__version__ = "0.1.0"
