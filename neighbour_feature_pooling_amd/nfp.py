"""NFPPooling — drop-in for models/pooling/nfp.py::NFPPooling (nfp.py:15-375).

Same constructor, same forward contract ([B,C,H,W] -> [B, k*k-1, H', W']), same
public attributes, same error on an unknown measure; the work is done by the
fused gfx950 kernels behind include/nfp.h instead of two frozen depthwise convs
and a chain of ATen ops.  The module has no trainable parameters (as in the
reference, whose two conv weights are frozen: nfp.py:61,82).
"""
import torch
import torch.nn as nn

from . import _abi
from .functional import NfpConfig, nfp, nfp_multi_radius

_DISPATCH = set(_abi.MEASURES) | set(_abi.MEASURE_ALIASES)


class NFPPooling(nn.Module):
    def __init__(self, in_channels, R=1, measure='norm', p=1, stride=1, padding=0,
                 dilation=1, bias=False, padding_mode='reflect', similarity=True,
                 eps=1e-6, input_size=224, q_scs=1e-6):
        super().__init__()
        if bias:
            # nfp.py:46,57 would give both frozen-weight convs TRAINABLE random biases; no caller uses it.
            raise NotImplementedError("NFPPooling(bias=True) is not supported by the HIP implementation")
        if padding_mode not in _abi.PAD_MODES:
            raise ValueError(f"padding_mode must be one of {_abi.PAD_MODES}, got {padding_mode!r}")
        self.in_size = input_size
        self.measure = measure.lower()          # nfp.py:21
        self.in_channels = in_channels
        self.R = R
        self.stride = stride
        self.padding = padding
        self.padding_mode = padding_mode
        self.similarity = similarity
        self.p = p
        self.dilation = dilation
        self.bias = bias
        self.eps = eps
        self.q_scs = q_scs
        self.kernel_size = int(2 * self.R + 1)          # nfp.py:38
        self.out_channels = int(self.kernel_size ** 2 - 1)  # nfp.py:39
        # nfp.py:74 tests the RAW string, nfp.py:85 the lower-cased one: 'Norm' yields |neighbour|.
        self._diff_weights = measure in ('norm', 'rmse', 'mahalanobis')
        if self.measure not in _DISPATCH:
            raise RuntimeError(f'Similarity measure {self.measure} not implemented')  # nfp.py:120

    # -- the op ------------------------------------------------------------------------------
    @property
    def config(self):
        """Frozen view of the op-defining attributes (they are plain, assignable attributes as in the
        reference, so the view is rebuilt only when one of them changed)."""
        sig = (self.R, self.measure, self.p, self.stride, self.padding, self.dilation, self.padding_mode,
               self.similarity, self.eps, self.q_scs, self._diff_weights)
        cached = self.__dict__.get("_cfg_cache")
        if cached is not None and cached[0] == sig:
            return cached[1]
        cfg = self._build_config()
        if not torch.compiler.is_compiling():      # (no attribute mutation inside a traced forward)
            self.__dict__["_cfg_cache"] = (sig, cfg)
        return cfg

    def _build_config(self):
        return NfpConfig(R=int(self.R), measure=_abi.MEASURE_ALIASES.get(self.measure, self.measure),
                         p=self.p, stride=int(self.stride), padding=int(self.padding),
                         dilation=int(self.dilation), padding_mode=self.padding_mode,
                         similarity=bool(self.similarity), eps=float(self.eps), q_scs=float(self.q_scs),
                         diff_weights=self._diff_weights)

    def forward(self, x):
        if x.dim() == 4 and x.shape[1] != self.in_channels:
            raise RuntimeError(f"NFPPooling expected input with {self.in_channels} channels, "
                               f"got {x.shape[1]} channels instead")
        return nfp(x, self.config)

    @property
    def output_size(self):
        """nfp.py:125-130 (square-only helper based on input_size)."""
        return (self.in_size + 2 * self.padding - self.dilation * (self.kernel_size - 1) - 1) // self.stride + 1

    def extra_repr(self):
        return (f"in_channels={self.in_channels}, R={self.R}, measure={self.measure!r}, p={self.p}, "
                f"stride={self.stride}, padding={self.padding}, dilation={self.dilation}, "
                f"padding_mode={self.padding_mode!r}, similarity={self.similarity}")

    # -- checkpoint compatibility ------------------------------------------------------------
    # A reference NFPPooling state-dict holds its two frozen conv weights,
    # comp_neighbors.weight [C*N,1,k,k] and center_value.weight [C,1,k,k] (nfp.py:42-82).
    # They are constants of (C, R, measure); emit them so reference code can load our
    # checkpoints, and accept (ignore) them when loading reference checkpoints.
    def _frozen_weights(self):
        C, k, N, R = int(self.in_channels), self.kernel_size, self.out_channels, int(self.R)
        centre = torch.zeros(C, 1, k, k)
        centre[:, :, R, R] = 1
        comp = torch.zeros(C * N, 1, k, k)
        taps = [t for t in range(k * k) if t != (k * k) // 2]
        ky = torch.tensor([t // k for t in taps]).repeat(C)
        kx = torch.tensor([t % k for t in taps]).repeat(C)
        rows = torch.arange(C * N)
        if self._diff_weights:
            comp[:, :, R, R] = 1
            comp[rows, 0, ky, kx] = -1
        else:
            comp[rows, 0, ky, kx] = 1
        return comp, centre

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        super()._save_to_state_dict(destination, prefix, keep_vars)
        comp, centre = self._frozen_weights()
        destination[prefix + 'comp_neighbors.weight'] = comp
        destination[prefix + 'center_value.weight'] = centre

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys,
                              unexpected_keys, error_msgs):
        for key in ('comp_neighbors.weight', 'center_value.weight'):
            state_dict.pop(prefix + key, None)
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys,
                                      unexpected_keys, error_msgs)


class EnhancedNFPPooling(NFPPooling):
    """Constructible stand-in for models.pooling.enhanced_nfp.EnhancedNFPPooling, which
    models/nfp_heads.py:6 and models/vittiny_models_new.py:7 import but the reference does
    not ship.  Call sites pass (in_channels=, R=, measure=, padding=[, **kw]) — see
    nfp_heads.py:18-23,58-63,88-93,208-214 — which NFPPooling accepts; unknown extra
    keyword arguments are ignored.  Parity for this class is unpinned (no reference code)."""

    def __init__(self, in_channels, R=1, measure='cosine', padding=0, **kw):
        known = {k: kw[k] for k in ('p', 'stride', 'dilation', 'bias', 'padding_mode', 'similarity',
                                    'eps', 'input_size', 'q_scs') if k in kw}
        super().__init__(in_channels, R=R, measure=measure, padding=padding, **known)


class MultiRadiusNFPPooling(nn.Module):
    """The NFP part of models/nfp_heads.py::MultiRadiusNFPHead (nfp_heads.py:80-118): one EnhancedNFPPooling per radius
    on the SAME feature map, concatenated along the channel axis (nfp_heads.py:88-93, 109-110).  `nfp_blocks` holds the
    per-radius layers exactly as the reference's ModuleList does (so state dicts line up); forward() returns
    torch.cat([blk(x) for blk in nfp_blocks], dim=1) — for R_list = (1, 2) on the GPU from one fused pass over x."""

    def __init__(self, in_channels, R_list=(1, 2), measure="cosine", **kw):
        super().__init__()
        self.nfp_blocks = nn.ModuleList([EnhancedNFPPooling(in_channels=in_channels, R=R, measure=measure, padding=R, **kw)
                                         for R in R_list])
        self.in_channels = in_channels
        self.out_channels = sum(b.out_channels for b in self.nfp_blocks)

    def forward(self, x):
        blocks = list(self.nfp_blocks)
        if len(blocks) == 2 and all(isinstance(b, NFPPooling) for b in blocks):
            if x.dim() == 4 and x.shape[1] != self.in_channels:
                raise RuntimeError(f"MultiRadiusNFPPooling expected input with {self.in_channels} channels, "
                                   f"got {x.shape[1]} channels instead")
            return nfp_multi_radius(x, blocks[0].config, blocks[1].config)
        return torch.cat([b(x) for b in blocks], dim=1)
