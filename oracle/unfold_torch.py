"""TEST INFRASTRUCTURE — PyTorch restatement of the reference's own CPU execution path.

The reference computes NFP as: pad -> frozen one-hot depthwise conv C -> C*N (neighbours,
nfp.py:42-50,64-82) and C -> C (centre, nfp.py:53-61) -> view [B,C,N,H,W] (nfp.py:136-139)
-> F.cosine_similarity(dim=1) / LA.norm(dim=1) (nfp.py:145,156), autograd for backward.
This file issues the same ATen op sequence (so it costs what the reference costs on the
same cores) and is what bench.py times as `cpu_baseline` on the GPU box, where
/root/reference does not exist.  tests/test_oracle_golden.py checks it against the
reference's golden outputs.  Cosine and Norm only (the two measures on the hot path).
"""
import torch
import torch.nn.functional as F

_PAD = {"zeros": "constant", "reflect": "reflect", "replicate": "replicate", "circular": "circular"}


def selector_weights(C, R, diff):
    """One-hot depthwise kernels: comp [C*N,1,k,k] picks neighbour n of its channel (or
    centre - neighbour when `diff`), centre [C,1,k,k] picks the centre tap."""
    k = 2 * R + 1
    taps = [t for t in range(k * k) if t != (k * k) // 2]
    comp = torch.zeros(C, len(taps), k, k)
    for n, t in enumerate(taps):
        comp[:, n, t // k, t % k] = -1.0 if diff else 1.0
    if diff:
        comp[:, :, R, R] = 1.0
    centre = torch.zeros(C, 1, k, k)
    centre[:, 0, R, R] = 1.0
    return comp.view(C * len(taps), 1, k, k), centre


class UnfoldNFP:
    def __init__(self, C, R=1, measure="cosine", p=2, stride=1, padding=0, dilation=1,
                 padding_mode="reflect", similarity=True, eps=1e-6):
        assert measure in ("cosine", "norm")
        self.C, self.R, self.measure, self.p = C, R, measure, p
        self.stride, self.padding, self.dilation, self.mode = stride, padding, dilation, padding_mode
        self.similarity, self.eps = similarity, eps
        self.N = (2 * R + 1) ** 2 - 1
        self.w_comp, self.w_centre = selector_weights(C, R, diff=(measure == "norm"))

    def _conv(self, x, w):
        if self.padding > 0:
            x = F.pad(x, (self.padding,) * 4, mode=_PAD[self.mode])
        return F.conv2d(x, w.to(x.dtype), None, self.stride, 0, self.dilation, groups=self.C)

    def __call__(self, x):
        nb = self._conv(x, self.w_comp)
        B, _, Ho, Wo = nb.shape
        nb = nb.reshape(B, self.C, self.N, Ho, Wo)
        if self.measure == "norm":
            r = torch.linalg.norm(nb, ord=self.p, dim=1)
            return -r if self.similarity else r
        ce = self._conv(x, self.w_centre).unsqueeze(2)
        r = F.cosine_similarity(ce, nb, dim=1, eps=self.eps)
        return r if self.similarity else 1 - r
