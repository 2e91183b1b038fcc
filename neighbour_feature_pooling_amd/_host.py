"""NFP on HOST (CPU) tensors, in plain torch ops.

Why this exists: the reference's heads call their NFP layer on a CPU dummy inside
__init__ to probe the output channels (models/nfp_heads.py:24-27,95-97,166-167,
models/resnet18.py:22-25), so a drop-in module must accept CPU tensors.  It is
NOT a fallback: a CUDA tensor never reaches this file (functional.nfp raises if
the HIP library cannot serve it), and nothing here touches oracle/.

Formulation: pad once, then every kernel tap is a strided slice of the padded
map (no depthwise one-hot convolutions as in nfp.py:42-82); the measures follow
nfp.py:141-374 on the stacked [B,C,N,Ho,Wo] neighbours; autograd differentiates.
"""
import torch
import torch.nn.functional as F

_PAD = {"zeros": "constant", "reflect": "reflect", "replicate": "replicate", "circular": "circular"}


def _taps(x, cfg):
    R, pad, s, d = cfg.R, cfg.padding, cfg.stride, cfg.dilation
    k = 2 * R + 1
    xp = F.pad(x, (pad, pad, pad, pad), mode=_PAD[cfg.padding_mode]) if pad > 0 else x
    Hp, Wp = xp.shape[-2:]
    span = d * (k - 1) + 1
    Ho, Wo = (Hp - span) // s + 1, (Wp - span) // s + 1
    if Ho < 1 or Wo < 1:
        raise RuntimeError(f"Kernel size can't be greater than actual input size ({Hp}x{Wp} padded, span {span})")

    def tap(ky, kx):
        return xp[:, :, ky * d: ky * d + (Ho - 1) * s + 1: s, kx * d: kx * d + (Wo - 1) * s + 1: s]

    centre = tap(R, R)
    neigh = torch.stack([tap(t // k, t % k) for t in range(k * k) if t != (k * k) // 2], dim=2)
    return centre.unsqueeze(2), neigh  # [B,C,1,Ho,Wo], [B,C,N,Ho,Wo]


def nfp_host(x, cfg):
    a, b = _taps(x, cfg)
    m, sim, eps = cfg.measure, cfg.similarity, cfg.eps
    v = (a - b) if cfg.diff_weights else b  # what the reference's comp_neighbors conv yields
    if m == "norm":
        r = torch.linalg.norm(v, ord=cfg.p, dim=1)
        return -r if sim else r
    if m == "rmse":
        r = torch.sqrt(torch.mean(v ** 2, dim=1))
        return -r if sim else r
    if m == "cosine":
        r = F.cosine_similarity(a, b, dim=1, eps=eps)
        return r if sim else 1 - r
    if m == "dot":
        r = (a * b).sum(1)
        return r if sim else -r
    if m == "attention":
        r = F.softmax((a * b).sum(1), dim=1)
        return r if sim else -r
    if m == "geman":
        q = (a - b) ** 2
        r = (q / (q + eps)).mean(1)
        return r if sim else 1 - r
    if m == "emd":
        r = (a - b).abs().sum(1)
        return -r if sim else r
    if m == "canberra":
        r = ((a - b).abs() / (a.abs() + b.abs() + eps)).sum(1)
        return -r if sim else r
    if m in ("hellinger", "squaredchord"):
        q = ((a.abs() + eps).sqrt() - (b.abs() + eps).sqrt()) ** 2
        r = torch.sqrt(0.5 * q.sum(1)) if m == "hellinger" else q.sum(1)
        return -r if sim else r
    if m == "chisquared1":
        r = ((a - b) ** 2 / (a.abs() + b.abs() + eps)).sum(1)
        return -r if sim else r
    if m == "chisquared2":
        r = ((a - b) ** 2 / (a.abs() + eps)).sum(1)
        return -r if sim else r
    if m == "gfc":
        r = (a * b).sum(1) / (torch.norm(a, dim=1) * torch.norm(b, dim=1) + eps)
        return r if sim else -r
    if m == "pearson":
        ac, bc = a - a.mean(1, keepdim=True), b - b.mean(1, keepdim=True)
        r = (ac * bc).sum(1) / torch.sqrt((ac ** 2).sum(1) * (bc ** 2).sum(1) + eps)
        return r if sim else -r
    if m == "jeffrey":
        ca, cb = a.abs() + eps, b.abs() + eps
        r = (ca * torch.log(ca / cb) + cb * torch.log(cb / ca)).sum(1)
        return -r if sim else r
    if m == "smith":
        A, Bn = a.abs(), b.abs()
        r = 1 - torch.minimum(A, Bn).sum(1) / (torch.minimum(A.sum(1), Bn.sum(1)) + eps)
        return r if sim else -r
    if m == "scs":
        # nfp.py:359-374 — the (B,N,H,W)/(B,1,N,H,W) division broadcasts to (B,B,N,H,W) and
        # mean(dim=1) then averages over the batch; reproduced as written in the reference.
        na = torch.norm(a, dim=1, keepdim=True) + cfg.q_scs
        nb = torch.norm(b, dim=1, keepdim=True) + cfg.q_scs
        c = (a * b).sum(1) / (na * nb)
        r = torch.nan_to_num(torch.sign(c) * c.abs() ** cfg.p, nan=0.0, posinf=0.0, neginf=0.0)
        if not sim:
            r = 1 - r
        return r.mean(dim=1)
    raise RuntimeError(f"Similarity measure {m} not implemented")
