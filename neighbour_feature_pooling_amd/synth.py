"""Deterministic synthetic feature maps.

Integer-hash based (splitmix64 -> sum of four 16-bit uniforms), so the same
(shape, seed, kind) gives bit-identical float32 arrays on every machine — no
libm transcendental, no torch RNG.  Fixtures under tests/golden/ store only the
reference's OUTPUTS; the inputs are regenerated from (shape, seed, kind).
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(z):
    z = (z + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def feature_map(shape, seed=0, kind="normal"):
    """float32 ndarray of `shape`.

    kind: 'normal'  ~ unit-variance bell (Irwin-Hall of 4 uniforms, |x| <= 3.47)
          'relu'    max(normal, 0)  — what a ResNet stage hands to NFP (~50 % zeros)
          'uniform' U(-1, 1)
          'smooth'  normal + a per-channel offset, so neighbouring pixels correlate
    """
    n = int(np.prod(shape))
    with np.errstate(over="ignore"):
        ctr = np.arange(n, dtype=np.uint64) + np.uint64(seed) * np.uint64(0x632BE59BD9B4E019)
        h = _splitmix64(_splitmix64(ctr))
    m16 = np.uint64(0xFFFF)
    if kind == "uniform":
        v = ((h >> np.uint64(11)).astype(np.float64) + 0.5) / float(1 << 53) * 2.0 - 1.0
    else:
        s = ((h & m16) + ((h >> np.uint64(16)) & m16) + ((h >> np.uint64(32)) & m16)
             + (h >> np.uint64(48))).astype(np.float64)
        v = (s - 2.0 * 65535.0) / 65536.0 * 1.7320508075688772
    v = v.reshape(shape)
    if kind == "relu":
        v = np.maximum(v, 0.0)
    elif kind == "smooth":
        B, C = shape[0], shape[1]
        off = feature_map((B, C) + (1,) * (len(shape) - 2), seed + 7919, "normal").astype(np.float64)
        v = 0.35 * v + off
    elif kind not in ("normal", "uniform"):
        raise ValueError(kind)
    return v.astype(np.float32)
