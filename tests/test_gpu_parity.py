"""Parity of the HIP path (through the C ABI, via the nn.Module) on a real MI355X.

Bar (BASELINE.json north_star): <= 1e-5 relative, fp32, against the reference PyTorch NFP on
identical inputs.  "Relative" is to the tensor's max magnitude (max|a-ref| / max|ref|): cosine
maps of random features have entries arbitrarily close to 0, where an element-wise ratio is
meaningless for ANY fp32 implementation.  bf16 storage is compared at 2e-2 (bf16 has 8 bits).

References used: (1) the committed golden vectors = outputs of the real reference;
(2) the CPU oracle on the same seeded inputs; (3) size-independent properties at the
BASELINE.json headline size.
"""
import numpy as np
import pytest
import torch

import cases as K
from conftest import assert_matches_golden, golden_out_shape, load_golden, nfp_switch, rel_err, same_nan_pattern

pytestmark = pytest.mark.gpu

TOL = 1e-5
# every measure of the reference's dispatch chain (nfp.py:85-120) except SharpenedCosine, whose
# reference implementation mixes batch elements (nfp.py:359-374) and has no kernel
HIP_MEASURES = {"norm", "cosine", "dot", "rmse", "geman", "attention", "emd", "canberra", "hellinger",
                "chisquared1", "chisquared2", "gfc", "pearson", "jeffrey", "squaredchord", "smith"}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    from neighbour_feature_pooling_amd import _abi
    _abi.load()
    return torch.device("cuda:0")


def _launches():
    from neighbour_feature_pooling_amd import _abi
    return _abi.load().nfp_launch_count()


def run_hip(c, dev, x_np=None, dtype=torch.float32, channels_last=False):
    from neighbour_feature_pooling_amd import NFPPooling
    x_np = K.make_input(c) if x_np is None else x_np
    x = torch.from_numpy(x_np).to(dev).to(dtype)
    if channels_last:
        x = x.contiguous(memory_format=torch.channels_last)
    x.requires_grad_(True)
    m = NFPPooling(c["shape"][1], **c["ctor"])
    n0 = _launches()
    out = m(x)
    go_np = K.make_grad_out(c, tuple(out.shape))
    out.backward(torch.from_numpy(go_np).to(dev).to(dtype))
    torch.cuda.synchronize()
    assert _launches() >= n0 + 2, "the HIP kernels did not run"
    if channels_last:
        assert x.grad.is_contiguous(memory_format=torch.channels_last)
    return out.detach().float().cpu().numpy(), x.grad.float().cpu().numpy(), go_np


def _supported(c):
    return c["ctor"]["measure"].lower() in HIP_MEASURES


@pytest.fixture(params=["auto", "generic", "atomic"])
def variant(request, monkeypatch):
    """'auto' = the dispatcher's choice (hot-path kernels where they apply); 'generic' forces the
    any-geometry kernels of nfp_gather.h (fwd_pairs / bwd_gather); 'atomic' (the switches' historical name) forces
    the table-free kernels of last resort (nfp_direct.h) that serve maps too large for those kernels' LDS tables — every
    implementation is held to the same bar on every case."""
    nfp_switch(monkeypatch, "NFP_FORCE_GENERIC", "0" if request.param == "auto" else "1")
    nfp_switch(monkeypatch, "NFP_BWD_ATOMIC", "1" if request.param == "atomic" else "0")
    nfp_switch(monkeypatch, "NFP_FWD_SCALAR", "1" if request.param == "atomic" else "0")
    nfp_switch(monkeypatch, "NFP_BWD_BANDS", None)
    return request.param


@pytest.mark.parametrize("name", [c["name"] for c in K.CASES if _supported(c)])
def test_hip_matches_reference_golden(name, dev, variant):
    c = K.BY_NAME[name]
    g = load_golden(name)
    out, gx, _ = run_hip(c, dev)
    assert_matches_golden(out, gx, g, TOL)


@pytest.mark.parametrize("name", [c["name"] for c in K.CASES if _supported(c)])
def test_hip_matches_oracle(name, dev, oracle_lib, variant):
    c = K.BY_NAME[name]
    x = K.make_input(c)
    out, gx, go = run_hip(c, dev)
    ref_out = oracle_lib.forward(x, **c["ctor"])
    ref_gx = oracle_lib.backward(x, go, **c["ctor"])
    assert rel_err(np.nan_to_num(out), np.nan_to_num(ref_out)) <= TOL
    assert rel_err(np.nan_to_num(gx), np.nan_to_num(ref_gx)) <= TOL


@pytest.mark.parametrize("name", ["c1_cos_k3_2x64x14x14", "c2_cos_k3_4x512x7x7", "c5_l2_k5_4x192x14x14",
                                  "geo_cos_stride2", "geo_cos_circular", "geo_l2_zeros", "cos_k5_selfpairs_2x24x5x5"])
def test_channels_last_input_read_in_place(name, dev, variant):
    c = K.BY_NAME[name]
    out0, gx0, _ = run_hip(c, dev)
    out1, gx1, _ = run_hip(c, dev, channels_last=True)
    assert rel_err(out1, out0) <= 2e-6
    assert rel_err(gx1, gx0) <= 2e-6


@pytest.mark.parametrize("name", ["c5_l2_k5_bf16in_4x192x14x14", "c5_cos_k5_2x192x14x14", "c2_cos_k3_4x512x7x7"])
def test_bf16_storage_fp32_accumulate(name, dev, oracle_lib, variant):
    """bf16 load/store, f32 arithmetic: compare with the oracle run on the SAME bf16-rounded inputs."""
    c = K.BY_NAME[name]
    xb = K._bf16_round(K.make_input(c))
    out, gx, go = run_hip(c, dev, x_np=xb, dtype=torch.bfloat16)
    gob = K._bf16_round(go)
    ref_out = oracle_lib.forward(xb, **c["ctor"])
    ref_gx = oracle_lib.backward(xb, gob, **c["ctor"])
    assert rel_err(out, ref_out) <= 1e-2      # one bf16 rounding of the output
    assert rel_err(gx, ref_gx) <= 2e-2        # backward reads the bf16-rounded saved output


def test_unbuilt_measure_fails_loudly(dev):
    """A CUDA tensor is never served by anything but the HIP library."""
    from neighbour_feature_pooling_amd import NFPPooling, _abi
    supported_now = set()
    for meas in _abi.MEASURES:
        if meas == "scs":
            continue
        m = NFPPooling(8, R=1, measure=meas, padding=1)
        n0 = _launches()
        try:
            m(torch.randn(1, 8, 5, 5, device=dev))
            assert _launches() > n0
            supported_now.add(meas)
        except _abi.NfpUnsupported:
            pass
    assert HIP_MEASURES == supported_now, (HIP_MEASURES ^ supported_now)
    # SharpenedCosine: the reference's batch-mixing behaviour, as torch ops on the GPU, with a warning
    n0 = _launches()
    with pytest.warns(RuntimeWarning, match="mixes batch elements"):
        c = K.BY_NAME["m_scs_p2"]
        out = NFPPooling(c["shape"][1], **c["ctor"])(torch.from_numpy(K.make_input(c)).to(dev))
    assert _launches() == n0
    assert rel_err(out.cpu().numpy(), load_golden("m_scs_p2")["out"]) <= TOL
    n0 = _launches()
    with pytest.raises(_abi.NfpUnsupported, match="float32 or bfloat16"):   # refused before anything is launched
        NFPPooling(8, padding=1, measure="cosine")(torch.randint(0, 9, (1, 8, 5, 5), device=dev, dtype=torch.int32))
    assert _launches() == n0
    for p in (float("inf"), 0, -2):           # LA.norm orders with other semantics: refused, never mis-computed
        with pytest.raises(_abi.NfpUnsupported, match="norm order"):
            NFPPooling(8, padding=1, measure="norm", p=p)(torch.randn(1, 8, 5, 5, device=dev))


@pytest.mark.parametrize("p", [0.5, 1, 1.5, 4])
def test_general_norm_orders(p, dev):
    """LA.norm(ord=p) for finite p > 0 (nfp.py:145); the reference's default is p=1."""
    from neighbour_feature_pooling_amd import NFPPooling
    m = NFPPooling(16, R=1, measure="norm", p=p, padding=1)
    x = torch.randn(3, 16, 6, 7, dtype=torch.float64).abs().add_(0.1) * torch.sign(torch.randn(3, 16, 6, 7, dtype=torch.float64))
    x = x.float().double()                     # both sides see the same fp32-representable input
    go = torch.randn(3, 8, 6, 7, dtype=torch.float64).float().double()
    xr = x.clone().requires_grad_(True)
    ref = m(xr)                                # float64 host formulation
    (gref,) = torch.autograd.grad(ref, xr, go)
    xd = x.float().to(dev).requires_grad_(True)
    out = m(xd)
    (gx,) = torch.autograd.grad(out, xd, go.float().to(dev))
    assert rel_err(out.detach().cpu().numpy(), ref.detach().numpy()) <= TOL
    e = rel_err(gx.cpu().numpy(), gref.numpy())
    assert e <= (5e-5 if p < 1 else TOL), e    # |v|^(p-1) amplifies fp32 rounding of v = a - b for p < 1


# ---- properties at the BASELINE.json headline size [64,512,7,7] k=3 cosine -------------------

@pytest.fixture(scope="module")
def headline(dev):
    from neighbour_feature_pooling_amd import NFPPooling
    from neighbour_feature_pooling_amd.synth import feature_map
    x = torch.from_numpy(feature_map((64, 512, 7, 7), 13)).to(dev)
    m = NFPPooling(512, R=1, measure="cosine", padding=1)
    return m, x


def test_headline_uses_hot_path_kernels(headline):
    from neighbour_feature_pooling_amd import _abi
    m, x = headline
    x = x.clone().requires_grad_(True)
    out = m(x)
    assert _abi.load().nfp_last_variant().decode().startswith("fwd_band<R1,cos,f32,nchw>x4")
    out.sum().backward()
    torch.cuda.synchronize()
    assert _abi.load().nfp_last_variant().decode().startswith("bwd_fast")


def test_headline_bounds_and_symmetry(headline):
    m, x = headline
    out = m(x)
    assert out.shape == (64, 8, 7, 7)
    assert out.abs().max().item() <= 1.0 + 1e-6
    # interior symmetry: sim(p, p+d) == sim(p+d, p) i.e. out[n][y,x] == out[7-n][y+dy,x+dx]
    offs = [(-1, -1), (-1, 0), (-1, 1), (0, -1), (0, 1), (1, -1), (1, 0), (1, 1)]
    for n, (dy, dx) in enumerate(offs):
        a = out[:, n, 1:6, 1:6]
        b = out[:, 7 - n, 1 + dy:6 + dy, 1 + dx:6 + dx]
        assert (a - b).abs().max().item() <= 2e-7
    # reflect padding: at y=0 the "up" neighbour IS the "down" neighbour
    assert torch.equal(out[:, 1, 0, :], out[:, 6, 0, :])
    assert torch.equal(out[:, 3, :, 0], out[:, 4, :, 0])


def test_headline_batch_independence_and_determinism(headline):
    m, x = headline
    out = m(x)
    assert torch.equal(m(x), out)                                # bitwise reproducible
    perm = torch.randperm(64, device=x.device)
    assert torch.equal(m(x[perm].contiguous()), out[perm])       # an image's result does not depend on its neighbours
    # a different batch size may split an image into a different number of row bands (nfp_band.h), i.e. sum the
    # channels in a different order: equal to float32 rounding, bitwise equal when the split is the same
    assert (m(x[5:9].contiguous()) - out[5:9]).abs().max().item() <= 2e-6
    assert torch.equal(m(x[:48].contiguous()), out[:48])         # 48 and 64 images: four bands each, the same split


def test_headline_dissimilarity_is_one_minus(headline):
    from neighbour_feature_pooling_amd import NFPPooling
    m, x = headline
    md = NFPPooling(512, R=1, measure="cosine", padding=1, similarity=False)
    assert (md(x) - (1 - m(x))).abs().max().item() <= 2.4e-7  # one rounding of 1 - s


def test_headline_backward_linear_in_grad_out_and_scale_invariant(headline):
    m, x = headline
    x = x.clone().requires_grad_(True)
    g1 = torch.randn(64, 8, 7, 7, device=x.device)
    g2 = torch.randn(64, 8, 7, 7, device=x.device)
    out = m(x)
    a, = torch.autograd.grad(out, x, g1, retain_graph=True)
    b, = torch.autograd.grad(out, x, g2, retain_graph=True)
    ab, = torch.autograd.grad(out, x, 2.0 * g1 - 3.0 * g2)
    assert ((2.0 * a - 3.0 * b) - ab).abs().max().item() <= 1e-5 * ab.abs().max().item()
    # cosine is invariant to per-pixel scaling => <grad_x, x> summed over channels is 0 per pixel
    radial = (ab * x).sum(1)
    assert radial.abs().max().item() <= 1e-4 * (ab.abs() * x.abs()).sum(1).max().item()


def test_headline_matches_fp64_torch_on_device(headline):
    """Independent check at full size: the module vs the torch formulation run in float64 on the GPU."""
    from neighbour_feature_pooling_amd._host import nfp_host
    m, x = headline
    x32 = x.clone().requires_grad_(True)
    x64 = x.double().requires_grad_(True)
    go = torch.randn(64, 8, 7, 7, device=x.device)
    out = m(x32)
    out.backward(go)
    ref = nfp_host(x64, m.config)
    ref.backward(go.double())
    assert rel_err(out.detach().cpu().numpy(), ref.detach().cpu().numpy()) <= TOL
    assert rel_err(x32.grad.cpu().numpy(), x64.grad.cpu().numpy()) <= TOL


def test_l2_k5_properties_at_vit_size(dev):
    """C5 geometry [B,192,14,14] k=5 L2: symmetry d(p,q)=d(q,p), non-positivity, zero on identical maps."""
    from neighbour_feature_pooling_amd import NFPPooling
    from neighbour_feature_pooling_amd.synth import feature_map
    m = NFPPooling(192, R=2, measure="norm", p=2, padding=2)
    x = torch.from_numpy(feature_map((32, 192, 14, 14), 77)).to(dev)
    out = m(x)
    assert out.shape == (32, 24, 14, 14) and out.max().item() <= 0.0
    k = 5
    taps = [t for t in range(k * k) if t != 12]
    for n, t in enumerate(taps):
        dy, dx = t // k - 2, t % k - 2
        a = out[:, n, 2:12, 2:12]
        b = out[:, 23 - n, 2 + dy:12 + dy, 2 + dx:12 + dx]
        assert (a - b).abs().max().item() <= 1e-5
    const = torch.ones(2, 192, 14, 14, device=dev) * torch.randn(2, 192, 1, 1, device=dev)
    const.requires_grad_(True)
    oc = m(const)
    assert oc.abs().max().item() == 0.0
    oc.sum().backward()
    assert const.grad.abs().max().item() == 0.0  # torch's subgradient at d == 0


def test_no_grad_and_requires_grad_false(dev):
    from neighbour_feature_pooling_amd import NFPPooling
    m = NFPPooling(16, padding=1, measure="cosine")
    x = torch.randn(2, 16, 7, 7, device=dev)
    with torch.no_grad():
        a = m(x)
    b = m(x)
    assert not b.requires_grad and torch.equal(a, b)


# ---- end-to-end configs (BASELINE.json configs 3 and 5 geometry) -------------------

def test_resnet18_nfp_train_step_eurosat_shape(dev):
    """Config 3 (BASELINE.json configs[2]) at its configured batch: ResNet18+NFP(cosine) on [256,13,64,64]; NFP sees
    [256,512,2,2] (reflect padding on a 2x2 map)."""
    from neighbour_feature_pooling_amd import _abi
    from neighbour_feature_pooling_amd.train import build, make_step, synthetic_batch
    torch.manual_seed(0)
    net = build("resnet18", num_classes=10, in_chans=13, image=64, device=dev)
    step, _ = make_step(net)
    x, y = synthetic_batch(256, 13, 64, 10, dev, torch.float32, 3)
    n0 = _abi.load().nfp_launch_count()
    l0 = step(x, y)
    for _ in range(4):
        l1 = step(x, y)
    torch.cuda.synchronize()
    assert _abi.load().nfp_launch_count() >= n0 + 10       # 5 x (fwd + bwd) through the HIP library
    assert torch.isfinite(l1) and l1.item() < l0.item()


def test_nfp_inside_network_matches_torch_formulation(dev):
    """Gradients that flow back through the HIP NFP into a backbone equal those of the torch formulation."""
    from neighbour_feature_pooling_amd import functional
    from neighbour_feature_pooling_amd._host import nfp_host
    from neighbour_feature_pooling_amd.train import build, synthetic_batch
    torch.manual_seed(1)
    net = build("resnet18", num_classes=6, in_chans=3, image=96, device=dev).eval()
    x, y = synthetic_batch(8, 3, 96, 6, dev, torch.float32, 5)
    crit = torch.nn.CrossEntropyLoss(label_smoothing=0.05)
    feat_grads = []
    hook = net.pool.register_forward_pre_hook(
        lambda mod, args: args[0].register_hook(lambda g: feat_grads.append(g.clone())) and None)
    import neighbour_feature_pooling_amd.pooling as pooling
    orig = pooling.nfp_pool
    try:
        n0 = _launches()
        crit(net(x), y).backward()
        assert _launches() >= n0 + 2
        g_hip = {k: p.grad.clone() for k, p in net.named_parameters()}
        net.zero_grad()
        # second pass: both pooled reductions from torch ops on the GPU instead of the HIP kernels
        pooling.nfp_pool = lambda t, cfg: (t.mean((2, 3)), nfp_host(t, cfg).mean((2, 3)))
        n0 = _launches()
        crit(net(x), y).backward()
        assert _launches() == n0
    finally:
        pooling.nfp_pool = orig
        hook.remove()
    # what the NFP backward hands to the backbone: the same forward features both times, so only the two NFP
    # implementations (and the head) differ
    assert len(feat_grads) == 2
    assert (feat_grads[0] - feat_grads[1]).abs().max().item() <= 1e-4 * feat_grads[1].abs().max().item()
    for k, p in net.named_parameters():
        ref = p.grad
        if k.startswith(("pool.", "fc.")):
            assert (g_hip[k] - ref).abs().max().item() <= 1e-4 * max(ref.abs().max().item(), 1e-6), k
        else:
            # through MIOpen's conv backward kernels, whose choice (Winograd / implicit GEMM, atomic split-K)
            # and rounding are not the same from one pass to the next: a sanity bound only
            assert (g_hip[k] - ref).norm().item() <= 2e-2 * max(ref.norm().item(), 1e-6), k


def test_vit_tiny_nfp_bf16_k5_l2_step(dev):
    """Config 5 geometry: ViT-Tiny tokens -> 14x14x192 grid, NFP k=5 L2, bf16 storage."""
    from neighbour_feature_pooling_amd import NFPPooling, _abi
    from neighbour_feature_pooling_amd.train import build, make_step, synthetic_batch
    torch.manual_seed(0)
    layer = NFPPooling(192, R=2, measure="norm", p=2, padding=2)
    net = build("vit_tiny_patch16_224", num_classes=10, in_chans=3, image=224, nfp=layer, device=dev,
                dtype=torch.bfloat16)
    step, _ = make_step(net)
    x, y = synthetic_batch(8, 3, 224, 10, dev, torch.bfloat16, 9)
    losses = [step(x, y).item() for _ in range(3)]
    torch.cuda.synchronize()
    assert all(np.isfinite(losses))
    # the ViT head hands NFP a channels-last VIEW of the token matrix (batch stride (1+196)*192): read in place
    bv = _abi.load().nfp_last_variant().decode()
    assert bv.startswith("bwd_fast<R2,l2,bf16,nhwc"), bv
    tok = torch.randn(8, 197, 192, device=dev, dtype=torch.bfloat16)
    view = tok[:, 1:].transpose(1, 2).unflatten(2, (14, 14))
    assert not view.is_contiguous() and not view.is_contiguous(memory_format=torch.channels_last)
    out = layer(view)
    fv = _abi.load().nfp_last_variant().decode()
    assert fv.startswith("fwd_gram<R2,l2,bf16,nhwc"), fv
    assert torch.equal(out, layer(view.contiguous(memory_format=torch.channels_last)))


# ---- fused nfp_pooling tail (models/NFP_Pooling.py:27-31) ----------------------------

@pytest.mark.parametrize("layout,dtype", [("nchw", torch.float32), ("nhwc", torch.float32), ("nhwc", torch.bfloat16),
                                          ("nchw", torch.bfloat16)])
@pytest.mark.parametrize("shape,ctor", [
    ((64, 512, 7, 7), dict(R=1, measure="cosine", padding=1)),
    ((8, 512, 2, 2), dict(R=1, measure="cosine", padding=1)),
    ((4, 192, 14, 14), dict(R=2, measure="norm", p=2, padding=2)),
    ((3, 24, 5, 6), dict(R=1, measure="cosine", padding=1, padding_mode="zeros", similarity=False)),
    ((2, 960, 7, 7), dict(R=1, measure="norm", p=2, padding=1)),
    ((300, 512, 7, 7), dict(R=1, measure="cosine", padding=1)),       # one workgroup per image: more channels than threads
    ((600, 960, 7, 7), dict(R=1, measure="cosine", padding=1)),       # ... than twice the threads (staged grad(GAP) loop)
    ((1100, 64, 7, 7), dict(R=1, measure="cosine", padding=1)),       # four workgroups per CU: quarter-size workgroups, both passes
    ((1040, 64, 14, 14), dict(R=1, measure="norm", p=2, padding=1)),
])
def test_fused_pool_matches_composition(shape, ctor, layout, dtype, dev):
    """nfp_pool (one pass: GAP(x) and GAP(NFP(x)), NFP_Pooling.py:27-31) against the same two reductions composed from
    `nfp` and torch means, for both layouts and storage types; bf16 at the bf16 bound (the composition rounds the
    [B,N,H,W] map to bf16 before averaging it, the fused pass does not)."""
    from neighbour_feature_pooling_amd import NFPPooling, _abi
    from neighbour_feature_pooling_amd.functional import nfp, nfp_pool
    from neighbour_feature_pooling_amd.synth import feature_map
    m = NFPPooling(shape[1], **ctor)
    bf = dtype == torch.bfloat16
    x0 = torch.from_numpy(feature_map(shape, 91)).to(dev).to(dtype)
    if layout == "nhwc":
        x0 = x0.contiguous(memory_format=torch.channels_last)
    x1 = x0.clone(memory_format=torch.preserve_format).requires_grad_(True)
    x2 = x0.detach().float().contiguous().requires_grad_(True)     # composition in float32 on the same (rounded) input
    gap, nfpm = nfp_pool(x1, m.config)
    fv = _abi.load().nfp_last_variant().decode()
    assert fv.startswith(("fwd_band<", "fwd_gram<")) and ",pool>" in fv and (",nhwc" in fv) == (layout == "nhwc"), fv
    assert fv.startswith("fwd_gram<") == (bf and shape[1] % 16 == 0 and shape[1] >= 48), fv   # bf16: the matrix cores
    ref_gap, ref_nfpm = x2.mean((2, 3)), nfp(x2, m.config).mean((2, 3))
    tol_f, tol_b = (1e-2, 2e-2) if bf else (2e-6, 1e-5)
    assert rel_err(gap.detach().float().cpu().numpy(), ref_gap.detach().cpu().numpy()) <= tol_f
    assert rel_err(nfpm.detach().float().cpu().numpy(), ref_nfpm.detach().cpu().numpy()) <= tol_f
    gg = torch.from_numpy(feature_map(tuple(gap.shape), 92)).to(dev)
    gn = torch.from_numpy(feature_map(tuple(nfpm.shape), 93)).to(dev)
    ((gap.float() * gg).sum() + (nfpm.float() * gn).sum()).backward()
    torch.cuda.synchronize()
    bv = _abi.load().nfp_last_variant().decode()
    assert bv.startswith("bwd_fast<") and bv.endswith(",pool>"), bv
    ((ref_gap * gg).sum() + (ref_nfpm * gn).sum()).backward()
    torch.cuda.synchronize()
    assert x1.grad.is_contiguous(memory_format=torch.channels_last if layout == "nhwc" else torch.contiguous_format)
    assert rel_err(x1.grad.float().cpu().numpy(), x2.grad.cpu().numpy()) <= tol_b


def test_nfp_pooling_wrapper_on_gpu_matches_reference_golden(dev):
    """The wrapper (now on the fused kernels) against the reference's own nfp_pooling outputs."""
    from neighbour_feature_pooling_amd import _abi, nfp_pooling
    from neighbour_feature_pooling_amd.synth import feature_map
    g = load_golden("wrapper_nfp_pooling_4x64x7x7")
    B, C = 4, 64
    params = {"num_ftrs": {"m": C}, "Model_name": "m", "Dataset": "d", "num_classes": {"d": 10}, "input_size": 7}
    w = nfp_pooling(Params=params).to(dev)
    with torch.no_grad():
        w.nfp_proj.weight.copy_(torch.from_numpy(feature_map((C, 8), 201) * 0.3))
        w.nfp_proj.bias.copy_(torch.from_numpy(feature_map((C,), 202) * 0.1))
    x = torch.from_numpy(feature_map((B, C, 7, 7), 200)).to(dev).requires_grad_(True)
    y = w(x)
    assert ",pool>" in _abi.load().nfp_last_variant().decode()
    y.backward(torch.from_numpy(feature_map((B, C), 203)).to(dev))
    assert rel_err(y.detach().cpu().numpy(), g["y"]) <= 1e-5
    assert rel_err(x.grad.cpu().numpy(), g["gx"]) <= 1e-5
    assert rel_err(w.nfp_proj.weight.grad.cpu().numpy(), g["gw"]) <= 1e-5
    # the same wrapper on a channels-last bf16 feature map (what a bf16 ViT / channels-last ResNet hands over): fused
    # too, within the bf16 bound of the reference's float32 golden
    wb = nfp_pooling(Params=params).to(dev).to(torch.bfloat16)
    with torch.no_grad():
        wb.nfp_proj.weight.copy_(w.nfp_proj.weight)
        wb.nfp_proj.bias.copy_(w.nfp_proj.bias)
    xb = x.detach().to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    yb = wb(xb)
    fv = _abi.load().nfp_last_variant().decode()
    assert fv.startswith("fwd_gram<R1,cos,bf16,nhwc,pool>"), fv
    yb.backward(torch.from_numpy(feature_map((B, C), 203)).to(dev).to(torch.bfloat16))
    assert rel_err(yb.detach().float().cpu().numpy(), g["y"]) <= 3e-2
    assert rel_err(xb.grad.float().cpu().numpy(), g["gx"]) <= 3e-2


# ---- the C ABI called directly: raw device pointers + descriptor, no nn.Module / autograd in between ----

@pytest.mark.parametrize("name", ["c1_cos_k3_2x64x14x14", "c2_cos_k3_4x512x7x7", "c5_l2_k5_4x192x14x14",
                                  "geo_cos_stride2_dil2_pad0", "m_canberra", "m_pearson"])
def test_c_abi_direct_calls_match_reference_golden(name, dev):
    import ctypes
    from neighbour_feature_pooling_amd import _abi
    L = _abi.load()
    c = K.BY_NAME[name]
    g = load_golden(name)
    ctor = dict(R=1, measure="norm", p=1, stride=1, padding=0, dilation=1, padding_mode="reflect", similarity=True,
                eps=1e-6, q_scs=1e-6)
    ctor.update(c["ctor"])
    x = torch.from_numpy(K.make_input(c)).to(dev)
    d = _abi.NfpDesc()
    d.B, d.C, d.H, d.W = x.shape
    d.R, d.pad, d.stride, d.dilation = ctor["R"], ctor["padding"], ctor["stride"], ctor["dilation"]
    d.pad_mode = _abi.PAD_MODES.index(ctor["padding_mode"])
    d.measure = _abi.measure_id(ctor["measure"].lower())
    d.similarity = int(ctor["similarity"])
    d.diff_weights = int(ctor["measure"] in ("norm", "rmse", "mahalanobis"))
    d.dtype = _abi.F32
    d.p, d.eps, d.q_scs = ctor["p"], ctor["eps"], ctor["q_scs"]
    d.sxB, d.sxC, d.sxH, d.sxW = x.stride()
    n, ho, wo = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    assert L.nfp_output_shape(ctypes.byref(d), ctypes.byref(n), ctypes.byref(ho), ctypes.byref(wo)) == 0
    assert (d.B, n.value, ho.value, wo.value) == golden_out_shape(g)
    out = torch.empty(golden_out_shape(g), device=dev)
    saved = torch.empty(max(int(L.nfp_saved_floats(ctypes.byref(d))), 1), device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    assert L.nfp_forward(ctypes.byref(d), x.data_ptr(), out.data_ptr(), saved.data_ptr(), stream) == 0, L.nfp_last_error()
    go = torch.from_numpy(K.make_grad_out(c, golden_out_shape(g))).to(dev)
    gx = torch.empty_like(x)
    assert L.nfp_backward(ctypes.byref(d), x.data_ptr(), go.data_ptr(), out.data_ptr(), saved.data_ptr(), gx.data_ptr(),
                          stream) == 0, L.nfp_last_error()
    torch.cuda.synchronize()
    assert rel_err(out.cpu().numpy(), g["out"]) <= TOL
    if "gx" in g:
        assert rel_err(gx.cpu().numpy(), g["gx"]) <= TOL
    else:
        assert rel_err(gx.cpu().numpy().reshape(-1)[K.gx_sample_index(gx.numel())], g["gx_sample"]) <= TOL


# ---- shape sweep: hot-path staging / chunking corner cases against the float64 torch formulation ----

def _sweep_cases():
    import itertools, random
    rnd = random.Random(1234)
    cases = []
    # every P % 4, tiny and large channel counts, chunked slabs (C*P*4 beyond one LDS slab), B above/below CU count
    for (H, W) in [(2, 2), (2, 3), (3, 3), (3, 5), (5, 5), (7, 7), (6, 9), (14, 14), (11, 13), (16, 16), (22, 23), (5, 100)]:
        for R in (1, 2):
            if R >= H or R >= W:
                continue
            C = rnd.choice([4, 8, 36, 100, 192, 512])
            B = rnd.choice([1, 3, 17, 70])
            meas = rnd.choice(["cosine", "norm"])
            mode = rnd.choice(["reflect", "zeros", "replicate"])
            cases.append((B, C, H, W, R, meas, mode))
    cases += [(2, 2048, 7, 7, 1, "cosine", "reflect"),      # ResNet50_NFPPooling width (texture_pooling.py:547-561)
              (300, 64, 7, 7, 1, "cosine", "reflect"),       # more images than CUs: two workgroups per CU
              (1100, 32, 7, 7, 1, "cosine", "reflect"),      # from four per CU on: quarter-size forward workgroups
              (1030, 16, 14, 14, 2, "norm", "zeros"),        # ... 196 pixels in 256 threads
              (1030, 8, 20, 20, 1, "cosine", "replicate"),   # ... 400 pixels: stays at two per CU
              (5, 1000, 3, 3, 1, "norm", "zeros"),
              (1, 4, 4, 1, 1, "cosine", "replicate") if False else (1, 4, 4, 4, 1, "cosine", "replicate"),
              (3, 516, 9, 9, 2, "cosine", "reflect")]
    return cases


@pytest.mark.parametrize("B,C,H,W,R,meas,mode", _sweep_cases())
def test_shape_sweep_against_float64_formulation(B, C, H, W, R, meas, mode, dev):
    from neighbour_feature_pooling_amd import NFPPooling
    from neighbour_feature_pooling_amd._host import nfp_host
    from neighbour_feature_pooling_amd.synth import feature_map
    ctor = dict(R=R, measure=meas, padding=R, padding_mode=mode)
    if meas == "norm":
        ctor["p"] = 2
    m = NFPPooling(C, **ctor)
    x = torch.from_numpy(feature_map((B, C, H, W), 7 * H + W + C)).to(dev).requires_grad_(True)
    out = m(x)
    go = torch.from_numpy(feature_map(tuple(out.shape), 3 * H + W)).to(dev)
    gx, = torch.autograd.grad(out, x, go)
    x64 = x.detach().double().requires_grad_(True)
    ref = nfp_host(x64, m.config)
    gref, = torch.autograd.grad(ref, x64, go.double())
    assert rel_err(out.detach().cpu().numpy(), ref.detach().cpu().numpy()) <= TOL
    assert rel_err(gx.cpu().numpy(), gref.cpu().numpy()) <= TOL
    # channels-last view of the same tensor goes through the NHWC staging path
    xl = x.detach().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    outl = m(xl)
    gxl, = torch.autograd.grad(outl, xl, go)
    assert rel_err(outl.detach().cpu().numpy(), ref.detach().cpu().numpy()) <= TOL
    assert rel_err(gxl.cpu().numpy(), gref.cpu().numpy()) <= TOL


def _gather_cases():
    import random
    rnd = random.Random(4321)
    cases = []
    meas_pool = ["dot", "norm", "emd", "cosine", "gfc", "pearson", "smith", "canberra", "geman", "chisquared2"]
    for mode in ("reflect", "zeros", "replicate", "circular"):
        for (H, W, R, pad, stride, dil) in [(7, 7, 1, 1, 1, 1), (6, 9, 1, 3, 1, 1), (9, 8, 2, 1, 2, 1),
                                            (11, 10, 1, 2, 3, 2), (5, 12, 2, 4, 1, 2), (1, 9, 1, 0, 1, 1),
                                            (8, 1, 1, 0, 1, 1), (4, 4, 1, 3, 2, 1), (13, 6, 3, 2, 1, 1)]:
            k = 2 * R + 1
            if H + 2 * pad < dil * (k - 1) + 1 or W + 2 * pad < dil * (k - 1) + 1:
                continue
            if mode == "reflect" and (pad >= H or pad >= W):
                continue
            if mode == "circular" and (pad > H or pad > W):
                continue
            cases.append((rnd.choice([1, 3, 5]), rnd.choice([3, 5, 8, 18, 67]), H, W, R, pad, stride, dil, mode,
                          rnd.choice(meas_pool)))
    return cases


@pytest.mark.parametrize("B,C,H,W,R,pad,stride,dil,mode,meas", _gather_cases())
def test_gather_backward_geometry_sweep(B, C, H, W, R, pad, stride, dil, mode, meas, dev, monkeypatch):
    """The gather-form backward (nfp_gather.h) inverts pad/stride/dilation analytically: every padding mode,
    pixels that are the centre of none or of several outputs, 1-pixel-wide maps, channel counts that are not
    a multiple of 4 — against the float64 formulation, bitwise reproducible, and equal to the atomic fallback."""
    from neighbour_feature_pooling_amd import NFPPooling, _abi
    from neighbour_feature_pooling_amd._host import nfp_host
    nfp_switch(monkeypatch, "NFP_FORCE_GENERIC", "1")
    nfp_switch(monkeypatch, "NFP_BWD_ATOMIC", "0")
    nfp_switch(monkeypatch, "NFP_FWD_SCALAR", "0")
    ctor = dict(R=R, measure=meas, padding=pad, stride=stride, dilation=dil, padding_mode=mode)
    if meas == "norm":
        ctor["p"] = 1
    m = NFPPooling(C, **ctor)
    g = torch.Generator().manual_seed(H * 131 + W * 17 + C)
    x = (torch.rand(B, C, H, W, generator=g) + 0.25).to(dev).requires_grad_(True)   # positive: valid for every measure
    out = m(x)
    assert _abi.load().nfp_last_variant().decode() == "fwd_pairs"
    assert torch.equal(out, m(x))
    go = torch.randn(out.shape, generator=g).to(dev)
    gx, = torch.autograd.grad(out, x, go, retain_graph=True)
    assert _abi.load().nfp_last_variant().decode() == "bwd_gather"
    gx2, = torch.autograd.grad(out, x, go, retain_graph=True)
    assert torch.equal(gx, gx2)
    x64 = x.detach().double().requires_grad_(True)
    ref = nfp_host(x64, m.config)
    gref, = torch.autograd.grad(ref, x64, go.double())
    assert rel_err(out.detach().cpu().numpy(), ref.detach().cpu().numpy()) <= TOL
    assert rel_err(gx.cpu().numpy(), gref.cpu().numpy()) <= 2 * TOL
    # the banded kernel (maps too large for whole-image tables), forced here onto 3 bands of rows
    nfp_switch(monkeypatch, "NFP_BWD_BANDS", "3")
    gxb, = torch.autograd.grad(out, x, go, retain_graph=True)
    assert _abi.load().nfp_last_variant().decode() in ("bwd_gather_banded", "bwd_direct")
    assert rel_err(gxb.cpu().numpy(), gref.cpu().numpy()) <= 2 * TOL
    nfp_switch(monkeypatch, "NFP_BWD_BANDS", None)
    nfp_switch(monkeypatch, "NFP_BWD_ATOMIC", "1")
    nfp_switch(monkeypatch, "NFP_FWD_SCALAR", "1")
    gx3, = torch.autograd.grad(out, x, go)
    assert _abi.load().nfp_last_variant().decode() == "bwd_direct"
    assert rel_err(gx3.cpu().numpy(), gref.cpu().numpy()) <= 2 * TOL
    out3 = m(x)
    assert _abi.load().nfp_last_variant().decode() == "fwd_direct"
    assert rel_err(out3.detach().cpu().numpy(), ref.detach().cpu().numpy()) <= TOL


@pytest.mark.parametrize("B,C,H,W,ctor", [
    (2, 12, 60, 52, dict(R=1, measure="cosine", padding=1)),
    (1, 7, 75, 41, dict(R=2, measure="norm", p=1, padding=2, padding_mode="replicate")),
    (2, 8, 64, 64, dict(R=1, measure="canberra", padding=1, stride=2, padding_mode="zeros")),
    (1, 6, 90, 33, dict(R=1, measure="pearson", padding=2, dilation=2)),
    (3, 64, 56, 56, dict(R=1, measure="cosine", padding=1)),
])
def test_large_maps_use_the_banded_backward(B, C, H, W, ctor, dev, monkeypatch):
    """Maps whose whole-image pair tables exceed LDS: row-windowed forward tiles, row-banded gather backward.  (Cosine /
    L2 "same" maps of this size are served by the row-band kernels of nfp_tile.h since round 3 — tests/test_gpu_tile.py;
    the any-geometry kernels are forced here so that they stay covered on those shapes too.)"""
    nfp_switch(monkeypatch, "NFP_FORCE_GENERIC", "1")
    from neighbour_feature_pooling_amd import NFPPooling, _abi
    from neighbour_feature_pooling_amd._host import nfp_host
    m = NFPPooling(C, **ctor)
    g = torch.Generator().manual_seed(H + W)
    x = (torch.rand(B, C, H, W, generator=g) + 0.25).to(dev).requires_grad_(True)
    out = m(x)
    assert _abi.load().nfp_last_variant().decode() == "fwd_pairs"
    go = torch.randn(out.shape, generator=g).to(dev)
    gx, = torch.autograd.grad(out, x, go, retain_graph=True)
    assert _abi.load().nfp_last_variant().decode() == "bwd_gather_banded"
    gx2, = torch.autograd.grad(out, x, go)
    assert torch.equal(gx, gx2)
    x64 = x.detach().double().requires_grad_(True)
    ref = nfp_host(x64, m.config)
    gref, = torch.autograd.grad(ref, x64, go.double())
    assert rel_err(out.detach().cpu().numpy(), ref.detach().cpu().numpy()) <= TOL
    assert rel_err(gx.cpu().numpy(), gref.cpu().numpy()) <= 2 * TOL


def test_random_geometry_stress(dev, monkeypatch):
    """A few hundred random (geometry, measure, layout) draws through the default dispatch and through the forced
    row-banded backward, each against the float64 formulation: the index tables of nfp_gather.h are built
    analytically, this is the net under that arithmetic."""
    import random
    from neighbour_feature_pooling_amd import NFPPooling, _abi
    from neighbour_feature_pooling_amd._host import nfp_host
    rnd = random.Random(77)
    measures = sorted(HIP_MEASURES)
    done, variants = 0, set()
    for it in range(400):
        H, W = rnd.randint(1, 22), rnd.randint(1, 22)
        R, stride, dil = rnd.choice([1, 1, 2, 3]), rnd.choice([1, 1, 2, 3]), rnd.choice([1, 1, 2])
        pad = rnd.choice([0, R, R * dil, R * dil + 1, 1])
        mode = rnd.choice(["reflect", "zeros", "replicate", "circular"])
        k = 2 * R + 1
        if H + 2 * pad < dil * (k - 1) + 1 or W + 2 * pad < dil * (k - 1) + 1:
            continue
        if (mode == "reflect" and (pad >= H or pad >= W)) or (mode == "circular" and (pad > H or pad > W)):
            continue
        meas = rnd.choice(measures)
        B, C = rnd.randint(1, 3), rnd.choice([4, 5, 6, 8, 13, 32])   # (1-3 channels: several measures degenerate to ~eps)
        ctor = dict(R=R, measure=meas, padding=pad, stride=stride, dilation=dil, padding_mode=mode,
                    similarity=rnd.random() < 0.7)
        if meas == "norm":
            ctor["p"] = rnd.choice([1, 2, 3])
        m = NFPPooling(C, **ctor)
        g = torch.Generator().manual_seed(it)
        x = (torch.rand(B, C, H, W, generator=g) + 0.25).to(dev)
        if rnd.random() < 0.3:
            x = x.contiguous(memory_format=torch.channels_last)
        x.requires_grad_(True)
        nfp_switch(monkeypatch, "NFP_BWD_BANDS", None)
        out = m(x)
        go = torch.randn(out.shape, generator=g).to(dev)
        gx, = torch.autograd.grad(out, x, go, retain_graph=True)
        variants.add(_abi.load().nfp_last_variant().decode().split("<")[0])
        nfp_switch(monkeypatch, "NFP_BWD_BANDS", "2")
        gxb, = torch.autograd.grad(out, x, go)
        variants.add(_abi.load().nfp_last_variant().decode().split("<")[0])
        x64 = x.detach().double().requires_grad_(True)
        ref = nfp_host(x64, m.config)
        gref, = torch.autograd.grad(ref, x64, go.double())
        what = (it, tuple(x.shape), ctor)
        # an index error is an O(1) error; float32-vs-float64 conditioning of a few measures on few channels is not
        tol_g = 2e-4 if meas in ("pearson", "hellinger", "squaredchord", "jeffrey", "smith") else 5e-5
        o_, r_ = out.detach().cpu().numpy(), ref.detach().cpu().numpy()
        assert same_nan_pattern(o_, r_), what
        assert rel_err(np.nan_to_num(o_), np.nan_to_num(r_)) <= 1e-4, what
        gr_ = gref.cpu().numpy()
        for g_ in (gx.cpu().numpy(), gxb.cpu().numpy()):
            # a neighbour tap that folds onto its own centre has distance exactly 0: sqrt'(0) * 0 is NaN in torch
            # and here, at the same elements
            assert same_nan_pattern(g_, gr_), what
            assert rel_err(np.nan_to_num(g_), np.nan_to_num(gr_)) <= tol_g, what
        done += 1
    assert done > 250
    assert {"bwd_gather", "bwd_gather_banded", "bwd_fast"} <= variants, variants


@pytest.mark.parametrize("B,C,H,W,R,meas,mode,kind", [
    (4, 192, 14, 14, 2, "norm", "reflect", "randn"),        # config 5 geometry
    (4, 192, 14, 14, 2, "cosine", "reflect", "randn"),
    (3, 512, 7, 7, 1, "cosine", "reflect", "relu"),         # headline geometry, bf16 channels-last
    (3, 32, 5, 9, 1, "norm", "replicate", "randn"),
    (2, 48, 6, 5, 2, "cosine", "zeros", "relu"),
    (2, 16, 9, 9, 1, "norm", "zeros", "const"),             # identical neighbours: exactly zero distance
    (2, 192, 14, 14, 2, "norm", "reflect", "smooth"),       # nearly identical neighbours: the Gram form's hard case
    (1, 512, 16, 16, 2, "norm", "reflect", "randn"),        # image + tiles beyond LDS: fragments from global memory
    (2, 64, 1, 40, 1, "cosine", "replicate", "randn"),
    (5, 192, 14, 14, 2, "norm", "reflect", "nchw"),          # NCHW bf16: transposed into the same LDS image
    (5, 512, 7, 7, 1, "cosine", "reflect", "nchw"),
    (5, 48, 6, 5, 2, "cosine", "zeros", "nchw"),
    (260, 512, 7, 7, 1, "cosine", "reflect", "randn"),      # one workgroup per image: more Xt blocks than wavefronts
    (260, 256, 8, 8, 1, "norm", "reflect", "nchw"),         # ... and more NCHW pieces than four per thread
])
def test_matrix_core_forward_bf16(B, C, H, W, R, meas, mode, kind, dev, monkeypatch):
    """fwd_gram (nfp_mfma.h): banded Gram matrix on v_mfma_f32_32x32x16_bf16.  Same bf16 inputs, f32 accumulation:
    it must agree with the vector kernel to f32 rounding BEFORE the bf16 output rounding — i.e. the two bf16
    outputs may differ by at most one bf16 ulp, rarely — and with the float64 formulation to bf16 precision."""
    from neighbour_feature_pooling_amd import NFPPooling, _abi
    from neighbour_feature_pooling_amd._host import nfp_host
    nfp_switch(monkeypatch, "NFP_FORCE_GENERIC", "0")
    ctor = dict(R=R, measure=meas, padding=R, padding_mode=mode)
    if meas == "norm":
        ctor["p"] = 2
    m = NFPPooling(C, **ctor)
    g = torch.Generator().manual_seed(C + H)
    x = torch.randn(B, C, H, W, generator=g)
    if kind == "relu":
        x = x.relu()
    if kind == "const":
        x = torch.ones_like(x) * 0.37 + torch.arange(C).view(1, C, 1, 1) * 0.01
    if kind == "smooth":
        x = x.mean((2, 3), keepdim=True) + 0.01 * x
    x = x.to(dev).bfloat16()
    if kind != "nchw":
        x = x.contiguous(memory_format=torch.channels_last)
    x.requires_grad_(True)
    nfp_switch(monkeypatch, "NFP_MFMA", "1")
    n0 = _launches()
    out = m(x)
    assert _abi.load().nfp_last_variant().decode().startswith("fwd_gram<"), _abi.load().nfp_last_variant()
    assert _launches() == n0 + 1
    assert torch.equal(out, m(x))                                   # deterministic
    nfp_switch(monkeypatch, "NFP_MFMA", "0")
    out_v = m(x)
    assert _abi.load().nfp_last_variant().decode().startswith("fwd_band<")
    ref = nfp_host(x.detach().double(), m.config)
    sc = ref.abs().max().item()
    d = (out.float() - out_v.float()).abs()
    assert d.max().item() <= 2 ** -7 * sc                            # at most one bf16 ulp of the largest value apart
    assert (d > 0).float().mean().item() <= 0.02                     # and only where a rounding boundary is crossed
    assert rel_err(out.detach().float().cpu().numpy(), ref.float().cpu().numpy()) <= 1e-2
    if kind == "const":
        assert out[:, :, 1:-1, 1:-1].abs().max().item() == 0.0 if R == 1 else True
    # the backward: phase B on the matrix cores too (when C % 32 == 0), against the vector kernel and float64
    nfp_switch(monkeypatch, "NFP_MFMA", "1")
    go = torch.randn(out.shape, generator=g).to(dev).bfloat16()
    o1 = m(x)
    gx, = torch.autograd.grad(o1, x, go, retain_graph=True)
    bv = _abi.load().nfp_last_variant().decode()
    if C % 32 == 0 and bv.startswith(("bwd_fast", "bwd_gemm2")) and not (kind == "nchw" and (H * W) % 4 and B > 128):
        assert bv.endswith((",mfma>", ",mfma2>")), bv          # (maps whose phase-A tables exceed LDS go to the general kernels)
        gx2, = torch.autograd.grad(o1, x, go, retain_graph=True)
        assert torch.equal(gx, gx2)                                  # deterministic
        nfp_switch(monkeypatch, "NFP_MFMA", "0")
        gv, = torch.autograd.grad(o1, x, go, retain_graph=True)
        assert not _abi.load().nfp_last_variant().decode().endswith((",mfma>", ",mfma2>"))
        nfp_switch(monkeypatch, "NFP_MFMA", "1")
        # the two forms of the matrix-core phase B (nfp_fast.h: bwd_gemm_phase, bwd_gemm_phase3) hold the same weights in the
        # same hi / lo split: they differ by the order of the float32 sums only
        other = "0" if bv.endswith(",mfma2>") else "2"   # (2: the second form wherever its LDS fits, not only where it is faster)
        nfp_switch(monkeypatch, "NFP_GEMM3", other)
        g3, = torch.autograd.grad(o1, x, go, retain_graph=True)
        bv3 = _abi.load().nfp_last_variant().decode()
        nfp_switch(monkeypatch, "NFP_GEMM3", None)
        assert bv3.endswith(",mfma>" if other == "0" else (",mfma>", ",mfma2>")), bv3    # (the second form needs every row tile's weights in LDS)
        if kind not in ("const", "smooth"):
            assert (gx.float() - g3.float()).abs().max().item() <= 2 ** -7 * gx.float().abs().max().item()
        if kind not in ("const", "smooth"):
            assert (gx.float() - gv.float()).abs().max().item() <= 2 ** -6 * gv.float().abs().max().item()
    x64 = x.detach().double().requires_grad_(True)
    gref, = torch.autograd.grad(nfp_host(x64, m.config), x64, go.double())
    if kind not in ("const", "smooth"):                               # (those have |out| ~ 0: sqrt'(0))
        assert rel_err(gx.float().cpu().numpy(), gref.float().cpu().numpy()) <= 2e-2


# ---- the BASELINE.json workloads themselves (configs[3], configs[4]) against the oracle ------------------------

def _oracle_pair(oracle_lib, x_np, go_np, ctor):
    return oracle_lib.forward(x_np, **ctor), oracle_lib.backward(x_np, go_np, **ctor)


@pytest.mark.parametrize("kind", ["randn", "relu"])
def test_config4_nfp_shape_f32_against_oracle(kind, dev, oracle_lib):
    """configs[3]: ResNet18+NFP at 224x224, bs 256 per GPU => NFP sees [256,512,7,7] f32 (NCHW), cosine k=3,
    through the default dispatch; out and grad_x against the CPU oracle, <= 1e-5 of the tensor's max magnitude."""
    from neighbour_feature_pooling_amd import NFPPooling, _abi
    from neighbour_feature_pooling_amd.synth import feature_map
    ctor = dict(R=1, measure="cosine", padding=1)
    xh = feature_map((256, 512, 7, 7), 401)
    if kind == "relu":
        xh = np.maximum(xh, 0).astype(np.float32)
    goh = feature_map((256, 8, 7, 7), 402)
    x = torch.from_numpy(xh).to(dev).requires_grad_(True)
    n0 = _launches()
    out = NFPPooling(512, **ctor)(x)
    assert _abi.load().nfp_last_variant().decode().startswith("fwd_")
    fv = _abi.load().nfp_last_variant().decode()
    out.backward(torch.from_numpy(goh).to(dev))
    torch.cuda.synchronize()
    bv = _abi.load().nfp_last_variant().decode()
    assert _launches() == n0 + 2 and fv.startswith("fwd_band<R1,cos,f32,nchw>x1") and bv.startswith("bwd_fast<R1,cos,f32,nchw")
    ref_out, ref_gx = _oracle_pair(oracle_lib, xh, goh, ctor)
    assert rel_err(out.detach().cpu().numpy(), ref_out) <= TOL
    assert rel_err(x.grad.cpu().numpy(), ref_gx) <= TOL


@pytest.mark.parametrize("kind,scale", [("randn", None), ("smooth", 0.01), ("smooth", 0.002)])
def test_config5_nfp_shape_bf16_channels_last_against_oracle(kind, scale, dev, oracle_lib):
    """configs[4]: ViT-Tiny tokens => NFP sees [256,192,14,14] bf16 channels-last, k=5, L2, on the matrix-core
    kernels (fwd_gram, bwd_fast<...,mfma>).  Oracle on the SAME bf16-rounded inputs; bf16 bound: out within 1e-2 and
    grad_x within 2e-2 of the tensor's max magnitude (one bf16 rounding of out; the backward reads that rounded
    out).  'smooth' = neighbours nearly identical, where G_pp + G_qq - 2 G_pq cancels: gradients are checked there
    too, at the same bound."""
    from neighbour_feature_pooling_amd import NFPPooling, _abi
    from neighbour_feature_pooling_amd.synth import feature_map
    ctor = dict(R=2, measure="norm", p=2, padding=2)
    xh = feature_map((256, 192, 14, 14), 501)
    if kind == "smooth":
        xh = (xh.mean(axis=(2, 3), keepdims=True) + scale * xh).astype(np.float32)
    xh = K._bf16_round(xh)
    goh = K._bf16_round(feature_map((256, 24, 14, 14), 502))
    x = torch.from_numpy(xh).to(dev).bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    n0 = _launches()
    out = NFPPooling(192, **ctor)(x)
    fv = _abi.load().nfp_last_variant().decode()
    out.backward(torch.from_numpy(goh).to(dev).bfloat16())
    torch.cuda.synchronize()
    bv = _abi.load().nfp_last_variant().decode()
    assert _launches() == n0 + 2
    assert fv.startswith("fwd_gram<R2,l2,bf16,nhwc"), fv
    assert bv.startswith("bwd_fast<R2,l2,bf16,nhwc,mfma"), bv
    assert x.grad.is_contiguous(memory_format=torch.channels_last)
    ref_out, ref_gx = _oracle_pair(oracle_lib, xh, goh, ctor)
    assert rel_err(out.detach().float().cpu().numpy(), ref_out) <= 1e-2
    assert rel_err(x.grad.float().cpu().numpy(), ref_gx) <= 2e-2


def test_attention_bf16_and_batch_strided_views(dev, oracle_lib):
    """nfp.py:195-205 follows the input dtype: Attention on bf16 maps (raw dots kept in f32 scratch, probabilities
    rounded once); and a batch-strided NCHW view is read in place with a dense gradient."""
    from neighbour_feature_pooling_amd import NFPPooling, _abi
    from neighbour_feature_pooling_amd.synth import feature_map
    ctor = dict(R=1, measure="attention", padding=1)
    xh = K._bf16_round(feature_map((3, 32, 6, 5), 601) * 0.5)
    goh = K._bf16_round(feature_map((3, 8, 6, 5), 602))
    x = torch.from_numpy(xh).to(dev).bfloat16().requires_grad_(True)
    m = NFPPooling(32, **ctor)
    with torch.no_grad():
        o_ng = m(x)
    out = m(x)
    assert torch.equal(out, o_ng)
    out.backward(torch.from_numpy(goh).to(dev).bfloat16())
    ref_out, ref_gx = _oracle_pair(oracle_lib, xh, goh, ctor)
    assert rel_err(out.detach().float().cpu().numpy(), ref_out) <= 1e-2
    assert rel_err(x.grad.float().cpu().numpy(), ref_gx) <= 2e-2
    # batch-strided NCHW view (every other image of a larger buffer)
    big = torch.from_numpy(feature_map((8, 64, 7, 7), 603)).to(dev)
    view = big[::2].requires_grad_(True)
    assert not view.is_contiguous()
    mc = NFPPooling(64, R=1, measure="cosine", padding=1)
    o1 = mc(view)
    assert _abi.load().nfp_last_variant().decode().startswith("fwd_band<R1,cos,f32,nchw")
    g1, = torch.autograd.grad(o1, view, torch.ones_like(o1))
    dense = big[::2].contiguous().requires_grad_(True)
    o2 = mc(dense)
    g2, = torch.autograd.grad(o2, dense, torch.ones_like(o2))
    assert torch.equal(o1, o2) and torch.equal(g1, g2) and g1.is_contiguous()


def test_forward_and_backward_envelopes_agree(dev):
    """Round 1's forward served maps its backward refused (found out inside loss.backward()).  Now every descriptor
    the forward serves, the backward serves too (nfp_direct.h is the table-free last resort of both); and should a
    backward ever be unserved, functional.nfp refuses in forward() when a gradient is needed (`no_bwd` in the plan)."""
    import ctypes
    from neighbour_feature_pooling_amd import NFPPooling, _abi, functional
    L = _abi.load()
    buf = ctypes.create_string_buffer(1024)
    for (C, H, W, R, meas) in [(4, 400, 400, 1, "cosine"), (4, 300, 300, 2, "cosine"), (4, 200, 200, 3, "norm"),
                               (8, 160, 160, 1, "cosine"), (8, 144, 144, 2, "pearson"), (3, 2, 5000, 1, "dot")]:
        x = torch.zeros(1, C, H, W, device=dev)
        d = functional.make_desc(x, NFPPooling(C, R=R, measure=meas, padding=R).config)
        assert L.nfp_plan(ctypes.byref(d), 0, buf, 1024) == 0, L.nfp_last_error()
        assert L.nfp_plan(ctypes.byref(d), 1, buf, 1024) == 0, L.nfp_last_error()
    # the refusal path itself, with a plan whose backward is marked unserved
    m = NFPPooling(8, R=1, measure="cosine", padding=1)
    x = torch.randn(2, 8, 6, 6, device=dev, requires_grad=True)
    m(x)                                                        # builds the plan
    key = next(k for k in functional._PLANS if k[0] == (2, 8, 6, 6))
    d_, shape_, ns_, _ = functional._PLANS[key]
    functional._PLANS[key] = (d_, shape_, ns_, "test: backward unserved")
    try:
        n0 = _launches()
        with pytest.raises(_abi.NfpUnsupported, match="backward is not"):
            m(x)
        assert _launches() == n0
        with torch.no_grad():
            assert m(x).shape == (2, 8, 6, 6)
        assert m(x.detach()).shape == (2, 8, 6, 6)
    finally:
        del functional._PLANS[key]


# ---- configs[3] / configs[4] end to end: DDP over an RCCL process group (world size 1 on this box) ---------------

@pytest.fixture(scope="module")
def rccl_group(dev):
    import socket
    import torch.distributed as dist
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=dev)
    yield dist
    dist.destroy_process_group()


def _ddp_step(net, x, y, dev, steps=2):
    from torch.nn.parallel import DistributedDataParallel as DDP
    from neighbour_feature_pooling_amd import _abi
    from neighbour_feature_pooling_amd.train import make_step
    ddp = DDP(net, device_ids=[dev.index], gradient_as_bucket_view=True)
    step, _ = make_step(ddp)
    n0 = _abi.load().nfp_launch_count()
    losses = [step(x, y).item() for _ in range(steps)]
    torch.cuda.synchronize()
    assert _abi.load().nfp_launch_count() >= n0 + 2 * steps, "the HIP NFP kernels did not run inside the DDP step"
    assert all(np.isfinite(losses)), losses
    for p in ddp.parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all()
    return losses


def test_config4_resnet18_nfp_ddp_step_rccl_224(dev, rccl_group):
    """configs[3]: ResNet18+NFP(cosine), [256,3,224,224] per GPU, DistributedDataParallel on an RCCL ("nccl") group."""
    from neighbour_feature_pooling_amd import _abi
    from neighbour_feature_pooling_amd.train import build, synthetic_batch
    assert rccl_group.get_backend() == "nccl"
    torch.manual_seed(0)
    net = build("resnet18", num_classes=10, in_chans=3, image=224, device=dev)
    x, y = synthetic_batch(256, 3, 224, 10, dev, torch.float32, 11)
    _ddp_step(net, x, y, dev)
    bv = _abi.load().nfp_last_variant().decode()
    assert bv.startswith("bwd_fast<R1,cos,f32,nchw"), bv      # NFP saw [256,512,7,7]
    t = torch.ones(4, device=dev)
    rccl_group.all_reduce(t)                                    # the collective DDP's buckets use, on this group
    assert t.sum().item() == 4.0


def test_config5_vit_tiny_nfp_bf16_ddp_step_rccl_bs256(dev, rccl_group):
    """configs[4]: ViT-Tiny + NFP(k=5, L2) head on 14x14x192 tokens, bf16, bs 256, DDP on the RCCL group."""
    from neighbour_feature_pooling_amd import NFPPooling, _abi
    from neighbour_feature_pooling_amd.train import build, synthetic_batch
    torch.manual_seed(0)
    layer = NFPPooling(192, R=2, measure="norm", p=2, padding=2)
    net = build("vit_tiny_patch16_224", num_classes=10, in_chans=3, image=224, nfp=layer, device=dev,
                dtype=torch.bfloat16)
    x, y = synthetic_batch(256, 3, 224, 10, dev, torch.bfloat16, 12)
    _ddp_step(net, x, y, dev)
    bv = _abi.load().nfp_last_variant().decode()
    assert bv.startswith("bwd_fast<R2,l2,bf16,nhwc"), bv


# ---- large maps: the MultiStage feature maps of texture_pooling.py:211-268 and maps beyond every LDS table ---------

@pytest.mark.parametrize("name", ["ms_cos_112x112x16", "ms_cos_56x56x24", "ms_cos_28x28x40", "ms_cos_14x14x112",
                                  "ms_cos_7x7x960", "geo_cos_224x224", "geo_l2_k5_160x160"])
def test_multistage_and_large_maps_match_reference_golden(name, dev):
    """MobileNetV3_MultiStageNFP feeds NFP 112x112x16 ... 7x7x960 maps; 224x224 is the reference's default
    input_size; 160x160 with k = 5 is beyond the banded backward's tables.  Default dispatch, forward AND backward
    served, bitwise reproducible, against the real reference's outputs."""
    from neighbour_feature_pooling_amd import _abi
    c = K.BY_NAME[name]
    g = load_golden(name)
    out, gx, _ = run_hip(c, dev)
    fv_bv = _abi.load().nfp_last_variant().decode()
    assert_matches_golden(out, gx, g, TOL, 2 * TOL)
    out2, gx2, _ = run_hip(c, dev)
    assert np.array_equal(out, out2) and np.array_equal(gx, gx2), fv_bv


def test_direct_kernels_serve_what_no_table_kernel_can(dev):
    """A map wider than any LDS row window / pair table: the kernels of last resort (nfp_direct.h), against the float64
    formulation; and the forward/backward envelopes agree (nothing is refused in backward that forward served)."""
    from neighbour_feature_pooling_amd import NFPPooling, _abi
    from neighbour_feature_pooling_amd._host import nfp_host
    for (B, C, H, W, ctor) in [(1, 3, 6, 3000, dict(R=2, measure="cosine", padding=2)),
                               (1, 5, 300, 280, dict(R=1, measure="pearson", padding=1, padding_mode="circular")),
                               (2, 4, 150, 170, dict(R=3, measure="norm", p=2, padding=3, stride=2))]:
        m = NFPPooling(C, **ctor)
        gen = torch.Generator().manual_seed(H + W)
        x = (torch.rand(B, C, H, W, generator=gen) + 0.25).to(dev).requires_grad_(True)
        out = m(x)
        fv = _abi.load().nfp_last_variant().decode()
        go = torch.randn(out.shape, generator=gen).to(dev)
        gx, = torch.autograd.grad(out, x, go, retain_graph=True)
        bv = _abi.load().nfp_last_variant().decode()
        gx2, = torch.autograd.grad(out, x, go)
        assert torch.equal(gx, gx2), (fv, bv)
        x64 = x.detach().double().requires_grad_(True)
        ref = nfp_host(x64, m.config)
        gref, = torch.autograd.grad(ref, x64, go.double())
        assert rel_err(out.detach().cpu().numpy(), ref.detach().cpu().numpy()) <= 1e-4, (fv, bv)
        assert rel_err(gx.cpu().numpy(), gref.cpu().numpy()) <= 2e-4, (fv, bv)


# ---- both radii from one pass (models/nfp_heads.py:80-118) ---------------------------------------------------------

@pytest.mark.parametrize("shape,measure,mode,dtype,layout", [
    ((64, 512, 7, 7), "cosine", "reflect", torch.float32, "nchw"),      # the head's call shape (in_c = 512, 7x7)
    ((5, 64, 7, 7), "cosine", "reflect", torch.float32, "nhwc"),
    ((3, 48, 14, 14), "norm", "reflect", torch.float32, "nchw"),
    ((2, 24, 5, 6), "cosine", "zeros", torch.float32, "nchw"),
    ((2, 16, 3, 3), "cosine", "reflect", torch.float32, "nchw"),        # reflect folds taps onto the pixel itself
    ((4, 64, 7, 7), "cosine", "reflect", torch.bfloat16, "nhwc"),
])
def test_multi_radius_fused_matches_oracle_concatenation(shape, measure, mode, dtype, layout, dev, oracle_lib):
    """MultiRadiusNFPPooling(R_list=(1, 2)) on the GPU: one forward and one backward kernel produce
    cat([NFP_R1(x), NFP_R2(x)]) and its gradient; against the oracle run once per radius."""
    from neighbour_feature_pooling_amd import MultiRadiusNFPPooling, _abi
    from neighbour_feature_pooling_amd.synth import feature_map
    kw = dict(padding_mode=mode)
    if measure == "norm":
        kw["p"] = 2
    m = MultiRadiusNFPPooling(shape[1], R_list=(1, 2), measure=measure, **kw)
    bf = dtype == torch.bfloat16
    xh = feature_map(shape, 71)
    goh = feature_map((shape[0], 32, shape[2], shape[3]), 72)
    if bf:
        xh, goh = K._bf16_round(xh), K._bf16_round(goh)
    x = torch.from_numpy(xh).to(dev).to(dtype)
    if layout == "nhwc":
        x = x.contiguous(memory_format=torch.channels_last)
    x.requires_grad_(True)
    n0 = _launches()
    out = m(x)
    fv = _abi.load().nfp_last_variant().decode()
    out.backward(torch.from_numpy(goh).to(dev).to(dtype))
    torch.cuda.synchronize()
    bv = _abi.load().nfp_last_variant().decode()
    assert _launches() == n0 + 2, (fv, bv)                        # ONE forward and ONE backward kernel
    assert fv.startswith("fwd_band<R1+2,") and bv.startswith("bwd_fast<R1+2,"), (fv, bv)
    c1 = dict(R=1, measure=measure, padding=1, padding_mode=mode, **({"p": 2} if measure == "norm" else {}))
    c2 = dict(c1, R=2, padding=2)
    ref_out = np.concatenate([oracle_lib.forward(xh, **c1), oracle_lib.forward(xh, **c2)], axis=1)
    ref_gx = oracle_lib.backward(xh, goh[:, :8].copy(), **c1) + oracle_lib.backward(xh, goh[:, 8:].copy(), **c2)
    to, tg = (1e-2, 2e-2) if bf else (TOL, TOL)
    assert out.shape == ref_out.shape
    assert rel_err(out.detach().float().cpu().numpy(), ref_out) <= to
    assert rel_err(x.grad.float().cpu().numpy(), ref_gx) <= tg
    # and bitwise equal to the two single-radius maps where those run on the same kernels (float32)
    if not bf:
        sep = torch.cat([blk(x.detach()) for blk in m.nfp_blocks], dim=1)
        assert (out.detach() - sep).abs().max().item() <= 2e-6


def test_multi_radius_falls_back_to_two_passes(dev):
    """Radii other than (1, 2) or a measure without a fused kernel: the same concatenation from one kernel pair per
    radius — still HIP kernels, never a CPU path.  (Replicate padding is fused while the link tables hold its folds:
    nfp_workspace_bytes checks the exact bound per geometry.)"""
    from neighbour_feature_pooling_amd import MultiRadiusNFPPooling, _abi
    x = torch.randn(2, 16, 9, 9, device=dev)
    for kw, fused in ((dict(R_list=(1, 2), measure="cosine", padding_mode="replicate"), None),
                      (dict(R_list=(1, 2), measure="dot"), True),           # round 4: every hot measure is fused, p = 1 / EMD too
                      (dict(R_list=(1, 2), measure="norm", p=1), True), (dict(R_list=(1, 2), measure="emd"), True),
                      (dict(R_list=(1, 2), measure="canberra"), False), (dict(R_list=(1, 2), measure="norm", p=3), False),
                      (dict(R_list=(2, 3), measure="cosine"), False)):
        m = MultiRadiusNFPPooling(16, **kw)
        n0 = _launches()
        y = m(x)
        n = _launches() - n0
        assert n >= (2 if fused is False else 1)
        if fused is True:
            assert n == 1 and _abi.load().nfp_last_variant().decode().startswith("fwd_band<R1+2,"), kw
        ref = torch.cat([b(x) for b in m.nfp_blocks], dim=1)
        assert (y - ref).abs().max().item() <= 2e-6


# ---- C++ autograd nodes (csrc/nfp_torch.cpp) vs the Python nodes of functional.py ----------------------------------

def test_cpp_autograd_nodes_equal_python_nodes(dev):
    """Both node flavours drive the same C ABI: identical outputs and gradients, same launches, same error classes."""
    from neighbour_feature_pooling_amd import NFPPooling, _abi, functional
    from neighbour_feature_pooling_amd.functional import nfp_pool
    cpp = functional._cpp_nodes()
    assert cpp, "the C++ autograd nodes (_nfp_torch.so) are not built / do not load"
    cases = [((8, 64, 7, 7), dict(R=1, measure="cosine", padding=1), torch.float32, False),
             ((4, 192, 14, 14), dict(R=2, measure="norm", p=2, padding=2), torch.bfloat16, True),
             ((2, 12, 9, 8), dict(R=1, measure="canberra", padding=2, stride=2), torch.float32, False),
             ((3, 16, 6, 6), dict(R=1, measure="attention", padding=1), torch.bfloat16, False)]
    for shape, ctor, dtype, nhwc in cases:
        m = NFPPooling(shape[1], **ctor)
        x0 = (torch.rand(shape, device=dev) + 0.25).to(dtype)
        if nhwc:
            x0 = x0.contiguous(memory_format=torch.channels_last)
        res = []
        for flavour in (cpp, False):
            functional._CPP = flavour
            try:
                x = x0.clone(memory_format=torch.preserve_format).requires_grad_(True)
                n0 = _launches()
                out = m(x)
                go = torch.ones_like(out) * 0.5
                gx, = torch.autograd.grad(out, x, go)
                torch.cuda.synchronize()
                res.append((out.detach(), gx, _launches() - n0))
                if ctor["measure"] in ("cosine", "norm") and ctor.get("stride", 1) == 1:
                    x = x0.clone(memory_format=torch.preserve_format).requires_grad_(True)
                    gap, nfpm = nfp_pool(x, m.config)
                    (gap.sum() + 2.0 * nfpm.sum()).backward()
                    res[-1] += (gap.detach(), nfpm.detach(), x.grad)
                    x = x0.clone(memory_format=torch.preserve_format).requires_grad_(True)
                    gap, nfpm = nfp_pool(x, m.config)
                    nfpm.sum().backward()                       # gap unused: its gradient arrives undefined
                    res[-1] += (x.grad,)
            finally:
                functional._CPP = cpp
        a, b = res
        assert a[2] == b[2]
        for u, v in zip(a[:2] + a[3:], b[:2] + b[3:]):
            assert torch.equal(u, v)
    with pytest.raises(_abi.NfpUnsupported, match="norm order"):
        NFPPooling(8, padding=1, measure="norm", p=float("inf"))(torch.randn(1, 8, 5, 5, device=dev))


# ---- round 3: parity hardening -------------------------------------------------------------------------------------

@pytest.mark.parametrize("name", [c["name"] for c in K.CASES
                                  if c["ctor"]["measure"].lower() in ("norm", "rmse") and c["ctor"].get("p", 1) in (1, 2)])
def test_distance_maps_match_the_reference_element_wise(name, dev):
    """L2 / L1 / RMSE maps have one sign and similar magnitudes, so an element-wise bound means something (VERDICT r2,
    weak #2): |out - ref| <= 1e-5 |ref| + 1e-6 max|ref| against the real reference's golden output."""
    c = K.BY_NAME[name]
    g = load_golden(name)
    if "out" not in g:
        pytest.skip("large case: the fixture holds samples and sums only")
    out, _, _ = run_hip(c, dev)
    ref = g["out"].astype(np.float64)
    ok = np.isfinite(ref)
    assert np.all(np.abs(out.astype(np.float64) - ref)[ok] <= 1e-5 * np.abs(ref[ok]) + 1e-6 * np.abs(ref[ok]).max())


@pytest.mark.parametrize("shape,ctor,layout,dtype,fwd,bwd", [
    ((256, 512, 7, 7), dict(R=1, measure="cosine", padding=1), "nchw", torch.float32, "fwd_band<R1,cos,f32,nchw,pool>", "bwd_fast<R1,cos,f32,nchw,pool>"),
    ((256, 512, 7, 7), dict(R=1, measure="cosine", padding=1), "nhwc", torch.bfloat16, "fwd_gram<R1,cos,bf16,nhwc,pool>", "bwd_fast<R1,cos,bf16,nhwc,mfma,pool>"),
    ((256, 192, 14, 14), dict(R=2, measure="norm", p=2, padding=2), "nhwc", torch.bfloat16, "fwd_gram<R2,l2,bf16,nhwc,pool>", "bwd_fast<R2,l2,bf16,nhwc,mfma2,pool>"),
])
def test_fused_pooling_tail_against_the_oracle_at_config_shapes(shape, ctor, layout, dtype, fwd, bwd, dev, oracle_lib):
    """The pooled kernels the train steps of configs[3] / configs[4] actually run (fwd_band / fwd_gram <...,pool>,
    bwd_fast<...[,mfma],pool>) against the CPU ORACLE — not against this package's own nfp + mean composition
    (VERDICT r2, weak #1): gap, nfpm and grad_x for a loss that weights both pooled outputs."""
    from neighbour_feature_pooling_amd import NFPPooling, _abi
    from neighbour_feature_pooling_amd.functional import nfp_pool
    from neighbour_feature_pooling_amd.synth import feature_map
    B, C, H, W = shape
    m = NFPPooling(C, **ctor)
    N = m.out_channels
    xh = feature_map(shape, 611)
    if dtype == torch.bfloat16:
        xh = K._bf16_round(xh)
    wg, wn = feature_map((B, C), 612), feature_map((B, N), 613)
    x = torch.from_numpy(xh).to(dev).to(dtype)
    if layout == "nhwc":
        x = x.contiguous(memory_format=torch.channels_last)
    x.requires_grad_(True)
    L = _abi.load()
    gap, nfpm = nfp_pool(x, m.config)
    fv = L.nfp_last_variant().decode()
    ((gap * torch.from_numpy(wg).to(dev)).sum() + (nfpm * torch.from_numpy(wn).to(dev)).sum()).backward()
    torch.cuda.synchronize()
    bv = L.nfp_last_variant().decode()
    assert fv.startswith(fwd) and bv.startswith(bwd), (fv, bv)
    # oracle: the maps, their means, and the gradient of the same loss: grad_out[b,n,p] = wn[b,n] / P, plus wg[b,c] / P
    P = H * W
    ref_map = oracle_lib.forward(xh, **ctor)
    go = np.broadcast_to((wn / P)[:, :, None, None], ref_map.shape).astype(np.float32).copy()
    ref_gx = oracle_lib.backward(xh, go, **ctor) + (wg / P)[:, :, None, None]
    bf = dtype == torch.bfloat16
    assert rel_err(gap.detach().cpu().numpy(), xh.mean((2, 3))) <= (1e-5 if not bf else 1e-5)       # sums of the stored inputs, in f32
    assert rel_err(nfpm.detach().cpu().numpy(), ref_map.mean((2, 3))) <= (1e-5 if not bf else 1e-2)
    assert rel_err(x.grad.float().cpu().numpy(), ref_gx) <= (1e-5 if not bf else 2e-2)


def test_autocast_and_float16_follow_the_reference_policy(dev):
    """SURVEY a7 / VERDICT r2 item 7c.  Measured on this GPU with the reference's own op sequence (oracle/unfold_torch.py,
    profiles/r03_c_autocast_probe_reference_ops.jsonl): under torch.autocast the maps come back float32 (cosine_similarity
    and linalg.norm are on autocast's float32 list), gradients in the input's type; a float16 tensor outside autocast
    gives float16 maps.  The op here does the same, computing in float32."""
    from neighbour_feature_pooling_amd import NFPPooling
    from oracle.unfold_torch import UnfoldNFP
    for meas, kw in (("cosine", {}), ("norm", {"p": 2})):
        ctor = dict(R=1, measure=meas, padding=1, **kw)
        m = NFPPooling(64, **ctor)
        ref_m = UnfoldNFP(64, **ctor)
        ref_m.w_comp, ref_m.w_centre = ref_m.w_comp.to(dev), ref_m.w_centre.to(dev)
        x32 = torch.randn(2, 64, 7, 7, device=dev)
        exact = ref_m(x32)
        for ac in (torch.bfloat16, torch.float16):
            for xdt in (torch.float32, torch.bfloat16, torch.float16):
                x = x32.to(xdt).requires_grad_(True)
                xr = x32.to(xdt).requires_grad_(True)
                with torch.autocast("cuda", ac):
                    out, ref = m(x), ref_m(xr)
                (gx,) = torch.autograd.grad(out.float().sum(), x)
                (gr,) = torch.autograd.grad(ref.float().sum(), xr)
                assert out.dtype == ref.dtype == torch.float32 and gx.dtype == gr.dtype == xdt
                # both are within the reference's own low-precision error of the float32 result
                assert rel_err(out.detach().cpu().numpy(), exact.cpu().numpy()) <= 1e-2
        x = x32.half().requires_grad_(True)          # float16 outside autocast
        xr = x32.half().requires_grad_(True)
        ref_m.w_comp, ref_m.w_centre = ref_m.w_comp.half(), ref_m.w_centre.half()
        out, ref = m(x), ref_m(xr)
        (gx,) = torch.autograd.grad(out.float().sum(), x)
        assert out.dtype == ref.dtype == torch.float16 and gx.dtype == torch.float16
        assert rel_err(out.detach().float().cpu().numpy(), exact.cpu().numpy()) <= 2e-3


# ---- cases the suite did not reach -----------------------------------------------------------------

@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(8, 8, 9, 3), (28, 8, 9, 3), (5, 16, 11, 3), (3, 8, 13, 5)])
def test_band_forward_overlapped_last_block_below_the_band_origin(shape, dtype, dev, monkeypatch):
    """H*W % 4 != 0 with a last band that starts on an 8-aligned pixel above P - 4: the NCHW staging's overlapped last
    block begins below the band's slot origin (fwd_band's commit redirects those three slots).  Against the float64
    formulation, forward and backward."""
    from neighbour_feature_pooling_amd import NFPPooling, _abi
    from neighbour_feature_pooling_amd._host import nfp_host
    from neighbour_feature_pooling_amd.synth import feature_map
    nfp_switch(monkeypatch, "NFP_MFMA", "0")         # (bf16 with C % 16 == 0 would take the matrix-core forward)
    m = NFPPooling(shape[1], R=1, measure="cosine", padding=1)
    x = torch.from_numpy(feature_map(shape, 77)).to(dev).to(dtype).requires_grad_(True)
    out = m(x)
    assert _abi.load().nfp_last_variant().decode().startswith("fwd_band<R1,cos")
    go = torch.from_numpy(feature_map(tuple(out.shape), 78)).to(dev).to(dtype)
    gx, = torch.autograd.grad(out, x, go)
    x64 = x.detach().double().requires_grad_(True)
    ref = nfp_host(x64, m.config)
    gref, = torch.autograd.grad(ref, x64, go.double())
    to, tg = (TOL, TOL) if dtype == torch.float32 else (1e-2, 2e-2)
    assert rel_err(out.detach().float().cpu().numpy(), ref.detach().cpu().numpy()) <= to
    assert rel_err(gx.float().cpu().numpy(), gref.cpu().numpy()) <= tg


@pytest.mark.parametrize("shape", [(3, 32, 16, 16), (2, 64, 20, 20), (2, 32, 22, 23), (9, 256, 20, 20)])
@pytest.mark.parametrize("measure", ["cosine", "norm"])
def test_fused_callers_between_196_and_512_pixels(shape, measure, dev):
    """The fused pooling tail and the multi-radius maps with R = 2 above 14x14 (where round 2's nfp_pool_supported promised
    descriptors its launchers then refused, and where the multi-radius backward does not fit and training falls back to two
    passes while no_grad uses the fused kernel): each against the composition of the plain ops, forward and backward,
    and the no_grad result against the training-mode one."""
    from neighbour_feature_pooling_amd import MultiRadiusNFPPooling, NFPPooling
    from neighbour_feature_pooling_amd.functional import nfp, nfp_pool
    from neighbour_feature_pooling_amd.synth import feature_map
    B, C, H, W = shape
    ctor = dict(R=2, measure=measure, padding=2, **({"p": 2} if measure == "norm" else {}))
    m = NFPPooling(C, **ctor)
    x1 = torch.from_numpy(feature_map(shape, 31)).to(dev).requires_grad_(True)
    x2 = x1.detach().clone().requires_grad_(True)
    gap, nfpm = nfp_pool(x1, m.config)
    rg, rn = x2.mean((2, 3)), nfp(x2, m.config).mean((2, 3))
    assert rel_err(gap.detach().cpu().numpy(), rg.detach().cpu().numpy()) <= 2e-6
    assert rel_err(nfpm.detach().cpu().numpy(), rn.detach().cpu().numpy()) <= 2e-6
    wg = torch.from_numpy(feature_map((B, C), 32)).to(dev)
    wn = torch.from_numpy(feature_map((B, 24), 33)).to(dev)
    ((gap * wg).sum() + (nfpm * wn).sum()).backward()
    ((rg * wg).sum() + (rn * wn).sum()).backward()
    assert rel_err(x1.grad.cpu().numpy(), x2.grad.cpu().numpy()) <= TOL
    # multi-radius (1, 2): training mode (fused or two passes, whichever the backward allows) vs no_grad (fused forward)
    mr = MultiRadiusNFPPooling(C, R_list=(1, 2), measure=measure, **({"p": 2} if measure == "norm" else {}))
    x3 = x1.detach().clone().requires_grad_(True)
    y = mr(x3)
    with torch.no_grad():
        y0 = mr(x1.detach())
    ref = torch.cat([blk(x1.detach()) for blk in mr.nfp_blocks], dim=1)
    assert rel_err(y.detach().cpu().numpy(), ref.cpu().numpy()) <= 2e-6
    assert rel_err(y0.cpu().numpy(), ref.cpu().numpy()) <= 2e-6
    go = torch.from_numpy(feature_map(tuple(y.shape), 34)).to(dev)
    (g3,) = torch.autograd.grad(y, x3, go)
    x4 = x1.detach().clone().requires_grad_(True)
    (g4,) = torch.autograd.grad(torch.cat([blk(x4) for blk in mr.nfp_blocks], dim=1), x4, go)
    assert rel_err(g3.cpu().numpy(), g4.cpu().numpy()) <= TOL


def test_workspace_first_seen_inside_a_graph_capture(dev):
    """a geometry first seen under capture used to get its table fill RECORDED, not run, and the unfilled
    buffer cached.  Now that call runs without tables (general kernels, a RuntimeWarning) and nothing is cached: the
    replay and later eager calls agree with an eager reference."""
    from neighbour_feature_pooling_amd import NFPPooling, functional
    m = NFPPooling(8, R=1, measure="cosine", padding=1)
    x = torch.randn(2, 8, 6, 5, device=dev)
    for k in [k for k in functional._WORKSPACES if k[1:3] == (6, 5)]:   # (as if no earlier call had seen this geometry)
        del functional._WORKSPACES[k]
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with pytest.warns(RuntimeWarning, match="graph capture"):
            with torch.cuda.graph(g, stream=s):
                y = m(x)
        g.replay()
    torch.cuda.synchronize()
    assert not [k for k in functional._WORKSPACES if k[1:3] == (6, 5)]
    y2 = m(x)                                          # eager: builds the tables now
    assert [k for k in functional._WORKSPACES if k[1:3] == (6, 5)]
    assert rel_err(y.cpu().numpy(), y2.cpu().numpy()) <= 2e-6


def test_workspace_fill_is_ordered_before_a_plan_cache_hit_on_another_stream(dev):
    """the first call of a geometry enqueues its table fill on stream A; a same-shape call on stream B hits
    the plan cache — and used to launch without waiting for the fill.  Stream A is held up by a long sleep kernel in
    front of the fill, so an unordered B would read unwritten tables."""
    from neighbour_feature_pooling_amd import NFPPooling, functional
    m = NFPPooling(8, R=1, measure="cosine", padding=1)
    x = torch.randn(2, 8, 9, 11, device=dev)
    for k in [k for k in functional._WORKSPACES if k[1:3] == (9, 11)]:
        del functional._WORKSPACES[k]
    for k in [k for k in functional._PLANS if isinstance(k[0], tuple) and k[0][2:] == (9, 11)]:
        del functional._PLANS[k]
    torch.cuda.synchronize()
    a, b = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(a):
        torch.cuda._sleep(400_000_000)       # ~0.2 s in front of the fill
        ya = m(x)
    with torch.cuda.stream(b):
        yb = m(x)                            # plan-cache hit, tables possibly still unwritten
    torch.cuda.synchronize()
    assert functional._PENDING_FILLS == [] or all(r[1] is not None for r in functional._PENDING_FILLS)
    yc = m(x)
    torch.cuda.synchronize()
    assert functional._PENDING_FILLS == []
    assert torch.equal(ya, yc) and torch.equal(yb, yc)


@pytest.mark.parametrize("pooled", [False, True])
def test_torch_compile_fullgraph_runs_the_hip_kernels(pooled, dev):
    """A model holding NFPPooling / nfp_pooling compiles as ONE graph (fullgraph=True raises on any break) — the registered
    custom ops of _ops.py stand for the op inside it — and the compiled forward / backward launch the same HIP kernels
    and give the eager results; torch.library.opcheck validates schema, fake implementation and autograd registration."""
    from test_compile import Net
    from neighbour_feature_pooling_amd import _abi, _ops
    from neighbour_feature_pooling_amd.functional import NfpConfig
    torch.manual_seed(0)
    net = Net(pooled, device="cuda")
    x = torch.randn(4, 3, 9, 9, device=dev)
    ref = net(x)
    ref.square().sum().backward()
    g_ref = net.conv.weight.grad.clone()
    net.zero_grad()
    n0 = _launches()
    y = torch.compile(net, fullgraph=True, backend="aot_eager")(x)
    y.square().sum().backward()
    torch.cuda.synchronize()
    assert _launches() >= n0 + 2
    assert torch.allclose(y, ref, atol=1e-6) and torch.allclose(net.conv.weight.grad, g_ref, atol=1e-5)
    cfg = NfpConfig(R=1, measure="cosine", padding=1)
    xs = torch.randn(2, 16, 9, 9, device=dev, requires_grad=True)
    torch.library.opcheck(torch.ops.nfp_amd.nfp.default, (xs, *_ops.cfg_args(cfg), True),
                          test_utils=("test_schema", "test_faketensor", "test_autograd_registration"))
    torch.library.opcheck(torch.ops.nfp_amd.nfp_pool.default, (xs, *_ops.cfg_args(cfg), True, True),
                          test_utils=("test_schema", "test_faketensor", "test_autograd_registration"))


def test_float64_feature_maps_follow_the_input_type(dev, oracle_lib):
    """VERDICT r3 missing #8: the reference follows the input dtype (nfp.py:141-159); float64 CUDA tensors used to be
    refused.  They are computed in float32 (a RuntimeWarning says so once) and come back float64, gradients too."""
    import warnings
    from neighbour_feature_pooling_amd import NFPPooling
    from neighbour_feature_pooling_amd.synth import feature_map
    xh = feature_map((3, 32, 9, 8), 77)
    goh = feature_map((3, 8, 9, 8), 78)
    x = torch.from_numpy(xh).to(dev).double().requires_grad_(True)
    n0 = _launches()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        out = NFPPooling(32, R=1, measure="cosine", padding=1)(x)
    gx, = torch.autograd.grad(out, x, torch.from_numpy(goh).to(dev).double())
    assert out.dtype == torch.float64 and gx.dtype == torch.float64 and _launches() == n0 + 2
    ctor = dict(R=1, measure="cosine", padding=1)
    assert rel_err(out.detach().cpu().numpy(), oracle_lib.forward(xh, **ctor)) <= TOL
    assert rel_err(gx.cpu().numpy(), oracle_lib.backward(xh, goh, **ctor)) <= TOL


def _load_script(name):
    import importlib.util, os
    spec = importlib.util.spec_from_file_location(name, os.path.join(os.path.dirname(__file__), "..", "scripts", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_random_stress_of_the_table_kernels_at_large_batches(dev):
    """Batches of 1024+ images run quarter-size workgroups, four per CU, in both passes (csrc/nfp_hip.hip): random small
    maps, every hot measure incl. Norm p = 1, padding modes, layouts, storage types, plain and pooled, against the
    float64 formulation — NaN patterns included.  scripts/stress_big_batch.py is the long form."""
    import random
    sb = _load_script("stress_big_batch")
    rnd = random.Random(4242)
    for _ in range(24):
        ok, desc, errs, vs = sb.one_case(rnd, dev)
        assert ok, (desc, errs, vs)
        torch.cuda.empty_cache()


@pytest.mark.parametrize("form", ["2", "0"])
def test_random_stress_of_the_matrix_core_kernels(form, dev, monkeypatch):
    """bf16 maps of whole 32-channel tiles (fwd_gram, bwd_fast<...,mfma / mfma2>): random geometries up to 20 x 20, both
    radii, padding modes, layouts, batches up to 300, plain and pooled, against the float64 formulation — with the backward's
    second form (nfp_fast.h::bwd_gemm_phase3) wherever its LDS fits (NFP_GEMM3=2) and with the first form only (0).
    Long form: STRESS_C32=1 scripts/stress_big_batch.py (80 + 80 cases passed, profiles/r04_zc_…)."""
    import random
    sb = _load_script("stress_big_batch")
    for k, v in (("STRESS_C32", "1"), ("STRESS_B", "1,300"), ("STRESS_HW", "20")):
        monkeypatch.setenv(k, v)
    nfp_switch(monkeypatch, "NFP_GEMM3", form)
    rnd = random.Random(77 + int(form))
    seen = set()
    for _ in range(16):
        ok, desc, errs, vs = sb.one_case(rnd, dev)
        assert ok, (desc, errs, vs)
        seen.add("mfma2" if "mfma2" in vs[1] else ("mfma" if "mfma" in vs[1] else "other"))
        torch.cuda.empty_cache()
    nfp_switch(monkeypatch, "NFP_GEMM3", None)
    assert ("mfma2" if form == "2" else "mfma") in seen, seen
    assert form == "2" or "mfma2" not in seen, seen


def test_dot_product_backward_in_bf16_storage_has_no_nan(dev):
    """Round 3 found it with the stress above: DotProduct keeps no saved norms, so the table backward reads the OUTPUT MAP
    in their place; read as floats, a bf16 map holds NaN / Inf bit patterns, and `NaN * 0` reached the window weights.
    (The vector backward: C not a multiple of 32.)"""
    sb = _load_script("stress_tile")
    from neighbour_feature_pooling_amd import NFPPooling, _abi
    from neighbour_feature_pooling_amd._host import nfp_host
    from neighbour_feature_pooling_amd.synth import feature_map
    m = NFPPooling(60, R=1, measure="dot", padding=1, padding_mode="zeros")
    x = torch.from_numpy(feature_map((8, 60, 12, 5), 5)).to(dev).bfloat16().requires_grad_(True)
    out = m(x)
    go = torch.from_numpy(feature_map(tuple(out.shape), 6)).to(dev).bfloat16()
    gx, = torch.autograd.grad(out, x, go)
    assert _abi.load().nfp_last_variant().decode().startswith("bwd_fast<R1,dot,bf16,nchw")
    x64 = x.detach().double().requires_grad_(True)
    gref, = torch.autograd.grad(nfp_host(x64, m.config), x64, go.double())
    assert not torch.isnan(gx).any()
    assert sb.rel_err(gx.float().cpu().numpy(), gref.cpu().numpy()) <= 2e-2


@pytest.mark.parametrize("mode,R,shape", [("replicate", 1, (5, 32, 13, 16)), ("reflect", 2, (3, 16, 12, 10)), ("replicate", 2, (1030, 8, 6, 7))])
def test_rmse_has_no_subgradient_at_distance_zero_like_the_reference(mode, R, shape, dev):
    """RMSE (nfp.py:172-179) of a pixel and its own padded copy — replicate padding; reflect with R = 2, where tap -2 of
    column 1 lands on column 1 — is sqrt(0): torch's backward gives inf * 0 = NaN on that pixel, every channel.  The table
    kernels dropped such pairs (right for L2, whose gradient at 0 is 0) and returned numbers where the reference has NaN;
    they now add the pair's +-inf to both of the pixel's weights.  The NaN pattern and every other element must match."""
    sb = _load_script("stress_tile")
    from neighbour_feature_pooling_amd import NFPPooling, _abi
    from neighbour_feature_pooling_amd._host import nfp_host
    from neighbour_feature_pooling_amd.synth import feature_map
    m = NFPPooling(shape[1], R=R, measure="rmse", padding=R, padding_mode=mode)
    x = torch.from_numpy(feature_map(shape, 21)).to(dev).requires_grad_(True)
    out = m(x)
    go = torch.from_numpy(feature_map(tuple(out.shape), 22)).to(dev)
    gx, = torch.autograd.grad(out, x, go)
    assert _abi.load().nfp_last_variant().decode().startswith("bwd_fast<")
    x64 = x.detach().double().requires_grad_(True)
    gref, = torch.autograd.grad(nfp_host(x64, m.config), x64, go.double())
    assert torch.isnan(gref).any()
    assert sb.rel_err(gx.cpu().numpy(), gref.cpu().numpy()) <= 2e-5


def test_random_stress_of_the_one_pass_multi_radius_maps(dev):
    """MultiRadiusNFPPooling (nfp_heads.py:88-110) on random small maps, every hot measure, against the float64
    cat([NFP_R1(x), NFP_R2(x)]) — scripts/stress_multi_radius.py is the long form (200 cases passed)."""
    import random
    sm = _load_script("stress_multi_radius")
    rnd = random.Random(808)
    fused = 0
    for _ in range(24):
        ok, desc, errs, vs = sm.one_case(rnd, dev)
        fused += "R1+2" in vs[0]
        assert ok, (desc, errs, vs)
        torch.cuda.empty_cache()
    assert fused >= 6
