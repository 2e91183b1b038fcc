"""The nn.Module mirror of the reference interface (no GPU): constructor/attribute contract
(nfp.py:16-39,120,125-130), checkpoint keys, the CPU-tensor path used by the reference heads'
__init__ probes, and the nfp_pooling wrapper — all against the reference's golden outputs."""
import numpy as np
import pytest
import torch

import cases as K
from conftest import load_golden, rel_err, same_nan_pattern
from neighbour_feature_pooling_amd import EnhancedNFPPooling, NFPPooling, nfp_pooling
from neighbour_feature_pooling_amd.synth import feature_map


def test_constructor_contract():
    m = NFPPooling(512, R=1, measure='cosine', padding=1)
    assert (m.in_channels, m.kernel_size, m.out_channels, m.R) == (512, 3, 8, 1)
    assert (m.stride, m.padding, m.dilation, m.padding_mode, m.similarity, m.p, m.eps) == \
        (1, 1, 1, 'reflect', True, 1, 1e-6)
    assert NFPPooling(8, R=2).out_channels == 24
    assert NFPPooling(8, measure='Cosine').measure == 'cosine'
    assert sum(p.numel() for p in m.parameters()) == 0
    m.in_channels = 256  # models/resnet18.py:166 assigns it
    assert m.in_channels == 256
    with pytest.raises(RuntimeError, match="Similarity measure mahalanobis not implemented"):
        NFPPooling(8, measure='mahalanobis')
    with pytest.raises(RuntimeError, match="not implemented"):
        NFPPooling(8, measure='nope')


@pytest.mark.parametrize("size,pad,stride,dil,R", [(224, 0, 1, 1, 1), (7, 1, 1, 1, 1), (14, 2, 1, 1, 2), (9, 0, 2, 2, 1)])
def test_output_size_property(size, pad, stride, dil, R):
    m = NFPPooling(4, R=R, padding=pad, stride=stride, dilation=dil, input_size=size)
    k = 2 * R + 1
    assert m.output_size == (size + 2 * pad - dil * (k - 1) - 1) // stride + 1


def test_state_dict_has_reference_keys_and_round_trips():
    m = NFPPooling(6, R=1, measure='cosine', padding=1)
    sd = m.state_dict()
    assert list(sd) == ['comp_neighbors.weight', 'center_value.weight']
    assert sd['comp_neighbors.weight'].shape == (48, 1, 3, 3) and sd['center_value.weight'].shape == (6, 1, 3, 3)
    # one-hot selectors: neighbour n of every channel, row-major, centre skipped (nfp.py:64-80)
    w = sd['comp_neighbors.weight'].view(6, 8, 3, 3)
    taps = [(0, 0), (0, 1), (0, 2), (1, 0), (1, 2), (2, 0), (2, 1), (2, 2)]
    for n, (i, j) in enumerate(taps):
        assert w[:, n, i, j].eq(1).all() and w[:, n].sum().item() == 6
    assert sd['center_value.weight'][:, 0, 1, 1].eq(1).all()
    wn = NFPPooling(6, measure='norm', padding=1).state_dict()['comp_neighbors.weight'].view(6, 8, 3, 3)
    assert wn[:, :, 1, 1].eq(1).all() and wn[:, 0, 0, 0].eq(-1).all()
    NFPPooling(6, R=1, measure='cosine', padding=1).load_state_dict(sd, strict=True)
    outer = torch.nn.Sequential(NFPPooling(6, padding=1))
    outer.load_state_dict(torch.nn.Sequential(m).state_dict(), strict=True)


def test_enhanced_alias_accepts_head_call_sites():
    m = EnhancedNFPPooling(in_channels=32, R=1, measure="cosine", padding=1)
    with torch.no_grad():
        assert m(torch.randn(1, 32, 7, 7)).shape == (1, 8, 7, 7)       # nfp_heads.py:24-27
    m0 = EnhancedNFPPooling(in_channels=16, R=1, measure="cosine", padding=0, some_future_kw=3)
    with torch.no_grad():
        assert m0(torch.randn(1, 16, 5, 5)).shape == (1, 8, 3, 3)      # nfp_heads.py:166-167


def test_channel_mismatch_raises():
    with pytest.raises(RuntimeError, match="channels"):
        NFPPooling(8, padding=1)(torch.randn(1, 4, 5, 5))


@pytest.mark.parametrize("name", [c["name"] for c in K.CASES if np.prod(c["shape"]) <= 40000])
def test_cpu_tensor_path_matches_reference(name):
    c = K.BY_NAME[name]
    g = load_golden(name)
    x = torch.from_numpy(K.make_input(c)).requires_grad_(True)
    m = NFPPooling(c["shape"][1], **c["ctor"])
    out = m(x)
    assert tuple(out.shape) == g["out"].shape
    assert same_nan_pattern(out.detach().numpy(), g["out"])
    assert rel_err(np.nan_to_num(out.detach().numpy()), np.nan_to_num(g["out"])) <= 1e-5
    out.backward(torch.from_numpy(K.make_grad_out(c, tuple(out.shape))))
    gx = x.grad.numpy()
    if "gx" in g:
        assert same_nan_pattern(gx, g["gx"])
        assert rel_err(np.nan_to_num(gx), np.nan_to_num(g["gx"])) <= 1e-5


def test_nfp_pooling_wrapper_matches_reference():
    g = load_golden("wrapper_nfp_pooling_4x64x7x7")
    B, C = 4, 64
    params = {"num_ftrs": {"m": C}, "Model_name": "m", "Dataset": "d", "num_classes": {"d": 10}, "input_size": 7}
    w = nfp_pooling(Params=params)
    assert w.nfp_proj.weight.shape == (C, 8) and w.num_classes == 10 and w.model_name == "m"
    with torch.no_grad():
        w.nfp_proj.weight.copy_(torch.from_numpy(feature_map((C, 8), 201) * 0.3))
        w.nfp_proj.bias.copy_(torch.from_numpy(feature_map((C,), 202) * 0.1))
    x = torch.from_numpy(feature_map((B, C, 7, 7), 200)).requires_grad_(True)
    y = w(x)
    y.backward(torch.from_numpy(feature_map((B, C), 203)))
    assert rel_err(y.detach().numpy(), g["y"]) <= 1e-5
    assert rel_err(x.grad.numpy(), g["gx"]) <= 1e-5
    assert rel_err(w.nfp_proj.weight.grad.numpy(), g["gw"]) <= 1e-5
    assert rel_err(w.nfp_proj.bias.grad.numpy(), g["gb"]) <= 1e-5
    assert nfp_pooling().nfp_proj is None  # NFP_Pooling.py:23


def test_inner_layout_and_canonical_strides():
    """functional._dense: images dense in NCHW or channels-last order are read in place whatever the batch stride
    (ViT tokens behind a class token, texture_pooling.py:181-188); anything else is copied to NCHW once."""
    from neighbour_feature_pooling_amd import functional as F
    x = torch.randn(4, 8, 5, 6)
    assert F._dense(x)[1] == "nchw" and F._dense(x)[0] is x
    xl = x.contiguous(memory_format=torch.channels_last)
    assert F._dense(xl)[1] == "nhwc" and F._dense(xl)[0] is xl
    tok = torch.randn(4, 1 + 30, 8)
    view = tok[:, 1:].transpose(1, 2).unflatten(2, (5, 6))
    assert not view.is_contiguous(memory_format=torch.channels_last)
    v, lay = F._dense(view)
    assert v is view and lay == "nhwc"
    assert F._canonical_strides(v, lay) == (31 * 8, 1, 6 * 8, 8)
    every_other = torch.randn(8, 8, 5, 6)[::2]
    v, lay = F._dense(every_other)
    assert v is every_other and lay == "nchw" and F._canonical_strides(v, lay) == (2 * 240, 30, 6, 1)
    sliced = x[:, :, :, ::2]                       # not dense inside an image: copied
    v, lay = F._dense(sliced)
    assert v is not sliced and v.is_contiguous() and lay == "nchw"
    expanded = torch.randn(1, 8, 5, 6).expand(3, 8, 5, 6)   # batch stride 0: images overlap, copied
    assert F._dense(expanded)[0].is_contiguous() and F._dense(expanded)[0] is not expanded
    one = torch.randn(1, 1, 1, 1)                  # size-1 dimensions: arbitrary strides in torch, canonical here
    assert F._canonical_strides(*F._dense(one)) == (1, 1, 1, 1)
    d = F.make_desc(view, F.NfpConfig(R=1, measure="cosine", padding=1))
    assert (d.sxB, d.sxC, d.sxH, d.sxW, d.sgB) == (248, 1, 48, 8, 240)


def test_plan_cache_is_lru():
    from neighbour_feature_pooling_amd import functional as F
    F._PLANS.clear()
    for i in range(F._PLANS_MAX + 10):
        F._plans_put(("k", i), i)
        F._plans_get(("k", 0))                     # keep the first entry hot
    assert len(F._PLANS) == F._PLANS_MAX
    assert F._plans_get(("k", 0)) == 0 and F._plans_get(("k", 1)) is None
    F._PLANS.clear()


def test_multi_radius_module_on_cpu_is_the_concatenation():
    """MultiRadiusNFPPooling = torch.cat of one NFP layer per radius (nfp_heads.py:88-93,109-110)."""
    from neighbour_feature_pooling_amd import MultiRadiusNFPPooling
    m = MultiRadiusNFPPooling(16, R_list=(1, 2), measure="cosine")
    assert m.out_channels == 8 + 24 and len(m.nfp_blocks) == 2
    x = torch.randn(2, 16, 7, 7, requires_grad=True)
    y = m(x)
    ref = torch.cat([NFPPooling(16, R=1, measure="cosine", padding=1)(x), NFPPooling(16, R=2, measure="cosine", padding=2)(x)], 1)
    assert y.shape == (2, 32, 7, 7) and torch.equal(y, ref)
    y.sum().backward()
    assert x.grad is not None
    with torch.no_grad():   # the reference heads probe their NFP blocks with a CPU dummy in __init__ (nfp_heads.py:94-97)
        assert sum(b(torch.randn(1, 16, 7, 7)).shape[1] for b in m.nfp_blocks) == 32
    m3 = MultiRadiusNFPPooling(8, R_list=(1, 2, 3), measure="norm", p=2)
    assert m3(torch.randn(1, 8, 9, 9)).shape == (1, 8 + 24 + 48, 9, 9)
