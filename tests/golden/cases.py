"""Golden-vector case list (pure data).

Shared by make_golden.py (which runs the REAL reference on each case, in the
build container only) and by the tests (which regenerate the same inputs from
(shape, seed, kind, tweak) and compare against the stored reference outputs).

ctor = keyword arguments of NFPPooling(in_channels=C, **ctor)  (nfp.py:16-18).
"""
import numpy as np

FULL_GX_LIMIT = 131072  # store grad_x (and out) in full up to this many elements, else a strided sample
SAMPLE_STRIDE = 97


def case(name, shape, ctor, seed, kind="normal", tweak=None, go_kind="normal", full_limit=FULL_GX_LIMIT):
    return dict(name=name, shape=tuple(shape), ctor=dict(ctor), seed=seed, kind=kind, tweak=tweak,
                go_kind=go_kind, full_limit=full_limit)


COS = dict(R=1, measure="cosine", padding=1)
L2K5 = dict(R=2, measure="norm", p=2, padding=2)

CASES = [
    # --- BASELINE.json configs (SURVEY.md §8c2 ①-⑤) -------------------------------------
    case("c1_cos_k3_2x64x14x14", (2, 64, 14, 14), COS, 11),
    case("c2_cos_k3_4x512x7x7", (4, 512, 7, 7), COS, 12),
    case("c2_cos_k3_64x512x7x7_full", (64, 512, 7, 7), COS, 13),
    case("c2_cos_k3_relu_8x512x7x7", (8, 512, 7, 7), COS, 14, kind="relu"),
    case("c3_cos_k3_8x512x2x2", (8, 512, 2, 2), COS, 15),
    case("c5_l2_k5_4x192x14x14", (4, 192, 14, 14), L2K5, 16),
    case("c5_l2_k5_bf16in_4x192x14x14", (4, 192, 14, 14), L2K5, 16, tweak="bf16_round"),
    case("c5_cos_k5_2x192x14x14", (2, 192, 14, 14), dict(R=2, measure="cosine", padding=2), 17),
    case("cos_k5_selfpairs_2x24x5x5", (2, 24, 5, 5), dict(R=2, measure="cosine", padding=2), 18),
    case("cos_k5_selfpairs_1x8x3x3", (1, 8, 3, 3), dict(R=2, measure="cosine", padding=2), 19),
    case("l2_k5_selfpairs_1x8x3x3", (1, 8, 3, 3), L2K5, 19),
    case("pad0_cos_2x16x9x9", (2, 16, 9, 9), dict(R=1, measure="cosine", padding=0), 20),
    case("pad0_cos_probe_1x32x5x5", (1, 32, 5, 5), dict(R=1, measure="cosine", padding=0), 21),
    case("pad0_l2_2x16x9x9", (2, 16, 9, 9), dict(R=1, measure="norm", p=2, padding=0), 22),
    # --- edge cases (§8c2 ⑥) -------------------------------------------------------------
    case("edge_cos_zero_pixel", (2, 16, 5, 5), COS, 30, tweak="zero_pixel"),
    case("edge_cos_tiny_norms", (2, 16, 5, 5), COS, 31, tweak="tiny_norms"),
    case("edge_cos_relu_sparse", (2, 8, 6, 6), COS, 32, kind="relu"),
    case("edge_l2_identical_neighbours", (2, 16, 5, 5), dict(R=1, measure="norm", p=2, padding=1), 33,
         tweak="identical"),
    case("edge_cos_dissimilarity", (2, 32, 7, 7), dict(R=1, measure="cosine", padding=1, similarity=False), 34),
    case("edge_l2_dissimilarity", (2, 32, 7, 7), dict(R=1, measure="norm", p=2, padding=1, similarity=False), 35),
    case("edge_cos_eps_large", (2, 16, 5, 5), dict(R=1, measure="cosine", padding=1, eps=3.0), 36),
    # --- general geometry (§8f4) ---------------------------------------------------------
    case("geo_cos_stride2", (2, 16, 9, 9), dict(R=1, measure="cosine", padding=1, stride=2), 40),
    case("geo_cos_dil2", (2, 8, 9, 9), dict(R=1, measure="cosine", padding=2, dilation=2), 41),
    case("geo_cos_zeros", (2, 8, 6, 5), dict(R=1, measure="cosine", padding=1, padding_mode="zeros"), 42),
    case("geo_cos_replicate", (2, 8, 6, 5), dict(R=1, measure="cosine", padding=1, padding_mode="replicate"), 43),
    case("geo_cos_circular", (2, 8, 6, 5), dict(R=1, measure="cosine", padding=1, padding_mode="circular"), 44),
    case("geo_cos_nonsquare", (3, 12, 5, 8), COS, 45),
    case("geo_cos_pad2_R1", (2, 8, 6, 6), dict(R=1, measure="cosine", padding=2), 46),
    case("geo_cos_stride2_dil2_pad0", (2, 8, 11, 10), dict(R=1, measure="cosine", padding=0, stride=2, dilation=2), 47),
    case("geo_l2_zeros", (2, 8, 6, 5), dict(R=1, measure="norm", p=2, padding=1, padding_mode="zeros"), 48),
    case("geo_l2_stride2_replicate", (2, 8, 7, 9), dict(R=1, measure="norm", p=2, padding=1, stride=2,
                                                     padding_mode="replicate"), 49),
    case("geo_cos_k7", (1, 8, 9, 9), dict(R=3, measure="cosine", padding=3), 50),
    case("geo_cos_big_map", (1, 16, 40, 36), COS, 51),
    # the five feature maps MobileNetV3_MultiStageNFP feeds NFP(R=1, cosine, padding=1) at a 224x224 input
    # (texture_pooling.py:211-268; mobilenetv3_large_100 features_only: 16, 24, 40, 112, 960 channels)
    case("ms_cos_112x112x16", (1, 16, 112, 112), COS, 52, kind="relu", full_limit=16384),
    case("ms_cos_56x56x24", (1, 24, 56, 56), COS, 53, kind="relu", full_limit=16384),
    case("ms_cos_28x28x40", (2, 40, 28, 28), COS, 54, kind="relu", full_limit=16384),
    case("ms_cos_14x14x112", (2, 112, 14, 14), COS, 55, kind="relu", full_limit=16384),
    case("ms_cos_7x7x960", (2, 960, 7, 7), COS, 56, kind="relu", full_limit=16384),
    # the reference's default input_size: a 224x224 map (beyond one LDS channel slab)
    case("geo_cos_224x224", (1, 4, 224, 224), COS, 57, full_limit=16384),
    case("geo_l2_k5_160x160", (1, 4, 160, 160), L2K5, 58, full_limit=16384),
    # --- Norm variants (nfp.py:141-148, quirk at nfp.py:74) -------------------------------
    case("norm_p1_default", (2, 16, 7, 7), dict(R=1, measure="norm", padding=1), 60),
    case("norm_p3", (2, 16, 7, 7), dict(R=1, measure="norm", p=3, padding=1), 61),
    case("norm_capitalised_quirk", (2, 16, 7, 7), dict(R=1, measure="Norm", p=2, padding=1), 62),
    case("rmse", (2, 16, 7, 7), dict(R=1, measure="rmse", padding=1), 63),
    case("rmse_dissim", (2, 16, 7, 7), dict(R=1, measure="rmse", padding=1, similarity=False), 64),
]

# --- every other measure, small, both conventions -------------------------
for _i, _m in enumerate(["dot", "geman", "attention", "emd", "canberra", "hellinger", "chisquared1",
                         "chisquared2", "gfc", "pearson", "jeffrey", "squaredchord", "smith"]):
    CASES.append(case(f"m_{_m}", (2, 12, 5, 6), dict(R=1, measure=_m, padding=1), 100 + _i))
    CASES.append(case(f"m_{_m}_dissim_zeros", (2, 12, 5, 6),
                      dict(R=1, measure=_m, padding=1, similarity=False, padding_mode="zeros"), 130 + _i,
                      kind="relu" if _m in ("hellinger", "squaredchord", "jeffrey", "smith") else "normal"))
CASES.append(case("m_scs_p2", (3, 6, 4, 4), dict(R=1, measure="scs", p=2, padding=1), 160))
CASES.append(case("m_scs_p1_dissim", (2, 6, 4, 4), dict(R=1, measure="sharpened_cosine", p=1, padding=1,
                                                       similarity=False), 161))

# --- round 4: the row-band kernels (csrc/nfp_tile.h, maps above 512 pixels) in front of the reference ----------------------
# RESNET18_NFP_AT_LAYER's maps (models/resnet18.py:410-468: layer1 64 x 56 x 56, layer2 128 x 28 x 28), k = 5 L2 on the
# largest MultiStage map, the other padding modes, the other measures the row-band kernels serve, and the class default
# (Norm p = 1, nfp.py:16) / EMD on a large map.
CASES += [
    case("rn_cos_56x56x64", (2, 64, 56, 56), COS, 70, kind="relu", full_limit=16384),
    case("rn_cos_28x28x128", (2, 128, 28, 28), COS, 71, kind="relu", full_limit=16384),
    case("tile_l2_k5_112x112x16", (1, 16, 112, 112), L2K5, 72, full_limit=16384),
    case("tile_cos_k5_40x40x24", (2, 24, 40, 40), dict(R=2, measure="cosine", padding=2), 73, full_limit=16384),
    case("tile_cos_zeros_56x56x24", (2, 24, 56, 56), dict(R=1, measure="cosine", padding=1, padding_mode="zeros"), 74,
         full_limit=16384),
    case("tile_cos_replicate_56x56x24", (2, 24, 56, 56), dict(R=1, measure="cosine", padding=1, padding_mode="replicate"),
         75, full_limit=16384),
    case("tile_l2_replicate_k5_30x37x8", (1, 8, 30, 37), dict(R=2, measure="norm", p=2, padding=2, padding_mode="replicate"),
         76),
    case("tile_rmse_40x40x16", (2, 16, 40, 40), dict(R=1, measure="rmse", padding=1), 77, full_limit=16384),
    case("tile_gfc_40x40x16", (2, 16, 40, 40), dict(R=1, measure="gfc", padding=1), 78, full_limit=16384),
    case("tile_dot_40x40x16", (2, 16, 40, 40), dict(R=1, measure="dot", padding=1), 79, full_limit=16384),
    case("tile_norm_p1_40x40x16", (2, 16, 40, 40), dict(R=1, measure="norm", padding=1), 80, full_limit=16384),
    case("tile_norm_p1_k5_zeros_30x37x8", (1, 8, 30, 37), dict(R=2, measure="norm", padding=2, padding_mode="zeros"), 81),
    case("tile_emd_dissim_40x40x16", (2, 16, 40, 40), dict(R=1, measure="emd", padding=1, similarity=False), 82,
         full_limit=16384),
    case("tile_norm_p1_quirk_40x40x8", (1, 8, 40, 40), dict(R=1, measure="Norm", padding=1), 83, full_limit=16384),
    # Geman-McClure / Canberra / squared chord / chi-squared 1 (nfp.py:181-193, 218-227, 310-324, 243-252): one shared
    # instantiation of the row-band kernels (csrc/nfp_measures.h::kSymTerm)
    case("tile_canberra_40x40x16", (2, 16, 40, 40), dict(R=1, measure="canberra", padding=1), 84, full_limit=16384),
    case("tile_geman_k5_zeros_dissim_30x37x8", (1, 8, 30, 37),
         dict(R=2, measure="geman", padding=2, padding_mode="zeros", similarity=False), 85),
    case("tile_sqchord_replicate_40x40x8", (2, 8, 40, 40), dict(R=1, measure="squaredchord", padding=1, padding_mode="replicate"),
         86, kind="relu"),
    case("tile_chisq1_k5_56x56x24", (1, 24, 56, 56), dict(R=2, measure="chisquared1", padding=2), 87, full_limit=16384),
    case("tile_hellinger_40x40x16", (2, 16, 40, 40), dict(R=1, measure="hellinger", padding=1), 89, kind="relu", full_limit=16384),
    case("tile_jeffrey_k5_zeros_30x37x8", (1, 8, 30, 37), dict(R=2, measure="jeffrey", padding=2, padding_mode="zeros"), 90),
]

# --- pooled NFP: adaptive_avg_pool2d(NFPPooling(feat), 1) and its input gradient, what MobileNetV3_MultiStageNFP /
# MidNFP consume (models/texture_pooling.py:251-252, 320-321).  Fixture: "nfpm" [B,N] and grad_x (sampled) for the
# output gradient make_pool_grad gives.
POOL_CASES = [
    case("pool_ms_cos_112x112x16", (1, 16, 112, 112), COS, 90, kind="relu", full_limit=16384),
    case("pool_ms_cos_56x56x24", (2, 24, 56, 56), COS, 91, kind="relu", full_limit=16384),
    case("pool_ms_cos_14x14x112", (2, 112, 14, 14), COS, 92, kind="relu", full_limit=16384),
    case("pool_l2_k5_28x28x40", (2, 40, 28, 28), L2K5, 93, full_limit=16384),
]
POOL_BY_NAME = {c["name"]: c for c in POOL_CASES}


def make_pool_grad(c, n_maps):
    from neighbour_feature_pooling_amd.synth import feature_map
    return feature_map((c["shape"][0], n_maps), c["seed"] + 2000, "normal")


BY_NAME = {c["name"]: c for c in CASES}
assert len(BY_NAME) == len(CASES)


def _bf16_round(a):
    u = a.astype(np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def make_input(c):
    """Regenerate the case's input feature map (float32, NCHW) bit-exactly."""
    from neighbour_feature_pooling_amd.synth import feature_map
    x = feature_map(c["shape"], c["seed"], c["kind"])
    t = c["tweak"]
    if t == "zero_pixel":
        x[0, :, 2, 2] = 0.0
        x[1, :, 0, 0] = 0.0
    elif t == "tiny_norms":
        x[0, :, 1, 1] *= 1e-8      # norm far below eps -> clamp active
        x[0, :, 3, 2] *= 3e-7      # norm just around eps=1e-6
        x[1, :, 4, 4] *= 1e-7
    elif t == "identical":
        x[0, :, 2, 3] = x[0, :, 2, 2]
        x[1, :, 0, 1] = x[1, :, 0, 0]
        x[1, :, 1, 0] = x[1, :, 0, 0]
    elif t == "bf16_round":
        x = _bf16_round(x)
    elif t is not None:
        raise ValueError(t)
    return x


def make_grad_out(c, out_shape):
    from neighbour_feature_pooling_amd.synth import feature_map
    return feature_map(out_shape, c["seed"] + 1000, c["go_kind"])


def gx_sample_index(n):
    return np.arange(0, n, SAMPLE_STRIDE)
