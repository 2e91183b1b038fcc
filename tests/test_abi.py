"""C-ABI checks that need no GPU: the library loads, exports every symbol include/nfp.h
declares, and its host-side validation / shape arithmetic matches the reference's conv."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from neighbour_feature_pooling_amd import _abi
from neighbour_feature_pooling_amd.build import build_hip


@pytest.fixture(scope="module")
def lib():
    build_hip()
    return _abi.load()


def declared_functions():
    src = open(os.path.join(ROOT, "include", "nfp.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(nfp_[a-z_]+)\s*\(", src)))


def test_exports_every_declared_symbol(lib):
    names = declared_functions()
    assert len(names) >= 8
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/nfp.h but not exported"
    assert sorted(_abi.EXPORTS) == names


def test_abi_version_matches_header(lib):
    src = open(os.path.join(ROOT, "include", "nfp.h")).read()
    v = int(re.search(r"#define NFP_ABI_VERSION (\d+)", src).group(1))
    assert lib.nfp_abi_version() == v == _abi.ABI_VERSION


def test_desc_layout_matches_header():
    # 13 int32 + 3 float + 5 int64 + 1 pointer + 2 int32, naturally aligned
    assert ctypes.sizeof(_abi.NfpDesc) == 13 * 4 + 3 * 4 + 5 * 8 + 8 + 8
    assert _abi.NfpDesc.inner_R.offset == 112
    assert _abi.NfpDesc.sxB.offset == 64
    assert _abi.NfpDesc.sgB.offset == 96
    assert _abi.NfpDesc.ws.offset == 104


def _desc(shape, R=1, pad=1, stride=1, dil=1, mode="reflect", measure="cosine"):
    d = _abi.NfpDesc()
    d.B, d.C, d.H, d.W = shape
    d.R, d.pad, d.stride, d.dilation = R, pad, stride, dil
    d.pad_mode = _abi.PAD_MODES.index(mode)
    d.measure = _abi.measure_id(measure)
    d.similarity, d.diff_weights, d.dtype = 1, 0, 0
    d.p, d.eps, d.q_scs = 2.0, 1e-6, 1e-6
    B, C, H, W = shape
    d.sxB, d.sxC, d.sxH, d.sxW = C * H * W, H * W, W, 1
    return d


@pytest.mark.parametrize("shape,kw,expect", [
    ((64, 512, 7, 7), dict(), (8, 7, 7)),
    ((2, 64, 14, 14), dict(), (8, 14, 14)),
    ((8, 512, 2, 2), dict(), (8, 2, 2)),
    ((4, 192, 14, 14), dict(R=2, pad=2), (24, 14, 14)),
    ((1, 32, 5, 5), dict(pad=0), (8, 3, 3)),
    ((2, 16, 9, 9), dict(stride=2), (8, 5, 5)),
    ((2, 8, 11, 10), dict(pad=0, stride=2, dil=2), (8, 4, 3)),
    ((1, 8, 9, 9), dict(R=3, pad=3), (48, 9, 9)),
])
def test_output_shape(lib, shape, kw, expect):
    d = _desc(shape, **kw)
    n, ho, wo = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    assert lib.nfp_output_shape(ctypes.byref(d), ctypes.byref(n), ctypes.byref(ho), ctypes.byref(wo)) == 0
    assert (n.value, ho.value, wo.value) == expect


def test_invalid_descriptors_are_refused(lib):
    n = ctypes.c_int32()
    bad = [
        _desc((1, 4, 2, 2), pad=2),                 # reflect pad >= size (torch raises too)
        _desc((1, 4, 2, 2), R=2, pad=0),            # kernel larger than input
        _desc((1, 4, 5, 5), stride=0),
        _desc((1, 0, 5, 5)),
    ]
    for d in bad:
        rc = lib.nfp_output_shape(ctypes.byref(d), ctypes.byref(n), ctypes.byref(n), ctypes.byref(n))
        assert rc == -1
        assert lib.nfp_last_error()
    for p in (float("inf"), 0.0, -1.0, float("nan")):   # LA.norm orders without a kernel: refused, not mis-computed
        d = _desc((1, 4, 5, 5), measure="norm")
        d.p = p
        assert lib.nfp_output_shape(ctypes.byref(d), ctypes.byref(n), ctypes.byref(n), ctypes.byref(n)) == -2
        assert b"norm order" in lib.nfp_last_error()
    d = _desc((1, 4, 5, 5))
    d.measure = 99
    assert lib.nfp_forward(ctypes.byref(d), None, None, None, None) == -1
    d = _desc((1, 4, 5, 5))
    assert lib.nfp_forward(ctypes.byref(d), None, None, None, None) == -1  # null pointers
    assert b"null" in lib.nfp_last_error()


def test_saved_floats(lib):
    assert lib.nfp_saved_floats(ctypes.byref(_desc((64, 512, 7, 7)))) == 64 * 49
    assert lib.nfp_saved_floats(ctypes.byref(_desc((4, 192, 14, 14), R=2, pad=2, measure="norm"))) == 0


def test_plan_is_reentrant_and_last_variant_is_stable(lib):
    """include/nfp.h promises re-entrancy: two threads plan different descriptors in a loop while a third reads
    nfp_last_variant; every plan must describe ITS descriptor and the published variant must never change (nfp_plan
    publishes nothing) or be torn."""
    import threading
    before = lib.nfp_last_variant()
    want = {"a": (_desc((64, 512, 7, 7)), b"fwd_band<R1,cos,f32,nchw>"),
            "b": (_desc((4, 192, 14, 14), R=2, pad=2, measure="norm"), b"fwd_band<R2,l2,f32,nchw>")}
    want["b"][0].diff_weights = 1
    errors, stop = [], threading.Event()

    def planner(key):
        d, prefix = want[key]
        buf = ctypes.create_string_buffer(1024)
        for _ in range(3000):
            rc = lib.nfp_plan(ctypes.byref(d), 0, buf, len(buf))
            if rc != 0 or not buf.value.startswith(prefix):
                errors.append((key, rc, buf.value))
                return

    def reader():
        while not stop.is_set():
            v = lib.nfp_last_variant()
            if v != before:
                errors.append(("reader", v))
                return

    ts = [threading.Thread(target=planner, args=(k,)) for k in want] + [threading.Thread(target=reader)]
    for t in ts:
        t.start()
    for t in ts[:2]:
        t.join()
    stop.set()
    ts[2].join()
    assert not errors, errors[:3]


def test_launch_path_never_reads_the_environment():
    """The A/B switches are read when the library is loaded (and by nfp_reload_env), not per launch: `getenv` appears in
    read_env (csrc/nfp_launch.h) and nowhere else in the library's sources."""
    csrc = os.path.join(ROOT, "neighbour_feature_pooling_amd", "csrc")
    for name in os.listdir(csrc):
        src = open(os.path.join(csrc, name)).read()
        if name == "nfp_launch.h":
            body = src[src.index("void read_env()"):]
            src = src[:src.index("void read_env()")] + body[body.index("\n}\n") + 3:]
        assert "getenv" not in src, name


def test_env_switches_take_effect_only_through_reload(lib, monkeypatch):
    d = _desc((64, 512, 7, 7))
    buf = ctypes.create_string_buffer(1024)
    monkeypatch.setenv("NFP_FORCE_GENERIC", "1")
    assert lib.nfp_plan(ctypes.byref(d), 0, buf, len(buf)) == 0 and buf.value.startswith(b"fwd_band")
    lib.nfp_reload_env()
    assert lib.nfp_plan(ctypes.byref(d), 0, buf, len(buf)) == 0 and buf.value.startswith(b"fwd_pairs")
    monkeypatch.delenv("NFP_FORCE_GENERIC")
    lib.nfp_reload_env()
    assert lib.nfp_plan(ctypes.byref(d), 0, buf, len(buf)) == 0 and buf.value.startswith(b"fwd_band")


def test_workspace_bytes(lib):
    """Constant tables exist for the hot-path geometry (stride 1, pad = R, R <= 2, map <= 512 pixels) only; "same" maps
    above it get the pooled tail's arrival counters alone (ABI 6)."""
    assert lib.nfp_workspace_bytes(ctypes.byref(_desc((64, 512, 7, 7)))) > _abi.TICKET_BYTES
    assert lib.nfp_workspace_bytes(ctypes.byref(_desc((4, 192, 14, 14), R=2, pad=2, measure="norm"))) > 0
    assert lib.nfp_workspace_bytes(ctypes.byref(_desc((4, 192, 14, 14), R=2, pad=2, mode="replicate"))) > 0
    assert lib.nfp_workspace_bytes(ctypes.byref(_desc((2, 16, 9, 9), stride=2))) == 0
    assert lib.nfp_workspace_bytes(ctypes.byref(_desc((2, 16, 9, 9), pad=0))) == 0
    assert lib.nfp_workspace_bytes(ctypes.byref(_desc((2, 16, 40, 40)))) == _abi.TICKET_BYTES
    assert lib.nfp_workspace_bytes(ctypes.byref(_desc((2, 16, 40, 40), stride=2))) == 0
    assert lib.nfp_workspace_bytes(ctypes.byref(_desc((2, 16, 9, 9), mode="circular"))) == 0
    assert lib.nfp_workspace_bytes(ctypes.byref(_desc((1, 0, 5, 5)))) == -1
    # the same tables whatever B, C, measure: one buffer per geometry
    a = lib.nfp_workspace_bytes(ctypes.byref(_desc((1, 8, 7, 7))))
    assert a == lib.nfp_workspace_bytes(ctypes.byref(_desc((64, 512, 7, 7), measure="norm")))
    d = _desc((64, 512, 7, 7))
    assert lib.nfp_workspace_init(ctypes.byref(d), None, None) == -1


def test_cpp_autograd_module_builds_and_matches_the_descriptor(lib):
    """csrc/nfp_torch.cpp (C++ autograd nodes over the C ABI) builds against this torch, loads, exposes its two nodes
    and agrees with the ctypes binding on sizeof(nfp_desc); no compute without a GPU."""
    from neighbour_feature_pooling_amd import functional
    from neighbour_feature_pooling_amd.build import build_torch_ext
    build_torch_ext()
    functional._CPP = None
    cpp = functional._cpp_nodes()
    assert cpp and cpp.desc_bytes == ctypes.sizeof(_abi.NfpDesc)
    assert callable(cpp.nfp_apply) and callable(cpp.nfp_pool_apply)
