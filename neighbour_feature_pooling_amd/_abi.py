"""ctypes binding of libnfp_hip.so — the C ABI declared in include/nfp.h.

This is the whole boundary between Python and the HIP kernels: plain pointers,
sizes and one descriptor struct.  There is no CPU implementation behind it; if
the library is missing or a call fails, an exception is raised.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libnfp_hip.so")
ABI_VERSION = 6

MEASURES = ["norm", "cosine", "dot", "rmse", "geman", "attention", "emd", "canberra", "hellinger",
            "chisquared1", "chisquared2", "gfc", "pearson", "jeffrey", "squaredchord", "smith", "scs"]
MEASURE_ALIASES = {"sharpened_cosine": "scs"}
PAD_MODES = ["zeros", "reflect", "replicate", "circular"]
F32, BF16 = 0, 1
TICKET_BYTES = 4096 * 4       # the arrival counters at the head of a workspace (csrc/nfp_common.h: kTicketBytes)

EXPORTS = ["nfp_abi_version", "nfp_last_error", "nfp_output_shape", "nfp_saved_floats", "nfp_forward",
           "nfp_backward", "nfp_pool_supported", "nfp_pool_saved_floats", "nfp_pool_forward", "nfp_pool_backward", "nfp_launch_count",
           "nfp_last_variant", "nfp_plan", "nfp_reload_env", "nfp_workspace_bytes", "nfp_workspace_init", "nfp_time_next_launch"]


class NfpDesc(ctypes.Structure):
    """struct nfp_desc (include/nfp.h)."""
    _fields_ = [(n, ctypes.c_int32) for n in
                ("B", "C", "H", "W", "R", "pad", "stride", "dilation", "pad_mode", "measure",
                 "similarity", "diff_weights", "dtype")] + \
               [("p", ctypes.c_float), ("eps", ctypes.c_float), ("q_scs", ctypes.c_float)] + \
               [(n, ctypes.c_int64) for n in ("sxB", "sxC", "sxH", "sxW", "sgB")] + [("ws", ctypes.c_void_p), ("inner_R", ctypes.c_int32), ("reserved_", ctypes.c_int32)]


class NfpError(RuntimeError):
    pass


class NfpUnsupported(NfpError, NotImplementedError):
    pass


_lib = None


def load():
    """Load libnfp_hip.so (once).  Raises NfpError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NfpError(
            f"{LIB_PATH} is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `python -m neighbour_feature_pooling_amd.build`). There is no fallback for GPU tensors.")
    L = ctypes.CDLL(LIB_PATH)
    vp, dp = ctypes.c_void_p, ctypes.POINTER(NfpDesc)
    i32p = ctypes.POINTER(ctypes.c_int32)
    L.nfp_abi_version.restype = ctypes.c_int
    L.nfp_last_error.restype = ctypes.c_char_p
    L.nfp_last_variant.restype = ctypes.c_char_p
    L.nfp_launch_count.restype = ctypes.c_uint64
    L.nfp_reload_env.restype = None
    L.nfp_time_next_launch.argtypes = [vp, vp]
    L.nfp_time_next_launch.restype = None
    L.nfp_plan.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_char_p, ctypes.c_int32]
    L.nfp_plan.restype = ctypes.c_int
    L.nfp_output_shape.argtypes = [dp, i32p, i32p, i32p]
    L.nfp_saved_floats.argtypes = [dp]
    L.nfp_saved_floats.restype = ctypes.c_int64
    L.nfp_workspace_bytes.argtypes = [dp]
    L.nfp_workspace_bytes.restype = ctypes.c_int64
    L.nfp_workspace_init.argtypes = [dp, vp, vp]
    L.nfp_forward.argtypes = [dp, vp, vp, vp, vp]
    L.nfp_backward.argtypes = [dp, vp, vp, vp, vp, vp, vp]
    L.nfp_pool_supported.argtypes = [dp]
    if hasattr(L, "nfp_pool_saved_floats"):     # (absent only from an older library loaded for an A/B run)
        L.nfp_pool_saved_floats.argtypes = [dp]
        L.nfp_pool_saved_floats.restype = ctypes.c_int64
    L.nfp_pool_forward.argtypes = [dp, vp, vp, vp, vp, vp, vp]
    L.nfp_pool_backward.argtypes = [dp, vp, vp, vp, vp, vp, vp, vp]
    if L.nfp_abi_version() != ABI_VERSION:
        raise NfpError(f"libnfp_hip.so ABI {L.nfp_abi_version()} != binding {ABI_VERSION}; rebuild")
    _lib = L
    return L


def check(rc):
    if rc == 0:
        return
    msg = load().nfp_last_error().decode()
    if rc == -2:
        raise NfpUnsupported(f"libnfp_hip: {msg}")
    raise NfpError(f"libnfp_hip rc={rc}: {msg}")


def measure_id(name):
    name = MEASURE_ALIASES.get(name, name)
    return MEASURES.index(name)
