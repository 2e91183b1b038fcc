"""Build libnfp_hip.so (hipcc, gfx950) in-tree, next to this file.

hipcc cross-compiles without a GPU; the built .so is git-ignored but travels to
the GPU box with the source snapshot.
"""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(_HERE, "libnfp_hip.so")
SOURCES = ["nfp_hip.hip", "nfp_tile.hip"]   # two translation units, compiled in parallel
# -fno-slp-vectorize: left to itself hipcc packs adjacent scalar f32 FMAs of the channel loops into v_pk_fma_f32,
# which costs more issue time than it saves at two wavefronts per SIMD (headline forward 5.31 -> 5.10 us,
# [256,512,7,7] forward 7.8 -> 7.5 us; scripts/ab_flags.py)
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize",
               "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]


def _newest_source_mtime():
    m = os.path.getmtime(os.path.join(_HERE, "..", "include", "nfp.h"))
    for f in os.listdir(CSRC):
        m = max(m, os.path.getmtime(os.path.join(CSRC, f)))
    return m


def hipcc_path():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def compile_hip(out_lib, extra_flags=(), verbose=False):
    """hipcc: every source of SOURCES to an object (in parallel), then one shared library."""
    import concurrent.futures as cf
    import tempfile
    with tempfile.TemporaryDirectory(prefix="nfp_build_") as tmp:
        def one(src):
            obj = os.path.join(tmp, src + ".o")
            cmd = [hipcc_path()] + HIPCC_FLAGS + list(extra_flags) + ["-c", "-o", obj, os.path.join(CSRC, src)]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
            return obj
        with cf.ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
            objs = list(ex.map(one, SOURCES))
        cmd = [hipcc_path(), "--offload-arch=gfx950", "-fPIC", "-shared", "-fno-gpu-rdc", "-o", out_lib] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return out_lib


# Test build of the same sources with -DNFP_LDS_POISON (csrc/nfp_tile.h::lds_poison): the row-band kernels start on an LDS
# full of signalling NaNs.  tests/test_gpu_tile.py runs its oracle / golden cases once more on this library.
LIB_POISON = os.path.join(_HERE, "libnfp_hip_poison.so")


def build_hip(force=False, verbose=False, poison=True):
    """Compile csrc/*.hip -> libnfp_hip.so (and, side by side, the LDS-poison test build).  Returns the library path."""
    import concurrent.futures as cf
    newest = _newest_source_mtime()
    jobs = []
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < newest:
        jobs.append((LIB, ()))
    if poison and (force or not os.path.exists(LIB_POISON) or os.path.getmtime(LIB_POISON) < newest):
        jobs.append((LIB_POISON, ("-DNFP_LDS_POISON",)))

    def one(job):
        lib, flags = job
        compile_hip(lib + ".tmp", extra_flags=flags, verbose=verbose)
        os.replace(lib + ".tmp", lib)
    with cf.ThreadPoolExecutor(max_workers=2) as ex:
        list(ex.map(one, jobs))
    return LIB


TORCH_EXT = os.path.join(_HERE, "_nfp_torch.so")


def build_torch_ext(force=False, verbose=False):
    """Compile csrc/nfp_torch.cpp (C++ autograd nodes over the C ABI; host code only) -> _nfp_torch.so next to
    libnfp_hip.so, with g++ against libtorch.  Returns the module path."""
    import sysconfig
    import torch
    from torch.utils import cpp_extension as ce
    src = os.path.join(CSRC, "nfp_torch.cpp")
    newest = max(os.path.getmtime(src), os.path.getmtime(os.path.join(_HERE, "..", "include", "nfp.h")))
    if not force and os.path.exists(TORCH_EXT) and os.path.getmtime(TORCH_EXT) >= newest:
        return TORCH_EXT
    build_hip()
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           "-DTORCH_EXTENSION_NAME=_nfp_torch", "-DTORCH_API_INCLUDE_EXTENSION_H",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", "-Wno-deprecated-declarations"]
    cmd += [f"-I{p}" for p in ce.include_paths()] + [f"-I{rocm}/include", f"-I{sysconfig.get_paths()['include']}"]
    cmd += [src, "-o", TORCH_EXT + ".tmp"]
    for lp in ce.library_paths():
        cmd += [f"-L{lp}", f"-Wl,-rpath,{lp}"]
    cmd += ["-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch", "-ltorch_hip", "-ltorch_python",
            f"-L{_HERE}", "-lnfp_hip", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(TORCH_EXT + ".tmp", TORCH_EXT)
    return TORCH_EXT


if __name__ == "__main__":
    print(build_hip(force=True, verbose=True))
    print(build_torch_ext(force=True, verbose=True))
