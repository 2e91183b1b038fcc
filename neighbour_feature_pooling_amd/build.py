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
SOURCES = ["nfp_hip.hip"]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]


def _newest_source_mtime():
    m = os.path.getmtime(os.path.join(_HERE, "..", "include", "nfp.h"))
    for f in os.listdir(CSRC):
        m = max(m, os.path.getmtime(os.path.join(CSRC, f)))
    return m


def hipcc_path():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def build_hip(force=False, verbose=False):
    """Compile csrc/*.hip -> libnfp_hip.so.  Returns the library path."""
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= _newest_source_mtime():
        return LIB
    cmd = [hipcc_path()] + HIPCC_FLAGS + ["-o", LIB + ".tmp"] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build_hip(force=True, verbose=True))
