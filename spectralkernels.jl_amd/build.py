"""Builds libspectralkernels_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["sk_api.cpp", "sk_kernels.hip", "sk_rules.cpp", "sk_tables.cpp", "sk_k0.cpp"]
HEADERS = ["sk_device.h", "sk_internal.h", os.path.join("..", "..", "include", "spectralkernels_hip.h")]
LIB = os.path.join(HERE, "libspectralkernels_hip.so")


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fopenmp", "-shared",
           "-munsafe-fp-atomics", "-Wno-unused-value", "-o", LIB] + [os.path.join(CSRC, f) for f in SOURCES] + ["-lrocfft"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
