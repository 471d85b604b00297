"""Builds libspectralkernels_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

Each source is compiled to an object under spectralkernels.jl_amd/build/ (rebuilt when it or a header is newer),
then linked; `--force` rebuilds everything."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
SOURCES = ["sk_api.cpp", "sk_hostio.cpp", "sk_shard.cpp", "sk_joint.cpp", "sk_kernels.hip", "sk_rules.cpp", "sk_tables.cpp", "sk_k0.cpp"]
HEADERS = ["sk_device.h", "sk_internal.h", "sk_config.h", "sk_joint_kernels.inc", "sk_batch_kernels.inc", "sk_hankel_inner.inc", "sk_lean_math.h", os.path.join("..", "..", "include", "spectralkernels_hip.h")]
LIB = os.path.join(HERE, "libspectralkernels_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fopenmp", "-munsafe-fp-atomics", "-Wno-unused-value",
         "-Wno-pass-failed"]


def _mtime(p):
    return os.path.getmtime(p) if os.path.exists(p) else 0.0


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    srcs = [f for f in SOURCES if os.path.exists(os.path.join(CSRC, f))]
    hdr_t = max(_mtime(os.path.join(CSRC, h)) for h in HEADERS)
    jobs = []
    objs = []
    for f in srcs:
        src = os.path.join(CSRC, f)
        obj = os.path.join(OBJ, os.path.splitext(f)[0] + ".o")
        objs.append(obj)
        if force or _mtime(obj) < max(_mtime(src), hdr_t):
            jobs.append([hipcc] + FLAGS + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    if jobs:
        with ThreadPoolExecutor(max_workers=4) as ex:
            list(ex.map(run, jobs))
    if jobs or force or _mtime(LIB) < max(_mtime(o) for o in objs):
        run([hipcc, "--offload-arch=gfx950", "-fPIC", "-fopenmp", "-shared", "-o", LIB] + objs + ["-lrocfft"])
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)


def build_stamps(verbose=False):
    """Diagnostic build with in-kernel s_memtime stamps (tools/diag/stamps*.py): tools/diag/libsk_stamps.so."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    out = os.path.join(os.path.dirname(HERE), "tools", "diag", "libsk_stamps.so")
    cmd = [hipcc] + FLAGS + ["-DSK_STAMPS", "-shared", "-o", out] + [os.path.join(CSRC, f) for f in SOURCES] + ["-lrocfft"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


def build_variant(flags, name="libsk_var.so", verbose=False):
    """Diagnostic build with extra -D flags for same-box A/B runs (tools/diag/ab_libs.sh): tools/diag/<name>."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    out = os.path.join(os.path.dirname(HERE), "tools", "diag", name)
    cmd = [hipcc] + FLAGS + list(flags) + ["-shared", "-o", out] + [os.path.join(CSRC, f) for f in SOURCES] + ["-lrocfft"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out
