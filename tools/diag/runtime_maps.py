"""Which HIP / HSA / rocFFT runtime serves the library in this process?  python tools/diag/runtime_maps.py [torch-first]
Prints the mapped paths of libamdhip64 / libhsa-runtime64 / librocfft / libspectralkernels_hip after one small call."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 1 and sys.argv[1] == "torch-first":
    import torch  # noqa: F401
    torch.cuda.is_available()
import spectralkernels_jl_amd as sk
cfg = sk.AdaptiveKernelConfig(sk.SDF.matern(1, 1, 1.5), quadspec=(256, 16))
ks, _ = sk.kernel_values(cfg, 10.0 ** np.linspace(-4, 0, 1000), k0=np.pi / 2)
seen = set()
for line in open("/proc/self/maps"):
    path = line.split()[-1]
    if any(k in path for k in ("libamdhip64", "libhsa-runtime64", "librocfft", "libspectralkernels_hip", "libomp", "libgomp")):
        seen.add(os.path.realpath(path))
print("mode:", "torch imported first" if len(sys.argv) > 1 else "library only", "| result ok:", bool(np.isfinite(ks).all()))
for p in sorted(seen):
    print("  ", p)
