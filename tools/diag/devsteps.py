"""Device-resident steps of a bench workload (no torch): ms per step, error vs the closed form, event-timed kernels.
usage: devsteps.py [C2|C3|C5] [steps]   (run under rocprofv3 --kernel-trace --stats for per-kernel durations)"""
import os, sys, time
import numpy as np
if os.environ.get("SK_TORCH_FIRST"):  # torch's bundled HIP / rocFFT serve the library then (bench.py at N > 1)
    import torch
    torch.cuda.init()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import spectralkernels_jl_amd as sk
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
w = bench.WORKLOADS[wl]
kind, prm = w["sdf"]
if prm is None:
    from scipy import special
    q = (1 - 0.4) / 2
    prm = (1.0 / (10.0 ** (1 - 0.4 - 2 * 1.0) * special.beta(q, 1.0 - q)), 10.0, 0.5)
cfg = sk.AdaptiveKernelConfig(getattr(sk.SDF, kind)(*prm), batch_panels=int(os.environ.get("SK_BATCH", "0")), **w["kw"])
k0 = 1.0 if wl == "C3" else sk.compute_k0(cfg)
rs = bench.shard_grid(w, 0, 1)
n = rs.size
d_rs, d_ks, d_es = sk.DeviceArray.from_host(rs), sk.DeviceArray(n), sk.DeviceArray(n)
step = lambda: sk.kernel_values_device(cfg, d_rs.ptr, n, k0, d_ks.ptr, d_es.ptr)
for _ in range(5):
    step()
sk.synchronize()
t = time.perf_counter()
for _ in range(steps):
    step()
sk.synchronize()
dt = (time.perf_counter() - t) / steps
ks = d_ks.to_host()
err = None
if wl == "C2":
    err = np.abs(ks - k0 * (1 + 2 * np.pi * rs) * np.exp(-2 * np.pi * rs)).max() / k0
st = cfg.last_stats()
cfg.profile(True)
acc = {}
for _ in range(10):
    step()
    for k, v in cfg.kernel_times().items():
        a = acc.setdefault(k, [0.0, 0])
        a[0] += v["total_ms"]; a[1] += v["launches"]
print("%s SK_BATCH=%s: %.4f ms/step, err %s, panels %d, events(us): %s, checksum %.17g" % (
    wl, os.environ.get("SK_BATCH", "auto"), 1e3 * dt, err, st["n_panels"],
    {k: round(1e3 * a[0] / a[1], 2) for k, a in acc.items() if a[1]}, float(ks.sum())))
