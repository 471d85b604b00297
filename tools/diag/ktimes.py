"""Per-kernel HIP-event times of one bench workload (diagnostics): python tools/diag/ktimes.py [C2|C3|C5] [reps]"""
import sys, os, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
import spectralkernels_jl_amd as sk

wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
w = bench.WORKLOADS[wl]
kind, prm = w["sdf"]
if prm is None:
    from scipy import special
    q = (1 - 0.4) / 2
    prm = (1.0 / (10.0 ** (1 - 0.4 - 2 * 1.0) * special.beta(q, 1.0 - q)), 10.0, 0.5)
cfg = sk.AdaptiveKernelConfig(getattr(sk.SDF, kind)(*prm), **w["kw"])
k0 = 1.0 if wl == "C3" else sk.compute_k0(cfg)
rs = bench.shard_grid(w, 0, 1)
d_xs = torch.from_numpy(rs).cuda()
d_v = torch.empty_like(d_xs); d_e = torch.empty_like(d_xs)
st = torch.cuda.current_stream().cuda_stream
def step():
    sk.kernel_values_device(cfg, d_xs.data_ptr(), d_xs.numel(), k0, d_v.data_ptr(), d_e.data_ptr(), stream_ptr=st)
for _ in range(5):
    step()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(reps):
    step()
torch.cuda.synchronize()
wall = (time.perf_counter() - t) / reps * 1e3
ref = d_v.cpu().numpy().copy()
cfg.profile(True)
acc = {}
for _ in range(reps):
    step()
    for k, v in cfg.kernel_times().items():
        a = acc.setdefault(k, [0.0, 0])
        a[0] += v["total_ms"]; a[1] += v["launches"]
cfg.profile(False)
print(wl, "fuse" if not os.environ.get("SK_NO_FUSE") else "nofuse", "wall ms/call %.4f" % wall,
      {k: "%.2f us x %.1f" % (1e3 * a[0] / max(a[1], 1), a[1] / reps) for k, a in acc.items()}, cfg.last_stats(), flush=True)
np.save(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "gpurun_out", "ktimes_%s_%s.npy" % (wl, "nofuse" if os.environ.get("SK_NO_FUSE") else "fuse")), ref)
