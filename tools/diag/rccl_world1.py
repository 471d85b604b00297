"""RCCL and the library in one process (world size 1 -- one GPU box): the same calls bench.py makes at N > 1 with the
nccl backend (init with device_id, shared K(0) all-reduce on a device tensor, barrier + synchronize fences, all_gather
of the per-rank rows), torch imported first.  usage: python3 tools/diag/rccl_world1.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29531")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import numpy as np
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
import spectralkernels_jl_amd as sk
from spectralkernels_jl_amd import dist as skdist
import bench
w = bench.WORKLOADS["C2"]
kind, prm = w["sdf"]
cfg = sk.AdaptiveKernelConfig(getattr(sk.SDF, kind)(*prm), device=0, **w["kw"])
k0 = skdist.shared_k0(cfg, dist, torch.device("cuda", 0))
rs = bench.shard_grid(w, 0, 1, "strided")
n = rs.size
d_rs, d_ks, d_es = sk.DeviceArray.from_host(rs, device=0), sk.DeviceArray(n, 0), sk.DeviceArray(n, 0)
def fence():
    sk.synchronize(0); dist.barrier(); sk.synchronize(0)
step = lambda: sk.kernel_values_device(cfg, d_rs.ptr, n, k0, d_ks.ptr, d_es.ptr)
dt = bench.timed(step, 20, 5, fence)
# cost of the closing fence itself (inside the timed region by bench.py's contract)
def lap(f, k=50):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k):
        f()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / k
tok = torch.zeros(1, device="cuda")
def ar():
    dist.all_reduce(tok); torch.cuda.synchronize()
print("fence costs (us): dist.barrier %.1f, all_reduce(1 float)+synchronize %.1f, sk.synchronize %.1f" % (
    lap(dist.barrier), lap(ar), lap(lambda: sk.synchronize(0))))
t = torch.tensor([dt, float(n)], dtype=torch.float64, device="cuda")
allt = [torch.zeros_like(t)]
dist.all_gather(allt, t)
ks = d_ks.to_host()
err = np.abs(ks - k0 * (1 + 2 * np.pi * rs) * np.exp(-2 * np.pi * rs)).max() / k0
print("rccl world-1: k0 %.15g, %.4f ms/step, err %.2e, gathered %s" % (k0, 1e3 * dt / 20, err, allt[0].cpu().numpy()))
# where the fence's cost lands: one step right after a barrier against one step after a plain synchronize, and the
# closing fence after a busy period
def one_after(pre, k=30):
    tot = 0.0
    for _ in range(k):
        pre(); t0 = time.perf_counter(); step(); sk.synchronize(0); tot += time.perf_counter() - t0
    return 1e6 * tot / k
def closing(k=30):
    tot = 0.0
    for _ in range(k):
        for _ in range(5):
            step()
        t0 = time.perf_counter(); fence(); tot += time.perf_counter() - t0
    return 1e6 * tot / k
print("one step after fence %.1f us, after plain synchronize %.1f us; closing fence after 5 queued steps %.1f us (5 steps = %.1f us)" % (
    one_after(fence), one_after(lambda: sk.synchronize(0)), closing(), 5 * 119.3))
cfg.profile(True)
acc = {}
for _ in range(10):
    step()
    for k, v in cfg.kernel_times().items():
        a = acc.setdefault(k, [0.0, 0])
        a[0] += v["total_ms"]; a[1] += v["launches"]
cfg.profile(False)
print("events(us) with RCCL initialised:", {k: round(1e3 * a[0] / a[1], 2) for k, a in acc.items() if a[1]})
fence0 = lambda: sk.synchronize(0)
print("steps without the barrier in the fence: %.4f ms/step" % (1e3 * bench.timed(step, 200, 5, fence0) / 200))
dist.destroy_process_group()
print("after destroy_process_group: %.4f ms/step" % (1e3 * bench.timed(step, 200, 5, fence0) / 200))
