"""Diagnostic: in-kernel stamps of k_interp2 (SK_STAMPS build) on C2: where a workgroup's time goes and how the
launch fills the chip.  Stages: 0 entry, 1 state arrived, 2 lag arrived, 3 pre-barrier arithmetic done (window loads
in flight), 4 window in LDS + barrier, 5 weights, 6 taps + finish, 7 end (stores, reductions)."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["SPECTRALKERNELS_HIP_LIB"] = os.path.join(ROOT, "tools", "diag", "libsk_stamps.so")
os.environ["SK_STAMP_FILE"] = "/tmp/stamps.bin"
import spectralkernels_jl_amd as sk
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10 ** 6
cfg = sk.AdaptiveKernelConfig(sk.SDF.matern(1, 1, 1.5))
rs = 10.0 ** np.linspace(-6, 0, n)
d_rs, d_ks, d_es = sk.DeviceArray.from_host(rs), sk.DeviceArray(n), sk.DeviceArray(n)
for _ in range(4):
    sk.kernel_values_device(cfg, d_rs.ptr, n, np.pi / 2, d_ks.ptr, d_es.ptr)
raw = np.fromfile("/tmp/stamps.bin.interp", dtype=np.uint64).reshape(-1, 16)
a = raw[raw[:, 0] > 0].astype(np.int64)
print("workgroups stamped:", len(a), "window path:", int(a[:, 11].sum()))
t = a[:, :8]
d = np.diff(t, axis=1)
names = ["state wait", "lag wait", "pre-barrier arithmetic", "window->LDS + barrier", "weights", "taps + finish", "stores + reductions"]
print("stage durations in s_memtime ticks (mean / median / p90):")
for k, nm in enumerate(names):
    print("  %-26s %8.0f %8.0f %8.0f" % (nm, d[:, k].mean(), np.median(d[:, k]), np.percentile(d[:, k], 90)))
tot = t[:, 7] - t[:, 0]
print("  %-26s %8.0f %8.0f %8.0f" % ("workgroup total", tot.mean(), np.median(tot), np.percentile(tot, 90)))
rt0, rt1 = a[:, 8], a[:, 9]
base = rt0.min()
print("s_memrealtime (100 MHz, 10 ns ticks): launch span %d ticks; memtime ticks per realtime tick ~ %.2f" % (
    rt1.max() - base, np.median(tot[rt1 > rt0] / np.maximum(rt1 - rt0, 1)[rt1 > rt0])))
print("  workgroup entry quantiles (ticks after first entry):", np.percentile(rt0 - base, [0, 10, 25, 50, 75, 90, 100]).round(0))
print("  workgroup end quantiles:", np.percentile(rt1 - base, [0, 10, 25, 50, 75, 90, 100]).round(0))
hw = a[:, 10]
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
print("  distinct (se, sh, cu) seen:", len(set(zip(se.tolist(), sh.tolist(), cu.tolist()))))
# occupancy over time: number of workgroups alive per realtime tick
ev = np.zeros(int(rt1.max() - base) + 2)
for s0, s1 in zip(rt0 - base, rt1 - base):
    ev[int(s0):int(s1) + 1] += 1
print("  workgroups alive per 10 ns tick (every 100 ticks):", ev[::100].astype(int).tolist())
