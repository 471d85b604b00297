"""Diagnostic: in-kernel stamps of k_spread_b (SK_STAMPS build): state prologue phases and tile phases of the LAST step of a call."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["SPECTRALKERNELS_HIP_LIB"] = os.path.join(ROOT, "tools", "diag", "libsk_stamps.so")
os.environ["SK_STAMP_FILE"] = "/tmp/stamps.bin"
import spectralkernels_jl_amd as sk
wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
bp = int(sys.argv[2]) if len(sys.argv) > 2 else 0
if wl == "C5":
    cfg = sk.AdaptiveKernelConfig(sk.SDF.matern(2.14, 0.97, 0.89), tol=1e-12, quadspec=(256, 16), batch_panels=bp)
    rs = np.linspace(0.0, 5.1, 10 ** 5)
    k0 = sk.compute_k0(cfg)
else:
    cfg = sk.AdaptiveKernelConfig(sk.SDF.matern(1, 1, 1.5), batch_panels=bp)
    rs = 10.0 ** np.linspace(-6, 0, 10 ** 6)
    k0 = np.pi / 2
for _ in range(3):
    sk.kernel_values(cfg, rs, k0=k0)
print(cfg.last_stats())
raw = np.fromfile("/tmp/stamps.bin", dtype=np.uint64).reshape(-1, 8)
a = raw[:6000]
a = a[a[:, 0] > 0]
t = a[:, :5].astype(np.int64)
d = np.diff(t, axis=1)
print("items", len(a), "(ticks of s_memtime = 100 MHz: 1 tick = 10 ns); kb of the stamped step:", np.unique(a[:, 7]))
print("item phase means: loads+barrier %.1f  stage0 %.1f  gather0 %.1f  rest %.1f  total %.1f" % (
    d[:, 0].mean(), d[:, 1].mean(), d[:, 2].mean(), d[:, 3].mean(), (t[:, 4] - t[:, 0]).mean()))
tot = t[:, 4] - t[:, 0]
nd = 8
print("item total by position in the (heaviest-first) list, octiles:", [int(tot[i * len(tot) // nd:(i + 1) * len(tot) // nd].mean()) for i in range(nd)])
print("   sources per item, octiles:", [int(a[i * len(a) // nd:(i + 1) * len(a) // nd, 5].mean()) for i in range(nd)],
      " cells:", [int(a[i * len(a) // nd:(i + 1) * len(a) // nd, 6].mean()) for i in range(nd)])
for nm, col in (("loads", 0), ("stage0", 1), ("gather0", 2), ("rest", 3)):
    print("   %-8s octiles:" % nm, [int(d[i * len(d) // nd:(i + 1) * len(d) // nd, col].mean()) for i in range(nd)])
w = raw[6000:8000]
w = w[w[:, 0] > 0].astype(np.int64)
t0 = w[:, 0].min()
print("workgroups", len(w), "kernel span %d ticks" % (w[:, 2].max() - t0))
print("  entry quantiles", np.percentile(w[:, 0] - t0, [0, 50, 90, 100]))
print("  prologue (entry -> state ready + barrier) quantiles", np.percentile(w[:, 1] - w[:, 0], [0, 50, 90, 100]))
for nm, c0, c1 in (("entry->state loaded+reduced", 0, 4), ("apply", 4, 5), ("tails", 5, 6), ("build+publish", 6, 7), ("wave done->barrier passed", 7, 1)):
    print("    %-30s median %.0f  p90 %.0f" % (nm, np.median(w[:, c1] - w[:, c0]), np.percentile(w[:, c1] - w[:, c0], 90)))
print("  items per workgroup", np.bincount(w[:, 3].astype(int)))
print("  end quantiles", np.percentile(w[:, 2] - t0, [0, 25, 50, 75, 90, 100]))
print("  per-workgroup busy (end - entry) quantiles", np.percentile(w[:, 2] - w[:, 0], [0, 50, 90, 100]))
