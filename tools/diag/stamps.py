"""Diagnostic: run one C2 kernel_values with the SK_STAMPS build and summarise k_spread phase times."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from spectralkernels_jl_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "diag", "libsk_stamps.so")
os.environ["SK_STAMP_FILE"] = "/tmp/stamps.bin"
import spectralkernels_jl_amd as sk
cfg = sk.AdaptiveKernelConfig(sk.SDF.matern(1, 1, 1.5))
rs = 10.0 ** np.linspace(-6, 0, 10 ** 6)
for _ in range(3):
    sk.kernel_values(cfg, rs, k0=np.pi / 2)
a = np.fromfile("/tmp/stamps.bin", dtype=np.uint64).reshape(-1, 8)
a = a[a[:, 0] > 0]
t = a[:, :5].astype(np.int64)
d = np.diff(t, axis=1)
print("blocks", len(a), "kernel span (ticks)", t[:, 4].max() - t[:, 0].min())
print("phase means (ticks @100MHz? or shader clk): tables+cs %.0f  stage0 %.0f  gather0 %.0f  rest %.0f  total %.0f" % (
    d[:, 0].mean(), d[:, 1].mean(), d[:, 2].mean(), d[:, 3].mean(), (t[:, 4] - t[:, 0]).mean()))
order = np.argsort(t[:, 0])
st = t[order, 0] - t[:, 0].min()
en = t[order, 4] - t[:, 0].min()
print("start times quantiles", np.percentile(st, [0, 25, 50, 75, 100]))
print("end times quantiles", np.percentile(en, [0, 25, 50, 75, 100]))
tot = t[:, 4] - t[:, 0]
print("total per block quantiles", np.percentile(tot, [0, 25, 50, 75, 95, 100]))
nsrc = a[:, 5].astype(int)
for lo, hi in ((0, 128), (128, 257), (257, 2000)):
    m = (nsrc >= lo) & (nsrc < hi)
    if m.any():
        print("nsrc in [%d,%d): n=%d mean total %.0f, phases %s" % (lo, hi, m.sum(), tot[m].mean(), d[m].mean(axis=0).round(0)))

b = np.fromfile("/tmp/stamps.bin", dtype=np.uint64).reshape(-1, 8)[4096:]
b = b[b[:, 0] > 0]
t = b[:, :4].astype(np.int64)
rt = (b[:, 5] - b[:, 4]).astype(np.int64)
cyc = t[:, 3] - t[:, 0]
print("k_interp blocks", len(b), "cycles/block quantiles", np.percentile(cyc, [0, 50, 90, 100]))
print("  phases mean: loads+stage %.0f  compute %.0f  reduce %.0f" % ((t[:, 1] - t[:, 0]).mean(), (t[:, 2] - t[:, 1]).mean(), (t[:, 3] - t[:, 2]).mean()))
print("  realtime ticks (100 MHz) per block median %.0f -> shader clock = %.2f GHz" % (np.median(rt), np.median(cyc / np.maximum(rt, 1)) * 0.1))
print("  window sizes quantiles", np.percentile(b[:, 6].astype(np.int64), [0, 50, 90, 100]), "use_win frac", b[:, 7].mean())
