import os, sys, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import spectralkernels_jl_amd as sk
n = 10 ** 5
rs = np.linspace(0.0, 5.1, n)
cfg = sk.AdaptiveKernelConfig(sk.SDF.matern(2.14, 0.97, 0.89), tol=1e-12, quadspec=(256, 16))
k0 = sk.compute_k0(cfg)
for i in range(3):
    ks, es = sk.kernel_values(cfg, rs, k0=k0)
    print(i, cfg.last_stats(), sk.step_kinds(cfg), float(ks.sum()))
cfg4 = sk.AdaptiveKernelConfig(sk.SDF.matern(2.14, 0.97, 0.89), tol=1e-12, quadspec=(256, 16), batch_panels=4)
ks4, es4 = sk.kernel_values(cfg4, rs, k0=k0)
print("explicit 4:", cfg4.last_stats(), sk.step_kinds(cfg4), np.array_equal(ks, ks4), np.array_equal(es, es4, equal_nan=True))
d_rs, d_ks, d_es = sk.DeviceArray.from_host(rs), sk.DeviceArray(n), sk.DeviceArray(n)
for c, nm in ((cfg, "auto (hybrid)"), (cfg4, "explicit 4")):
    for _ in range(5):
        sk.kernel_values_device(c, d_rs.ptr, n, k0, d_ks.ptr, d_es.ptr)
    sk.synchronize()
    t = time.perf_counter()
    for _ in range(200):
        sk.kernel_values_device(c, d_rs.ptr, n, k0, d_ks.ptr, d_es.ptr)
    sk.synchronize()
    print(nm, "%.4f ms/call" % (1e3 * (time.perf_counter() - t) / 200), sk.step_kinds(c))
