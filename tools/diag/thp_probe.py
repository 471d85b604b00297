"""Does a MADV_HUGEPAGE hint on a fresh result array shorten its first-touch cost?  (host-only probe)"""
import ctypes, mmap, time, numpy as np
for f in ("enabled", "defrag", "shmem_enabled"):
    try:
        print(f, open("/sys/kernel/mm/transparent_hugepage/" + f).read().strip())
    except Exception as e:
        print(f, "unreadable", e)
libc = ctypes.CDLL(None, use_errno=True)
n = 2 * 10 ** 6
src = np.ones(n)
def touch(hint):
    ts = []
    for _ in range(10):
        a = np.empty(n)
        if hint:
            addr = a.ctypes.data
            lo = (addr + 4095) & ~4095
            rc = libc.madvise(ctypes.c_void_p(lo), ctypes.c_size_t(a.nbytes - (lo - addr)), 14)  # MADV_HUGEPAGE
        t = time.perf_counter(); a[:] = src; ts.append(time.perf_counter() - t)
        del a
    return 1e3 * np.median(ts)
print("first-touch copy of 16 MB: plain %.3f ms, with MADV_HUGEPAGE %.3f ms; into touched memory %.3f ms" % (
    touch(False), touch(True), (lambda a: (lambda t: (a.__setitem__(slice(None), src), 1e3 * (time.perf_counter() - t))[1])(time.perf_counter()))(np.zeros(n))))
