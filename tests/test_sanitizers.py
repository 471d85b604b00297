"""CPU tier: the threaded host code of the library (staging threads of sk_hostio.cpp, shard threads and the parallel merge
sort of sk_shard.cpp) under ThreadSanitizer and Address/UB-Sanitizer, driven without a device by
tools/sanitize/host_driver.cpp through the injected shard backend.  Never runs on the GPU box tier."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/clang++") or not os.path.exists("/opt/rocm/lib/llvm/lib/libarcher.so"),
                    reason="needs ROCm's clang++ with libomp / archer")
def test_threaded_host_code_is_clean_under_tsan_and_asan(tmp_path):
    r = subprocess.run([os.path.join(ROOT, "tools", "sanitize", "run_host_sanitizers.sh"), str(tmp_path)],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert r.stdout.count("sanitizer reports: 0, host_driver: all checks held") == 2, r.stdout
