"""CPU tier: resource metadata of the gfx950 code object inside libspectralkernels_hip.so.

The march kernels must not spill: a spilled VGPR is private-segment (scratch) traffic per lane and launch -- round 2's
k_spread<12,true> wrote 8.4 MB per launch that way, three times its algorithmic output.  Reads the AMDGPU metadata
notes of the extracted device bundle (llvm-objdump --offloading, llvm-readelf --notes); no GPU needed."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
MARCH = ("k_spread_b", "k_interp_b", "k_advance_b", "k_interp2", "k_joint_interp", "k_spread_joint")


def _kernels():
    lib = os.path.join(ROOT, "spectralkernels.jl_amd", "libspectralkernels_hip.so")
    if not os.path.exists(lib):
        import __graft_entry__
        __graft_entry__.build()
    objdump, readelf = os.path.join(LLVM, "llvm-objdump"), os.path.join(LLVM, "llvm-readelf")
    if not (os.path.exists(objdump) and os.path.exists(readelf)):
        pytest.skip("llvm-objdump / llvm-readelf of the ROCm toolchain not found")
    with tempfile.TemporaryDirectory() as tmp:
        shutil.copy(lib, tmp)
        subprocess.run([objdump, "--offloading", "libspectralkernels_hip.so"], cwd=tmp, check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        dev = [f for f in os.listdir(tmp) if f.endswith("gfx950")]
        assert len(dev) == 1, os.listdir(tmp)
        notes = subprocess.run([readelf, "--notes", dev[0]], cwd=tmp, check=True, capture_output=True, text=True).stdout
    out = {}
    for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk)
        if not name:
            continue
        f = lambda key: int(re.search(r"\." + key + r":\s+(\d+)", blk).group(1))
        dem = subprocess.run(["c++filt", name.group(1)], capture_output=True, text=True).stdout.strip()
        out[dem] = dict(vgpr=f("vgpr_count"), vgpr_spill=f("vgpr_spill_count"), sgpr_spill=f("sgpr_spill_count"),
                        scratch=f("private_segment_fixed_size"), lds=f("group_segment_fixed_size"))
    return out


def test_march_kernels_do_not_spill():
    ks = _kernels()
    march = {k: v for k, v in ks.items() if any(("sk::" + m + "<") in k or ("sk::" + m + "(") in k for m in MARCH)}
    # every width of the dim-1 march is there, batched and not
    assert sum("k_spread_b<" in k for k in march) >= 7 * 3 and sum("k_interp_b<" in k for k in march) >= 7 * 3, sorted(march)
    bad = {k: v for k, v in march.items() if v["vgpr_spill"] > 0 or v["scratch"] > 0}
    assert not bad, "kernels with spilled VGPRs / a private segment:\n" + "\n".join("%s %s" % kv for kv in sorted(bad.items()))


def test_default_config_kernels_keep_four_workgroups_per_cu():
    # BASELINE configs 1 / 2 (tol 1e-8 -> w = 12, two panels per step): the spread kernel at <= 128 VGPRs and <= 40 KB of
    # LDS (four 256-thread workgroups per CU), the interpolation kernel at <= 102 VGPRs (five waves per SIMD)
    ks = _kernels()
    sp = [v for k, v in ks.items() if "k_spread_b<12, 2>" in k]
    ip = [v for k, v in ks.items() if "k_interp_b<12, 2," in k]
    assert len(sp) == 1 and len(ip) == 1
    assert sp[0]["vgpr"] <= 128 and sp[0]["lds"] <= 40960, sp
    assert ip[0]["vgpr"] <= 102 and ip[0]["lds"] <= 32768, ip


def test_hankel_and_sort_kernels_resources():
    # dim >= 2 by boxes and the bucket sort (3f): no private segment anywhere; the far kernel at four
    # workgroups per CU (<= 128 VGPRs, <= 40 KB LDS), the node kernel of the default orders at three waves per SIMD
    # (<= 168 VGPRs: with the generic jn inlined it took all 256 -- one wave per SIMD, 200 us), the wave sort in registers
    ks = _kernels()
    hk = {k: v for k, v in ks.items() if "sk::k_hk_" in k or "sk::k_bk_" in k or "sk::k_spread_multi" in k}
    assert sum("k_hk_far<" in k for k in hk) >= 7 and sum("k_hk_nodes<" in k for k in hk) == 6, sorted(hk)
    bad = {k: v for k, v in hk.items() if v["vgpr_spill"] > 0 or v["scratch"] > 0}
    assert not bad, "\n".join("%s %s" % kv for kv in sorted(bad.items()))
    far = [v for k, v in hk.items() if "k_hk_far<12>" in k]
    assert len(far) == 1 and far[0]["vgpr"] <= 128 and far[0]["lds"] <= 40960, far
    for nu in (0, 1):
        nodes = [v for k, v in hk.items() if "k_hk_nodes<%d, " % nu in k]  # near-box and inner-box instantiation
        assert len(nodes) == 2 and all(v["vgpr"] <= 168 and v["lds"] == 0 for v in nodes), nodes
    wave = [v for k, v in hk.items() if "k_bk_sort_wave" in k]
    assert len(wave) == 1 and wave[0]["lds"] == 0 and wave[0]["vgpr"] <= 128, wave
