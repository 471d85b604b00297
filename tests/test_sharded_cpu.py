"""CPU tier of the multi-device entry (sk_kernel_values_sharded, SURVEY 8e "independent mode"): the split of the
sorted unique lag set into contiguous blocks, the host dedup / scatter for unsorted input and the error path, with
the device call of a shard replaced by the oracle (sk_sharded_create_injected).  On a multi-GPU node the same code
drives one config replica per device."""
import numpy as np
import pytest


def _backend(oracle, calls):
    ocfg = oracle.AdaptiveKernelConfig(oracle.SDF.matern(1.0, 1.0, 1.5), quadspec=(64, 4), nthreads=1)

    def be(shard, xs, k0, pd):
        calls.append((shard, xs.copy()))
        return oracle.kernel_values(ocfg, xs, k0=k0)

    return ocfg, be


@pytest.mark.parametrize("nshard", [1, 2, 3, 8])
def test_sorted_blocks_written_in_place(sk, oracle, nshard):
    calls = []
    ocfg, be = _backend(oracle, calls)
    sh = sk.ShardedKernelConfig(backend=be, n_shards=nshard)
    xs = np.sort(np.random.default_rng(1).uniform(0.0, 1.0, 41))
    v, e = sh.kernel_values(xs, k0=np.pi / 2)
    info = sh.last()
    # contiguous blocks that tile [0, n) with sizes differing by at most one; every shard saw exactly its block
    assert info[0]["lo"] == 0 and info[-1]["hi"] == xs.size
    assert all(a["hi"] == b["lo"] for a, b in zip(info, info[1:]))
    sizes = [d["hi"] - d["lo"] for d in info]
    assert max(sizes) - min(sizes) <= 1 and sum(sizes) == xs.size
    for shard, seen in calls:
        assert np.array_equal(seen, xs[info[shard]["lo"]:info[shard]["hi"]])
    assert sorted(c[0] for c in calls) == list(range(nshard))
    # each block marches on its own (its own r_hi, src/adaptive.jl:152): equal to the oracle on that block, and
    # within tol of the single march
    for d in info:
        bv, be_ = oracle.kernel_values(ocfg, xs[d["lo"]:d["hi"]], k0=np.pi / 2)
        assert np.array_equal(v[d["lo"]:d["hi"]], bv) and np.array_equal(e[d["lo"]:d["hi"]], be_, equal_nan=True)
    v1, _ = oracle.kernel_values(ocfg, xs, k0=np.pi / 2)
    assert np.abs(v - v1).max() <= 1e-8 * np.pi / 2


def test_unsorted_duplicates_zero_and_more_shards_than_lags(sk, oracle):
    calls = []
    ocfg, be = _backend(oracle, calls)
    sh = sk.ShardedKernelConfig(backend=be, n_shards=3)
    rng = np.random.default_rng(2)
    xs = np.r_[rng.uniform(0, 1, 30), 0.0, 0.25, 0.25, 0.0, -0.0]
    rng.shuffle(xs)
    v, e = sh.kernel_values(xs, k0=np.pi / 2)
    uniq = np.unique(xs)
    assert sum(c[1].size for c in calls) == uniq.size  # every unique lag evaluated once
    assert np.array_equal(np.concatenate([c[1] for c in sorted(calls, key=lambda c: c[0])]), uniq)
    v1, e1 = oracle.kernel_values(ocfg, xs, k0=np.pi / 2)
    assert np.abs(v - v1).max() <= 1e-8 * np.pi / 2
    assert np.all(v[xs == 0] == np.pi / 2) and np.all(np.isnan(e[xs == 0]))  # src/adaptive.jl:132-146
    assert np.all(v[xs == 0.25] == v[xs == 0.25][0])
    # fewer unique lags than shards: the surplus shards stay idle
    calls.clear()
    sh8 = sk.ShardedKernelConfig(backend=be, n_shards=8)
    v, _ = sh8.kernel_values(np.array([0.5, 0.1, 0.5]), k0=np.pi / 2)
    assert len(calls) == 2 and v[0] == v[2]
    assert [d["hi"] - d["lo"] for d in sh8.last()] == [1, 1, 0, 0, 0, 0, 0, 0]


def test_errors(sk, oracle):
    calls = []
    _, be = _backend(oracle, calls)
    sh = sk.ShardedKernelConfig(backend=be, n_shards=2)
    with pytest.raises(sk.SpectralKernelsError) as ex:
        sh.kernel_values(np.array([0.1, -1.0]), k0=1.0)
    assert ex.value.code == 1 and not calls
    with pytest.raises(sk.SpectralKernelsError):  # no device here: K(0) cannot be computed by an injected handle
        sh.kernel_values(np.array([0.1, 0.2]))

    def failing(shard, xs, k0, pd):
        if shard == 1:
            raise RuntimeError("boom")
        return np.zeros(xs.size), np.zeros(xs.size)

    shf = sk.ShardedKernelConfig(backend=failing, n_shards=2)
    with pytest.raises(sk.SpectralKernelsError) as ex:
        shf.kernel_values(np.linspace(0.1, 1, 10), k0=1.0)
    assert "shard 1" in str(ex.value)
    v, e = sk.ShardedKernelConfig(backend=be, n_shards=2).kernel_values(np.array([]), k0=1.0)
    assert v.size == 0


@pytest.mark.parametrize("nshard", [2, 8])
def test_strided_split_same_range_on_every_shard(sk, oracle, nshard):
    """sk_sharded_set_split(strided): shard s takes lags s, s + ns, ... of the sorted unique set -- every shard sees the
    whole lag range (same panel schedule, equal work), results scattered back in place."""
    calls = []
    ocfg, be = _backend(oracle, calls)
    sh = sk.ShardedKernelConfig(backend=be, n_shards=nshard, split="strided")
    rng = np.random.default_rng(4)
    xs = np.r_[np.sort(rng.uniform(0.0, 1.0, 57)), 0.3, 0.3, 0.0]
    rng.shuffle(xs)
    v, e = sh.kernel_values(xs, k0=np.pi / 2)
    uniq = np.unique(xs)
    for shard, seen in calls:
        assert np.array_equal(seen, uniq[shard::nshard])
    assert sorted(c[0] for c in calls) == list(range(nshard))
    sizes = [c[1].size for c in calls]
    assert max(sizes) - min(sizes) <= 1 and sum(sizes) == uniq.size
    # each shard's largest lag is within one stride of the global one: same first panel width up to that
    tops = [c[1][-1] for c in calls]
    assert min(tops) >= uniq[-nshard]
    v1, e1 = oracle.kernel_values(ocfg, xs, k0=np.pi / 2)
    assert np.abs(v - v1).max() <= 1e-8 * np.pi / 2
    assert np.array_equal(np.isnan(e), np.isnan(e1))


def test_large_unsorted_input_goes_through_the_parallel_host_sort(sk, oracle):
    # 2e5 lags with duplicates: the injected tier has no device, so the dedup is the parallel merge sort of sk_shard.cpp
    rng = np.random.default_rng(9)
    base = rng.uniform(0.0, 2.0, 150000)
    xs = np.r_[base, base[:50000], 0.0, -0.0]
    rng.shuffle(xs)
    seen = []
    be = lambda shard, x, k0, pd: (seen.append(x.copy()) or (2.0 * x + 1.0, x * 0.5))
    sh = sk.ShardedKernelConfig(backend=be, n_shards=4)
    v, e = sh.kernel_values(xs, k0=1.0)
    assert np.array_equal(v, 2.0 * np.abs(xs) + 1.0) and np.array_equal(e, np.abs(xs) * 0.5)
    allseen = np.concatenate(seen)
    assert np.array_equal(np.sort(allseen), np.unique(xs)) and all(np.all(np.diff(s) > 0) for s in seen)


@pytest.mark.parametrize("nshard", [2, 8])
def test_pairs_sharded_blocks_of_the_pair_list(sk, oracle, nshard):
    """sk_kernel_values_pairs_sharded (BASELINE config 4's call shape): points replicated, the pair list cut into one
    block per shard, each shard's lags formed / sorted / deduplicated by the shard itself; results in pair order."""
    rng = np.random.default_rng(12)
    pts = rng.uniform(0.0, 1.0, (40, 2))
    pairs = np.array([(i, j) for i in range(40) for j in range(i, 40) if (i + j) % 3 != 1], dtype=np.int64)
    seen = []

    def be(shard, x, k0, pd):
        assert np.all(np.diff(x) > 0)  # sorted unique, as a device shard's march gets them
        seen.append((shard, x.size))
        return np.exp(-x), x * x

    sh = sk.ShardedKernelConfig(backend=be, n_shards=nshard)
    v, e = sh.kernel_values_pairs(pts, pairs, k0=1.0)
    lag = np.sqrt(((pts[pairs[:, 0]] - pts[pairs[:, 1]]) ** 2).sum(1))
    assert np.allclose(v, np.exp(-lag), rtol=0, atol=1e-15) and np.allclose(e, lag * lag, rtol=0, atol=1e-15)
    info = sh.last()
    assert info[0]["lo"] == 0 and info[-1]["hi"] == len(pairs) and all(a["hi"] == b["lo"] for a, b in zip(info, info[1:]))
    assert sorted(s for s, _ in seen) == list(range(nshard))
    # the dense list of build_dense_cov_matrix (src/utils.jl:43-46): 0, then |p_i - p_j| for i < j
    seen.clear()
    vd, ed = sh.kernel_values_pairs(pts[:17], None, k0=1.0)
    dl = np.r_[0.0, [np.linalg.norm(pts[i] - pts[j]) for i in range(17) for j in range(i + 1, 17)]]
    assert vd.size == 1 + 17 * 16 // 2
    assert np.allclose(vd, np.exp(-dl), rtol=0, atol=1e-15) and np.allclose(ed, dl * dl, rtol=0, atol=1e-15)
    # 1-D points
    v1, _ = sh.kernel_values_pairs(pts[:9, 0], np.array([(0, 5), (8, 2), (3, 3)]), k0=1.0)
    assert np.allclose(v1, np.exp(-np.abs([pts[0, 0] - pts[5, 0], pts[8, 0] - pts[2, 0], 0.0])), rtol=0, atol=1e-15)
