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
