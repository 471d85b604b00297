"""World-size-2 gloo tests (CPU) of the multi-GPU plumbing: r-set sharding, the cooperative
K(0) all-reduce and result reassembly.  The per-shard compute is injected (the oracle stands in
for the HIP call, which needs a GPU); on the GPU box bench.py runs the same plumbing over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import spectralkernels_jl_amd as sk
        from spectralkernels_jl_amd import api, dist as skdist
        from oracle import sk_oracle as o
        args, _, _ = api._make_args(f=sk.SDF.matern(1, 1, 1.5), df=None, dim=1, alpha=0.0, tol=1e-8,
                                    derivative=False, logw=False, convergence_criteria="both", tail=None,
                                    quadspec=(64, 4), device=-1, nufft_width=0)

        class Cfg:  # shared_k0 only needs the argument block
            _args = args

        k0 = skdist.shared_k0(Cfg, dist)
        rs = 10.0 ** np.linspace(-4, 0, 101)
        ocfg = o.AdaptiveKernelConfig(o.SDF.matern(1, 1, 1.5), quadspec=(64, 4), nthreads=1)
        out = {}
        for strided in (True, False):
            if strided:
                mine = skdist.shard_strided(rs, rank, world)
            else:
                lo, hi = skdist.shard_bounds(rs.size, rank, world)
                mine = rs[lo:hi]
            ks, _ = o.kernel_values(ocfg, mine, k0=k0)
            out[strided] = skdist.gather_sharded(ks, rs.size, rank, world, dist, strided=strided)
        q.put((rank, k0, out[True], out[False]))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_sharding():
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sys.path.insert(0, ROOT)
    from oracle import sk_oracle as o
    rs = 10.0 ** np.linspace(-4, 0, 101)
    truth = np.pi / 2 * (1 + 2 * np.pi * rs) * np.exp(-2 * np.pi * rs)
    for rank, k0, strided, blocked in res:
        assert abs(k0 - np.pi / 2) < 1e-9  # parts of the K(0) quadrature sum to K(0)
        # every rank reassembles the same full vector; both shardings are within tol of the truth
        assert np.abs(strided - truth).max() / k0 <= 1e-8
        assert np.abs(blocked - truth).max() / k0 <= 1e-8
    assert np.array_equal(res[0][2], res[1][2]) and np.array_equal(res[0][3], res[1][3])


def test_shard_bounds_cover():
    sys.path.insert(0, ROOT)
    from spectralkernels_jl_amd import dist as skdist
    for n in (0, 1, 7, 8, 1000003):
        for world in (1, 2, 3, 8):
            b = [skdist.shard_bounds(n, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1


def _json_lines(out):
    import json
    lines = []
    for ln in out.splitlines():
        ln = ln.strip()
        if ln.startswith("{") and ln.endswith("}"):
            try:
                lines.append(json.loads(ln))
            except ValueError:
                pass
    return lines


@pytest.mark.parametrize("door", ["self_launch", "torchrun"])
def test_bench_gpus2_gloo_through_both_doors(door):
    """`python bench.py --gpus 2` (the driver's form for N = 1) must itself start the two ranks -- a child
    torch.distributed.run, never an exec -- and relay rank 0's ONE JSON line; the launcher form stays.  The march is
    left out (--plumbing-only: no device here); rendezvous, the K(0) all-reduce, the r-set split and the gathers run."""
    pytest.importorskip("torch")
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    tail = ["--gpus", "2", "--backend", "gloo", "--plumbing-only", "--workload", "C2", "--shard", "strided"]
    if door == "self_launch":
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + tail
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py")] + tail
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    j = lines[0]
    assert j["n_gpus"] == 2 and j["plumbing_only"] and j["value"] is None and j["backend"] == "gloo"
    assert abs(j["k0"] - np.pi / 2) < 1e-9  # the two halves of the K(0) quadrature, summed by the all-reduce
    assert [p["lags"] for p in j["per_rank"]] == [10 ** 6, 10 ** 6]
    # strided split: both ranks see the whole r-range (hence the same panel schedule)
    assert all(p["r_min"] < 2e-6 and p["r_max"] > 0.99 for p in j["per_rank"])


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    import subprocess
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--plumbing-only"], env=env,
                       capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert r.returncode != 0 and "does not match" in (r.stderr + r.stdout)
