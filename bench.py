#!/usr/bin/env python3
"""bench.py -- r-targets/sec through kernel_values on MI355X (BASELINE.json metric).

One "step" = one kernel_values pass over one batch of synthetic lags: BASELINE config 2 by default -- Matern
nu=3/2 SDF S(w) = (1+w^2)^-2, 10^6 log-spaced r in [1e-6, 1], tol = 1e-8, quadspec (4096, 16).

  value         span "device_resident": lags already in HBM when the timed region starts (sk_kernel_values_device),
                results left in HBM -- the contract's `value`
  host_to_host  span "host_to_host" (host entry -> results in host memory): the reference-shaped call
                (src/adaptive.jl:95-108: host Vector in, two FRESH host Vectors out) on the same lags through
                sk_kernel_values; headline leg = fresh result vectors as the API hands them out (pinned pool), then
                fresh pageable vectors, reused vectors, permuted input with duplicates, everything pinned;
                cpu_baseline is quoted against it

Multi-GPU: one process per GPU (torch.distributed over RCCL); every rank marches its own 10^6-lag shard of the
r-set (weak scaling, no data-path collective); the only exchange is the K(0) normaliser, whose quadrature is split
over the ranks and summed with one all-reduce at set-up.  --shard strided (default): rank r takes every N-th lag of
the N*10^6 grid (same r-range, same panel schedule on every rank); --shard blocked: contiguous blocks (each rank
its own r-range and schedule -- per-rank ms and panel counts in the JSON line show the imbalance).

At N = 1 no torch / second HIP runtime is loaded into the process: device buffers come from the library's own ABI.
Prints ONE JSON line on rank 0 (see DESIGN.md section "Measurement").
"""
import argparse
import ctypes.util
import json
import math
import os
import shutil
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
ROUND = "r04"

WORKLOADS = {
    # name: (sdf kind, params, config kwargs, r-grid builder, k0, truth)
    "C2": dict(desc="Matern nu=3/2 S=(1+w^2)^-2, 1e6 log-spaced r in [1e-6,1], tol=1e-8, quadspec=(4096,16)",
               sdf=("matern", (1.0, 1.0, 1.5)), kw=dict(tol=1e-8), n=10 ** 6, lo=-6.0, hi=0.0),
    "C4": dict(desc="dim=2 isotropic S=0.7(1.3^2+w^2)^-2.8 (scripts/simple_2d.jl), 1e6 pairwise lags |x_i-y_j| of 1e3 x 1e3 "
                    "splitmix64-seeded uniform points in [0,1]^2, tol=1e-8, quadspec=(4096,16); fast Hankel transform "
                    "(one asymptotic-series NUFFT per panel + Chebyshev-compressed near / inner boxes)",
               sdf=("matern", (0.7, 1.3, 1.8, 2.0)), kw=dict(tol=1e-8, dim=2), n=10 ** 6, lo=None, hi=None),
    "C5": dict(desc="Matern (2.14,0.97,0.89), 1e5 uniform r in [0,5.1], tol=1e-12, quadspec=(256,16)",
               sdf=("matern", (2.14, 0.97, 0.89)), kw=dict(tol=1e-12, quadspec=(256, 16)), n=10 ** 5, lo=None, hi=5.1),
    "C3": dict(desc="singular long-memory SDF alpha=0.4 (rho=10, nu=0.5, K(0)=1), 1e6 log-spaced r in [1e-8,1e-2], "
                    "tol=1e-8, quadspec=(256,16)",
               sdf=("matern", None), kw=dict(tol=1e-8, alpha=0.4, quadspec=(256, 16)), n=10 ** 6, lo=-8.0, hi=-2.0),
}


def splitmix64(seed, count):
    """Documented deterministic stream: splitmix64 -> uniform doubles in [0,1)."""
    out = np.empty(count)
    x = np.uint64(seed)
    with np.errstate(over="ignore"):
        for i in range(count):
            x = x + np.uint64(0x9E3779B97F4A7C15)
            z = x
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            z = z ^ (z >> np.uint64(31))
            out[i] = float(z >> np.uint64(11)) * (1.0 / 9007199254740992.0)
    return out


def c4_points(m=1000):
    u = splitmix64(123, 4 * m).reshape(4, m)
    return np.stack([u[0], u[1]], 1), np.stack([u[2], u[3]], 1)


def shard_grid(w, rank, world, mode="strided"):
    """This rank's lags of the world*n global set: every world-th lag (strided) or a contiguous block (blocked)."""
    n = w["n"]
    if w["hi"] is None:  # C4: pairwise distances between two seeded point sets, rows of X sharded over ranks
        m = int(round(np.sqrt(n)))
        X, Y = c4_points(m)
        rows = np.arange(rank, m, world) if world > 1 else np.arange(m)
        d = np.sqrt(((X[rows, None, :] - Y[None, :, :]) ** 2).sum(-1)).ravel()
        return np.ascontiguousarray(d)
    if w["lo"] is None:  # uniform grid including r = 0 (test/matern_sdf.jl:5)
        g = np.linspace(0.0, w["hi"], n * world)
    else:
        g = 10.0 ** np.linspace(w["lo"], w["hi"], n * world)
    if mode == "blocked":
        return np.ascontiguousarray(g[rank * n:(rank + 1) * n])
    return np.ascontiguousarray(g[rank::world])


def reference_probe():
    """Is the true reference path (Julia + FINUFFT, README.md:38-48) runnable on this box?  SURVEY 8(d)."""
    found = {"julia": shutil.which("julia"), "libfinufft": ctypes.util.find_library("finufft")}
    try:
        import finufft  # noqa: F401
        found["python_finufft"] = True
    except Exception:
        found["python_finufft"] = False
    ok = bool(found["julia"]) or bool(found["libfinufft"]) or found["python_finufft"]
    return ("available" if ok else "unavailable"), found


def cpu_quota():
    """CPUs the cgroup lets this job use at once (cpu.max / cfs_quota_us), or None.  The GPU boxes show 256 logical CPUs
    under a quota of 16: a 64-thread run of the port is throttled, not slow."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else max(1, -(-int(q) // int(per)))
    except Exception:
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else max(1, -(-q // per))
    except Exception:
        return None


def kernels_sha():
    """Identity of the kernel sources a PMC / instruction-mix summary under profiles/ belongs to."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "spectralkernels.jl_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".inc", ".h", ".cpp")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def timed(fn, steps, warmup, fence):
    for _ in range(warmup):
        fn()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    fence()
    return time.perf_counter() - t0


def self_launch(argv, gpus):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks ourselves, as a CHILD process
    (python -m torch.distributed.run, rendezvous on 127.0.0.1), issued before this process has imported the HIP
    library or torch -- never an exec, never after a GPU call -- and hand back the child's exit code.  The ranks
    inherit stdout: rank 0's JSON line is the one line this invocation prints."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.run(cmd, env=env).returncode


def plumbing_only(args, w, sk, skdist, dist, rank, world):
    """Everything of a multi-rank run but the march (no device needed): rendezvous, the cooperative K(0) all-reduce
    (host quadrature, sk_host_compute_k0_part), this rank's share of the r-set, the barriers and the per-rank gather
    that the real run uses; one JSON line on rank 0 with `value` null.  tests/test_dist_cpu.py drives it through the
    self-launch door (`python bench.py --gpus 2 --backend gloo --plumbing-only`)."""
    from spectralkernels_jl_amd import api
    kind, prm = w["sdf"]
    kw = dict(w["kw"])
    a, _, _ = api._make_args(f=getattr(sk.SDF, kind)(*prm), df=None, dim=kw.get("dim", 1), alpha=kw.get("alpha", 0.0),
                             tol=kw["tol"], derivative=False, logw=False, convergence_criteria="both", tail=None,
                             quadspec=kw.get("quadspec", (4096, 16)), device=-1, nufft_width=0)

    class ArgsOnly:  # shared_k0 needs the argument block only
        _args = a

    k0 = skdist.shared_k0(ArgsOnly, dist, None)
    rs = shard_grid(w, rank, world, args.shard)
    if dist is not None:
        import torch
        dist.barrier()
        t = torch.tensor([float(rs.size), float(rs[0]), float(rs[-1])], dtype=torch.float64)
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        rows = [x.numpy() for x in allt]
        dist.barrier()
    else:
        rows = [np.array([float(rs.size), rs[0], rs[-1]])]
    if rank == 0:
        print(json.dumps({"metric": "r-targets/sec at tol=%g" % kw["tol"], "value": None, "unit": "r-targets/s",
                          "n_gpus": world, "plumbing_only": True, "k0": k0, "backend": args.backend if dist is not None else None,
                          "config": {"workload": args.workload + ": " + w["desc"], "sharding": args.shard},
                          "per_rank": [{"rank": i, "lags": int(r[0]), "r_min": r[1], "r_max": r[2]} for i, r in enumerate(rows)]}))
    if dist is not None:
        dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="C2", choices=sorted(WORKLOADS))
    ap.add_argument("--shard", default="strided", choices=("strided", "blocked"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-leg", action="store_true")
    ap.add_argument("--no-launch-floor", action="store_true",
                    help="skip the launch-floor measurement (profiling passes: its empty launches would enter the kernel averages)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="lags in the CPU-baseline sample (0 = auto)")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"),
                    help="torch.distributed backend at N > 1: nccl (= RCCL over xGMI; the driver's runs) or gloo "
                         "(rehearsal of the multi-rank plumbing with the ranks sharing the visible GPUs)")
    ap.add_argument("--plumbing-only", action="store_true",
                    help="rehearsal of the multi-rank plumbing without a device (CPU test tier): rendezvous, the K(0) "
                         "all-reduce, the r-set split, barriers and the per-rank gather run; no march is launched and "
                         "`value` is null")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # the driver's form for N = 1 (`python bench.py --gpus N`) must work for N > 1 too
        raise SystemExit(self_launch(sys.argv[1:], args.gpus))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    dist = torch = None
    if world > 1:
        # torch (and with it RCCL) only in the multi-process case; it is imported before the HIP library so that
        # one HIP runtime serves both ("One HIP runtime per process")
        import torch
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
            local_rank = local_rank % max(1, torch.cuda.device_count())

    import spectralkernels_jl_amd as sk
    from spectralkernels_jl_amd import dist as skdist

    w = WORKLOADS[args.workload]
    if args.plumbing_only:
        return plumbing_only(args, w, sk, skdist, dist, rank, world)
    kind, prm = w["sdf"]
    if prm is None:  # C3: normalise K(0) = 1 with the closed form (scripts/singular_matern_demo.jl:21)
        from scipy import special
        q = (1 - 0.4) / 2
        prm = (1.0 / (10.0 ** (1 - 0.4 - 2 * 1.0) * special.beta(q, 1.0 - q)), 10.0, 0.5)
    cfg = sk.AdaptiveKernelConfig(getattr(sk.SDF, kind)(*prm), device=local_rank, **w["kw"])

    # shared K(0) normaliser: quadrature split over ranks + one all-reduce (RCCL over xGMI)
    if args.workload == "C3":
        k0 = 1.0  # normalised by the closed form; the reference's own QuadGK call fails on this SDF (demo TODO)
    else:
        k0 = skdist.shared_k0(cfg, dist, torch.device("cuda", local_rank) if (torch and args.backend == "nccl") else None)

    rs = shard_grid(w, rank, world, args.shard)
    n = rs.size
    d_rs = sk.DeviceArray.from_host(rs, device=local_rank)
    d_ks, d_errs = sk.DeviceArray(n, local_rank), sk.DeviceArray(n, local_rank)

    def step():
        sk.kernel_values_device(cfg, d_rs.ptr, n, k0, d_ks.ptr, d_errs.ptr)

    def fence():
        sk.synchronize(local_rank)
        if dist is not None:
            dist.barrier()
            sk.synchronize(local_rank)

    dt_local = timed(step, args.steps, args.warmup, fence)
    dt = dt_local
    per_rank = None
    st = cfg.last_stats()
    kinds = sk.step_kinds(cfg)
    if dist is not None:
        t = torch.tensor([dt_local, float(st["n_panels"]), float(st["n_outer"]), float(n), float(st["sum_active"])],
                         dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        rows = [x.cpu().numpy() for x in allt]
        dt = max(r[0] for r in rows)
        per_rank = [{"rank": i, "ms_per_step": 1e3 * r[0] / args.steps, "n_panels": int(r[1]), "n_outer": int(r[2]),
                     "lags": int(r[3]), "abar": r[4] / max(1.0, r[3])} for i, r in enumerate(rows)]
        n_all = int(sum(r[3] for r in rows))
    else:
        n_all = n
    ms_per_step = 1e3 * dt / args.steps
    value = n_all / (dt / args.steps)

    # ---- accuracy of this rank's shard against the closed form
    ks = d_ks.to_host()
    if args.workload == "C2":
        truth = k0 * (1 + 2 * np.pi * rs) * np.exp(-2 * np.pi * rs)
        max_rel_err = float(np.abs(ks - truth).max() / abs(k0))
    elif args.workload in ("C4", "C5"):
        from oracle import sk_oracle as o
        d = 2 if args.workload == "C4" else 1
        max_rel_err = float(np.abs(ks - o.matern_cov(rs, prm[:3], d=d)).max() / abs(k0))
    else:
        max_rel_err = None

    # ---- the reference-shaped call: host arrays in, host arrays out (rank 0, N = 1)
    host = None
    if rank == 0 and world == 1 and not args.no_host_leg:
        hsteps, hwarm = max(5, min(args.steps, 30)), 3
        nofence = lambda: None  # sk_kernel_values returns with the results in the caller's arrays
        legs = {}
        io_bytes = 24.0 * n

        def leg(name, xs, into=None, results="pinned"):
            out = {}
            if into is None:  # fresh result vectors every call, as the reference's kernel_values returns them
                fn = lambda: out.__setitem__("r", sk.kernel_values(cfg, xs, k0=k0, results=results))
            else:             # caller-provided result arrays (the C ABI's contract), reused across calls
                fn = lambda: out.__setitem__("r", sk.kernel_values_into(cfg, xs, k0, into[0], into[1]))
            dth = timed(fn, hsteps, hwarm, nofence) / hsteps
            tm = sk.host_timing(cfg)
            legs[name] = {"ms_per_step": 1e3 * dth, "value": xs.size / dth, "GB/s_effective": io_bytes / dth / 1e9,
                          "split_ms": {k: round(v, 4) for k, v in tm.items()}, "unique_lags": cfg.last_stats()["n_unique"]}
            return out["r"]

        # the reference's contract first: sorted pageable lags in, two fresh ordinary (pageable) vectors out -- the
        # headline of this span; then the same with the vectors taken from the API's capped pinned pool
        hv, _ = leg("fresh_pageable_result_vectors", rs, results="pageable")
        host_err = float(np.abs(np.asarray(hv) - ks).max() / abs(k0))
        del hv
        leg("fresh_pinned_pool_result_vectors", rs, results="pinned")
        leg("reused_result_vectors", rs, into=(np.empty(n), np.empty(n)))
        prng = np.random.default_rng(7)
        perm = prng.permutation(n)
        xs_p = np.ascontiguousarray(np.r_[rs[perm[: n - n // 10]], rs[perm[: n // 10]]])  # 10 % duplicates, shuffled
        prng.shuffle(xs_p)
        pv, _ = leg("permuted_with_duplicates", xs_p, into=(np.empty(n), np.empty(n)))
        truth_p = k0 * (1 + 2 * np.pi * xs_p) * np.exp(-2 * np.pi * xs_p) if args.workload == "C2" else None
        px, pk, pe = sk.pinned_empty(n), sk.pinned_empty(n), sk.pinned_empty(n)
        px[:] = rs
        leg("all_pinned_arrays", px, into=(pk, pe))
        head = legs["fresh_pageable_result_vectors"]
        host = {"span": "host_to_host",
                "what": "sk_kernel_values: pageable host array in, two host arrays out (src/adaptive.jl:95-108), %d steps.  "
                        "Headline = fresh_pageable_result_vectors: the two result vectors are new ordinary (pageable) numpy "
                        "vectors per call, as the reference returns them; other legs: fresh vectors from the API's capped "
                        "pool of pinned vectors / caller-provided reused vectors / permuted input with duplicates / all "
                        "three arrays pinned" % hsteps,
                "ms_per_step": head["ms_per_step"], "value": head["value"], "unit": "r-targets/s",
                "GB/s_effective": head["GB/s_effective"], "legs": legs,
                "max_abs_diff_vs_device_leg_over_k0": host_err,
                "permuted_max_rel_err_vs_closed_form": (float(np.abs(pv - truth_p).max() / abs(k0))
                                                        if truth_p is not None else None),
                "pinned_leg_max_abs_diff_over_k0": float(np.abs(pk - ks).max() / abs(k0)),
                "copy_threads": int(os.environ.get("SK_COPY_THREADS", "8"))}

    # ---- per-kernel roofline from HIP events on the kernels' own stream (separate, untimed passes)
    cfg.profile(True)
    acc = None
    for _ in range(10):
        step()
        kt = cfg.kernel_times()
        if acc is None:
            acc = {k: dict(v) for k, v in kt.items()}
        else:
            for k in kt:
                for f in ("total_ms", "launches", "active_items"):
                    acc[k][f] += kt[k][f]
    cfg.profile(False)
    st = cfg.last_stats()
    # dominant kernel: the interpolation launch (largest share of device time in the rocprofv3 kernel stats,
    # profiles/*_summary.txt; the HIP-event brackets add the launch gap to whichever kernel follows a host
    # synchronisation, so the event totals of interpolation and spread are too close to rank them reliably)
    batched = w["kw"].get("dim", 1) == 1
    if batched:
        dom = "interp" if acc.get("interp", {}).get("launches", 0) > 0 else max(acc, key=lambda k: acc[k]["total_ms"])
    else:  # dim >= 2 (fast Hankel transform): whatever the events say takes longest -- spread, FFT or interpolation
        # ("near" = the boxes: six small compute-bound kernels with no byte model of their own; their share is reported)
        dom = max((k for k in acc if k not in ("advance", "near")), key=lambda k: acc[k]["total_ms"])
    hk_names = {"gen": "k_gen", "spread": "k_spread_multi", "fft": "rocfft (batch 2 x series terms, length n2)",
                "interp": "k_hk_far", "near": "boxes: k_hk_moments / k_hk_msum / k_hk_eqw / k_hk_nodes / k_hk_icheb / k_hk_iinterp"}
    dom_name = ("k_interp_b" if dom == "interp" else "k_" + dom) if batched else hk_names.get(dom, "k_" + dom)
    M = cfg.quadspec[0] * cfg.quadspec[1]
    n1 = M  # spread grid of this geometry; the batched FFT plan transforms 4 K grids of n1
    # interpolation, per STEP (a step evaluates one or more panels of equal width) and active lag: read r (8) and the
    # confirmed sums (16, not on the very first step), write the sums after the step (16) = 40 B; the final outputs (16)
    # once per lag, when it retires.  The batched kernel also stores the sums after the step's FIRST panel (16 B more
    # when a step has several panels): overhead, counted in `traffic`, not here.
    interp_bytes_total = 10 * (40.0 * st["sum_active_steps"] + 16.0 * st["n_unique"])  # over the 10 profiled calls
    panels_per_step = st["n_panels"] / max(1, st["n_steps"])
    rl = {}
    for k, v in acc.items():
        if v["launches"] == 0:
            continue
        avg_ms = v["total_ms"] / v["launches"]
        items = v["active_items"] / v["launches"]
        if not batched:
            # fast Hankel transform, per launch: the series terms of the panel through the fixed spread geometry
            NT = {8: 6, 10: 8, 12: 10, 14: 13, 16: 16}.get(cfg.nufft_width, 10)  # hk_terms(w), sk_device.h
            n2 = 2 * n1
            if k == "interp":    # the lags above r_c: r, the running sums in and out (their share of sum_active)
                b = 40.0 * 10 * st["sum_active"] / v["launches"]
            elif k == "spread":  # 3M (strength, node) pairs + positions in; NT terms x 2 rules x n2 complex out
                b = 24.0 * 3 * M + NT * 2 * n2 * 16.0
            elif k == "fft":     # NT x 2 grids of n2, read and written once
                b = 2 * NT * 2 * n2 * 16.0
            elif k == "gen":
                b = 48.0 * M + 48.0 * M
            else:                # boxes: Bessel evaluations between equivalent sources and Chebyshev nodes, L2-resident
                b = 0.0
        elif k == "interp":
            b = interp_bytes_total / v["launches"]
        elif k == "gen":
            b = 48.0 * M + 48.0 * M  # read node + weight of both rules (M + 2M) x (8 + 8), write 3M complex strengths
        elif k == "spread":
            # read 3M source positions + Gauss weights (the strengths are computed while staging) and the tile list;
            # write, per panel of the step, the even / odd grids of both rules: 4 n1 complex
            b = (24.0 * 2 * M + 4 * 16.0 * n1 * panels_per_step) if batched else (48.0 * M + 24.0 * M + 16.0 * items)
        elif k == "fft":
            b = 2 * 16.0 * items  # read + write the grids once
        else:
            b = 0.0
        rl[k] = dict(avg_us=1e3 * avg_ms, bytes=b, gbps=(b / (avg_ms * 1e-3) / 1e9) if avg_ms > 0 else 0.0,
                     launches_per_step=v["launches"] / 10.0)
    peak = 8000.0
    # HBM bytes per launch of the dominant kernel from the PMC summary of this round -- only if that summary was taken
    # from THESE kernel sources (kernels_sha) and names this kernel; a stale number is worse than none
    traffic, traffic_src = None, None
    try:
        pj = json.load(open(os.path.join(ROOT, "profiles", "%s_%s_summary.json" % (ROUND, args.workload.lower()))))
        if pj.get("kernels_sha") == kernels_sha():
            for kname, tv in pj.get("traffic", {}).items():
                if dom_name in kname:
                    traffic = tv["hbm_bytes_per_launch"]
                    traffic_src = "profiles/%s_%s_summary.json (rocprofv3 --pmc, same kernel sources)" % (ROUND, args.workload.lower())
    except Exception:
        pass
    floor = {} if args.no_launch_floor else cfg.launch_floor()
    roofline = {"kernel": dom_name, "bound": "hbm", "achieved": rl[dom]["gbps"], "peak": peak, "unit": "GB/s",
                "frac": rl[dom]["gbps"] / peak, "traffic": traffic, "traffic_source": traffic_src, "avg_us": rl[dom]["avg_us"],
                "algorithmic_bytes_per_launch": rl[dom]["bytes"],
                "per_kernel": {k: {"avg_us": round(v["avg_us"], 3), "GB/s": round(v["gbps"], 1),
                                   "launches_per_step": v["launches_per_step"]} for k, v in rl.items()},
                # what a launch costs before it does any work, measured in this run with the same event brackets:
                # an empty kernel, the fused spread launch reduced to its step-state prologue, k_interp reduced to
                # its state loads (DESIGN.md "Launch floor")
                "launch_floor_us": floor}
    if not batched:
        roofline["note"] = ("dim >= 2: per panel one k_spread_multi / rocFFT (batch 2 x series terms) / k_hk_far triple for the lags "
                            "above r_c plus the two compressed boxes (event-timed on one stream here; the march runs them beside "
                            "the spread and the FFT); `kernel` is the launch kind with the largest event time")
        roofline["per_kernel_share"] = {k: round(acc[k]["total_ms"] / max(1e-12, sum(a["total_ms"] for a in acc.values())), 3)
                                        for k in acc if acc[k]["launches"]}
    if "spread" in roofline["per_kernel"] and batched:
        roofline["per_kernel"]["spread"]["launch_floor_us"] = floor.get("spread_prologue_only")
        # north_star: ">= 40 % of the HBM roofline in the NUFFT spread step"
        roofline["per_kernel"]["spread"]["frac"] = round(rl["spread"]["gbps"] / peak, 4)
        roofline["per_kernel"]["spread"]["target_frac"] = 0.40
        roofline["per_kernel"]["spread"]["panels_per_launch"] = round(panels_per_step, 3)
    # the dominant kernel's other roofline: vector-instruction issue.  Dynamic instructions per wave from the committed
    # PMC passes (profiles/r02_interp_instmix.json), one wave instruction per SIMD every 4 cycles, 1024 SIMDs at the
    # 2.4 GHz peak clock -- the bound that explains why the kernel sits at a fifth of the HBM peak
    try:
        mixj = json.load(open(os.path.join(ROOT, "profiles", "%s_interp_instmix.json" % ROUND)))
        mix = mixj.get(str(cfg.nufft_width)) if mixj.get("kernels_sha") == kernels_sha() else None
        if dom == "interp" and mix:
            # waves of one launch: 256 lags per workgroup of four waves, over the lags active in a step (a wave evaluates
            # every panel of the step); issue cycles per wave: 4 per FP64 instruction, 2 per other vector instruction
            # (16 / 32 lanes per clock and SIMD), 1024 SIMDs at the 2.4 GHz peak clock
            waves = 4.0 * math.ceil(st["sum_active_steps"] / max(1, st["n_steps"]) / 256.0)
            f64 = mix["fp64_fma_add_mul_per_wave"]
            cyc = 4.0 * f64 + 2.0 * (mix["valu_per_wave"] - f64)
            issue_us = waves * cyc / (1024 * 2.4e3)
            roofline["valu_issue"] = {"valu_insts_per_wave": mix["valu_per_wave"], "fp64_insts_per_wave": f64,
                                      "waves_per_launch": waves, "issue_cycles_per_wave": cyc,
                                      "issue_us_at_peak_clock": issue_us,
                                      "frac_of_issue_peak": issue_us / rl["interp"]["avg_us"]}
    except Exception:
        pass
    # whole-step view: algorithmic bytes per lag (24 + 40*abar + 48*M*Q/N) over the step time
    abar = st["sum_active"] / max(1, st["n_unique"])
    bytes_per_lag = 24.0 + 40.0 * abar + 48.0 * M * st["n_panels"] / max(1, n)  # SURVEY 8(d), per panel evaluation
    roofline["step_GB/s"] = bytes_per_lag * n / (1e3 * dt_local / args.steps * 1e-3) / 1e9
    roofline["step_frac"] = roofline["step_GB/s"] / peak
    roofline["bytes_per_lag"] = bytes_per_lag

    # ---- CPU baseline: the oracle (a port of the reference's algorithm, not FINUFFT) on the GPU box's
    # host cores: the whole workload with the oracle's CPU type-3 NUFFT, plus a small exact-direct-sum sample
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import sk_oracle as o
        ref_state, ref_found = reference_probe()
        avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        quota = cpu_quota()
        usable = min(avail, quota) if quota else avail
        # thread count of the port: the box may expose far more logical CPUs than this job's share of them (the cgroup
        # quota: an over-subscribed run is throttled -- 17x slower once); one trial run each at 16 and 64 threads, both
        # capped at what the job may use, keep the faster
        cands = sorted({min(usable, 16), min(usable, 64)})
        trial = {}
        fast_ok = w["kw"].get("dim", 1) == 1  # the oracle has no fast Hankel transform: dim >= 2 is direct sums only
        for T in cands:
            tcfg = o.AdaptiveKernelConfig(getattr(o.SDF, kind)(*prm), use_nufft=True, nthreads=T, **w["kw"])
            o.gauss_rules(tcfg.quadspec[0], tcfg.p)
            if fast_ok and len(cands) > 1:
                tc = time.perf_counter()
                o.kernel_values(tcfg, rs, k0=k0)
                trial[T] = time.perf_counter() - tc
        nthr = min(trial, key=trial.get) if trial else cands[-1]
        ocfg = o.AdaptiveKernelConfig(getattr(o.SDF, kind)(*prm), use_nufft=True, nthreads=nthr, **w["kw"])
        best, reps, oks = None, 0, None
        t_all = time.perf_counter()
        while fast_ok and (reps < 3 or (time.perf_counter() - t_all < 10.0 and reps < 20)):
            tc = time.perf_counter()
            oks, _ = o.kernel_values(ocfg, rs, k0=k0)
            dtc = time.perf_counter() - tc
            best = dtc if best is None else min(best, dtc)
            reps += 1
        ns = args.cpu_sample or (2000 if M > 10000 else 20000)
        idx = np.unique(np.r_[np.linspace(0, n - 1, ns).astype(int), n - 1])
        dcfg = o.AdaptiveKernelConfig(getattr(o.SDF, kind)(*prm), nthreads=nthr, **w["kw"])
        tc = time.perf_counter()
        dks, _ = o.kernel_values(dcfg, rs[idx], k0=k0)
        tdir = time.perf_counter() - tc
        if fast_ok:
            cpu = {"value": n / best, "unit": "r-targets/s", "cores": ocfg.nthreads, "kind": "port",
                   "sample": "all %d lags, host array in / host arrays out, oracle with its CPU type-3 NUFFT (OpenMP), best "
                             "of %d runs, %.2f s each; exact direct sums on %d of the lags: %.0f r-targets/s" % (
                                 n, reps, best, idx.size, idx.size / tdir),
                   "max_abs_diff_vs_gpu_over_k0": float(np.abs(oks - ks).max() / abs(k0)),
                   "direct_sum_sample_diff_vs_gpu_over_k0": float(np.abs(dks - ks[idx]).max() / abs(k0))}
        else:
            cpu = {"value": idx.size / tdir, "unit": "r-targets/s", "cores": dcfg.nthreads, "kind": "port",
                   "sample": "%d of the %d lags, oracle with exact direct Bessel sums (OpenMP; the oracle has no fast "
                             "Hankel transform), %.1f s" % (idx.size, n, tdir),
                   "max_abs_diff_vs_gpu_over_k0": float(np.abs(dks - ks[idx]).max() / abs(k0))}
        cpu["logical_cpus"] = avail
        cpu["cpu_quota"] = quota  # CPUs the cgroup grants at once (None: no quota): the ceiling of any thread count
        cpu["thread_trials_s"] = {str(k): round(v, 3) for k, v in trial.items()}
        cpu["reference"] = ref_state  # Julia + FINUFFT (README.md:38-48) on this box?
        cpu["reference_probe"] = ref_found
        if host is not None:  # like against like: both spans are host array -> host arrays
            cpu["gpu_host_to_host_over_cpu"] = host["value"] / cpu["value"]

    if rank == 0:
        out = {"metric": "r-targets/sec at tol=%g" % w["kw"]["tol"], "value": value, "unit": "r-targets/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
               "scaling": "strong" if args.workload == "C4" and world > 1 else "weak", "vs_baseline": None, "dtype": "f64",
               "data": "synthetic", "span": "device_resident",
               # `value` is the device-resident span (lags in HBM when the timed region starts, the bench contract);
               # SURVEY 8(d) / BASELINE.md 2.3 quote the metric host entry -> results in host memory: that figure of the
               # same run is baseline_span_value (= host_to_host.value: fresh pageable result vectors), PCIe-inclusive
               "value_span_matches_baseline": False,
               "baseline_span": "host_to_host",
               "baseline_span_value": host["value"] if host is not None else None,
               "baseline_span_ms_per_step": host["ms_per_step"] if host is not None else None,
               "config": {"workload": args.workload + ": " + w["desc"], "lags_per_gpu": n,
                          "sharding": "r-set %s over ranks, K(0) all-reduce at set-up only" % args.shard,
                          "nufft_width": cfg.nufft_width},
               "max_rel_err_vs_closed_form": max_rel_err,
               "panels": {"outer": st["n_outer"], "nufft_pairs": st["n_panels"], "abar": abar, "steps": st["n_steps"],
                          "steps_rerun": st["n_rerun"], "steps_launched": st["n_steps_launched"], "host_syncs": st["n_syncs"],
                          # launches of the last march that ran the one-panel kernels inside a full-width march / that
                          # those kernels skipped (launched again at the full width): DESIGN.md 3g
                          "one_panel_launches": kinds["one_panel_launches"], "one_panel_skipped": kinds["skipped"]},
               "rccl_ranks": world if (dist is not None and args.backend == "nccl") else 0, "per_rank": per_rank,
               "host_to_host": host, "roofline": roofline, "cpu_baseline": cpu}
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
