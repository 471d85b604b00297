"""Multi-GPU plumbing for the kernel_values path: one process per GPU, the r-set is sharded
(no data-path collective); the only shared quantity is the K(0) normaliser (src/adaptive.jl:97),
whose quadrature is split over the ranks and summed with one all-reduce (RCCL on GPUs, gloo in
the CPU tests)."""
import ctypes as C

import numpy as np

from . import _lib as L


def shard_bounds(n, rank, world):
    """Contiguous block [lo, hi) of n sorted lags owned by `rank` (sizes differ by at most 1)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_strided(xs, rank, world):
    """Every world-th lag: all ranks see the same r-range, hence the same panel schedule."""
    return np.ascontiguousarray(np.asarray(xs)[rank::world])


def k0_part(args, rank, world):
    """This rank's share of compute_k0's integral over t in [rank/world, (rank+1)/world), w = t/(1-t)."""
    part = C.c_double()
    code = L.load().sk_host_compute_k0_part(C.byref(args), float(rank) / world, float(rank + 1) / world,
                                            C.byref(part))
    if code != 0:
        raise RuntimeError(L.last_error())
    return part.value


def shared_k0(cfg, dist=None, device=None):
    """K(0) of `cfg`, computed cooperatively: each rank integrates its slice, one all-reduce sums."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return k0_part(cfg._args, 0, 1)
    import torch
    t = torch.tensor([k0_part(cfg._args, dist.get_rank(), dist.get_world_size())], dtype=torch.float64,
                     device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_sharded(local_vals, n_total, rank, world, dist, strided=True):
    """Reassembles per-rank results into the global order (used by tests and by callers that want
    the full vector on every rank; the bench never gathers)."""
    import torch
    sizes = [len(range(r, n_total, world)) if strided else (shard_bounds(n_total, r, world)[1] -
                                                             shard_bounds(n_total, r, world)[0]) for r in range(world)]
    mx = max(sizes)
    buf = torch.zeros(mx, dtype=torch.float64)
    buf[:len(local_vals)] = torch.as_tensor(np.asarray(local_vals), dtype=torch.float64)
    outs = [torch.zeros(mx, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(outs, buf)
    full = np.zeros(n_total)
    for r in range(world):
        v = outs[r][:sizes[r]].numpy()
        if strided:
            full[r::world] = v
        else:
            lo, hi = shard_bounds(n_total, r, world)
            full[lo:hi] = v
    return full
