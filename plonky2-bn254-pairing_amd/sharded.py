"""Multi-GPU sharding of independent pairings: one process per GPU, contiguous
slices of the SoA batch per rank, NO data-path collective -- the units are independent.  The only
optional exchanges are the north star's scatter of inputs from rank 0 and gather of Fq12 outputs to
rank 0 (torch.distributed: backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

`compute(g1_soa, g2_soa, n_local)` is the per-rank hot path: the HIP engine in production
(default), anything with the same signature in tests.
"""
import importlib

import numpy as np


def shard_bounds(n, world, rank):
    """Contiguous slice [lo, hi) of rank `rank`; sizes differ by at most one."""
    lo = (n * rank) // world
    hi = (n * (rank + 1)) // world
    return lo, hi


def slice_soa(buf, words, n, lo, hi):
    """Rows [lo, hi) of every limb plane of an SoA batch (plane-major: words*4... planes of length n)."""
    planes = np.asarray(buf, dtype=np.uint64).reshape(words, n)
    return np.ascontiguousarray(planes[:, lo:hi]).reshape(-1)


def _default_compute(g1, g2, n_local):
    pkg = importlib.import_module("plonky2-bn254-pairing_amd")
    return pkg.pairing_batch(g1, g2, n_local)


def pairing_sharded(g1, g2, n, dist=None, compute=None, scatter_from_root=False, gather_to_root=True):
    """Every rank returns its slice's result; rank 0 additionally returns the gathered SoA batch
    (48*n u64) when gather_to_root.  g1/g2 must be valid on every rank unless scatter_from_root,
    in which case only rank 0's are read and the slices travel by point-to-point send/recv."""
    compute = compute or _default_compute
    if dist is None or not dist.is_initialized():
        out = compute(np.asarray(g1, dtype=np.uint64), np.asarray(g2, dtype=np.uint64), n)
        return out, out
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    lo, hi = shard_bounds(n, world, rank)
    n_local = hi - lo
    if scatter_from_root:
        if rank == 0:
            for r in range(1, world):
                rlo, rhi = shard_bounds(n, world, r)
                for buf, words in ((g1, 8), (g2, 16)):
                    t = torch.from_numpy(slice_soa(buf, words, n, rlo, rhi).view(np.int64))
                    dist.send(t, dst=r)
            l1, l2 = slice_soa(g1, 8, n, lo, hi), slice_soa(g2, 16, n, lo, hi)
        else:
            t1 = torch.empty(8 * n_local, dtype=torch.int64)
            t2 = torch.empty(16 * n_local, dtype=torch.int64)
            dist.recv(t1, src=0)
            dist.recv(t2, src=0)
            l1, l2 = t1.numpy().view(np.uint64), t2.numpy().view(np.uint64)
    else:
        l1, l2 = slice_soa(g1, 8, n, lo, hi), slice_soa(g2, 16, n, lo, hi)
    local = compute(l1, l2, n_local) if n_local else np.zeros(0, dtype=np.uint64)
    gathered = None
    if gather_to_root:
        if rank == 0:
            full = np.zeros((48, n), dtype=np.uint64)
            full[:, lo:hi] = np.asarray(local, dtype=np.uint64).reshape(48, n_local)
            for r in range(1, world):
                rlo, rhi = shard_bounds(n, world, r)
                t = torch.empty(48 * (rhi - rlo), dtype=torch.int64)
                if rhi > rlo:
                    dist.recv(t, src=r)
                    full[:, rlo:rhi] = t.numpy().view(np.uint64).reshape(48, rhi - rlo)
            gathered = full.reshape(-1)
        elif n_local:
            dist.send(torch.from_numpy(np.asarray(local, dtype=np.uint64).view(np.int64).copy()), dst=0)
    return local, gathered
