"""world_size-2 gloo test (CPU) of the multi-GPU sharding driver: slices, scatter, gather.
The per-rank compute is injected (the CPU oracle) -- the sharding logic is what is under test."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle_compute(g1, g2, n_local):
    pk = H.pkg()
    out = H.oracle_pairing(pk.layout.to_aos(g1, 8), pk.layout.to_aos(g2, 16), n_local)
    return pk.layout.to_soa(out, 48)


def _worker(rank, world, port, n, g1, g2, want, scatter, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    import importlib
    sh = importlib.import_module("plonky2-bn254-pairing_amd.sharded")
    a, b = (g1, g2) if (rank == 0 or not scatter) else (None, None)
    local, gathered = sh.pairing_sharded(a, b, n, dist=dist, compute=_oracle_compute, scatter_from_root=scatter)
    lo, hi = sh.shard_bounds(n, world, rank)
    ok = np.array_equal(np.asarray(local).reshape(48, hi - lo), want.reshape(48, n)[:, lo:hi])
    if rank == 0:
        ok = ok and np.array_equal(gathered, want)
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("scatter", [False, True])
def test_sharded_two_ranks(scatter):
    pk = H.pkg()
    n = 5                                   # ragged: ranks get 2 and 3 pairings
    P, Q = H.subgroup_points(n, seed=31)
    g1a, g2a = H.g1_aos(P), H.g2_aos(Q)
    want = pk.layout.to_soa(H.oracle_pairing(g1a, g2a, n), 48)
    g1, g2 = pk.layout.to_soa(g1a, 8), pk.layout.to_soa(g2a, 16)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + (1 if scatter else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, g1, g2, want, scatter, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def test_shard_bounds_cover_batch():
    sys.path.insert(0, ROOT)
    import importlib
    sh = importlib.import_module("plonky2-bn254-pairing_amd.sharded")
    for n in (0, 1, 7, 8, 65536):
        for w in (1, 2, 4, 8):
            b = [sh.shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
