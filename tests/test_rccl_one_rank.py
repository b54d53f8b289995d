"""RCCL on the one GPU of the lease (round 5): every call bench.py's configs[4] flow and `sharded.py` make into
torch.distributed's "nccl" backend (= RCCL on ROCm), executed with a world of ONE rank.  A one-rank world moves nothing
between GPUs -- no scaling is measured here and none is claimed -- but the library is loaded, the communicator is built with
`device_id=` and the bounded timeout bench.py passes, the collectives run as device kernels on device tensors, and the grouped
point-to-point path of `sharded._P2P` / `_run` carries a rank's whole configs[4] shard (2^21 lanes: 8 + 16 + 48 plane views of
16 MiB each, one `batch_isend_irecv` group) from the rank to itself, bit for bit.

Unit that is sharded: `pairing(p, q)`, /root/reference/src/pairing.rs:20-22 (independent units).

Each test runs in a child process (its own process group; the pytest process keeps no communicator)."""
import importlib
import json
import os
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sh():
    sys.path.insert(0, ROOT)
    return importlib.import_module("plonky2-bn254-pairing_amd.sharded")


def _one_rank_worker(port, log2, q):
    from datetime import timedelta
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, timeout=timedelta(seconds=120))
    sh = _sh()
    res = {"backend": str(dist.get_backend()), "moves_device_memory": bool(sh._moves_device_memory(dist))}
    # --- the collectives of bench.py (run_rank: all_reduce MAX of the elapsed time, all_gather of the per-rank record,
    #     broadcast of rank 0's verdict, barrier), on device tensors as there
    t = torch.tensor([1.25], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    res["all_reduce"] = float(t.item()) == 1.25
    mine = torch.tensor([1.0, 2.0, 3.0, 4.0], dtype=torch.float64, device=dev)
    allr = [torch.zeros_like(mine)]
    dist.all_gather(allr, mine)
    res["all_gather"] = bool(torch.equal(allr[0], mine))
    v = torch.tensor([3], dtype=torch.int32, device=dev)
    dist.broadcast(v, src=0)
    res["broadcast"] = int(v.item()) == 3
    dist.barrier()
    # --- the grouped point-to-point path: a whole configs[4] shard from this rank to itself, plane by plane, ONE group
    n = 1 << log2
    g = torch.Generator(device=dev).manual_seed(0xB2540055)
    src = {w: torch.randint(-(1 << 62), 1 << 62, (w * n,), dtype=torch.int64, device=dev, generator=g) for w in (8, 16, 48)}
    dst = {w: torch.zeros(w * n, dtype=torch.int64, device=dev) for w in (8, 16, 48)}
    # ragged column window, as a peer's slice of a bigger batch would be: columns [lo, hi) of every plane
    lo, hi = 0, n
    ops = []
    for w in (8, 16, 48):
        ops += [sh._P2P(dist, "send", pl, 0) for pl in sh._planes(src[w], w, n, lo, hi)]
        ops += [sh._P2P(dist, "recv", pl, 0) for pl in sh._planes(dst[w], w, n, lo, hi)]
    res["p2p_ops"] = len(ops)
    res["p2p_bounced_through_host"] = any(x.bounce is not None for x in ops)
    res["plane_bytes"] = 8 * (hi - lo)
    sh._run(dist, ops)
    torch.cuda.synchronize(dev)
    res["p2p_equal"] = all(bool(torch.equal(src[w], dst[w])) for w in (8, 16, 48))
    # a strict sub-window of the columns (non-zero offset into every plane), fresh destination
    lo2, hi2 = n // 3 + 1, n - 5
    d2 = torch.zeros(48 * n, dtype=torch.int64, device=dev)
    ops = [sh._P2P(dist, "send", pl, 0) for pl in sh._planes(src[48], 48, n, lo2, hi2)]
    ops += [sh._P2P(dist, "recv", pl, 0) for pl in sh._planes(d2, 48, n, lo2, hi2)]
    sh._run(dist, ops)
    torch.cuda.synchronize(dev)
    a, b = src[48].view(48, n), d2.view(48, n)
    res["p2p_window_equal"] = bool(torch.equal(a[:, lo2:hi2], b[:, lo2:hi2])) and int(b[:, :lo2].abs().sum()) == 0 and int(b[:, hi2:].abs().sum()) == 0
    # --- the functions bench.py calls, end to end with one rank: scatter (local copy), HIP engine, gather
    m = 3 * 256 + 7
    pk = H.pkg()
    st = torch.cuda.current_stream(dev)
    g1 = torch.zeros(8 * m, dtype=torch.int64, device=dev)
    g2 = torch.zeros(16 * m, dtype=torch.int64, device=dev)
    pk.generate_pairs_dev(0xB2540056, g1, g2, m, 0, st)
    local, gathered = sh.pairing_sharded(g1, g2, m, dist=dist, scatter_from_root=True, chunk=512, device=dev)
    ref = torch.empty(48 * m, dtype=torch.int64, device=dev)
    pk.pairing_batch_dev(g1, g2, ref, m, 0, st)
    pk.last_status(0, st)
    res["sharded_equal"] = bool(torch.equal(gathered, ref)) and bool(torch.equal(local, ref))
    dist.barrier()
    dist.destroy_process_group()
    q.put(res)


@pytest.mark.gpu
def test_rccl_collectives_and_grouped_p2p_with_one_rank():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_one_rank_worker, args=(35000 + os.getpid() % 2000, 21, q))
    p.start()
    res = q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert "nccl" in res["backend"].lower() and res["moves_device_memory"] is True
    assert res["all_reduce"] and res["all_gather"] and res["broadcast"]
    assert res["p2p_ops"] == 2 * (8 + 16 + 48) and res["plane_bytes"] == 16 << 20
    assert res["p2p_bounced_through_host"] is False          # device memory travels as device memory under RCCL
    assert res["p2p_equal"] and res["p2p_window_equal"] and res["sharded_equal"]


@pytest.mark.gpu
def test_bench_configs4_flow_over_rccl_with_one_rank():
    """bench.py's own N > 1 code path (process group with `device_id` and the bounded timeout, scatter, timed steps, the
    all_reduce / all_gather / broadcast / barrier sequence, gather, oracle gate) over backend nccl with a world of one rank, at the
    configs[4] shard size (2^21 lanes)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_PORT=str(37000 + os.getpid() % 2000))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--dist-at-one", "--steps", "2", "--warmup", "1", "--no-power"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-1000:], p.stderr[-3000:])
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 1 and rec["rccl_ranks"] == 1 and rec["dist_backend"] == "nccl" and rec["verified_vs_oracle"] is True
    assert rec["config"]["pairings_per_gpu"] == 1 << 21
    assert len(rec["per_rank"]["kernel_ms_avg"]) == 1 and rec["per_rank"]["kernel_ms_avg"][0] > 0
    assert "exchange" in rec and "extra" not in rec and "cpu_baseline" not in rec


def test_a_failing_rank_ends_the_job_with_a_reason():
    """A rank that dies before the rendezvous (here: rank 1 of a two-rank gloo job on the stand-in engine, killed by an injected
    fault) must end the job quickly with its reason on stderr -- not hold rank 0 in `init_process_group` for ten minutes."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(BENCH_TEST_ENGINE="fake_engine:Engine", BENCH_DIST_BACKEND="gloo", OMP_NUM_THREADS="1", BENCH_TEST_FAIL_RANK="1",
               BENCH_DIST_TIMEOUT_S="60")
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--log2-batch", "4"],
                       env=env, capture_output=True, text=True, timeout=400)
    assert p.returncode != 0
    assert "rank 1 of 2 failed" in p.stderr and "injected fault" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert time.time() - t0 < 200
