"""Multi-rank sharding: slices, scatter from rank 0, compute on the shards, gather to rank 0 -- the exchange
steps sit between launches (the pairing kernel fills the chip: profiles/r03_coresidency.txt).

  * world_size-2 / 3 `gloo` runs on CPU tensors with an injected per-rank compute (the CPU oracle) -- the sharding logic is
    what is under test here;
  * `-m gpu`, two ranks on ONE GPU: the same function bodies with the HIP engine on device-resident shards; gloo carries the
    slices (device tensors bounce through pinned host buffers -- RCCL refuses two ranks on one device): `pairing_sharded`
    directly, and bench.py's whole configs[4] flow (`--gpus 2`: generate on rank 0, scatter, timed HIP compute per rank,
    gather, oracle gate on the peer's slice);
  * `-m gpu`, backend "nccl" (RCCL) on device tensors, one rank per visible GPU -- skipped on a one-GPU box, where it would
    move nothing;
  * bench.py's own launcher (`--gpus 2` with no WORLD_SIZE in the environment) on CPU with a stand-in engine."""
import importlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sh():
    sys.path.insert(0, ROOT)
    return importlib.import_module("plonky2-bn254-pairing_amd.sharded")


def _oracle_compute(g1, g2, m):
    """Stand-in for the HIP engine in the CPU tests: same signature (SoA int64 tensors in, SoA int64 tensor out)."""
    pk = H.pkg()
    a = pk.layout.to_aos(g1.numpy().view(np.uint64), 8)
    b = pk.layout.to_aos(g2.numpy().view(np.uint64), 16)
    out = H.oracle_pairing(a, b, m)
    return torch.from_numpy(pk.layout.to_soa(out, 48).view(np.int64).copy())


def _worker(rank, world, port, n, g1, g2, want, scatter, chunk, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sh = _sh()
    a, b = (g1, g2) if (rank == 0 or not scatter) else (None, None)
    local, gathered = sh.pairing_sharded(a, b, n, dist=dist, compute=_oracle_compute, scatter_from_root=scatter, chunk=chunk,
                                         device=torch.device("cpu"))
    lo, hi = sh.shard_bounds(n, world, rank)
    ok = torch.equal(local.view(48, hi - lo), want.view(48, n)[:, lo:hi])
    if rank == 0:
        ok = ok and torch.equal(gathered, want)
    else:
        ok = ok and gathered is None
    # the stand-alone scatter / gather pair bench.py uses for configs[4]
    l1 = torch.empty(8 * (hi - lo), dtype=torch.int64)
    l2 = torch.empty(16 * (hi - lo), dtype=torch.int64)
    sh.scatter_inputs(g1 if rank == 0 else None, g2 if rank == 0 else None, n, l1, l2, dist)
    ok = ok and torch.equal(l1.view(8, hi - lo), g1.view(8, n)[:, lo:hi]) and torch.equal(l2.view(16, hi - lo), g2.view(16, n)[:, lo:hi])
    full = sh.gather_outputs(local, n, dist)
    ok = ok and (torch.equal(full, want) if rank == 0 else full is None)
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def _spawn(world, n, scatter, chunk, seed):
    pk = H.pkg()
    P, Q = H.subgroup_points(max(n, 1), seed=seed)
    g1a, g2a = H.g1_aos(P[:n]), H.g2_aos(Q[:n])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64).copy())
    want = t(pk.layout.to_soa(H.oracle_pairing(g1a, g2a, n), 48)) if n else torch.empty(0, dtype=torch.int64)
    g1, g2 = t(pk.layout.to_soa(g1a, 8)), t(pk.layout.to_soa(g2a, 16))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() * 7 + n * 13 + world * 101 + (1 if scatter else 0)) % 3000
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, g1, g2, want, scatter, chunk, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(r, True) for r in range(world)]


@pytest.mark.parametrize("scatter", [False, True])
def test_sharded_two_ranks(scatter):
    """ragged: ranks get 2 and 3 pairings; chunk = 2 lanes, so rank 1 sends a 2-lane and a 1-lane chunk"""
    _spawn(2, 5, scatter, chunk=2, seed=31)


def test_sharded_three_ranks_with_an_empty_slice():
    """n = 2 over 3 ranks: rank 0's slice is empty (scatter and gather skip it consistently on both sides)"""
    _spawn(3, 2, True, chunk=8, seed=32)


def test_shard_bounds_cover_batch():
    sh = _sh()
    for n in (0, 1, 7, 8, 65536, (1 << 24)):
        for w in (1, 2, 3, 4, 8):
            b = [sh.shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
    # BASELINE configs[4]: 2^24 over 8 GPUs = 2^21 each
    assert [hi - lo for lo, hi in (sh.shard_bounds(1 << 24, 8, r) for r in range(8))] == [1 << 21] * 8


def test_bench_launcher_two_ranks_cpu():
    """`python bench.py --gpus 2` with no torch.distributed environment starts two ranks itself (a child torch.distributed.run,
    before anything touches a GPU) and runs the configs[4] flow -- scatter from rank 0, timed compute on the shards, gather --
    here over gloo with a stand-in engine (tests/fake_engine.py)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(BENCH_TEST_ENGINE="fake_engine:Engine", BENCH_DIST_BACKEND="gloo", OMP_NUM_THREADS="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--log2-batch", "6"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["rccl_ranks"] == 2 and rec["steps"] == 2 and rec["scaling"] == "weak"
    assert rec["config"]["pairings_per_gpu"] == 64 and rec["config"]["pairings_total"] == 128
    assert rec["exchange"]["bytes_scattered"] == 192 * 64 and rec["exchange"]["bytes_gathered"] == 384 * 64
    assert "TEST ENGINE" in rec["data"]
    # a world size that contradicts --gpus is an error, not a silent one-rank run
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env2,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)


def test_bench_launcher_eight_ranks_cpu():
    """The shape of the driver's 8-GPU run (`--gpus 8`: BASELINE.json configs[4], 2^24 over 8 ranks), rehearsed on CPU over gloo
    with the stand-in engine: eight ranks, shard bounds, the 7-peer grouped plane-wise scatter and gather, every peer's slice
    checked on rank 0 (the gathered batch equals the whole-batch recomputation), per-rank kernel times and memory in the line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(BENCH_TEST_ENGINE="fake_engine:Engine", BENCH_DIST_BACKEND="gloo", OMP_NUM_THREADS="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--log2-batch", "5"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 8 and rec["rccl_ranks"] == 8 and rec["scaling"] == "weak"
    assert rec["config"]["pairings_per_gpu"] == 32 and rec["config"]["pairings_total"] == 256
    assert rec["exchange"]["bytes_scattered"] == 192 * 7 * 32 and rec["exchange"]["bytes_gathered"] == 384 * 7 * 32
    assert rec["gathered_equals_whole_batch_recomputation"] is True
    pr = rec["per_rank"]
    assert all(len(pr[k]) == 8 for k in ("kernel_ms_avg", "kernel_ms_min", "kernel_ms_max", "peak_device_bytes"))
    assert rec["value"] > 0 and abs(rec["value"] - 256 * 2 / (rec["ms_per_step"] * 2e-3)) < 1e-6 * rec["value"]


@pytest.mark.parametrize("mode", ["peer", "all", "op"])
def test_bench_p2p_group_modes_three_ranks_cpu(mode):
    """BENCH_P2P_GROUP switches how the scatter / gather transfers are grouped (one batch_isend_irecv group per peer -- the default --, one per
    step, or one operation at a time) without a code change: same bytes, same places -- the gathered batch equals the whole-batch
    recomputation in every mode, and the N > 1 line says which mode ran, what share of a step the exchange would be, and how long rank 0
    kept its peers waiting after the timed steps."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(BENCH_TEST_ENGINE="fake_engine:Engine", BENCH_DIST_BACKEND="gloo", OMP_NUM_THREADS="1", BENCH_P2P_GROUP=mode)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "1", "--warmup", "0", "--log2-batch", "5"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    rec = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert rec["gathered_equals_whole_batch_recomputation"] is True
    ex = rec["exchange"]
    assert ex["p2p_group"] == mode and 0.0 < ex["exchange_share_of_step"] < 1.0
    assert abs(rec["value_incl_exchange"] - rec["config"]["pairings_total"] / ((rec["ms_per_step"] + ex["scatter_ms"] + ex["gather_ms"]) * 1e-3)) < 1e-6 * rec["value"]
    post = rec["rank0_post_steps_s"]
    assert 0.0 <= post["seconds"] < post["limit"] == 30.0


def test_bench_unknown_p2p_group_mode_is_an_error():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(BENCH_TEST_ENGINE="fake_engine:Engine", BENCH_DIST_BACKEND="gloo", OMP_NUM_THREADS="1", BENCH_P2P_GROUP="pairs", BENCH_DIST_TIMEOUT_S="30")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--log2-batch", "4"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "p2p group mode" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


@pytest.mark.parametrize("dead", [1, 0])
def test_bench_rank_dying_inside_the_gather_ends_the_job_with_its_reason(dead):
    """A rank that dies AFTER the timed steps, inside gather_outputs (its peer is blocked in the matching receive / send), must end the whole
    job within the collective timeout, with its reason on stderr and no measurement line -- not leave the other rank waiting."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(BENCH_TEST_ENGINE="fake_engine:Engine", BENCH_DIST_BACKEND="gloo", OMP_NUM_THREADS="1", BENCH_TEST_FAIL_IN_GATHER=str(dead),
               BENCH_DIST_TIMEOUT_S="60")
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--log2-batch", "4"],
                       env=env, capture_output=True, text=True, timeout=300)
    dt = time.time() - t0
    assert p.returncode != 0
    assert "BENCH_TEST_FAIL_IN_GATHER" in p.stderr and f"rank {dead} of 2 failed" in p.stderr, p.stderr[-2000:]
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert dt < 60 + 45, f"the job took {dt:.0f} s to end"          # the collective timeout + torchrun's own teardown


def test_bench_refuses_the_engine_hook_outside_pytest():
    """BENCH_TEST_ENGINE is honoured only under pytest: a plain environment cannot make bench.py print a `value` from a stand-in."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "PYTEST_CURRENT_TEST")}
    env.update(BENCH_TEST_ENGINE="fake_engine:Engine", BENCH_DIST_BACKEND="gloo", OMP_NUM_THREADS="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--log2-batch", "4"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "refused outside pytest" in (p.stderr + p.stdout)
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


# ------------------------------------------------------------------------------------------------ GPU: nccl = RCCL
def _nccl_worker(rank, world, port, n, chunk, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    sh = _sh()
    pk = H.pkg()
    g1 = g2 = None
    if rank == 0:
        g1 = torch.zeros(8 * n, dtype=torch.int64, device=dev)
        g2 = torch.zeros(16 * n, dtype=torch.int64, device=dev)
        pk.generate_pairs_dev(0xB2540006, g1, g2, n, rank, torch.cuda.current_stream(dev))
        pk.last_status(rank, torch.cuda.current_stream(dev))
    local, gathered = sh.pairing_sharded(g1, g2, n, dist=dist, scatter_from_root=True, chunk=chunk, device=dev)
    pk.last_status(rank, torch.cuda.current_stream(dev))
    ok = True
    if rank == 0:
        ref = torch.empty(48 * n, dtype=torch.int64, device=dev)
        pk.pairing_batch_dev(g1, g2, ref, n, rank, torch.cuda.current_stream(dev))
        pk.last_status(rank, torch.cuda.current_stream(dev))
        ok = torch.equal(gathered, ref)
        pos = [0, n // 2, n - 1]
        g1h = g1.view(8, n)[:, pos].cpu().numpy().view(np.uint64).reshape(-1).copy()
        g2h = g2.view(16, n)[:, pos].cpu().numpy().view(np.uint64).reshape(-1).copy()
        want = H.oracle_pairing(pk.layout.to_aos(g1h, 8), pk.layout.to_aos(g2h, 16), len(pos), threads=3)
        got = gathered.view(48, n)[:, pos].cpu().numpy().view(np.uint64).reshape(-1).copy()
        ok = ok and np.array_equal(pk.layout.to_aos(got, 48), want)
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_sharded_nccl_on_visible_gpus():
    """backend nccl (RCCL), device tensors, the HIP engine on each rank's own device and stream: one rank per visible GPU."""
    world = torch.cuda.device_count()
    if world < 2:
        pytest.skip("one visible GPU: RCCL would move nothing (the two-rank flow runs over gloo on the shared GPU instead)")
    n = 3 * 4096 + 77
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31000 + os.getpid() % 2000
    procs = [ctx.Process(target=_nccl_worker, args=(r, world, port, n, 4096, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(r, True) for r in range(world)]


def _shared_gpu_worker(rank, world, port, n, chunk, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sh = _sh()
    pk = H.pkg()
    st = torch.cuda.current_stream(dev)
    g1 = g2 = None
    if rank == 0:
        g1 = torch.zeros(8 * n, dtype=torch.int64, device=dev)
        g2 = torch.zeros(16 * n, dtype=torch.int64, device=dev)
        pk.generate_pairs_dev(0xB2540007, g1, g2, n, 0, st)
        pk.last_status(0, st)
    local, gathered = sh.pairing_sharded(g1, g2, n, dist=dist, compute=sh.hip_compute(0), scatter_from_root=True, chunk=chunk, device=dev)
    pk.last_status(0, st)
    lo, hi = sh.shard_bounds(n, world, rank)
    ok = local.is_cuda and local.numel() == 48 * (hi - lo)
    if rank == 0:
        ref = torch.empty(48 * n, dtype=torch.int64, device=dev)
        pk.pairing_batch_dev(g1, g2, ref, n, 0, st)
        pk.last_status(0, st)
        ok = ok and gathered.is_cuda and torch.equal(gathered, ref)
        pos = [0, n // 2 + 3, n - 1]                      # the last two lie in rank 1's slice
        g1h = g1.view(8, n)[:, pos].cpu().numpy().view(np.uint64).reshape(-1).copy()
        g2h = g2.view(16, n)[:, pos].cpu().numpy().view(np.uint64).reshape(-1).copy()
        want = H.oracle_pairing(pk.layout.to_aos(g1h, 8), pk.layout.to_aos(g2h, 16), len(pos), threads=3)
        got = gathered.view(48, n)[:, pos].cpu().numpy().view(np.uint64).reshape(-1).copy()
        ok = ok and np.array_equal(pk.layout.to_aos(got, 48), want)
    else:
        ok = ok and gathered is None
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_pairing_sharded_two_ranks_on_one_gpu():
    """Two ranks, HIP engine, device-resident shards on the one visible GPU; gloo carries the slices through pinned host
    buffers.  Ragged: 3 launches of <= 1024 lanes per rank."""
    n = 2 * 2500 + 1
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33000 + os.getpid() % 2000
    procs = [ctx.Process(target=_shared_gpu_worker, args=(r, 2, port, n, 1024, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


@pytest.mark.gpu
def test_bench_configs4_flow_two_ranks_on_one_gpu():
    """bench.py --gpus 2 on the HIP engine (BASELINE.json configs[4] at 2^14 per rank): rank 0 generates 2 x 2^14 pairs on
    the device, scatters, both ranks time their device-resident shard, outputs are gathered and the oracle gate checks
    positions of rank 1's slice.  BENCH_SHARE_GPU maps both ranks to the one GPU; gloo carries the slices."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(BENCH_SHARE_GPU="1", BENCH_DIST_BACKEND="gloo", OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--log2-batch", "14",
                        "--no-power"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-1000:], p.stderr[-3000:])
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["rccl_ranks"] == 2 and rec["dist_backend"] == "gloo" and rec["verified_vs_oracle"] is True
    assert rec["config"]["pairings_per_gpu"] == 1 << 14 and rec["config"]["pairings_total"] == 1 << 15
    assert rec["exchange"]["bytes_scattered"] == 192 << 14 and rec["exchange"]["bytes_gathered"] == 384 << 14
    assert "TEST ENGINE" not in rec["data"] and rec["value"] > 0
    assert rec["config"]["ranks_share_one_gpu"] is True


def test_bench_power_sampler_reads_rocm_smi(tmp_path, monkeypatch):
    """bench.py's package-power sampler: parses `rocm-smi --showpower / --showmaxpower` (a stand-in script on PATH here),
    drops the ramp samples, and reports nothing when the tool is absent."""
    import time
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    fake = tmp_path / "rocm-smi"
    fake.write_text("#!/bin/sh\ncase \"$1\" in\n"
                    "  --showpower) echo 'GPU[0]\t\t: Current Socket Graphics Package Power (W): 1321.0';;\n"
                    "  --showmaxpower) echo 'GPU[0]\t\t: Max Graphics Package Power (W): 1400.0';;\nesac\n")
    fake.chmod(0o755)
    monkeypatch.setenv("PATH", str(tmp_path) + os.pathsep + os.environ["PATH"])
    with bench.PowerSampler() as ps:
        time.sleep(0.5)
    s = ps.summary()
    assert s and s["avg_w"] == 1321.0 and s["max_w"] == 1321.0 and s["limit_w"] == 1400.0 and s["samples"] >= 1
    monkeypatch.setenv("PATH", str(tmp_path / "nowhere"))
    with bench.PowerSampler() as ps:
        time.sleep(0.1)
    assert ps.summary() is None


@pytest.mark.gpu
def test_rank0_generates_the_whole_configs4_batch_in_one_call():
    """What rank 0 of the 8-GPU run does before the scatter (bench.py, BASELINE.json configs[4]): 2^24 pairs from ONE bn254_generate_pairs_dev call
    (3.2 GB; the plane offsets reach 2^31 bytes) -- points at the ends and in the middle equal [s]G1, [t]G2 for the generator's stated scalars, and
    the pairings of the LAST rank's 2^21-lane slice equal the oracle."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "exp", "gen_2_24.py")], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "ok: 2^24 pairs" in p.stdout, (p.stdout[-500:], p.stderr[-1500:])
