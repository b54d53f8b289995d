#!/usr/bin/env python3
"""bench.py -- BN254 pairings/sec on MI355X (BASELINE.json metric), one JSON line on rank 0.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU)

A step = one pass of the hot path (`pairing()` = final_exp_native(miller_loop_native(Q, P)),
src/pairing.rs:20-22) over one batch of synthetic subgroup points that is already resident in
HBM.  Workload at N = 1: BASELINE.json configs[1], 2^16 independent pairings.  For N > 1 every
rank owns its own 2^16 batch (independent units, no data-path collective): weak scaling.

The JSON line also carries
  roofline     -- integer-VALU bound (2,286,160 mul32 of algorithmic work per pairing)
                  against the calibrated v_mad_u64_u32 issue peak of gfx950 (tools/valu_calib.hip,
                  profiles/valu_calib_r01.txt); kernel time from HIP events on the launch stream
  cpu_baseline -- the C oracle (a port of the reference's schedule) timed on the host cores on a
                  bounded sample, rank 0, N = 1 only.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W_MUL32_PER_PAIRING = 2_286_160          # 16,810 fqmul x 136 mul32
W_FQMUL_PER_PAIRING = 16_810
# Calibrated peak (tools/valu_calib.hip on MI355X, profiles/valu_calib_r01.txt): v_mad_u64_u32 issues at
# ~554 G wave-instructions/s chip-wide at 8 waves/SIMD = 35.5 T mul32/s.  (SURVEY's nominal quarter-rate
# assumption was 9.83 T mul32/s; the measured rate is 3.6x that.)
PEAK_MUL32_PER_S = 554e9 * 64
NOMINAL_PEAK_MUL32_PER_S = 9.83e12
LOG2_BATCH = 16
PMC_SUMMARY = os.path.join(ROOT, "profiles", "r01_final_pmc.json")     # rocprofv3 --pmc passes of this same command (tools/gpu_profile.sh)


def pmc_traffic(log2_batch):
    """HBM bytes per launch of the dominant kernel from the committed PMC summary (FETCH_SIZE/WRITE_SIZE, corrected as the
    MI355X guide prescribes); None when the summary is missing or was taken at another batch size."""
    try:
        with open(PMC_SUMMARY) as f:
            notes = json.load(f)["_notes"]
        if log2_batch != LOG2_BATCH:
            return None, None
        return notes["hbm_bytes_per_launch_corrected"], notes.get("valu_insts_per_wave")
    except Exception:
        return None, None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--log2-batch", type=int, default=LOG2_BATCH, help="pairings per GPU per step = 2^k (default: configs[1])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--extra", action="store_true", help="also time configs[2] (2^20 pairings) and configs[3] (Groth16: 2^18 groups x 4 pairs); "
                                                        "reported under \"extra\" in the same JSON line")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def cpu_baseline(pkg, g1_soa, g2_soa, n_avail, seconds):
    """Oracle (port of the reference schedule) on the host cores, bounded sample of the same inputs."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers as H
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    chunk = 16 * cores
    g1a = pkg.layout.to_aos(g1_soa, 8)
    g2a = pkg.layout.to_aos(g2_soa, 16)
    H.oracle_pairing(g1a[:8], g2a[:16], 1)          # one-time constant tables outside the timed region
    done, t0 = 0, time.time()
    while done + chunk <= n_avail and (time.time() - t0) < seconds:
        H.oracle_pairing(g1a[8 * done: 8 * (done + chunk)], g2a[16 * done: 16 * (done + chunk)], chunk, threads=cores)
        done += chunk
    dt = time.time() - t0
    return {"value": done / dt, "unit": "pairings/s", "cores": cores, "kind": "port",
            "sample": f"first {done} pairings of the same batch, C oracle (reference schedule: affine G2 steps with an inversion "
                      f"per step, NAF pow with divisions), {cores} pthreads, {dt:.1f} s"}


def extra_configs(pkg, torch, dev, local_rank, stream):
    """BASELINE.json configs[2] and configs[3] on one GPU (not the headline `value`)."""
    out = {}

    def timed(fn, reps=2):
        fn()
        torch.cuda.synchronize(dev)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for _ in range(reps):
            fn()
        b.record(stream)
        torch.cuda.synchronize(dev)
        return a.elapsed_time(b) / reps

    n = 1 << 20
    g1 = torch.zeros(8 * n, dtype=torch.int64, device=dev)
    g2 = torch.zeros(16 * n, dtype=torch.int64, device=dev)
    o = torch.zeros(48 * n, dtype=torch.int64, device=dev)
    pkg.generate_pairs_dev(0xB2540002, g1, g2, n, device=local_rank, stream=stream)
    ms = timed(lambda: pkg.pairing_batch_dev(g1, g2, o, n, device=local_rank, stream=stream))
    out["2^20 independent pairings"] = {"ms": ms, "pairings_per_s": n / (ms * 1e-3)}
    groups, k = 1 << 18, 4
    o2 = torch.zeros(48 * groups, dtype=torch.int64, device=dev)
    ms = timed(lambda: pkg.multi_pairing_batch_dev(g1, g2, o2, groups, k, True, device=local_rank, stream=stream))
    out["Groth16 shape: 2^18 groups x 4 pairs, shared final exp"] = {"ms": ms, "groups_per_s": groups / (ms * 1e-3),
                                                                      "pairs_per_s": groups * k / (ms * 1e-3)}
    pkg.last_status(local_rank, stream)
    return out


def main():
    args = parse()
    import numpy as np
    import torch
    import __graft_entry__
    pkg = __graft_entry__.build()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback by design)")
    if os.environ.get("BENCH_SHARE_GPU"):          # test hook: several ranks on one GPU (RCCL refuses that: use gloo)
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    n = 1 << args.log2_batch
    g1 = torch.zeros(8 * n, dtype=torch.int64, device=dev)
    g2 = torch.zeros(16 * n, dtype=torch.int64, device=dev)
    out = torch.zeros(48 * n, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream(dev)
    pkg.generate_pairs_dev(0xB2540001 + 7919 * rank, g1, g2, n, device=local_rank, stream=stream)
    pkg.last_status(local_rank, stream)

    def step():
        pkg.pairing_batch_dev(g1, g2, out, n, device=local_rank, stream=stream)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    if dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in evs:
        a.record(stream)
        step()
        b.record(stream)
    torch.cuda.synchronize(dev)
    if dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    pkg.last_status(local_rank, stream)
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kern_ms = sorted(a.elapsed_time(b) for a, b in evs)
    kern_avg_ms = sum(kern_ms) / len(kern_ms)

    if rank == 0:
        total = n * world * args.steps
        value = total / elapsed
        per_gpu_rate = n / (kern_avg_ms * 1e-3)
        achieved = per_gpu_rate * W_MUL32_PER_PAIRING
        traffic, insts_per_wave = pmc_traffic(args.log2_batch)
        rec = {
            "metric": "BN254 pairings/sec (whole node)", "value": value, "unit": "pairings/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "i32x10 (254-bit Montgomery, signed radix-2^27 limbs, integer VALU)",
            "data": "synthetic: on-device [s]G1, [t]G2 subgroup points, SplitMix64 scalars, seed 0xB2540001",
            "config": {"workload": f"2^{args.log2_batch} independent pairings per GPU per step "
                                   f"(BASELINE.json configs[1]; pairing() = final_exp_native(miller_loop_native))",
                       "pairings_per_gpu": n, "layout": "SoA limb-major u64 Montgomery, inputs resident in HBM"},
            "roofline": {"bound": "valu-int32-mul", "achieved": achieved / 1e12, "peak": PEAK_MUL32_PER_S / 1e12, "unit": "T mul32/s",
                         "frac": achieved / PEAK_MUL32_PER_S, "traffic": traffic,
                         "traffic_note": "HBM bytes per launch, PMC FETCH_SIZE x2 + WRITE_SIZE (profiles/r01_final_pmc.json, separate --pmc passes "
                                         "of this command); algorithmic bytes per launch = 576 B x pairings; the excess is the final "
                                         "exponentiation's scratch registers (~2 % of 8 TB/s)",
                         "algorithmic_bytes_per_launch": 576 * n,
                         "kernel": "k3_pairing", "kernel_ms_avg": kern_avg_ms, "kernel_ms_min": kern_ms[0],
                         "work_per_unit": f"{W_MUL32_PER_PAIRING} mul32 = {W_FQMUL_PER_PAIRING} fqmul x 136 per pairing (SURVEY.md 8d)",
                         "frac_of_nominal_quarter_rate_peak": achieved / NOMINAL_PEAK_MUL32_PER_S,
                         "valu_issue": None if not insts_per_wave else {
                             "wave_instr_per_s": insts_per_wave * (n / 64) / (kern_avg_ms * 1e-3), "peak_wave_instr_per_s": 1024 * 2.4e9 / 4,
                             "frac": insts_per_wave * (n / 64) / (kern_avg_ms * 1e-3) / (1024 * 2.4e9 / 4),
                             "note": "all VALU instructions the kernel issues (SQ_INSTS_VALU per wave) against 1024 SIMDs x 2.4 GHz / 4 cycles"},
                         "hbm": {"bound": "hbm", "achieved": 576 * n / (kern_avg_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                                 "frac": 576 * n / (kern_avg_ms * 1e-3) / 8e12, "traffic": traffic,
                                 "note": "the schema's HBM view of the same kernel: algorithmic bytes (576 B/pairing) over the launch time "
                                         "against 8 TB/s -- three orders of magnitude from the bound (SURVEY.md 8d: integer-VALU bound)"},
                         "hbm_note": "algorithmic HBM bytes are 576 B/pairing (<0.01% of 8 TB/s): not the bound"},
        }
        if args.extra and world == 1:
            rec["extra"] = extra_configs(pkg, torch, dev, local_rank, stream)
        if world == 1 and not args.no_cpu_baseline:
            m = min(n, 1 << 14)
            g1h = g1.cpu().numpy().view(np.uint64).reshape(8, n)[:, :m].reshape(-1).copy()
            g2h = g2.cpu().numpy().view(np.uint64).reshape(16, n)[:, :m].reshape(-1).copy()
            rec["cpu_baseline"] = cpu_baseline(pkg, g1h, g2h, m, args.cpu_seconds)
            # sanity: the first few GPU results equal the oracle's on the same inputs
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import helpers as H
            k = 8
            want = H.oracle_pairing(pkg.layout.to_aos(g1h, 8)[:8 * k], pkg.layout.to_aos(g2h, 16)[:16 * k], k)
            got = pkg.layout.to_aos(out.cpu().numpy().view(np.uint64).reshape(48, n)[:, :k].reshape(-1).copy(), 48)
            rec["verified_vs_oracle"] = bool(np.array_equal(got, want))
        print(json.dumps(rec), flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
