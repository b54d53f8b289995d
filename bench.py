#!/usr/bin/env python3
"""bench.py -- BN254 pairings/sec on MI355X (BASELINE.json metric), one JSON line on rank 0.

  python bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path (`pairing()` = final_exp_native(miller_loop_native(Q, P)),
/root/reference/src/pairing.rs:20-22) over one batch of synthetic subgroup points that is already
resident in HBM when the timed region starts.

  N = 1   BASELINE.json configs[2]: 2^20 independent pairings on one MI355X (the largest single-GPU
          configuration).  configs[1] (2^16) and configs[3] (Groth16 shape, 2^18 groups x 4 pairs) are
          timed too and reported under "extra" (not the headline `value`).
  N > 1   BASELINE.json configs[4]: 2^21 pairings per GPU (2^24 on 8 GPUs), one process per GPU.  Rank 0
          generates the whole batch on its device and SCATTERS the contiguous slices over RCCL
          (plonky2-bn254-pairing_amd/sharded.py: grouped isend/irecv on device tensors); the K timed steps are
          the per-rank compute on the resident shard (`value` = all ranks' pairings / max-over-ranks time,
          weak scaling); afterwards the Fq12 outputs are GATHERED to rank 0.  Scatter and gather times are
          reported separately ("exchange"), together with the world size RCCL actually saw ("rccl_ranks").

With `--gpus N` and no WORLD_SIZE in the environment this script starts the N ranks itself (a child
`python -m torch.distributed.run`, started BEFORE this process imports torch or touches the GPU) and exits
with the child's code; under the driver's own torch.distributed.run launch it reads RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* from the environment.

The JSON line also carries
  roofline     -- integer-VALU bound (2,286,160 mul32 of algorithmic work per pairing):
                  `frac` against the NOMINAL issue peak (1024 SIMDs x 2.4 GHz / 4 cycles x 64 lanes = 39.3 T mul32/s), and
                  beside it the fraction of the v_mad_u64_u32 issue rate CALIBRATED ON THIS LEASE (tools/valu_calib --mad-only,
                  a child process of this run); kernel time from HIP events on the launch stream
  cpu_baseline -- the C oracle (a port of the reference's schedule) timed on the host cores on a
                  bounded sample (single thread and all cores), rank 0, N = 1 only.
The process exits non-zero (and prints no headline line) when the GPU results do not match the oracle.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W_MUL32_PER_PAIRING = 2_286_160          # 16,810 fqmul x 136 mul32
W_FQMUL_PER_PAIRING = 16_810
W_MUL32_PER_GROTH16_GROUP = 4_572_184    # 33,619 fqmul x 136 mul32 (k = 4, shared final exp)
# Calibrated peak (tools/valu_calib.hip on MI355X, profiles/valu_calib_r01.txt): v_mad_u64_u32 issues at
# ~554 G wave-instructions/s chip-wide at 8 waves/SIMD = 35.5 T mul32/s.  (SURVEY's nominal quarter-rate
# assumption was 9.83 T mul32/s; the measured rate is 3.6x that.)
PEAK_MUL32_PER_S = 554e9 * 64
NOMINAL_PEAK_MUL32_PER_S = 9.83e12
# The nominal issue peak: one VALU instruction per SIMD per 4 cycles at the 2.4 GHz boost clock, 1024 SIMDs x 64 lanes =
# 39.3 T mul32/s.  The calibrated figure above is what a pure multiply-add stream reaches at the clock the package holds at its
# power limit (the MI355X guide has no integer-VALU peak); both fractions are reported.
NOMINAL_ISSUE_PEAK_MUL32_PER_S = 1024 * 2.4e9 / 4 * 64
LOG2_SINGLE = 20                         # configs[2]
LOG2_PER_GPU_MULTI = 21                  # configs[4]: 2^24 over 8 GPUs
PMC_SUMMARY = os.environ.get("BENCH_PMC_SUMMARY", os.path.join(ROOT, "profiles", "r06_pmc.json"))
PMC_SUMMARY_GROTH16 = os.environ.get("BENCH_PMC_SUMMARY_GROTH16", os.path.join(ROOT, "profiles", "r06_groth16_pmc.json"))
CALIB_R01_MUL32_PER_S = PEAK_MUL32_PER_S   # profiles/valu_calib_r01.txt (one box, round 1): kept for comparison with earlier rounds only
KERNEL_HEADER = os.path.join(ROOT, "plonky2-bn254-pairing_amd", "csrc", "pairing_asm_gen.h")


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--log2-batch", type=int, default=None,
                    help="pairings per GPU per step = 2^k (default: 20 at N = 1 = configs[2], 21 at N > 1 = configs[4])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-power", action="store_true", help="do not sample the package power (rocm-smi) during the timed steps")
    ap.add_argument("--no-extra", action="store_true", help="skip configs[1] and configs[3] (reported under \"extra\" at N = 1)")
    ap.add_argument("--no-host-path", action="store_true",
                    help="skip the PCIe-inclusive host-pointer calls of `extra` (profiling runs: their chunk launches would enter the per-launch counter averages)")
    ap.add_argument("--no-scalar-latency", action="store_true",
                    help="skip the one-item calls of the scalar signatures in \"extra\" (profiling runs: their small launches would enter the kernel averages)")
    ap.add_argument("--cpu-seconds", type=float, default=16.0, help="wall-clock budget of the cpu_baseline legs (single thread + all cores)")
    ap.add_argument("--dist-at-one", action="store_true",
                    help="run the N > 1 flow (process group, scatter, collectives, gather) with whatever world size the environment gives, "
                         "world size 1 included: how the RCCL side of configs[4] is exercised on a one-GPU box (tests/test_rccl_one_rank.py)")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ launcher
def launch_ranks(args, argv):
    """--gpus N without a torch.distributed environment: start N fresh rank processes.  Nothing in this process has
    imported torch or initialised HIP (a process that has must not exec / fork GPU work on this pool)."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


# ------------------------------------------------------------------------------------------------ cpu baseline
def host_cpu_info():
    info = {"nproc": os.cpu_count(), "affinity": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None, "cgroup_cpu_max": None}
    for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(p) as f:
                info["cgroup_cpu_max"] = f.read().strip()
            break
        except OSError:
            pass
    return info


def usable_cores(info):
    cores = info["affinity"] or info["nproc"] or 1
    q = info["cgroup_cpu_max"]
    if q:
        parts = q.split()
        try:
            if len(parts) == 2 and parts[0] != "max":
                cores = max(1, min(cores, int(int(parts[0]) / int(parts[1]))))
            elif len(parts) == 1 and int(parts[0]) > 0:
                cores = max(1, min(cores, int(parts[0]) // 100000))
        except ValueError:
            pass
    return cores


def cpu_baseline(pkg, g1_soa, g2_soa, n_avail, seconds):
    """Oracle (port of the reference schedule) on the host cores, bounded sample of the same inputs: one thread, then all
    usable cores with >= 64 pairings per thread per call (thread start-up is noise at that size)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers as H
    info = host_cpu_info()
    cores = usable_cores(info)
    g1a = pkg.layout.to_aos(g1_soa, 8)
    g2a = pkg.layout.to_aos(g2_soa, 16)
    H.oracle_pairing(g1a[:8], g2a[:16], 1)          # one-time constant tables outside the timed region
    # single thread
    done1, t0, budget1 = 0, time.time(), seconds * 0.35
    while done1 + 32 <= n_avail and (time.time() - t0) < budget1:
        H.oracle_pairing(g1a[8 * done1: 8 * (done1 + 32)], g2a[16 * done1: 16 * (done1 + 32)], 32, threads=1)
        done1 += 32
    dt1 = time.time() - t0
    single = done1 / dt1
    # all cores
    chunk = min(n_avail, 64 * cores)
    done, t0, budget = 0, time.time(), seconds * 0.65
    while (time.time() - t0) < budget:
        lo = done % max(1, n_avail - chunk + 1)
        H.oracle_pairing(g1a[8 * lo: 8 * (lo + chunk)], g2a[16 * lo: 16 * (lo + chunk)], chunk, threads=cores)
        done += chunk
    dt = time.time() - t0
    return {"value": done / dt, "unit": "pairings/s", "cores": cores, "kind": "port",
            "single_thread": {"value": single, "unit": "pairings/s", "cores": 1, "sample": f"{done1} pairings, {dt1:.1f} s"},
            "all_cores": {"value": done / dt, "unit": "pairings/s", "cores": cores, "per_thread": done / dt / cores,
                          "sample": f"{done} pairings in calls of {chunk} ({chunk // cores} per thread), {dt:.1f} s"},
            "host": info,
            "sample": f"pairings of the same batch through the C oracle (reference schedule: affine G2 steps with an inversion per "
                      f"step, NAF pow with divisions; constants cached): {done1} on one thread in {dt1:.1f} s, {done} on {cores} pthreads in {dt:.1f} s"}


def kernel_header_sha16():
    """Identifies the kernel code a PMC summary was collected on (tools/summarize_prof.py records the same digest)."""
    import hashlib
    try:
        with open(KERNEL_HEADER, "rb") as f:
            return hashlib.sha256(f.read()).hexdigest()[:16]
    except OSError:
        return None


def pmc_summary(log2, kern_avg_ms, path=None):
    """Counter-derived fields come from the committed rocprofv3 --pmc summary of this same command (collected in separate
    passes as the MI355X guide prescribes), NOT from this run: they are labelled as such -- and dropped (None, with the reason)
    when the summary cannot describe this run: another batch size, another generation of the kernel code (digest of
    pairing_asm_gen.h), or a kernel time more than 5 % away from the one rocprofv3 saw."""
    path = path or PMC_SUMMARY
    try:
        with open(path) as f:
            notes = json.load(f).get("_notes", {})
    except Exception:
        return {}, "no PMC summary at " + os.path.relpath(path, ROOT)
    if notes.get("log2_batch") != log2:
        return {}, f"PMC summary is for 2^{notes.get('log2_batch')} lanes, this run for 2^{log2}"
    if notes.get("kernel_header_sha16") != kernel_header_sha16():
        return {}, "PMC summary was collected on other kernel code (pairing_asm_gen.h digest differs)"
    ref = notes.get("kernel_ms_avg_rocprof")
    if not ref or abs(kern_avg_ms - ref) > 0.05 * ref:
        return {}, f"kernel time of this run ({kern_avg_ms:.2f} ms) is more than 5 % from the profiled one ({ref} ms)"
    return notes, None


def calibrate_this_lease():
    """The calibrated peak, re-taken on the lease this run has (round 5): tools/valu_calib --mad-only as a CHILD process (a pure
    v_mad_u64_u32 stream at eight waves per SIMD, half a second, wall time from HIP events).  None (with the reason) when the binary
    is missing or fails: the fraction of the nominal issue peak does not depend on it."""
    exe = os.path.join(ROOT, "tools", "valu_calib")
    if not os.path.exists(exe):
        return None, "tools/valu_calib not built"
    try:
        p = subprocess.run([exe, "--mad-only", "--json"], capture_output=True, text=True, timeout=120)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        if p.returncode != 0 or not line:
            return None, f"tools/valu_calib failed (rc {p.returncode}): {(p.stdout + p.stderr)[-200:]}"
        rec = json.loads(line[0])
        rec["when"] = time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime())
        rec["host"] = os.uname().nodename
        return rec, None
    except Exception as e:      # noqa: BLE001
        return None, f"tools/valu_calib: {type(e).__name__}: {e}"


def under_profiler():
    """rocprofv3's preloaded library rides into child processes and initialises the GPU there: no rocm-smi (an `env python3` script, i.e. an
    exec after that initialisation) from a profiled run"""
    return any(k.startswith(("ROCP", "ROCPROF")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")


def box_id():
    """What identifies the box a line was measured on (the pool's boxes differ by a few per cent in the clock they hold)."""
    info = {"host": os.uname().nodename}
    if under_profiler():
        return info
    try:
        txt = subprocess.run(["rocm-smi", "--showuniqueid", "--showserial"], capture_output=True, text=True, timeout=10).stdout
        import re
        m = re.search(r"Unique ID:\s*(\S+)", txt)
        if m:
            info["gpu_unique_id"] = m.group(1)
        m = re.search(r"Serial Number:\s*(\S+)", txt)
        if m:
            info["gpu_serial"] = m.group(1)
    except Exception:           # noqa: BLE001
        pass
    return info


# ------------------------------------------------------------------------------------------------ one rank
def load_engine():
    """The HIP engine, or -- for the CPU launcher tests only -- an injected stand-in (BENCH_TEST_ENGINE=module:attr under tests/).
    The hook is honoured only inside a pytest run (PYTEST_CURRENT_TEST, which pytest sets and the launcher's children inherit):
    anywhere else it is an error, so no environment can make this script print a `value` that is not a measurement."""
    hook = os.environ.get("BENCH_TEST_ENGINE")
    if hook and not os.environ.get("PYTEST_CURRENT_TEST"):
        raise SystemExit("bench.py: BENCH_TEST_ENGINE is a test hook and is refused outside pytest (PYTEST_CURRENT_TEST is not set)")
    if hook:
        import importlib
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        mod, attr = hook.split(":")
        return getattr(importlib.import_module(mod), attr)(), True
    import __graft_entry__
    return __graft_entry__.build(), False


class PowerSampler:
    """Package power (rocm-smi) sampled by a host thread while the timed steps run: k_pairing is power-bound (DESIGN.md
    section 3), so the line carries the watts it ran at.  Pure observation -- no GPU setting is touched."""

    def __init__(self):
        import threading
        self.samples, self.stop, self.limit = [], threading.Event(), None
        self.thread = threading.Thread(target=self._run, daemon=True)

    @staticmethod
    def _read(flag, key):
        import re
        try:
            txt = subprocess.run(["rocm-smi", flag], capture_output=True, text=True, timeout=5).stdout
            m = re.search(key + r"[^:]*:\s*([0-9.]+)", txt)
            return float(m.group(1)) if m else None
        except Exception:
            return None

    def _run(self):
        while not self.stop.is_set():
            w = self._read("--showpower", r"Power \(W\)")
            if w is not None:
                self.samples.append(w)

    def __enter__(self):
        self.thread.start()
        return self

    def __exit__(self, *exc):
        self.stop.set()
        self.thread.join(timeout=10)
        self.limit = self._read("--showmaxpower", r"Max Graphics Package Power \(W\)")

    def summary(self):
        s_ = self.samples[1:-1] if len(self.samples) > 4 else self.samples        # drop the ramps
        if not s_:
            return None
        return {"avg_w": sum(s_) / len(s_), "max_w": max(s_), "samples": len(s_), "limit_w": self.limit, "source": "rocm-smi --showpower during the timed steps"}


def spot_check(pkg, torch, g1, g2, out, n, positions, threads):
    """GPU results at `positions` equal the oracle's on the same inputs (bit-exact)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers as H
    pos = torch.as_tensor(positions, device=g1.device)
    g1h = g1.view(8, n)[:, pos].cpu().numpy().view(np.uint64).reshape(-1).copy()
    g2h = g2.view(16, n)[:, pos].cpu().numpy().view(np.uint64).reshape(-1).copy()
    got = out.view(48, n)[:, pos].cpu().numpy().view(np.uint64).reshape(-1).copy()
    want = H.oracle_pairing(pkg.layout.to_aos(g1h, 8), pkg.layout.to_aos(g2h, 16), len(positions), threads=threads)
    return bool(np.array_equal(pkg.layout.to_aos(got, 48), want))


def extra_configs(pkg, torch, dev, local_rank, stream, g1, g2, n, sample_power=False, scalar_latency=True, calib_peak=None, host=True):
    """BASELINE.json configs[1] and configs[3] on one GPU (not the headline `value`); inputs: the resident 2^20 batch."""
    out = {}

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize(dev)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for _ in range(reps):
            fn()
        b.record(stream)
        torch.cuda.synchronize(dev)
        return a.elapsed_time(b) / reps

    m = 1 << 16
    sel = lambda t, planes: t.view(planes, n)[:, :m].contiguous().view(-1)
    s1, s2 = sel(g1, 8), sel(g2, 16)
    o = torch.zeros(48 * m, dtype=torch.int64, device=dev)
    ms = timed(lambda: pkg.pairing_batch_dev(s1, s2, o, m, device=local_rank, stream=stream), 10)
    out["configs[1]: 2^16 independent pairings"] = {"ms": ms, "pairings_per_s": m / (ms * 1e-3),
                                                    "roofline_frac": m / (ms * 1e-3) * W_MUL32_PER_PAIRING / NOMINAL_ISSUE_PEAK_MUL32_PER_S,
                                                    "roofline_frac_of_calibrated_peak": (m / (ms * 1e-3) * W_MUL32_PER_PAIRING / calib_peak) if calib_peak else None,
                                                    "roofline_frac_of_calibrated_peak_r01": m / (ms * 1e-3) * W_MUL32_PER_PAIRING / CALIB_R01_MUL32_PER_S}
    groups, k = 1 << 18, 4
    assert groups * k <= n
    o2 = torch.zeros(48 * groups, dtype=torch.int64, device=dev)
    ms = timed(lambda: pkg.multi_pairing_batch_dev(g1, g2, o2, groups, k, True, device=local_rank, stream=stream), 2)
    ach = groups / (ms * 1e-3) * W_MUL32_PER_GROTH16_GROUP
    notes, stale = pmc_summary(18, ms, PMC_SUMMARY_GROTH16)
    ipi = notes.get("valu_wave_insts_per_work_item")
    g16 = {"ms": ms, "groups_per_s": groups / (ms * 1e-3), "pairs_per_s": groups * k / (ms * 1e-3),
           "roofline_frac": ach / NOMINAL_ISSUE_PEAK_MUL32_PER_S, "kernel": "k_mpairing",
           "roofline": {"bound": "valu-int32-mul", "achieved": ach / 1e12, "peak": NOMINAL_ISSUE_PEAK_MUL32_PER_S / 1e12, "unit": "T mul32/s",
                        "frac": ach / NOMINAL_ISSUE_PEAK_MUL32_PER_S, "frac_of_nominal_issue_peak": ach / NOMINAL_ISSUE_PEAK_MUL32_PER_S,
                        "frac_of_calibrated_peak": (ach / calib_peak) if calib_peak else None, "calibrated_peak_this_lease": (calib_peak / 1e12) if calib_peak else None,
                        "frac_of_calibrated_peak_r01": ach / CALIB_R01_MUL32_PER_S,
                        "traffic": notes.get("hbm_bytes_per_launch_corrected"), "algorithmic_bytes_per_launch": (192 * k + 384) * groups,
                        "work_per_unit": f"{W_MUL32_PER_GROTH16_GROUP} mul32 per 4-pair group (SURVEY.md 8d)", "kernel": "k_mpairing", "kernel_ms_avg": ms,
                        "pmc_summary": os.path.relpath(PMC_SUMMARY_GROTH16, ROOT), "pmc_summary_kernel_ms": notes.get("kernel_ms_avg_rocprof"),
                        "pmc_dropped": stale,
                        "valu_issue": None if not ipi else {
                            "wave_instr_per_work_item": ipi, "wave_instr_per_s": ipi * (groups / 64) / (ms * 1e-3),
                            "frac": ipi * (groups / 64) / (ms * 1e-3) / (1024 * 2.4e9 / 4),
                            "issue_utilisation_at_measured_clock": notes.get("valu_issue_utilisation_at_measured_clock")}}}
    if sample_power:                # ~1.5 s of back-to-back launches: long enough for rocm-smi to see what this kernel draws
        with PowerSampler() as ps:
            for _ in range(24):
                pkg.multi_pairing_batch_dev(g1, g2, o2, groups, k, True, device=local_rank, stream=stream)
            torch.cuda.synchronize(dev)
        g16["package_power"] = ps.summary()
    out["configs[3]: Groth16 shape, 2^18 groups x 4 pairs, shared final exp"] = g16
    if hasattr(pkg, "pairing_fixed_g2_batch_dev"):
        # the same function when three of a group's four G2 points are THE SAME for the whole batch (a Groth16 verifier's beta, gamma, delta): their point
        # steps are done once (a line table), the per-group work of a fixed pair is one line scaling + one sparse multiplication per step
        kf = 3
        g2var = g2.view(16, groups, k)[:, :, 0].contiguous().view(-1)
        g2fix = g2.view(16, n)[:, 1:1 + kf].contiguous().view(-1)
        table = torch.zeros(pkg.g2_lines_bytes(kf) // 8, dtype=torch.int64, device=dev)
        pkg.g2_lines_dev(g2fix, kf, table, device=local_rank, stream=stream)
        o3 = torch.zeros(48 * groups, dtype=torch.int64, device=dev)
        ms_f = timed(lambda: pkg.pairing_fixed_g2_batch_dev(g1, g2var, table, kf, o3, groups, device=local_rank, stream=stream), 2)
        exp = g2.view(16, groups, k).clone()
        for j in range(kf):
            exp[:, :, 1 + j] = g2fix.view(16, kf)[:, j:j + 1]
        pkg.multi_pairing_batch_dev(g1, exp.contiguous().view(-1), o2, groups, k, True, device=local_rank, stream=stream)
        torch.cuda.synchronize(dev)
        out["configs[3] with a fixed verifying key: 2^18 groups of 1 + 3 pairs, three G2 points shared by the batch (bn254_pairing_fixed_g2_batch_dev)"] = {
            "ms": ms_f, "groups_per_s": groups / (ms_f * 1e-3), "kernel": "k_fpairing", "same_limbs_as_k_mpairing_on_the_expanded_pairs": bool(torch.equal(o2, o3)),
            "speedup_over_four_free_pairs": g16["ms"] / ms_f,
            "work_normalised_frac_of_nominal_issue_peak": groups / (ms_f * 1e-3) * W_MUL32_PER_GROTH16_GROUP / NOMINAL_ISSUE_PEAK_MUL32_PER_S,
            "note": "the algorithmic work of the four-pair product (SURVEY.md 8d) over this kernel's time: the fixed pairs' point steps are not executed per group at all"}
        if hasattr(pkg, "pairing_fixed_g2_check_target_batch_dev"):
            # a Groth16 verifier's whole check: gamma, delta fixed, e(alpha, beta) held as the comparison target: 1 + 2 pairs per proof, one verdict byte out
            kv = 2
            g1v = g1.view(8, groups, k)[:, :, :1 + kv].contiguous().view(-1)
            tab2 = torch.zeros(pkg.g2_lines_bytes(kv) // 8, dtype=torch.int64, device=dev)
            pkg.g2_lines_dev(g2fix.view(16, kf)[:, :kv].contiguous().view(-1), kv, tab2, device=local_rank, stream=stream)
            pkg.pairing_fixed_g2_batch_dev(g1v, g2var, tab2, kv, o3, groups, device=local_rank, stream=stream)
            target = o3.view(48, groups)[:, 0].contiguous().cpu().numpy().view("uint64")               # group 0's own product: exactly one group must match
            verdict = torch.zeros(groups, dtype=torch.uint8, device=dev)
            ms_v = timed(lambda: pkg.pairing_fixed_g2_check_target_batch_dev(g1v, g2var, tab2, kv, target, verdict, groups, device=local_rank, stream=stream), 2)
            torch.cuda.synchronize(dev)
            out["Groth16 verifier check: 2^18 proofs, e(A, B) e(L, gamma) e(C, delta) == e(alpha, beta) with gamma, delta fixed (bn254_pairing_fixed_g2_check_target_batch_dev)"] = {
                "ms": ms_v, "proofs_per_s": groups / (ms_v * 1e-3), "kernels": "k_fpairing + k_is_one", "speedup_over_four_free_pairs": g16["ms"] / ms_v,
                "verdicts_as_expected": bool(int(verdict[0]) == 1 and int(verdict.sum()) == 1)}
            # groups whose G2 points are ALL fixed (g2_var = None): a KZG / PLONK opening check e(P_1, [tau] G2) e(P_2, G2) -- no point step at all
            g1k = g1.view(8, groups, k)[:, :, :kv].contiguous().view(-1)
            ms_k = timed(lambda: pkg.pairing_fixed_g2_check_target_batch_dev(g1k, None, tab2, kv, None, verdict, groups, device=local_rank, stream=stream), 2)
            exp2 = g2fix.view(16, kf)[:, :kv].reshape(16, 1, kv).expand(16, groups, kv).contiguous().view(-1)
            same = None
            if not under_profiler():           # (a k = 2 launch of k_mpairing would be averaged into the profile of the four-pair launches)
                pkg.pairing_fixed_g2_batch_dev(g1k, None, tab2, kv, o3, groups, device=local_rank, stream=stream)
                o4 = torch.zeros_like(o3)
                pkg.multi_pairing_batch_dev(g1k, exp2, o4, groups, kv, True, device=local_rank, stream=stream)
                torch.cuda.synchronize(dev)
                same = bool(torch.equal(o3, o4))
                del o4
            out["KZG-style opening check: 2^18 groups of 2 pairs, both G2 points fixed for the batch (bn254_pairing_fixed_g2_check_target_batch_dev, g2_var = NULL)"] = {
                "ms": ms_k, "checks_per_s": groups / (ms_k * 1e-3), "kernels": "k_fpairing (no own pair) + k_is_one",
                "same_limbs_as_k_mpairing_on_the_expanded_pairs": same}
            del tab2, g1v, verdict, g1k, exp2
        del exp, o3, table
    if hasattr(pkg, "set_wide_groups"):
        # ONE group of all 2^20 pairs (an aggregated check: multi_miller_loop_native on one Vec, miller_loop_native.rs:324-326, + final_exp_native): the group
        # is spread over the lanes in chunks, the chunks' Miller values multiplied in a tree, one final exponentiation
        one = torch.zeros(48, dtype=torch.int64, device=dev)
        ms_a = timed(lambda: pkg.multi_pairing_batch_dev(g1, g2, one, 1, n, True, device=local_rank, stream=stream), 2)
        halves = torch.zeros(96, dtype=torch.int64, device=dev)
        pkg.multi_pairing_batch_dev(g1, g2, halves, 2, n // 2, False, device=local_rank, stream=stream)          # the same pairs as two groups: two Miller values
        prod, chk = torch.zeros(48, dtype=torch.int64, device=dev), torch.zeros(48, dtype=torch.int64, device=dev)
        pkg.fq12_mul_batch_dev(halves.view(48, 2)[:, 0].contiguous(), halves.view(48, 2)[:, 1].contiguous(), prod, 1, local_rank, stream)
        pkg.final_exp_batch_dev(prod, chk, 1, local_rank, stream)
        torch.cuda.synchronize(dev)
        out["one group of 2^20 pairs (an aggregated check): final_exp_native(multi_miller_loop_native(all pairs))"] = {
            "ms": ms_a, "pairs_per_s": n / (ms_a * 1e-3), "kernels": "k_mmiller_u over 65 536 lanes of 16 pairs + a multiplication tree (k_tree_split, k_op) + k_cvm (final exponentiation)",
            "equals_the_product_of_its_two_halves": bool(torch.equal(one, chk)) and int(one.abs().sum()) != 0}
        del one, halves, prod, chk
    # data formats either side of the path: element-major <-> limb-major on the device (HBM-bound: every word read once, written once)
    HBM_PEAK = 8.0e12
    lay = {}
    o3 = torch.empty(48 * n, dtype=torch.int64, device=dev)
    for name, src, words, fn, order in (("G1 elems -> planes", g1, 8, pkg.soa_from_elems_dev, 0), ("G2 elems -> planes", g2, 16, pkg.soa_from_elems_dev, 0),
                                        ("Fq12 planes -> ark Fq12 elems", o3, 48, pkg.soa_to_elems_dev, pkg.FQ12_ARK)):
        dst = torch.empty(words * n, dtype=torch.int64, device=dev)
        ms = timed(lambda: fn(src, dst, words, n, order, local_rank, stream), 20)
        lay[name] = {"ms": ms, "GB_per_s": 2 * 8 * words * n / (ms * 1e-3) / 1e9, "frac_of_hbm_peak": 2 * 8 * words * n / (ms * 1e-3) / HBM_PEAK}
        del dst
    out["layout kernels at 2^20 elements (roofline bound: hbm, 8 TB/s)"] = lay
    if host:
        out["host-resident batches: one host-pointer call, copies both ways included (PCIe-inclusive; never `value`)"] = host_path(pkg, torch, g1, g2, n, local_rank, dev)
    # the reference's functions are SCALAR (one pairing / one group per call): wall time of one call on one item, launch to completion,
    # on the throughput kernel (one item per lane) and on the lane-cooperative kernel that small batches take
    if scalar_latency and hasattr(pkg, "set_stream_latency"):
        import time as _t
        # the kernel selection of THIS stream only (bn254_set_stream_latency): the process-wide defaults are not touched
        o4 = torch.zeros(48, dtype=torch.int64, device=dev)
        f4 = torch.zeros(48, dtype=torch.int64, device=dev)
        one = lambda t, planes, cnt: t.view(planes, n)[:, :cnt].contiguous().view(-1)
        p1, q1, p4, q4 = one(g1, 8, 1), one(g2, 16, 1), one(g1, 8, 4), one(g2, 16, 4)
        pkg.set_stream_latency(0, -1, local_rank, stream)
        pkg.miller_loop_batch_dev(p1, q1, f4, 1, device=local_rank, stream=stream)
        calls = {"pairing(p, q)": lambda: pkg.pairing_batch_dev(p1, q1, o4, 1, device=local_rank, stream=stream),
                 "miller_loop_native(q, p)": lambda: pkg.miller_loop_batch_dev(p1, q1, o4, 1, device=local_rank, stream=stream),
                 "final_exp_native(f)": lambda: pkg.final_exp_batch_dev(f4, o4, 1, device=local_rank, stream=stream),
                 "final_exp_native(multi_miller_loop_native(4 pairs))": lambda: pkg.multi_pairing_batch_dev(p4, q4, o4, 1, 4, True, device=local_rank, stream=stream)}
        lat = {}
        for name, fn in calls.items():
            row = {}
            for kern, thr in (("throughput_kernel_ms", 0), ("lane_cooperative_kernel_ms", 1 << 20)):
                pkg.set_stream_latency(thr, -1, local_rank, stream)
                ts = []
                for _ in range(7):
                    torch.cuda.synchronize(dev)
                    t0 = _t.perf_counter()
                    fn()
                    torch.cuda.synchronize(dev)
                    ts.append(_t.perf_counter() - t0)
                row[kern] = min(ts[2:]) * 1e3
                row.setdefault("_ref", o4.clone())
                row["same_limbs"] = bool(torch.equal(row["_ref"], o4))
            del row["_ref"]
            lat[name] = row
        pkg.set_stream_latency(pkg.LATENCY_INHERIT, -1, local_rank, stream)
        out["scalar signatures: one item per call, wall ms (launch to completion, inputs resident)"] = lat
    pkg.last_status(local_rank, stream)
    return out


def host_path(pkg, torch, g1, g2, n, local_rank, dev):
    """What a binding of src/pairing.rs:20-22 gets: its callers hold their values in HOST memory, so its batch calls are the host-pointer
    entry points (chunked two-worker pipeline above 65 536 units, copies under the kernels).  Wall time of one call, second of two, from
    pageable numpy arrays and from page-locked ones (bn254_alloc_pinned); limb-major, and element-major with ark's Fq12 order on the way out."""
    import time as _t
    import numpy as np
    res = {}
    h1 = g1.cpu().numpy().view(np.uint64).copy()
    h2 = g2.cpu().numpy().view(np.uint64).copy()
    shapes = (("2^16 pairings", 1 << 16, 1), ("2^20 pairings", n, 1), ("Groth16 shape: 2^18 groups x 4 pairs", n // 4, 4))
    for label, units, k in shapes:
        np_ = units * k
        s1 = np.ascontiguousarray(h1.reshape(8, n)[:, :np_]).reshape(-1)
        s2 = np.ascontiguousarray(h2.reshape(16, n)[:, :np_]).reshape(-1)
        row = {}
        for fmt in ("limb-major", "element-major, ark Fq12 out"):
            a1, a2 = (s1, s2) if fmt == "limb-major" else (pkg.layout.to_aos(s1, 8), pkg.layout.to_aos(s2, 16))
            for mem in ("pageable", "page-locked"):
                if mem == "pageable":
                    b1, b2, bo = a1, a2, np.empty(48 * units, dtype=np.uint64)
                else:
                    b1, b2, bo = pkg.alloc_pinned(a1.size), pkg.alloc_pinned(a2.size), pkg.alloc_pinned(48 * units)
                    b1[:], b2[:] = a1, a2
                if fmt == "limb-major":
                    call = (lambda: pkg.pairing_batch(b1, b2, units, device=local_rank, out=bo)) if k == 1 else \
                           (lambda: pkg.multi_pairing_batch(b1, b2, units, k, True, device=local_rank, out=bo))
                else:
                    call = (lambda: pkg.pairing_batch_elems(b1, b2, units, out_order=pkg.FQ12_ARK, device=local_rank, out=bo)) if k == 1 else \
                           (lambda: pkg.multi_pairing_batch_elems(b1, b2, units, k, True, out_order=pkg.FQ12_ARK, device=local_rank, out=bo))
                call()
                ts = []
                for _ in range(2):
                    t0 = _t.perf_counter()
                    call()
                    ts.append(_t.perf_counter() - t0)
                row[f"{fmt}, {mem}"] = {"ms": min(ts) * 1e3, "pairings_per_s": np_ / min(ts), "units_per_s": units / min(ts)}
                if mem == "page-locked":
                    for b in (b1, b2, bo):
                        pkg.free_pinned(b)
        res[label] = row
    if hasattr(pkg, "pairing_fixed_g2_check_batch_elems"):
        # a Groth16 verifier's check from host structs: 2^18 proofs of 1 + 2 pairs (gamma, delta fixed, e(alpha, beta) as the target), element-major in,
        # one verdict byte per proof out -- the fixed-G2 kernel behind the same two-worker pipeline
        units, kv = n // 4, 2
        e1 = pkg.layout.to_aos(np.ascontiguousarray(h1.reshape(8, n)[:, :units * (1 + kv)]).reshape(-1), 8)
        e2 = pkg.layout.to_aos(np.ascontiguousarray(h2.reshape(16, n)[:, :units]).reshape(-1), 16)
        ef = pkg.layout.to_aos(np.ascontiguousarray(h2.reshape(16, n)[:, units:units + kv]).reshape(-1), 16)
        target = np.zeros(48, dtype=np.uint64)
        pkg.pairing_fixed_g2_check_batch_elems(e1, e2, ef, kv, units, target=target, device=local_rank)
        ts = []
        for _ in range(2):
            t0 = _t.perf_counter()
            pkg.pairing_fixed_g2_check_batch_elems(e1, e2, ef, kv, units, target=target, device=local_rank)
            ts.append(_t.perf_counter() - t0)
        res["Groth16 verifier check: 2^18 proofs (1 + 2 pairs, fixed gamma / delta, target), element-major, pageable"] = {
            "ms": min(ts) * 1e3, "proofs_per_s": units / min(ts), "units_per_s": units / min(ts)}
    return res


def run_rank(args):
    import numpy as np
    import torch
    pkg, test_engine = load_engine()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if os.environ.get("BENCH_TEST_FAIL_RANK") == str(rank) and os.environ.get("PYTEST_CURRENT_TEST"):
        raise RuntimeError("injected fault (BENCH_TEST_FAIL_RANK, a pytest-only hook): this rank dies before the rendezvous")
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    on_gpu = not test_engine
    if on_gpu:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (no CPU fallback by design)")
        if os.environ.get("BENCH_SHARE_GPU"):      # test hook: several ranks on one GPU (RCCL refuses that: use gloo)
            local_rank = local_rank % torch.cuda.device_count()
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        stream = torch.cuda.current_stream(dev)
        sync = lambda: torch.cuda.synchronize(dev)
    else:
        dev, stream, sync = torch.device("cpu"), None, (lambda: None)
    dist = None
    multi = world > 1 or args.dist_at_one          # the configs[4] flow: process group, scatter, collectives, gather
    if multi:
        import torch.distributed as dist
        from datetime import timedelta
        # a bounded rendezvous and bounded collectives: a rank that died (build, out of memory) must end the job with a reason in
        # seconds, not hold its peers for the default ten minutes (the longest gap between two collectives here is the timed steps)
        tmo = timedelta(seconds=float(os.environ.get("BENCH_DIST_TIMEOUT_S", "120")))
        if "RANK" not in os.environ:               # --dist-at-one started by hand: a one-rank group of its own
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(29400 + os.getpid() % 500))
            os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK=str(local_rank))
        if backend == "nccl" and on_gpu:
            dist.init_process_group(backend="nccl", device_id=dev, timeout=tmo)
        else:
            dist.init_process_group(backend=backend, timeout=tmo)
    sharded = __import__("importlib").import_module("plonky2-bn254-pairing_amd.sharded") if multi else None

    log2 = args.log2_batch if args.log2_batch is not None else (LOG2_PER_GPU_MULTI if multi else LOG2_SINGLE)
    n = 1 << log2                                   # pairings per GPU per step
    n_total = n * world
    exchange = None
    g1 = torch.zeros(8 * n, dtype=torch.int64, device=dev)
    g2 = torch.zeros(16 * n, dtype=torch.int64, device=dev)
    out = torch.zeros(48 * n, dtype=torch.int64, device=dev)
    full = None
    if not multi:
        pkg.generate_pairs_dev(0xB2540001, g1, g2, n, device=local_rank, stream=stream)
        pkg.last_status(local_rank, stream)
    else:
        # configs[4]: the whole batch exists on rank 0 only; slices travel over RCCL (device tensors, grouped isend/irecv)
        if rank == 0:
            f1 = torch.zeros(8 * n_total, dtype=torch.int64, device=dev)
            f2 = torch.zeros(16 * n_total, dtype=torch.int64, device=dev)
            pkg.generate_pairs_dev(0xB2540001, f1, f2, n_total, device=local_rank, stream=stream)
            pkg.last_status(local_rank, stream)
            full = (f1, f2)
        sync()
        dist.barrier()
        t0 = time.perf_counter()
        sharded.scatter_inputs(full[0] if full else None, full[1] if full else None, n_total, g1, g2, dist)
        sync()
        dist.barrier()
        exchange = {"scatter_ms": (time.perf_counter() - t0) * 1e3}

    def step():
        pkg.pairing_batch_dev(g1, g2, out, n, device=local_rank, stream=stream)

    for _ in range(args.warmup):
        step()
    sync()
    if dist:
        dist.barrier()
    sync()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)] if on_gpu else []
    import contextlib
    # (not under a profiler: its preloaded library would ride into the rocm-smi child processes)
    profiled = under_profiler()
    power = PowerSampler() if (on_gpu and rank == 0 and not args.no_power and not profiled) else None
    with (power or contextlib.nullcontext()):
        t0 = time.perf_counter()
        if on_gpu:
            for a, b in evs:
                a.record(stream)
                step()
                b.record(stream)
        else:
            for _ in range(args.steps):
                step()
        sync()
        if dist:
            dist.barrier()
        sync()
        elapsed = time.perf_counter() - t0
    pkg.last_status(local_rank, stream)
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if (backend == "nccl" and on_gpu) else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kern_ms = sorted(a.elapsed_time(b) for a, b in evs) if on_gpu else [elapsed / args.steps * 1e3]
    kern_avg_ms = sum(kern_ms) / len(kern_ms)
    per_rank = None
    if dist:
        # every rank's own kernel times and peak memory travel to rank 0: a straggler or a rank close to its memory limit is
        # visible in the N > 1 line (the timed region itself is the max over ranks, as the contract says)
        mem = float(torch.cuda.max_memory_allocated(dev)) if on_gpu else 0.0
        mine = torch.tensor([kern_avg_ms, kern_ms[0], kern_ms[-1], mem], dtype=torch.float64, device=dev if (backend == "nccl" and on_gpu) else "cpu")
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [[float(x) for x in t.tolist()] for t in allr]

    gathered = None
    if multi:
        sync()
        dist.barrier()
        t0 = time.perf_counter()
        if os.environ.get("BENCH_TEST_FAIL_IN_GATHER") == str(rank) and os.environ.get("PYTEST_CURRENT_TEST"):
            raise RuntimeError("injected fault (BENCH_TEST_FAIL_IN_GATHER, a pytest-only hook): this rank dies after the timed steps, inside the gather")
        gathered = sharded.gather_outputs(out, n_total, dist, device=dev)
        sync()
        dist.barrier()
        exchange["gather_ms"] = (time.perf_counter() - t0) * 1e3
        exchange["p2p_group"] = sharded.p2p_group_mode()
        if on_gpu and rank == 0:
            exchange["rank0_peak_device_bytes_after_gather"] = int(torch.cuda.max_memory_allocated(dev))
        exchange["bytes_scattered"] = 192 * (n_total - n)
        exchange["bytes_gathered"] = 384 * (n_total - n)
        exchange["note"] = ("rank 0 -> peers: G1/G2 slices, peers -> rank 0: Fq12 outputs; torch.distributed batch_isend_irecv on device "
                            "tensors (backend nccl = RCCL over xGMI; under gloo the device tensors bounce through pinned host buffers); "
                            "wall-clock incl. the staging copies, outside the timed steps")

    rc = 0
    t_post = time.perf_counter()        # from here to the verdict broadcast the peers wait for rank 0 (calibration child, oracle gate)
    if rank == 0:
        total = n_total * args.steps
        value = total / elapsed
        per_gpu_rate = n / (kern_avg_ms * 1e-3)
        achieved = per_gpu_rate * W_MUL32_PER_PAIRING
        notes, stale = pmc_summary(log2, kern_avg_ms)
        traffic = notes.get("hbm_bytes_per_launch_corrected")
        calib, calib_err = calibrate_this_lease() if (on_gpu and not os.environ.get("BENCH_NO_CALIB")) else (None, "not measured (test engine or BENCH_NO_CALIB)")
        calib_peak = calib["mul32_per_s"] if calib else None
        insts_per_item = notes.get("valu_wave_insts_per_work_item")        # wave-instructions per 64 pairings (one wave's lanes)
        cfg_name = "configs[2]" if (not multi and log2 == 20) else ("configs[4]" if (world > 1 and log2 == 21) else "custom size")
        rec = {
            "metric": "BN254 pairings/sec (whole node)", "value": value, "unit": "pairings/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "i32 limbs (254-bit Montgomery field, signed reduced-radix limbs, 32x32->64 integer multiply-add on the VALU)",
            "data": "synthetic: on-device [s]G1, [t]G2 subgroup points, SplitMix64 scalars, seed 0xB2540001" + (" [TEST ENGINE, not a measurement]" if test_engine else ""),
            "box": box_id() if on_gpu else None,
            "config": {"workload": f"2^{log2} independent pairings per GPU per step, {world} GPU(s): BASELINE.json {cfg_name}; "
                                   f"pairing() = final_exp_native(miller_loop_native)",
                       "pairings_per_gpu": n, "pairings_total": n_total, "layout": "SoA limb-major u64 Montgomery, inputs resident in HBM",
                       "ranks_share_one_gpu": bool(on_gpu and world > 1 and os.environ.get("BENCH_SHARE_GPU")),
                       "sharding": None if not multi else "contiguous slices per rank, scattered from / gathered to rank 0 over RCCL outside the timed steps; no data-path collective"},
            "roofline": {"bound": "valu-int32-mul", "achieved": achieved / 1e12, "peak": NOMINAL_ISSUE_PEAK_MUL32_PER_S / 1e12, "unit": "T mul32/s",
                         "frac": achieved / NOMINAL_ISSUE_PEAK_MUL32_PER_S, "traffic": traffic,
                         "peak_note": "the NOMINAL integer multiply-add issue peak: 1024 SIMDs x 2.4 GHz / 4 cycles x 64 lanes = 39.3 T mul32/s (the MI355X guide has "
                                      "no integer-VALU figure); the peak a pure multiply-add stream actually reaches on THIS lease is measured beside it "
                                      "(peaks.calibrated_this_lease, frac_of_calibrated_peak)",
                         "frac_of_calibrated_peak": (achieved / calib_peak) if calib_peak else None,
                         "frac_of_calibrated_peak_r01": achieved / CALIB_R01_MUL32_PER_S,
                         "achieved_note": "work-normalised: SURVEY.md 8(d)'s algorithmic mul32 per pairing x pairings/s (kernel time from HIP events), "
                                          "not the multiply-adds the kernel executes (the figure is the REFERENCE's schedule: its digit table of 6x+2 with 25 additions; "
                                          "the kernel walks a 21-addition form of the same number -- the pairing's value does not see the chain, DESIGN.md section 4.2 -- "
                                          "and executes 2.43 M multiply-adds of 3.43 M instructions per pairing)",
                         "traffic_note": f"HBM bytes per launch from the committed PMC summary {os.path.relpath(PMC_SUMMARY, ROOT)} (separate --pmc passes of this "
                                         "command; raw and corrected counters there), not measured in this run; algorithmic bytes per launch = 576 B x pairings"
                                         + (f"; DROPPED: {stale}" if stale else ""),
                         "pmc_summary_kernel_ms": notes.get("kernel_ms_avg_rocprof"), "kernel_header_sha16": kernel_header_sha16(),
                         "algorithmic_bytes_per_launch": 576 * n,
                         "kernel": "k_pairing", "kernel_ms_avg": kern_avg_ms, "kernel_ms_min": kern_ms[0],
                         "work_per_unit": f"{W_MUL32_PER_PAIRING} mul32 = {W_FQMUL_PER_PAIRING} fqmul x 136 per pairing (SURVEY.md 8d)",
                         "frac_of_nominal_quarter_rate_peak": achieved / NOMINAL_PEAK_MUL32_PER_S,
                         "frac_of_nominal_issue_peak": achieved / NOMINAL_ISSUE_PEAK_MUL32_PER_S,
                         "peaks": {"calibrated_this_lease": None if not calib else dict(calib, value=calib_peak / 1e12, frac=achieved / calib_peak),
                                   "calibrated_this_lease_missing": calib_err,
                                   "calibrated_r01": {"value": CALIB_R01_MUL32_PER_S / 1e12, "frac": achieved / CALIB_R01_MUL32_PER_S,
                                                      "what": "round 1's single measurement on another box (profiles/valu_calib_r01.txt): the denominator of the fractions "
                                                              "quoted in rounds 1-4, kept for comparison only"},
                                   "nominal_issue": {"value": NOMINAL_ISSUE_PEAK_MUL32_PER_S / 1e12, "frac": achieved / NOMINAL_ISSUE_PEAK_MUL32_PER_S,
                                                     "what": "1024 SIMDs x 2.4 GHz / 4 cycles x 64 lanes"}, "unit": "T mul32/s"},
                         "valu_issue": None if not insts_per_item else {
                             "source": "SQ_INSTS_VALU per wave work item (64 pairings) from the committed PMC summary x this run's kernel time",
                             "wave_instr_per_work_item": insts_per_item,
                             "wave_instr_per_s": insts_per_item * (n / 64) / (kern_avg_ms * 1e-3), "peak_wave_instr_per_s": 1024 * 2.4e9 / 4,
                             "frac": insts_per_item * (n / 64) / (kern_avg_ms * 1e-3) / (1024 * 2.4e9 / 4),
                             "issue_utilisation_at_measured_clock": notes.get("valu_issue_utilisation_at_measured_clock"),
                             "note": "all VALU instructions the kernel issues against 1024 SIMDs x 2.4 GHz (nominal) / 4 cycles; the profile's own "
                                     "clock (GRBM_GUI_ACTIVE / kernel time) gives the utilisation at the clock the package actually held"},
                         "package_power": power.summary() if power else None,
                         "hbm": {"bound": "hbm", "achieved": 576 * n / (kern_avg_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                                 "frac": 576 * n / (kern_avg_ms * 1e-3) / 8e12, "traffic": traffic,
                                 "note": "the schema's HBM view of the same kernel: algorithmic bytes (576 B/pairing) over the launch time "
                                         "against 8 TB/s -- three orders of magnitude from the bound (SURVEY.md 8d: integer-VALU bound)"}},
        }
        if multi:
            rec["rccl_ranks"] = dist.get_world_size()
            rec["per_rank"] = {"kernel_ms_avg": [r[0] for r in per_rank], "kernel_ms_min": [r[1] for r in per_rank],
                               "kernel_ms_max": [r[2] for r in per_rank], "peak_device_bytes": [r[3] for r in per_rank],
                               "note": "HIP-event time of each rank's own k_pairing launches over the timed steps; torch.cuda.max_memory_allocated per rank "
                                       "(rank 0 also holds the whole batch and, after the timed steps, the gathered outputs)"}
            rec["dist_backend"] = backend
            rec["exchange"] = exchange
            step_ms = elapsed / args.steps * 1e3
            rec["value_incl_exchange"] = n_total / ((step_ms + exchange["scatter_ms"] + exchange["gather_ms"]) * 1e-3)
            exchange["exchange_share_of_step"] = (exchange["scatter_ms"] + exchange["gather_ms"]) / (step_ms + exchange["scatter_ms"] + exchange["gather_ms"])
            exchange["overlap"] = ("none by design: the pairing kernel holds every register and 144 KiB of LDS of every CU, a communication kernel cannot "
                                   "co-reside (profiles/r03_coresidency.txt); one scatter + one gather per step would add exchange_share_of_step to a step")
        # correctness gate: oracle spot checks on every run (first / last lanes, a work-item boundary, other ranks' slices)
        if on_gpu:
            threads = min(32, usable_cores(host_cpu_info()))
            pos = sorted({0, 1, 255, 256, n // 2 + 77, n - 257, n - 1, 65535 % n, 65536 % n})
            ok = spot_check(pkg, torch, g1, g2, out, n, pos, threads)
            if multi:
                # every peer's slice: its first lane, one in the middle, its last lane -- against the oracle on rank 0's copy of the inputs
                gp = sorted({r * n + off for r in range(1, world) for off in (0, n // 2 + 5, n - 1)})
                ok = ok and (not gp or spot_check(pkg, torch, full[0], full[1], gathered, n_total, gp, threads))
                ok = ok and bool(torch.equal(gathered.view(48, n_total)[:, :n], out.view(48, n)))
                rec["verified_positions_in_peer_slices"] = len(gp)
            rec["verified_vs_oracle"] = ok
            if not ok:
                print("bench.py: GPU results differ from the oracle -- no measurement reported", file=sys.stderr)
                rc = 3
        if not on_gpu and multi:
            # launcher tests (stand-in engine): the gathered batch must be what the engine makes of the WHOLE input on rank 0 --
            # every peer's slice, every lane (shard bounds, scatter and gather all have to be right for that)
            ref = torch.zeros(48 * n_total, dtype=torch.int64)
            pkg.pairing_batch_dev(full[0], full[1], ref, n_total)
            rec["gathered_equals_whole_batch_recomputation"] = bool(torch.equal(ref, gathered))
            if not rec["gathered_equals_whole_batch_recomputation"]:
                print("bench.py: gathered outputs differ from the whole-batch recomputation", file=sys.stderr)
                rc = 3
        if rc == 0 and on_gpu and not multi and not args.no_extra and log2 == LOG2_SINGLE:
            # (the other configurations and consumer shapes are measured beside the headline, never instead of it: a failure there -- an allocation on a
            # crowded box, say -- is reported in the line, the headline measurement above stands)
            try:
                rec["extra"] = extra_configs(pkg, torch, dev, local_rank, stream, g1, g2, n, sample_power=power is not None,
                                               scalar_latency=not args.no_scalar_latency, calib_peak=calib_peak, host=not args.no_host_path)
            except Exception as e:      # noqa: BLE001
                import traceback
                traceback.print_exc()
                rec["extra"] = {"error": f"{type(e).__name__}: {e}"}
        if rc == 0 and on_gpu and not multi and not args.no_cpu_baseline:
            m = min(n, 1 << 15)
            g1h = g1.view(8, n)[:, :m].cpu().numpy().view(np.uint64).reshape(-1).copy()
            g2h = g2.view(16, n)[:, :m].cpu().numpy().view(np.uint64).reshape(-1).copy()
            try:
                rec["cpu_baseline"] = cpu_baseline(pkg, g1h, g2h, m, args.cpu_seconds)
            except Exception as e:      # noqa: BLE001 -- (a reported baseline, not the measurement: its failure must not cost the line)
                import traceback
                traceback.print_exc()
                rec["cpu_baseline"] = {"value": None, "unit": "pairings/s", "cores": 0, "kind": "port", "sample": "", "error": f"{type(e).__name__}: {e}"}
        if multi:
            # what the peers sit through in the verdict broadcast below: must stay well inside the collective timeout
            tmo_s = float(os.environ.get("BENCH_DIST_TIMEOUT_S", "120"))
            rec["rank0_post_steps_s"] = {"seconds": time.perf_counter() - t_post, "limit": tmo_s / 4,
                                         "what": "rank 0 alone between the last collective of the timed part and the verdict broadcast: same-lease calibration "
                                                 "(child process) + oracle gate on its own and the peers' slices; the peers wait in dist.broadcast "
                                                 f"(timeout BENCH_DIST_TIMEOUT_S = {tmo_s:g} s)"}
            if rec["rank0_post_steps_s"]["seconds"] > tmo_s / 4:
                print(f"bench.py: rank 0 spent {rec['rank0_post_steps_s']['seconds']:.1f} s alone (> a quarter of BENCH_DIST_TIMEOUT_S = {tmo_s:g} s): "
                      "raise the timeout or set BENCH_NO_CALIB=1", file=sys.stderr, flush=True)
        if rc == 0:
            print(json.dumps(rec), flush=True)
    if dist:
        # every rank leaves with rank 0's verdict (a failed oracle gate must not leave the peers waiting in a barrier)
        t = torch.tensor([rc], dtype=torch.int32, device=dev if (backend == "nccl" and on_gpu) else "cpu")
        dist.broadcast(t, src=0)
        rc = int(t.item())
        dist.barrier()
        dist.destroy_process_group()
    return rc


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, argv))
    try:
        rc = run_rank(args)
    except SystemExit:
        raise
    except BaseException as e:      # noqa: BLE001 -- a rank must not die silently: its peers wait in a collective
        import traceback
        traceback.print_exc()
        print(f"bench.py: rank {os.environ.get('RANK', '0')} of {os.environ.get('WORLD_SIZE', '1')} failed -- {type(e).__name__}: {e}; "
              f"leaving with code 1 so that the launcher ends the other ranks", file=sys.stderr, flush=True)
        sys.stdout.flush()
        os._exit(1)                 # no destructors: a process group in a broken state can hang in its own teardown
    sys.exit(rc)


if __name__ == "__main__":
    main()
