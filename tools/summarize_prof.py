#!/usr/bin/env python3
"""summarize_prof.py <rocprofv3 output dir> <tag> -- per-launch PMC averages of the dominant kernel -> JSON summary.

Writes <dir>/<tag>_pmc.json and <dir>/<tag>_kernel_stats.csv (copy both to profiles/).  HBM bytes follow
/opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE count kilobytes (x1024); on gfx950 FETCH_SIZE
under-reports wide coalesced reads by 2x, so the read side is doubled."""
import collections
import csv
import glob
import json
import os
import sys

KERNEL = os.environ.get("PROF_KERNEL", "::k_pairing(")
LOG2_BATCH = int(os.environ.get("PROF_LOG2_BATCH", "20"))          # lanes per launch = units (pairings, or k-pair groups)
PAIRS_PER_UNIT = int(os.environ.get("PROF_K", "1"))                # k of the multi-pairing kernels (Groth16 shape: 4)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_SIMD = 1024                                                      # MI355X: 256 CUs x 4 SIMDs


def header_sha16():
    import hashlib
    try:
        with open(os.path.join(ROOT, "plonky2-bn254-pairing_amd", "csrc", "pairing_asm_gen.h"), "rb") as f:
            return hashlib.sha256(f.read()).hexdigest()[:16]
    except OSError:
        return None


def main():
    out, tag = sys.argv[1], sys.argv[2]
    agg = collections.defaultdict(list)
    for f in sorted(glob.glob(os.path.join(out, "pmc_*counter_collection.csv"))):
        for r in csv.DictReader(open(f)):
            if KERNEL in r.get("Kernel_Name", ""):
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    s = {k: sum(v) / len(v) for k, v in agg.items()}
    notes = {"kernel": f"{KERNEL}, n=2^{LOG2_BATCH}, per-launch averages (rocprofv3 --pmc, separate passes; bench.py --steps 3 --warmup 1 --no-extra --no-cpu-baseline)"}
    stats = glob.glob(os.path.join(out, "trace_kernel_stats.csv"))
    if stats:
        rows = list(csv.DictReader(open(stats[0])))
        for r in rows:
            if KERNEL in r["Name"]:
                notes["kernel_ms_avg_rocprof"] = float(r["AverageNs"]) / 1e6
                notes["kernel_calls"] = int(r["Calls"])
        with open(os.path.join(out, f"{tag}_kernel_stats.csv"), "w") as g:
            g.write(open(stats[0]).read())
    if "FETCH_SIZE" in s and "WRITE_SIZE" in s:
        notes["FETCH_SIZE/WRITE_SIZE unit"] = "KB; gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md) -> doubled"
        notes["hbm_bytes_per_launch_corrected"] = (2 * s["FETCH_SIZE"] + s["WRITE_SIZE"]) * 1024
        notes["hbm_bytes_per_launch_raw_counters"] = (s["FETCH_SIZE"] + s["WRITE_SIZE"]) * 1024
        notes["log2_batch"] = LOG2_BATCH
        # per unit: k pairs in (192 B each), one Fq12 out (384 B)
        notes["algorithmic_bytes_per_launch"] = (192 * PAIRS_PER_UNIT + 384) << LOG2_BATCH
    notes["pairs_per_unit"] = PAIRS_PER_UNIT
    notes["kernel_header_sha16"] = header_sha16()
    if "SQ_WAVE_CYCLES" in s and "SQ_ACTIVE_INST_VALU" in s:
        notes["valu_busy_fraction_of_wave_cycles"] = s["SQ_ACTIVE_INST_VALU"] / s["SQ_WAVE_CYCLES"]
        notes["valu_insts_per_wave"] = s["SQ_INSTS_VALU"] / s["SQ_WAVES"]
        notes["valu_wave_insts_per_launch"] = s["SQ_INSTS_VALU"]
        notes["valu_wave_insts_per_work_item"] = s["SQ_INSTS_VALU"] / ((1 << LOG2_BATCH) / 64)   # one wave's 64 lanes = 64 units
        # (SQ_ACTIVE_INST_VALU equals SQ_INSTS_VALU on this part at one wave per SIMD: their ratio says nothing.)  Cycles come from
        # the clock the package actually held: GRBM_GUI_ACTIVE is summed over the 8 XCDs.
        if "GRBM_GUI_ACTIVE" in s and notes.get("kernel_ms_avg_rocprof"):
            cyc = s["GRBM_GUI_ACTIVE"] / 8
            notes["shader_cycles_per_launch"] = cyc
            notes["shader_clock_ghz_measured"] = cyc / (notes["kernel_ms_avg_rocprof"] * 1e-3) / 1e9
            # each SIMD issues one VALU instruction of its lone wave per 4 cycles at best
            notes["valu_issue_utilisation_at_measured_clock"] = 4 * s["SQ_INSTS_VALU"] / (N_SIMD * cyc)
    if "SQ_WAIT_ANY" in s and "SQ_WAVE_CYCLES" in s:
        notes["wait_fraction"] = s["SQ_WAIT_ANY"] / s["SQ_WAVE_CYCLES"]
    if "SQ_ACTIVE_INST_LDS" in s and "SQ_WAVE_CYCLES" in s:
        notes["lds_fraction"] = s["SQ_ACTIVE_INST_LDS"] / s["SQ_WAVE_CYCLES"]
    s["_notes"] = notes
    with open(os.path.join(out, f"{tag}_pmc.json"), "w") as g:
        json.dump(s, g, indent=1)
    print(json.dumps(notes, indent=1))


if __name__ == "__main__":
    main()
