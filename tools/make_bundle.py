#!/usr/bin/env python3
"""make_bundle.py <bundle dir> <tag> -- joins what tools/gpu_bundle.sh collected on ONE lease into <tag>_bundle.json: the calibrated
multiply-add peak, the bench line, the profiled kernel times, the PMC-derived figures, the in-kernel clock and the box id, with the
two roofline fractions of each kernel recomputed from those numbers alone (so a reader can redo them by hand):

    achieved = units per launch x algorithmic mul32 per unit / kernel time          (2 286 160 per pairing,
    frac_nominal    = achieved / (1024 SIMDs x 2.4 GHz / 4 cycles x 64 lanes)         4 572 184 per four-pair group)
    frac_calibrated = achieved / (calibrated wave-instructions per second x 64)

and the cross-check the review asks for: the profiled kernel time must not exceed bench.py's ms_per_step on the same lease."""
import json
import os
import sys

W_PAIRING, W_GROUP4 = 2_286_160, 4_572_184
NOMINAL = 1024 * 2.4e9 / 4 * 64


def load(path):
    try:
        with open(path) as f:
            txt = f.read().strip()
        return json.loads(txt) if txt else None
    except (OSError, ValueError):
        return None


def main():
    out, tag = sys.argv[1], sys.argv[2]
    bench = load(os.path.join(out, "bench.json"))
    c0, c1 = load(os.path.join(out, "calib_start.json")), load(os.path.join(out, "calib_end.json"))
    pmc = load(os.path.join(out, f"{tag}_pmc.json")) or {}
    pmc_g = load(os.path.join(out, f"{tag}_groth16_pmc.json")) or {}
    stamps = load(os.path.join(out, "clock_stamp.json")) or {}
    calib = [c["mul32_per_s"] for c in (c0, c1) if c]
    calib_peak = sum(calib) / len(calib) if calib else None
    n1, ng = pmc.get("_notes", {}), pmc_g.get("_notes", {})

    def kernel(notes, units, work, bench_ms):
        ms = notes.get("kernel_ms_avg_rocprof")
        if not ms:
            return None
        ach = units * work / (ms * 1e-3)
        return {"kernel_ms_avg_rocprof": ms, "kernel_calls_profiled": notes.get("kernel_calls"), "bench_ms": bench_ms,
                "profiled_ms_le_bench_ms": (ms <= bench_ms) if bench_ms else None,
                "achieved_T_mul32_per_s": ach / 1e12, "frac_of_nominal_issue_peak": ach / NOMINAL,
                "frac_of_calibrated_peak": ach / calib_peak if calib_peak else None,
                "frac_from_bench_ms_nominal": (units * work / (bench_ms * 1e-3) / NOMINAL) if bench_ms else None,
                "frac_from_bench_ms_calibrated": (units * work / (bench_ms * 1e-3) / calib_peak) if (bench_ms and calib_peak) else None,
                "hbm_bytes_per_launch_corrected": notes.get("hbm_bytes_per_launch_corrected"),
                "algorithmic_bytes_per_launch": notes.get("algorithmic_bytes_per_launch"),
                "valu_wave_insts_per_work_item": notes.get("valu_wave_insts_per_work_item"),
                "valu_busy_fraction_of_wave_cycles": notes.get("valu_busy_fraction_of_wave_cycles"),
                "valu_issue_utilisation_at_measured_clock": notes.get("valu_issue_utilisation_at_measured_clock"),
                "shader_clock_ghz_from_GRBM_GUI_ACTIVE": notes.get("shader_clock_ghz_measured"), "wait_fraction": notes.get("wait_fraction"),
                "kernel_header_sha16": notes.get("kernel_header_sha16")}

    g16 = None
    if bench:
        for k, v in (bench.get("extra") or {}).items():
            if k.startswith("configs[3]:"):
                g16 = v
    bundle = {
        "what": "one lease: calibration -> bench.py --steps 20 -> rocprofv3 trace + PMC passes (k_pairing, k_mpairing) -> in-kernel clock -> calibration",
        "tag": tag, "box": (bench or {}).get("box"), "calibration": {"start": c0, "end": c1, "peak_mul32_per_s_mean": calib_peak,
                                                                        "nominal_issue_peak_mul32_per_s": NOMINAL},
        "bench_line": bench,
        "k_pairing (configs[2], 2^20 pairings per launch)": kernel(n1, 1 << 20, W_PAIRING, (bench or {}).get("roofline", {}).get("kernel_ms_avg")),
        "k_mpairing (configs[3], 2^18 groups x 4 pairs per launch)": kernel(ng, 1 << 18, W_GROUP4, (g16 or {}).get("ms")),
        "in_kernel_clock": stamps or None,
        "formulae": {"achieved": "units per launch x algorithmic mul32 per unit / kernel time; 2286160 per pairing, 4572184 per 4-pair group (SURVEY.md 8d)",
                     "nominal": "1024 SIMDs x 2.4e9 / 4 x 64 = 39.3216e12", "calibrated": "tools/valu_calib --mad-only: wave-instructions/s x 64, mean of start and end"},
    }
    path = os.path.join(out, f"{tag}_bundle.json")
    with open(path, "w") as f:
        json.dump(bundle, f, indent=1)
    brief = {k: bundle[k] for k in bundle if k.startswith("k_")}
    print(json.dumps({"calibrated_peak_T": calib_peak / 1e12 if calib_peak else None, "bench_value": (bench or {}).get("value"),
                      "bench_ms_per_step": (bench or {}).get("ms_per_step"), **brief}, indent=1)[:4000])


if __name__ == "__main__":
    main()
