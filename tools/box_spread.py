#!/usr/bin/env python3
"""box_spread.py <out.json> <bench line files...> -- the spread of the SAME build over several fresh GPU leases.

Each input file holds the JSON line of one `python bench.py --steps 5 --warmup 1 --no-cpu-baseline` run on its own gpurun lease (the
boxes of the pool differ by a few per cent in the clock they hold under this load: DESIGN.md section 6).  Output: per configuration
(BASELINE.json configs[1], [2], [3]) the kernel time of every lease and median / min / max, with the roofline fractions against both
peaks, plus the kernel-code digest the runs share."""
import json
import statistics
import sys

W1, W4 = 2_286_160, 4_572_184
PEAK_CAL, PEAK_NOM = 554e9 * 64, 1024 * 2.4e9 / 4 * 64


def main():
    out, files = sys.argv[1], sys.argv[2:]
    runs = []
    for f in files:
        for line in open(f):
            if line.startswith('{"metric'):
                runs.append((f, json.loads(line)))
    assert runs, "no bench lines"
    digests = {r["roofline"]["kernel_header_sha16"] for _, r in runs}
    assert len(digests) == 1, f"runs of different kernel code: {digests}"
    rows = {"configs[2]: 2^20 independent pairings (headline)": [], "configs[1]: 2^16 independent pairings": [],
            "configs[3]: Groth16 shape, 2^18 groups x 4 pairs": []}
    for f, r in runs:
        rows["configs[2]: 2^20 independent pairings (headline)"].append(r["roofline"]["kernel_ms_avg"])
        for k, v in r.get("extra", {}).items():
            if k.startswith("configs[1]:"):
                rows["configs[1]: 2^16 independent pairings"].append(v["ms"])
            if k.startswith("configs[3]:"):
                rows["configs[3]: Groth16 shape, 2^18 groups x 4 pairs"].append(v["ms"])
    units = {"configs[2]: 2^20 independent pairings (headline)": (1 << 20, W1), "configs[1]: 2^16 independent pairings": (1 << 16, W1),
             "configs[3]: Groth16 shape, 2^18 groups x 4 pairs": (1 << 18, W4)}
    cal = [((r["roofline"].get("peaks") or {}).get("calibrated_this_lease") or {}).get("mul32_per_s") for _, r in runs]
    res = {"what": "one build, one `bench.py --steps 5 --warmup 1 --no-cpu-baseline` run per fresh gpurun lease", "leases": len(runs),
           "boxes": [r.get("box") for _, r in runs],
           "calibrated_peak_T_mul32_per_s_per_lease": [c / 1e12 if c else None for c in cal],
           "headline_frac_of_own_lease_calibration": [(r["roofline"].get("frac_of_calibrated_peak")) for _, r in runs],
           "kernel_header_sha16": digests.pop(), "verified_vs_oracle": all(r.get("verified_vs_oracle") for _, r in runs),
           "package_power_w_avg": [(r["roofline"].get("package_power") or {}).get("avg_w") for _, r in runs], "configs": {}}
    for name, ms in rows.items():
        if not ms:
            continue
        n, w = units[name]
        frac = lambda t, peak: n / (t * 1e-3) * w / peak
        res["configs"][name] = {"kernel_ms": ms, "median_ms": statistics.median(ms), "min_ms": min(ms), "max_ms": max(ms),
                                "spread_pct": 100 * (max(ms) - min(ms)) / statistics.median(ms),
                                "units_per_s_median": n / (statistics.median(ms) * 1e-3),
                                "roofline_frac_calibrated_peak_r01": {"median": frac(statistics.median(ms), PEAK_CAL), "best": frac(min(ms), PEAK_CAL), "worst": frac(max(ms), PEAK_CAL)},
                                "roofline_frac_nominal_issue_peak": {"median": frac(statistics.median(ms), PEAK_NOM), "best": frac(min(ms), PEAK_NOM), "worst": frac(max(ms), PEAK_NOM)}}
    with open(out, "w") as g:
        json.dump(res, g, indent=1)
    print(json.dumps(res["configs"], indent=1))


if __name__ == "__main__":
    main()
