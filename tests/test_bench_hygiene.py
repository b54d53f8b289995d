"""bench.py takes `roofline.traffic` / `valu_issue` from the committed rocprofv3 PMC summary only when that summary describes
the run: same batch size, same kernel code (digest of pairing_asm_gen.h), kernel time within 5 % of the profiled one.
tools/summarize_prof.py: algorithmic bytes per unit for k-pair runs, clock-derived issue utilisation."""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_pmc_summary_is_dropped_when_it_does_not_describe_the_run(tmp_path, monkeypatch):
    bench = importlib.import_module("bench")
    digest = bench.kernel_header_sha16()
    assert digest and len(digest) == 16
    notes = {"log2_batch": 20, "kernel_header_sha16": digest, "kernel_ms_avg_rocprof": 110.0, "hbm_bytes_per_launch_corrected": 2.5e10,
             "valu_wave_insts_per_work_item": 3.66e6}
    p = tmp_path / "pmc.json"
    p.write_text(json.dumps({"_notes": notes}))
    monkeypatch.setattr(bench, "PMC_SUMMARY", str(p))
    got, why = bench.pmc_summary(20, 112.0)                   # within 5 %
    assert why is None and got["hbm_bytes_per_launch_corrected"] == 2.5e10
    got, why = bench.pmc_summary(20, 117.0)                   # 6.4 % away
    assert got == {} and "5 %" in why
    got, why = bench.pmc_summary(21, 110.0)                   # another batch size
    assert got == {} and "2^20" in why
    p.write_text(json.dumps({"_notes": dict(notes, kernel_header_sha16="0" * 16)}))
    got, why = bench.pmc_summary(20, 110.0)                   # other kernel code
    assert got == {} and "digest" in why
    monkeypatch.setattr(bench, "PMC_SUMMARY", str(tmp_path / "missing.json"))
    got, why = bench.pmc_summary(20, 110.0)
    assert got == {} and "no PMC summary" in why


def test_committed_pmc_summaries_are_consistent():
    """the committed summaries carry the corrected fields; the Groth16 one counts 4 pairs in and one Fq12 out per unit"""
    for name, k, log2 in (("r03_pmc.json", 1, 20), ("r03_groth16_pmc.json", 4, 18), ("r04_pmc.json", 1, 20), ("r04_groth16_pmc.json", 4, 18),
                         ("r05_pmc.json", 1, 20), ("r05_groth16_pmc.json", 4, 18), ("r06_pmc.json", 1, 20), ("r06_groth16_pmc.json", 4, 18)):
        with open(os.path.join(ROOT, "profiles", name)) as f:
            n = json.load(f)["_notes"]
        assert n["pairs_per_unit"] == k and n["log2_batch"] == log2
        assert n["algorithmic_bytes_per_launch"] == (192 * k + 384) << log2
        assert "valu_wave_insts_per_work_item" in n and "cycles_per_valu_instruction_active" not in n
        assert 0.5 < n["valu_issue_utilisation_at_measured_clock"] < 1.0 and 1.5 < n["shader_clock_ghz_measured"] < 2.5
        assert len(n["kernel_header_sha16"]) == 16


def test_bench_reads_the_summaries_of_the_shipped_kernels():
    """bench.py's default PMC summaries (k_pairing and the Groth16 shape) were collected on the kernel code that is in the tree:
    their digest is the digest of the committed pairing_asm_gen.h -- otherwise `traffic` / `valu_issue` would be dropped on every run."""
    bench = importlib.import_module("bench")
    for path in (bench.PMC_SUMMARY, bench.PMC_SUMMARY_GROTH16):
        with open(path) as f:
            assert json.load(f)["_notes"]["kernel_header_sha16"] == bench.kernel_header_sha16(), os.path.basename(path)
