#!/usr/bin/env python3
"""instr_histogram.py [out.json] -- where the instructions of one pairing go.

Runs the generated k_pairing (or, KGEN_HIST_KERNEL=mpairing, the k = 4 multi-pairing kernel) on ONE lane in the instruction
simulator (tools/ksim.py) on the golden inputs and counts every executed instruction by CLASS (what kind of work it is) and
by PHASE (which part of the algorithm it belongs to).  A wave executes the same stream for its 64 lanes, so the counts are
wave-instructions per work item of 64 pairings = instructions per pairing per lane; rocprofv3's SQ_INSTS_VALU / (n / 64)
measures the VALU part of the same number on the hardware.

The profiling build carries LM_* labels at the phase changes inside routines (KGEN_MARKERS=1: a label costs at most an
alignment s_nop, 0.01 % here); everything else is the shipped instruction stream.  Reference cost centres:
/root/reference/src/miller_loop_native.rs:46-96,151-173, final_exp_native.rs:56-84,130-169."""
import json
import os
import re
import sys

os.environ["KGEN_MARKERS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests"), ROOT]

import kgen4_prog as K4P  # noqa: E402
import ksim as S  # noqa: E402
import test_kgen4 as T  # noqa: E402

P_LIMB = re.compile(r"s(3[6-9]|4[0-4])$")


def klass(op, a):
    if op == "v_mad_i64_i32":
        x, y = a[2], a[3]
        if y == "-1":
            return "mad: digit extraction (accumulator -= digit)"
        if P_LIMB.match(y) or P_LIMB.match(x):
            return "mad: reduction (m_i p_j of the Montgomery pass, q p_j of redn)"
        if x[0] == "v" and y[0] == "v":
            return "mad: limb product"
        return "mad: 64-bit chain with a constant coefficient (xi, 3t - 2z, 12 E^2, scale)"
    if op == "v_mad_u64_u32":
        return "mad: other"
    if op == "v_mul_lo_u32":
        return "v_mul_lo_u32 (Montgomery digit m = lo * n0')"
    if op == "v_mul_hi_i32":
        return "v_mul_hi_i32 (redn quotient)"
    if op in ("v_lshl_add_u64", "v_sub_co_u32_e32", "v_subb_co_u32_e32", "v_add_co_u32_e32", "v_addc_co_u32_e32"):
        return "64-bit combination (Karatsuba U +- W)"
    if op in ("v_bfe_i32", "v_bfe_u32", "v_ashrrev_i64", "v_ashrrev_i32_e32", "v_and_b32_e32", "v_alignbit_b32", "v_lshrrev_b32_e32", "v_lshl_or_b32",
              "v_lshrrev_b64"):
        return "digit / carry handling (bfe, ashr, and)"
    if op.startswith("v_accvgpr"):
        return "AGPR moves (v_accvgpr_read / write)"
    if op == "v_mov_b32_e32":
        return "v_mov_b32"
    if op in ("v_add_u32_e32", "v_sub_u32_e32", "v_subrev_u32_e32", "v_lshl_add_u32", "v_lshlrev_b32_e32", "v_add3_u32"):
        return "32-bit limb-wise add / sub / shift"
    if op.startswith("v_"):
        return "other VALU (cmp, cndmask, or, ...)"
    if op.startswith("ds_"):
        return "LDS (ds_read / ds_write)"
    if op.startswith("global_"):
        return "VMEM (global_load / global_store)"
    if op == "s_nop":
        return "s_nop"
    if op == "s_waitcnt":
        return "s_waitcnt"
    if op in ("s_call_b64", "s_setpc_b64", "s_branch") or op.startswith("s_cbranch"):
        return "branches (s_call / s_setpc / s_branch / s_cbranch)"
    return "SALU"


def strip(label):
    return re.sub(r"_\d+$", "", label.replace("_%=", ""))


class Phases:
    def __init__(self):
        self.mk = {}
        self.counts = {}

    def phase(self, region, stack):
        region = strip(region)
        l2 = [(i, strip(t)) for i, (_, t) in enumerate(stack) if t.startswith("L2_")]
        in_powx = any(t.startswith("L3_powx") for _, t in stack)
        if not l2:
            if in_powx:
                return "x-powers: control"
            return "kernel I/O: loads, cvtin / cvtout, stores, item loop"
        d, name = l2[0]
        if len(stack) == d + 1:                      # executing in the L2 frame itself: markers / routine start set the sub-phase
            if region.startswith("LM_"):
                self.mk[d] = re.sub(r"_\d+$", "", region[3:])
            elif region.startswith("L2_"):
                self.mk[d] = None
        mk = self.mk.get(d)
        if name in ("L2_dblmul", "L2_dblfirst", "L2_dblmul_s", "L2_addmul", "L2_addmul_last", "L2_addmul_s"):
            if mk in ("mul034", "mul235"):
                return f"sparse multiplication mul_by_{mk[3:]} (miller_loop_native.rs:46-96)"
            kind = "doubling" if "dbl" in name else "addition"
            return f"G2 {kind} step + line coefficients (projective; reference :10-44, :157,167)"
        if name == "L2_sqr":
            return "f^2: fq12_sqr in the Miller loop (:153)"
        if name in ("L2_prefetch",):
            return "multi-pairing: pair-state prefetch"
        if name in ("L2_inv", "L2_fqinv"):
            return "easy part: Fq12 inversion (final_exp_native.rs:200)"
        if name.startswith("L2_frob"):
            return "Frobenius maps (final_exp_native.rs:17-54)"
        if name in ("L2_cyc", "L2_cycN"):
            return ("x-powers: cyclotomic squarings (pow_native, :56-84)" if in_powx else "y-chain: cyclotomic squarings (:153-166)")
        if name in ("L2_mul_body", "L2_mulG", "L2_mulGc", "L2_mulG_w", "L2_mulGc_w", "L2_pfB", "L2_mulL", "L2_mulLc"):
            return ("x-powers: fq12_mul (table b^5 b^9 b^13 + digits)" if in_powx else "easy part + y-chain: fq12_mul (:135-166, :198-205)")
        if name in ("L2_stG", "L2_ldG", "L2_ldGc", "L2_conjF", "L2_redF", "L2_stL", "L2_ldL", "L2_cpB"):
            return "Fq12 register moves: scratch (stG / ldG), on-chip register (stL / ldL), operand copy (cpB), conj"
        if name in ("L2_descale", "L2_sqscale"):
            return "line-scale tracking (exact miller_loop_native value)"
        return name

    def hook(self, op, a, region, stack):
        key = (self.phase(region, stack), klass(op, a))
        self.counts[key] = self.counts.get(key, 0) + 1


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else None
    which = os.environ.get("KGEN_HIST_KERNEL", "pairing")
    vec = T.H.load_golden("bn254_vectors.json")
    ph = Phases()
    orig_run = S.run

    def run_hooked(lines, m, *a_, **k_):
        m.hook = ph.hook
        return orig_run(lines, m, *a_, **k_)

    S.run = run_hooked
    if which == "pairing":
        g1, g2 = T._inputs(vec, 1)
        kb = K4P.KernelBuilder(do_miller=True, do_fexp=True)
        out, m = T.run_kernel(kb, g1, g2, profile=True)
        assert out == T.HX(vec["pairing"][1])
        k, unit = 1, "pairing"
    else:
        k = 4
        idx = [0, 1, 2, 3]

        def soa(rows):
            n = len(rows)
            o = [0] * (len(rows[0]) * 4 * n)
            for i, el in enumerate(rows):
                for c, x in enumerate(el):
                    for l, w in enumerate(T.R.limbs4(T.R.to_mont(x))):
                        o[(c * 4 + l) * n + i] = w
            return o
        g1, g2 = soa([T.HX(vec["g1"][i]) for i in idx]), soa([T.HX(vec["g2"][i]) for i in idx])
        kb = K4P.KernelBuilder(do_miller=True, do_fexp=True, multi=True)
        out, m = T.run_kernel(kb, g1, g2, k=k, profile=True)
        unit = "4-pair group (Groth16 shape)"
    S.run = orig_run
    total = sum(ph.counts.values())
    by_class, by_phase = {}, {}
    for (p_, c_), n in ph.counts.items():
        by_class[c_] = by_class.get(c_, 0) + n
        by_phase.setdefault(p_, {})[c_] = n
    valu = sum(n for c_, n in by_class.items() if c_.startswith(("mad", "v_", "64-bit", "digit", "AGPR", "32-bit", "other VALU")))
    mads = sum(n for c_, n in by_class.items() if c_.startswith("mad"))
    rec = {
        "what": f"dynamic instructions of one {unit} on one lane (= wave-instructions per 64 {unit}s), tools/ksim.py on the generated kernel text",
        "kernel": "k_pairing" if which == "pairing" else "k_mpairing, k = 4",
        "total_instructions": total, "valu_instructions": valu, "multiply_adds_v_mad_i64_i32": mads,
        "algorithmic_mul32_per_unit_SURVEY_8d": 2_286_160 if which == "pairing" else 4_572_184,
        "simulator_count_incl_nop_cycles": m.count,
        "by_class": dict(sorted(by_class.items(), key=lambda kv: -kv[1])),
        "by_phase": {p_: {"total": sum(d.values()), "share": round(sum(d.values()) / total, 4), "by_class": dict(sorted(d.items(), key=lambda kv: -kv[1]))}
                     for p_, d in sorted(by_phase.items(), key=lambda kv: -sum(kv[1].values()))},
    }
    txt = json.dumps(rec, indent=1)
    if out_path:
        with open(out_path, "w") as f:
            f.write(txt + "\n")
    print(f"{total} instructions, {valu} VALU, {mads} multiply-adds")
    for c_, n in rec["by_class"].items():
        print(f"  {n:9d} {100 * n / total:5.1f} %  {c_}")
    for p_, d in rec["by_phase"].items():
        print(f"  {d['total']:9d} {100 * d['share']:5.1f} %  {p_}")


if __name__ == "__main__":
    main()
