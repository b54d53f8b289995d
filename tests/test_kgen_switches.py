"""Generator switches that are NOT the shipped default still build, certify and compute the right values (one lane, simulator):

  KGEN_FISSION=1   the split Miller loop (built, verified, measured, not adopted -- its line traffic costs more clock
                   than its cycles save).  Kept as a reproducible experiment: one pairing of k_pairing against the golden fixtures, with
                   the call sequence the bound certification walked (a k = 2 group of k_mpairing: `_run("multi", ...)` by hand).
  all round-4 switches (and round 5's shorter chain of the Miller loop) off: the generator still emits round 3's kernels (the A/B baseline `lib_base.so` of profiles/r04_ab.txt).
The switches are read at import time, so each case runs in a fresh interpreter."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASE = r'''
import sys
sys.path[:0] = [r"%(root)s/tools", r"%(root)s/tests", r"%(root)s"]
import kgen4_prog as K4P
import test_kgen4 as T
import helpers as H
vec = H.load_golden("bn254_vectors.json")
HX = T.HX
which = sys.argv[1]
if which == "single":
    g1, g2 = T._inputs(vec, 5)
    out, m = T.run_kernel(K4P.KernelBuilder(do_miller=True, do_fexp=True), g1, g2)
    assert out == HX(vec["pairing"][5]), "pairing value"
else:
    g = [x for x in vec["groups"] if x["k"] == 2][0]
    g1, g2 = T._soa([HX(vec["g1"][i]) for i in g["idx"]]), T._soa([HX(vec["g2"][i]) for i in g["idx"]])
    out, m = T.run_kernel(K4P.KernelBuilder(do_miller=True, do_fexp=True, multi=True), g1, g2, k=2)
    assert out == HX(g["pairing"]), "multi-pairing value"
assert m.max_acc < (1 << 63)
print("ok", m.count)
'''


def _run(which, env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    p = subprocess.run([sys.executable, "-c", CASE % {"root": ROOT}, which], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    return int(p.stdout.split()[-1])


def test_split_miller_loop_builds_and_is_exact():
    n_split = _run("single", {"KGEN_FISSION": "1"})
    assert 3_480_000 < n_split < 3_600_000                 # the shipped kernel's work (3.52 M instructions), plus the line stores / loads
    # (the k-pair kernel of the same switch was exact too when the experiment was run -- profiles/r04_ab.txt; the variant is not shipped
    # and its second simulation, 45 s, is no longer part of the suite: _run("multi", {"KGEN_FISSION": "1"}))


def test_round3_baseline_switches():
    off = {k: "0" for k in ("KGEN_MUL6_KEEP_DIFFS", "KGEN_DBL_LAZY_Y3", "KGEN_CYC_WIDE_M", "KGEN_FQINV_WIDE_M", "KGEN_MUL3_KEEP_DY", "KGEN_ADD_INJECT",
                            "KGEN_BOUSTRO", "KGEN_INV_FUSED", "KGEN_INV_SAFEGCD", "KGEN_DIGIT_ADD", "KGEN_SHORT_CHAIN")}
    n_r3 = _run("single", off)
    # round 3: 3.672 M instructions per pairing (profiles/r03_instr_histogram.json); the x-powers' digit set is a constant of the generator, not
    # a switch: with round 5's {1, 15, 19} (3 x 20.5 k instructions less than {1, 5, 9, 13}) the same switches give 3.617 M
    assert 3_605_000 < n_r3 < 3_630_000
