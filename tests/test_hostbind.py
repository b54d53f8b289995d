"""A program compiled against the host binding (include/bn254_pairing.hpp, the C++ stand-in for rust-shim/: same function
names, argument order and panic behaviour as the reference's pub fns) and linked to the C-ABI library runs the scalar entry
points on the golden inputs:  pairing (pairing.rs:20), miller_loop_native / multi_miller_loop_native
(miller_loop_native.rs:320,324), final_exp_native, frobenius_map_native, pow_native, frob_coeffs, get_naf
(final_exp_native.rs:209,17,56,183,86)."""
import os
import subprocess

import pytest

import helpers as H
from helpers import R

HX = lambda xs: [int(x, 16) for x in xs]
SRC = os.path.join(H.ROOT, "tests", "hostbind", "hostbind_demo.cpp")


def _build(tmp_path):
    pk = H.pkg()
    if not os.path.exists(pk.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    exe = str(tmp_path / "hostbind_demo")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(H.ROOT, "include"), SRC, "-o", exe,
                           "-L", os.path.dirname(pk.LIB_PATH), "-l:" + os.path.basename(pk.LIB_PATH),
                           "-Wl,-rpath," + os.path.dirname(pk.LIB_PATH)])
    return exe


def _words(ints):
    return " ".join("%x" % w for c in ints for w in H.fq_words(c))


def _run(exe, lines):
    p = subprocess.run([exe], input="\n".join(lines) + "\n", capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    return [l.split() for l in p.stdout.splitlines()]


def _ints(tokens):
    """hex u64 limbs -> canonical integers (4 limbs per Fq, Montgomery R = 2^256)"""
    w = [int(t, 16) for t in tokens]
    return [R.from_mont(sum(w[4 * i + l] << (64 * l) for l in range(4))) for i in range(len(w) // 4)]


def test_host_binding_program_host_only_calls(tmp_path):
    """No GPU needed: frob_coeffs (host table) and get_naf (host logic) through the compiled binding."""
    exe = _build(tmp_path)
    vec = H.load_golden("bn254_vectors.json")
    out = _run(exe, ["frobc %x" % k for k in range(13)] + ["naf 1 %x" % R.BN_X, "naf 1 ffffffffffffffff"])
    for k in range(12):
        assert out[k][0] == "frobc" and _ints(out[k][1:]) == HX(vec["consts"]["frob_coeffs"][str(k)])
    assert _ints(out[12][1:]) == HX(vec["consts"]["frob_coeffs"]["0"])         # index 12 = index 0 (frobenius_map_native reduces mod 12)
    assert [int(x) for x in out[13][1:]] == R.get_naf([R.BN_X])
    assert out[14] == ["panic", "-5"]                                          # the reference's assert at final_exp_native.rs:123


@pytest.mark.gpu
def test_host_binding_program_on_golden_inputs(tmp_path):
    exe = _build(tmp_path)
    vec = H.load_golden("bn254_vectors.json")
    lines, want = ["reserve 40 4"], [("reserved", [])]       # bn254_reserve(device 0, NULL stream, 64 units, 4 pairs): nothing below allocates
    for i in (0, 5):
        P, Q = HX(vec["g1"][i]), HX(vec["g2"][i])
        lines.append("pairing " + _words(P) + " " + _words(Q))
        want.append(("pairing", R.myfq12_to_ark(HX(vec["pairing"][i]))))
        lines.append("miller " + _words(P) + " " + _words(Q))
        want.append(("miller", HX(vec["miller"][i])))
    g = vec["groups"][1]
    lines.append("multi %x " % g["k"] + " ".join(_words(HX(vec["g1"][i])) + " " + _words(HX(vec["g2"][i])) for i in g["idx"]))
    want.append(("multi", HX(g["miller"])))
    a = HX(vec["fq12_in"][1])
    lines.append("fexp " + _words(a))
    want.append(("fexp", HX(vec["final_exp"][1])))
    lines.append("frob 3 " + _words(a))
    want.append(("frob", HX(vec["frobenius"]["3"][1])))
    lines.append("pow 1 %x " % R.BN_X + _words(a))
    want.append(("pow", HX(vec["pow_x"][1])))
    lines.append("pow 1 0 " + _words(a))                      # zero exponent: pow_native returns a (final_exp_native.rs:56-84)
    want.append(("pow", a))
    lines.append("pow 0 " + _words(a))                        # empty exponent vector: likewise
    want.append(("pow", a))
    # batch forms: std::vector<G1Affine> / <G2Affine> go to the engine as they are (element-major entry points)
    idx = list(range(7))
    lines.append("batch %x " % len(idx) + " ".join(_words(HX(vec["g1"][i])) + " " + _words(HX(vec["g2"][i])) for i in idx))
    want += [("bark", R.myfq12_to_ark(HX(vec["pairing"][i]))) for i in idx] + [("bmy", HX(vec["pairing"][i])) for i in idx]
    lines.append("fexp " + _words([0] * 12))                  # final_exp_native(0): the reference panics (division by zero)
    out = _run(exe, lines)
    # the page-locked forms of the same batch (pinned_vector + *_into, HostRegistration on the caller's vectors): same limbs, and the
    # library recognises both kinds of memory -- and that the registration ended with its guard
    pin = [g for g in out if g and g[0] == "bpin"]
    assert pin == [["bpin", "1", "1", "1", "0"]], pin
    out = [g for g in out if not (g and g[0] == "bpin")]
    for (tag, w), got in zip(want, out):
        assert got[0] == tag and _ints(got[1:]) == w, tag
    assert out[len(want)] == ["panic", "-4"]


@pytest.mark.gpu
def test_host_binding_product_check(tmp_path):
    """multi_pairing_check_batch through the binding: e(aP, Q) e(-P, aQ) == 1 (the shape of final_exp_native.rs:245-263) in
    group 0, an unrelated pair of pairs in group 1."""
    exe = _build(tmp_path)
    a = 0x1234567
    P, Q = R.G1_GEN, R.G2_GEN
    aP, aQ = R.g1_mul(P, a), R.g2_mul(Q, a)
    negP = (P[0], (-P[1]) % R.P)
    pts = [(aP, Q), (negP, aQ), (aP, Q), (P, aQ)]
    fl = lambda q: [q[0][0], q[0][1], q[1][0], q[1][1]]
    line = "check 2 2 " + " ".join(_words(list(p)) + " " + _words(fl(q)) for p, q in pts)
    out = _run(exe, [line])
    assert out[0] == ["check", "1", "0"]
