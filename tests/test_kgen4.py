"""The generated gfx950 kernels (tools/kgen4.py: L1 field routines on balanced radix-2^29 limbs; tools/kgen4_prog.py: L2/L3)
executed on ONE lane by the instruction-level simulator tools/ksim.py, against big-int arithmetic and the golden fixtures.
This is the CPU-side check of the product's instruction stream (the GPU parity tests run the same text on the hardware):
results, 64-bit operand alignment, uninitialised-register reads, the VALU-writes-SGPR -> VALU-reads hazard distance and any
signed 64-bit overflow of a column accumulator are all trapped here."""
import os
import random
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import asmcore as AC  # noqa: E402
import kgen4 as K4  # noqa: E402
import kgen4_prog as K4P  # noqa: E402
import ksim as S  # noqa: E402
import helpers as H  # noqa: E402
from helpers import R  # noqa: E402

P = AC.P_INT
NL, LB = K4.NL, K4.LB
HX = lambda xs: [int(x, 16) for x in xs]  # noqa: E731
G1B, G2B, FINB, OUTB, SCR, STAT = 0x100000, 0x200000, 0x300000, 0x400000, 0x1000000, 0x500000


def _sval(limbs):
    v = 0
    for i, l in enumerate(limbs):
        l = l - (1 << 32) if l >> 31 else l
        v += l << (LB * i)
    return v


def _redundant(x, rng, slack, k_p=3):
    """A signed redundant representation of x + k p with limb borrows of up to `slack` units of 2^29."""
    y = x + (rng.randrange(-k_p, k_p + 1) if slack else 0) * P
    l = K4.bal_limbs(y)
    for i in range(NL - 1):
        b = rng.randrange(-slack, slack + 1) if slack else 0
        l[i] += b << LB
        l[i + 1] -= b
    return l


def _m4(vals, rng, slack):
    m = S.Machine()
    for i in range(NL):
        m.s[K4.S_P + i] = K4.P_L[i] & 0xFFFFFFFF
    m.s[K4.S_N0] = K4.N0P
    m.s[K4.S_REDN] = K4.REDN_C
    m.s[K4.S_HALF], m.s[K4.S_HALF + 1] = 1 << 28, 0          # the digit extraction's rounding constant (kgen4.DIGIT_ADD)
    m.s[K4.S_M30] = (-30) & 0xFFFFFFFF
    for j, x in enumerate(vals):
        for i, w in enumerate(_redundant(x, rng, slack)):
            m.v[NL * j + i] = w & 0xFFFFFFFF
    return m


def _body(n):
    e = AC.Emitter()
    K4.routine_body(e, n)
    return e.finalize()


def _is_norm(m, regs):
    return all(-K4.HALF <= _sval([m.v[r]]) < K4.HALF for r in regs[:-1])


def test_l1_routines():
    """Redundant signed operands (negative limbs, limbs beyond 29 bits, value offsets by multiples of p): results are
    checked mod p; normalised / reduced outputs are checked as such."""
    rng = random.Random(7)
    RPI = pow(K4.RP, -1, P)
    B = {n: _body(n) for n in K4.L1V4_NAMES}

    def rnd():
        return rng.choice([0, 1, P - 1, P - 2]) if rng.random() < 0.25 else rng.randrange(P)

    def val(m, j):
        return _sval([m.v[NL * j + i] for i in range(NL)])

    for t in range(30):
        a0, a1, b0, b1 = rnd(), rnd(), rnd(), rnd()
        sl = [0, 1, 1][t % 3]          # limb borrows of up to one radix unit: |limb| <= 3 * 2^28
        want = {"mulfq": (a0 * b0 * RPI, a1 * b0 * RPI), "fqmul": (a0 * b0 * RPI, None), "fqsqr": (a0 * a0 * RPI, None),
                "add": (a0 + b0, a1 + b1), "sub": (a0 - b0, a1 - b1), "rsub": (b0 - a0, b1 - a1), "dbl": (2 * a0, 2 * a1),
                "neg": (-a0, -a1), "negc1": (a0, -a1), "norm": (a0, a1), "redn": (a0, a1), "mulxi": (9 * a0 - a1, a0 + 9 * a1),
                "mulxir": (9 * a0 - a1, a0 + 9 * a1)}
        if sl <= 1:
            want["mul"] = ((a0 * b0 - a1 * b1) * RPI, (a0 * b1 + a1 * b0) * RPI)
        if sl == 0:
            want["sqr"] = ((a0 * a0 - a1 * a1) * RPI, 2 * a0 * a1 * RPI)
        for name, (w0, w1) in want.items():
            m = _m4([a0, a1, b0, b1], rng, sl)
            S.run_block(B[name], m)
            assert (val(m, 0) - w0) % P == 0, name
            if w1 is not None:
                assert (val(m, 1) - w1) % P == 0, name
            if name in ("norm", "redn", "mulxi", "mulxir", "mul", "sqr", "mulfq"):
                assert _is_norm(m, list(range(NL))) and _is_norm(m, list(range(NL, 2 * NL))), name
            if name in ("redn", "mulxir"):
                assert all(abs(val(m, j)) < 0.52 * P for j in range(2)), name
            if name in ("mul", "sqr", "mulfq") and sl == 0:
                assert all(abs(val(m, j)) < 0.6 * P for j in range(2)), name          # inputs below p: sum/R' +- p/2
    f2m = lambda x, y: ((x[0] * y[0] - x[1] * y[1]) % P, (x[0] * y[1] + x[1] * y[0]) % P)
    f2a = lambda x, y: ((x[0] + y[0]) % P, (x[1] + y[1]) % P)
    sub = lambda x, y: ((x[0] - y[0]) % P, (x[1] - y[1]) % P)
    xi = lambda x: ((9 * x[0] - x[1]) % P, (9 * x[1] + x[0]) % P)

    def put(m, blk, el, neg=False):
        for h in range(2):
            for i, w in enumerate(K4.bal_limbs(el[h] if el[h] < P // 2 or True else el[h] - P)):
                m.v[blk + NL * h + i] = (-w if neg else w) & 0xFFFFFFFF

    # mul3: A <- A*B + H0*H1 + H2*H3
    for t in range(12):
        ops = [(rnd(), rnd()) for _ in range(6)]
        m = _m4([], rng, 0)
        for blk, el in zip((K4.A0, K4.B0, K4.HOME0, K4.HOME0 + K4.SLOT_DW, K4.HOME0 + 2 * K4.SLOT_DW, K4.HOME0 + 3 * K4.SLOT_DW), ops):
            put(m, blk, el)
        S.run_block(B["mul3"], m)
        w = f2a(f2a(f2m(ops[0], ops[1]), f2m(ops[2], ops[3])), f2m(ops[4], ops[5]))
        for h in range(2):
            assert (val(m, h) - w[h] * RPI) % P == 0, ("mul3", t, h)
        assert _is_norm(m, list(range(NL))) and _is_norm(m, list(range(NL, 2 * NL)))
    # mul2a: A <- A + H0*H1 + H2*H3 (the value in A enters the upper half of the column sums: added AS IT IS, not divided by R')
    for t in range(12):
        ops = [(rnd(), rnd()) for _ in range(5)]
        m = _m4([], rng, 0)
        for blk, el in zip((K4.A0, K4.HOME0, K4.HOME0 + K4.SLOT_DW, K4.HOME0 + 2 * K4.SLOT_DW, K4.HOME0 + 3 * K4.SLOT_DW), ops):
            put(m, blk, el)
        S.run_block(B["mul2a"], m)
        w = f2a(f2m(ops[1], ops[2]), f2m(ops[3], ops[4]))
        for h in range(2):
            assert (val(m, h) - w[h] * RPI - ops[0][h]) % P == 0, ("mul2a", t, h)
        assert _is_norm(m, list(range(NL))) and _is_norm(m, list(range(NL, 2 * NL)))
    # sqr4c / sqr4cx: Fq4 squaring of the cyclotomic squaring with the Granger-Scott recombination (zc, zd in home blocks 3, 4)
    for t in range(12):
        a, b, zc, zd = [(rnd(), rnd()) for _ in range(4)]
        for name in ("sqr4c", "sqr4cx"):
            m = _m4([], rng, 0)
            for r in range(K4.HOME0, K4.HOME0 + 3 * K4.SLOT_DW):
                m.v[r] = rng.getrandbits(32)     # scratch blocks hold garbage
            neg = t % 3 == 1                     # conjugates arrive as limb-wise negations
            put(m, K4.A0, a, neg)
            put(m, K4.B0, b, neg)
            put(m, K4.HOME0 + 3 * K4.SLOT_DW, zc)
            put(m, K4.HOME0 + 4 * K4.SLOT_DW, zd)
            S.run_block(B[name], m)
            r0 = f2a(f2m(a, a), xi(f2m(b, b)))
            tt = f2m(a, b)
            if name == "sqr4cx":
                tt = xi(tt)
            want = [3 * r0[0] * RPI - 2 * zc[0], 3 * r0[1] * RPI - 2 * zc[1], 6 * tt[0] * RPI + 2 * zd[0], 6 * tt[1] * RPI + 2 * zd[1]]
            for j in range(4):
                assert (val(m, j) - want[j]) % P == 0, (name, t, j)
                assert _is_norm(m, list(range(NL * j, NL * j + NL))) and abs(val(m, j)) < 0.52 * P, (name, t, j)
    # mul6 (fused Fq6 multiplication, schoolbook with lazy reduction, Karatsuba inside the a1 / a2 products): a in home blocks
    # 0..2 (also as unnormalised sums of two normalised values), b normalised in home blocks 3..5
    for t in range(10):
        a = [(rnd(), rnd()) for _ in range(3)]
        b = [(rnd(), rnd()) for _ in range(3)]
        m = _m4([], rng, 0)
        for r in list(range(K4.HOME0 + 6 * K4.SLOT_DW, K4.HOME0 + 8 * K4.SLOT_DW)) + list(range(0, 2 * K4.SLOT_DW)):
            m.v[r] = rng.getrandbits(32)
        book = {r: rng.getrandbits(32) for r in (K4.V_IDX8, K4.V_IDX, K4.V_TID, K4.V_FLAG)}     # parked in spare AGPRs meanwhile
        for r, x in book.items():
            m.v[r] = x
        for k, el in enumerate(a + b):
            blk = K4.HOME0 + K4.SLOT_DW * k
            if t % 2 and k < 3:                  # a sum of two normalised values, limb by limb
                part = (rng.randrange(P), rng.randrange(P))
                put(m, blk, part)
                rest = K4.bal_limbs((el[0] - part[0]) % P) + K4.bal_limbs((el[1] - part[1]) % P)
                for i in range(K4.SLOT_DW):
                    m.v[blk + i] = (m.v[blk + i] + rest[i]) & 0xFFFFFFFF
            else:
                put(m, blk, el)
        S.run_block(B["mul6"], m)
        assert all(m.v[r] == x for r, x in book.items())
        v = lambda i, j: f2m(a[i], b[j])
        want = [f2a(v(0, 0), xi(f2a(v(1, 2), v(2, 1)))), f2a(f2a(v(0, 1), v(1, 0)), xi(v(2, 2))), f2a(f2a(v(0, 2), v(1, 1)), v(2, 0))]
        where = [K4.HOME0 + 6 * K4.SLOT_DW, K4.HOME0 + 2 * K4.SLOT_DW, K4.A0]        # c0 -> home 6, c1 -> home 2, c2 -> A
        for c in range(3):
            for h in range(2):
                regs = list(range(where[c] + NL * h, where[c] + NL * h + NL))
                x = _sval([m.v[r] for r in regs])
                assert (x - want[c][h] * RPI) % P == 0, ("mul6", t, c, h)
                assert _is_norm(m, regs) and abs(x) < (1.6, 1.2, 0.6)[c] * P       # Prog._mul6_regs' bounds for operands below p, 2 p
    # extreme operands: every limb at the largest magnitude the routines accept (the simulator traps any signed 64-bit overflow
    # of a column accumulator; an int32 overflow shows up as a wrong residue elsewhere)
    top = K4.HALF
    H_ = lambda k: K4.HOME0 + K4.SLOT_DW * k
    cases = (("mul6", {**{H_(k): 2 for k in range(3)}, **{H_(k): 1 for k in range(3, 6)}}),
             ("sqr4c", {K4.A0: 1, K4.B0: 1, H_(3): 1, H_(4): 1}), ("sqr4cx", {K4.A0: 1, K4.B0: 1, H_(3): 1, H_(4): 1}),
             ("mul", {K4.A0: 2.5, K4.B0: 2.5}), ("mul3", {K4.A0: 1, K4.B0: 1, H_(0): 1, H_(1): 1, H_(2): 1, H_(3): 1}),
             ("mul3", {K4.A0: 2, K4.B0: 1, H_(0): 2, H_(1): 1, H_(2): 2, H_(3): 1}), ("sqr", {K4.A0: 1.8}),
             ("mul2a", {K4.A0: 3.9, H_(0): 2, H_(1): 1, H_(2): 2, H_(3): 1}))
    for name, mags in cases:
        for pattern in (lambda i: 1, lambda i: -1, lambda i: 1 if i % 2 else -1, lambda i: 1 if (i // 2) % 2 else -1):
            m = _m4([], rng, 0)
            for r in range(0, K4.HOME0 + 9 * K4.SLOT_DW):
                m.v[r] = rng.getrandbits(32) if r >= 2 * K4.SLOT_DW else 0
            for blk, mag_ in mags.items():
                for i in range(K4.SLOT_DW):
                    m.v[blk + i] = int(pattern(i) * mag_ * top) & 0xFFFFFFFF
            S.run_block(B[name], m)
            assert m.max_acc < (1 << 63)
    # redn on large representatives (x + t p) with unnormalised limbs of any int32 magnitude
    for t in range(40):
        xs = [rnd() + rng.randrange(-300, 300) * P for _ in range(2)]
        m = _m4([], rng, 0)
        for j, x in enumerate(xs):
            l = K4.bal_limbs(x)
            for i in range(NL - 1):
                bw = rng.randrange(-3, 4) if t % 2 else 0
                l[i] += bw << LB
                l[i + 1] -= bw
            for i in range(NL):
                m.v[NL * j + i] = l[i] & 0xFFFFFFFF
        S.run_block(B["redn"], m)
        for j in range(2):
            assert (val(m, j) - xs[j]) % P == 0 and abs(val(m, j)) < 0.52 * P, (t, j)
            assert _is_norm(m, list(range(NL * j, NL * j + NL)))
    # boundary conversions: ark 4 x u64 Montgomery (R = 2^256) <-> internal; cvtout is canonical
    for t in range(30):
        x = rnd()
        ext = (x << 256) % P
        m = _m4([], rng, 0)
        for i in range(8):
            m.v[i] = (ext >> (32 * i)) & 0xFFFFFFFF
        S.run_block(B["cvtin"], m)
        assert (val(m, 0) - x * K4.RP) % P == 0 and _is_norm(m, list(range(NL)))
        y = (x * K4.RP) % P + rng.randrange(-60, 61) * P          # any representative the kernels may hold
        m2 = _m4([], rng, 0)
        for i, w in enumerate(K4.bal_limbs(y)):
            m2.v[i] = w & 0xFFFFFFFF
        S.run_block(B["cvtout"], m2)
        assert sum(m2.v[i] << (32 * i) for i in range(8)) == ext


# ---------------------------------------------------------------- whole kernels on one lane
def _concretize(lines):
    ops = {"%0": "s[2:3]", "%1": "s[4:5]", "%2": "s[6:7]", "%3": "s[8:9]", "%4": "s10", "%5": "s11", "%6": "s[12:13]", "%7": "s14", "%8": "s[16:17]",
           "%9": "v255", "%10": "s18", "%11": "s19"}
    out = []
    for l in lines:
        l = re.sub(r"%(1[01]|\d)(?!\d)", lambda mo: ops["%" + mo.group(1)], l)
        out.append(l.replace("_%=", "_0"))
    return out


def _first_diff(a, b):
    for i, (x, y) in enumerate(zip(a, b)):
        if x != y:
            return f"call {i}: executed {a[max(0, i - 3):i + 3]} certified {b[max(0, i - 3):i + 3]}"
    return f"lengths {len(a)} vs {len(b)}"


def run_kernel(kb, g1=None, g2=None, fin=None, k=1, check_seq=True, profile=False):
    lines = _concretize(kb.build()) + ["s_endpgm"]
    m = S.Machine()

    def put64(base, words):
        for i, w in enumerate(words):
            m.gmem[base + 8 * i] = w & 0xFFFFFFFF
            m.gmem[base + 8 * i + 4] = (w >> 32) & 0xFFFFFFFF

    for base, words in ((G1B, g1), (G2B, g2), (FINB, fin)):
        if words is not None:
            put64(base, words)
    for name, val in (("s[2:3]", G1B), ("s[4:5]", G2B), ("s[6:7]", FINB), ("s[8:9]", OUTB), ("s10", 1), ("s11", k), ("s[12:13]", SCR),
                      ("s14", 256 * K4.SLOT_BYTES), ("s[16:17]", STAT), ("s18", 0), ("s19", 1)):
        m.sset(name, val)
    m.v[255] = 0
    m.call_log = []
    if profile:
        m.profile = {}
    S.run(lines, m)
    if check_seq:
        # the statically certified call sequence (value bounds, tools/kgen4_prog.py: certify_values) is the one executed
        rep = kb.certify_values(k_pairs=k)
        log = [re.sub(r"_\d+$", "", x) for x in m.call_log]
        assert log == rep["sequence"], _first_diff(log, rep["sequence"])
        assert rep["max_stored"] <= K4P.V_CAP
    out = []
    for c in range(12):
        v = 0
        for l in range(4):
            a = OUTB + (c * 4 + l) * 8
            v |= (m.gmem[a] | (m.gmem[a + 4] << 32)) << (64 * l)
        out.append(R.from_mont(v))
    return out, m


@pytest.fixture(scope="module")
def vec():
    return H.load_golden("bn254_vectors.json")


def _inputs(vec, i):
    g1 = [w for c in HX(vec["g1"][i]) for w in R.limbs4(R.to_mont(c))]
    g2 = [w for c in HX(vec["g2"][i]) for w in R.limbs4(R.to_mont(c))]
    return g1, g2


def test_miller_kernel_exact(vec):
    g1, g2 = _inputs(vec, 3)
    out, m = run_kernel(K4P.KernelBuilder(do_miller=True, do_fexp=False, track=True), g1, g2)
    assert out == HX(vec["miller"][3]) and STAT not in m.gmem
    assert m.max_acc < (1 << 63)


def test_final_exp_kernel(vec):
    x = HX(vec["fq12_in"][2])
    fin = [w for c in x for w in R.limbs4(R.to_mont(c))]
    out, m = run_kernel(K4P.KernelBuilder(do_miller=False, do_fexp=True), fin=fin)
    assert out == HX(vec["final_exp"][2])
    out, m = run_kernel(K4P.KernelBuilder(do_miller=False, do_fexp=True), fin=[0] * 48)
    assert m.gmem.get(STAT) == 1            # zero input: the reference panics


def test_pairing_kernel_generators(vec):
    """BASELINE.json configs[0]: e(G1gen, G2gen) through the fused default kernel."""
    g1, g2 = _inputs(vec, 0)
    out, m = run_kernel(K4P.KernelBuilder(do_miller=True, do_fexp=True), g1, g2)
    assert out == HX(vec["pairing"][0])


def _soa(rows):
    n = len(rows)
    out = [0] * (len(rows[0]) * 4 * n)
    for i, el in enumerate(rows):
        for c, x in enumerate(el):
            for l, w in enumerate(R.limbs4(R.to_mont(x))):
                out[(c * 4 + l) * n + i] = w
    return out


def test_multi_pairing_kernel_resident_points(vec):
    """k = 4 (the Groth16 shape) shared-f kernel, exact multi_miller_loop_native value (tracked scale).  k <= 4: the pairs'
    evaluation points stay packed in LDS slots 0, 1, 6, 7, the stream carries R only and prefetches Q for the addition steps."""
    g = vec["groups"][3]
    k, idx = g["k"], g["idx"]
    assert k == 4
    g1, g2 = _soa([HX(vec["g1"][i]) for i in idx]), _soa([HX(vec["g2"][i]) for i in idx])
    out, m = run_kernel(K4P.KernelBuilder(do_miller=True, do_fexp=False, track=True, multi=True), g1, g2, k=k)
    assert out == HX(g["miller"])


def test_multi_pairing_kernel_untracked_with_final_exp(vec):
    """k_mpairing itself (k = 4, untracked, final exponentiation on the shared f): the kernel whose streamed loop keeps pair 0's R and
    -- round 5 -- pair 1's X and Y on chip (LDS slots 2 and 7; pair 3's packed evaluation point moves into the LDS behind the eight
    slots and two VGPRs).  Expected value: final_exp_native of the golden multi_miller_loop_native value."""
    g = vec["groups"][3]
    k, idx = g["k"], g["idx"]
    assert k == 4
    kb = K4P.KernelBuilder(do_miller=True, do_fexp=True, track=False, multi=True)
    assert len(kb.r1_slots()) == 2 and kb.p3_moved()
    g1, g2 = _soa([HX(vec["g1"][i]) for i in idx]), _soa([HX(vec["g2"][i]) for i in idx])
    out, m = run_kernel(kb, g1, g2, k=k)
    assert out == R.final_exp_native(HX(g["miller"])) and STAT not in m.gmem


def test_multi_miller_kernel_untracked(vec):
    """k_mmiller_u: the k-pair Miller loop WITHOUT the line scale and without a final exponentiation (the chunks of the spread route, whose values are
    multiplied and exponentiated later): its value differs from multi_miller_loop_native's by a factor the final exponentiation kills -- and it walks the
    short chain.  final_exp_native of it must be final_exp_native of the golden Miller value."""
    g = [x for x in vec["groups"] if x["k"] == 3][0]
    k, idx = g["k"], g["idx"]
    kb = K4P.KernelBuilder(do_miller=True, do_fexp=False, track=False, multi=True)
    assert kb.naf == K4P.SIX_U_PLUS_2_SHORT
    g1, g2 = _soa([HX(vec["g1"][i]) for i in idx]), _soa([HX(vec["g2"][i]) for i in idx])
    out, m = run_kernel(kb, g1, g2, k=k)
    assert out != HX(g["miller"]) and R.final_exp_native(out) == R.final_exp_native(HX(g["miller"])) and STAT not in m.gmem


def test_multi_pairing_kernel_streamed_points(vec):
    """k = 5: more pairs than stay on chip -- P travels with R through the prefetch buffer, Q is fetched in the addition steps.
    Expected value: the product of the single Miller values (the reference's own T1, miller_loop_native.rs:336-348)."""
    idx = [1, 2, 5, 8, 11]
    g1, g2 = _soa([HX(vec["g1"][i]) for i in idx]), _soa([HX(vec["g2"][i]) for i in idx])
    out, m = run_kernel(K4P.KernelBuilder(do_miller=True, do_fexp=False, track=True, multi=True), g1, g2, k=len(idx))
    want = HX(vec["miller"][idx[0]])
    for i in idx[1:]:
        want = R.fq12_mul(want, HX(vec["miller"][i]))
    assert out == want


def test_helper_kernel(vec):
    """k_op: MyFq12 Mul, frobenius_map_native and pow_native (general, non-unitary elements)."""
    xs = [HX(x) for x in vec["fq12_in"]]
    fq12_words = lambda x: [w for c in x for w in R.limbs4(R.to_mont(c))]
    kb = K4P.KernelBuilder(helper=True)
    a, b = xs[1], xs[2]
    # Mul: a * b (golden fq12_mul[i] = fq12_in[i] * fq12_in[i + 1])
    out, m = run_kernel(kb, g1=fq12_words(b), fin=fq12_words(a), k=kb.OP_MUL, check_seq=False)
    assert out == HX(vec["fq12_mul"][1]) and STAT not in m.gmem
    for power in (1, 2, 3, 6, 11):
        out, m = run_kernel(kb, fin=fq12_words(a), k=kb.OP_FROB | power << 8, check_seq=False)
        assert out == HX(vec["frobenius"][str(power)][1]), f"frobenius power {power}"
    # pow_native(a, [BN_X]) with the reference's NAF (get_naf), -1 digits divide
    naf = R.get_naf([R.BN_X])
    while naf[-1] == 0:
        naf.pop()
    assert naf[-1] == 1
    packed = bytes((d & 0xFF) for d in naf) + b"\0" * 8
    words = [int.from_bytes(packed[8 * i: 8 * i + 8], "little") for i in range(len(packed) // 8)]
    out, m = run_kernel(kb, g2=words, fin=fq12_words(a), k=kb.OP_POW | 1 << 8 | len(naf) << 16, check_seq=False)
    assert out == HX(vec["pow_x"][1]) and STAT not in m.gmem
    # an exponent without -1 digits never divides: a = 0 is not an error (0^5 = 0)
    out, m = run_kernel(kb, g2=[0x0000000000010001], fin=[0] * 48, k=kb.OP_POW | 3 << 16, check_seq=False)
    assert out == [0] * 12 and STAT not in m.gmem
    kb.certify_helper()


def test_generate_kernel():
    """k_generate on one lane: P = [s] G1, Q = [t] G2 by the fixed-base radix-16 method (table of tools/gen_tables.py), scalars
    from SplitMix64 as plonky2-bn254-pairing_amd.generator_scalars states; affine canonical output."""
    import gen_tables as GT
    pk = H.pkg()
    kb = K4P.KernelBuilder(generate=True)
    lines = _concretize(kb.build()) + ["s_endpgm"]
    TAB = 0x600000
    seed, n = 0xB2540001, 300
    words = GT.all_words()
    for item, tid in ((1, 1),):                          # second work item, lane 1
        m = S.Machine()
        for i, w in enumerate(words):
            m.gmem[TAB + 4 * i] = w & 0xFFFFFFFF
        for name, val in (("s[2:3]", G1B), ("s[4:5]", G2B), ("s[6:7]", TAB), ("s[8:9]", seed), ("s10", n), ("s11", 1), ("s[12:13]", SCR),
                          ("s14", 256 * K4.SLOT_BYTES), ("s[16:17]", STAT), ("s18", item), ("s19", 4)):
            m.sset(name, val)
        m.v[255] = tid
        S.run(lines, m)
        idx = item * 256 + tid
        s, t = pk.generator_scalars(seed, idx)
        P_, Q_ = R.g1_mul(R.G1_GEN, s % R.R_ORDER), R.g2_mul(R.G2_GEN, t % R.R_ORDER)

        def rd(base, planes):
            out = []
            for c in range(planes):
                v = 0
                for l in range(4):
                    a = base + ((c * 4 + l) * n + idx) * 8
                    v |= (m.gmem[a] | (m.gmem[a + 4] << 32)) << (64 * l)
                out.append(R.from_mont(v))
            return out
        assert rd(G1B, 2) == list(P_)
        assert rd(G2B, 4) == [Q_[0][0], Q_[0][1], Q_[1][0], Q_[1][1]]
        print("generate kernel:", m.count, "instructions per pair")


def _fq2(x):
    return (x[0] % P, x[1] % P)


def test_l1_fused_point_steps():
    """dblstep / addstep (fused G2 steps + line coefficients) against the projective formulas in big integers."""
    rng = random.Random(11)
    RPI = pow(K4.RP, -1, P)
    f2m = lambda x, y: ((x[0] * y[0] - x[1] * y[1]) % P, (x[0] * y[1] + x[1] * y[0]) % P)
    f2a = lambda x, y: ((x[0] + y[0]) % P, (x[1] + y[1]) % P)
    f2s = lambda x, y: ((x[0] - y[0]) % P, (x[1] - y[1]) % P)
    f2k = lambda x, k: (x[0] * k % P, x[1] * k % P)
    xi = lambda x: ((9 * x[0] - x[1]) % P, (9 * x[1] + x[0]) % P)
    inv82 = pow(82, -1, P)
    three_b = (81 * inv82 % P, (-9 * inv82) % P)
    H_ = lambda k: K4.HOME0 + K4.SLOT_DW * k

    def put(m, blk, el):
        for h in range(2):
            for i, w in enumerate(K4.bal_limbs(K4.mont4(el[h]))):
                m.v[blk + NL * h + i] = w & 0xFFFFFFFF

    def get(m, blk):
        return tuple(_sval([m.v[blk + NL * h + i] for i in range(NL)]) * RPI % P for h in range(2))

    B = {n: _body(n) for n in ("dblstep", "addstep")}
    for t in range(6):
        X, Y, Z = [(rng.randrange(P), rng.randrange(P)) for _ in range(3)]
        px, py = rng.randrange(P), rng.randrange(P)
        m = _m4([], rng, 0)
        for r in range(0, K4.HOME0 + 9 * K4.SLOT_DW):
            m.v[r] = rng.getrandbits(32)
        put(m, H_(0), X); put(m, H_(1), Y); put(m, H_(2), Z)
        put(m, K4.B0, (px, 0))
        for i, w in enumerate(K4.bal_limbs(K4.mont4(py))):
            m.v[K4.B0 + NL + i] = w & 0xFFFFFFFF
        S.run_block(B["dblstep"], m)
        Bq, C = f2m(Y, Y), f2m(Z, Z)
        E = f2m(C, three_b)
        Fv = f2k(E, 3)
        Hh = f2k(f2m(Y, Z), 2)
        # the routine returns the new point scaled by xi^2 (no multiplication by the curve constant 3 b' = 9 / xi)
        xi2 = lambda x: xi(xi(x))
        want = {"X3": xi2(f2m(f2k(f2m(X, Y), 2), f2s(Bq, Fv))), "Y3": xi2(f2s(f2m(f2a(Bq, Fv), f2a(Bq, Fv)), f2k(f2m(E, E), 12))),
                "Z3": xi2(f2k(f2m(Bq, Hh), 4)), "L0": f2s(xi(Bq), f2k(C, 9)), "L3": f2k(Hh, py), "L4": f2k(f2m(X, X), (-3 * px) % P)}
        where = {"X3": H_(0), "Y3": H_(1), "Z3": H_(2), "L0": H_(7), "L3": H_(4), "L4": H_(5)}
        for k_, w in want.items():
            assert get(m, where[k_]) == w, ("dblstep", t, k_)
            if k_ != "L0":                       # L0 = xi B - N leaves as a limb-wise difference (two units)
                assert _is_norm(m, list(range(where[k_], where[k_] + NL))) and _is_norm(m, list(range(where[k_] + NL, where[k_] + 2 * NL)))
            if k_ in ("X3", "Y3", "Z3"):
                assert all(abs(_sval([m.v[where[k_] + NL * h + i] for i in range(NL)])) < 0.52 * P for h in range(2)), k_
        assert m.max_acc < (1 << 63)
    for t in range(6):
        X, Y, Z, x2, y2 = [(rng.randrange(P), rng.randrange(P)) for _ in range(5)]
        px, py = rng.randrange(P), rng.randrange(P)
        m = _m4([], rng, 0)
        for r in range(0, K4.HOME0 + 9 * K4.SLOT_DW):
            m.v[r] = rng.getrandbits(32)
        for k_, el in enumerate((X, Y, Z, x2, y2)):
            put(m, H_(k_), el)
        put(m, K4.B0, (px, 0))
        for i, w in enumerate(K4.bal_limbs(K4.mont4(py))):
            m.v[K4.B0 + NL + i] = w & 0xFFFFFFFF
        S.run_block(B["addstep"], m)
        th, mu = f2s(Y, f2m(y2, Z)), f2s(X, f2m(x2, Z))
        Cc, D = f2m(th, th), f2m(mu, mu)
        E, Fz, G = f2m(mu, D), f2m(Z, Cc), f2m(X, D)
        Hh = f2s(f2a(E, Fz), f2k(G, 2))
        want = {"X3": f2m(mu, Hh), "Y3": f2s(f2m(th, f2s(G, Hh)), f2m(E, Y)), "Z3": f2m(Z, E),
                "L2": f2k(mu, (-py) % P), "L3": f2k(th, px), "L5": f2s(f2m(X, y2), f2m(x2, Y))}
        where = {"X3": H_(6), "Y3": H_(4), "Z3": H_(2), "L2": H_(7), "L3": H_(8), "L5": K4.A0}
        for k_, w in want.items():
            assert get(m, where[k_]) == w, ("addstep", t, k_)
            assert _is_norm(m, list(range(where[k_], where[k_] + NL))) and _is_norm(m, list(range(where[k_] + NL, where[k_] + 2 * NL)))
        assert m.max_acc < (1 << 63)


def test_simulator_accumulator_check_has_teeth():
    """64-bit column accumulators may wrap transiently (kfips: the difference products come before the terms that cancel them);
    consuming a wrapped value is what the simulator must catch."""
    big = (1 << 31) - 1
    m = S.Machine()
    m.v[2], m.v[3], m.v[6], m.v[7] = big, big, (-big) & 0xFFFFFFFF, big
    wrap = [f"v_mad_i64_i32 v[4:5], vcc, v2, v3, {'0' if i == 0 else 'v[4:5]'}" for i in range(3)]          # 3 * 2^62: beyond 2^63
    back = ["v_mad_i64_i32 v[4:5], vcc, v6, v7, v[4:5]"] * 2                                                 # - 2 * 2^62: back inside
    S.run_block(wrap + back + ["v_ashrrev_i64 v[4:5], 29, v[4:5]"], m)
    assert m.transient_wraps >= 1 and (m.v[4] | (m.v[5] << 32)) == ((big * big) >> 29)
    m = S.Machine()
    m.v[2], m.v[3] = big, big
    with pytest.raises(S.SimError, match="wrapped"):
        S.run_block(wrap + ["v_ashrrev_i64 v[4:5], 29, v[4:5]"], m)
    m = S.Machine()
    m.v[2], m.v[3] = big, big
    with pytest.raises(S.SimError, match="wrapped"):
        S.run_block(wrap + ["v_mov_b32_e32 v8, v4"], m)


def test_subgroup_check_kernel():
    """k_subcheck on one lane (the instruction simulator): the G2 subgroup criterion of ark's `G2Affine::new` contract
    (/root/reference/src/miller_loop_native.rs:303,311) as [x + 1]Q + psi([x]Q) + psi^2([x]Q) == psi^3([2x]Q) in Jacobian coordinates --
    verdict word 0 for points of the r-torsion (golden inputs, a cofactor-cleared twist point), 1 for random points of the twist, equal to
    the big-int definition [r]Q == O each time; every routine keeps the value contract."""
    import random
    from test_point_checks import twist_point
    kb = K4P.KernelBuilder(subcheck=True)
    lines = _concretize(kb.build()) + ["s_endpgm"]
    for n_ in ("L2_sdbl", "L2_smadd", "L2_sfin"):
        assert kb._check_routine(n_) <= K4P.V_CAP
    vec = H.load_golden("bn254_vectors.json")
    rng = random.Random(11)
    tw = twist_point(rng)
    cases = [(tuple(map(tuple, (HX(vec["g2"][1])[:2], HX(vec["g2"][1])[2:]))), True), (tw, False),
             (R.g2_mul(tw, 2 * R.P - R.R_ORDER), True), (twist_point(rng), False)]
    for Q, want_in in cases:
        assert (R.g2_mul(Q, R.R_ORDER) is None) == want_in
        g2 = [w for c in (Q[0][0], Q[0][1], Q[1][0], Q[1][1]) for w in R.limbs4(R.to_mont(c))]
        m = S.Machine()
        for i, w in enumerate(g2):
            m.gmem[G2B + 8 * i] = w & 0xFFFFFFFF
            m.gmem[G2B + 8 * i + 4] = (w >> 32) & 0xFFFFFFFF
        for name, val in (("s[2:3]", G1B), ("s[4:5]", G2B), ("s[6:7]", FINB), ("s[8:9]", OUTB), ("s10", 1), ("s11", 1), ("s[12:13]", SCR),
                          ("s14", 256 * K4.SLOT_BYTES), ("s[16:17]", STAT), ("s18", 0), ("s19", 1)):
            m.sset(name, val)
        m.v[255] = 0
        S.run(lines, m)
        assert m.gmem[OUTB] == (0 if want_in else 1), (Q, m.gmem[OUTB])
        assert m.max_acc < (1 << 63) and STAT not in m.gmem
    print("subgroup check kernel:", m.count, "instructions per point")


@pytest.mark.parametrize("mode", [7])
def test_pairing_kernel_element_major_io(vec, mode):
    """The I/O layout bits of the kernels' k argument (bits 28..30: inputs element-major / output element-major / ... in ark's Fq12 order):
    k_pairing on lane 2 of a four-pair batch reads / writes the same limbs at the element-major addresses -- what a caller of
    src/pairing.rs:20-22 holds (`&[G1Affine]`, `&[G2Affine]`, `Vec<Fq12>`) goes to the kernel as it is."""
    n, lane = 4, 2
    rows1 = [HX(vec["g1"][i]) for i in range(n)]
    rows2 = [HX(vec["g2"][i]) for i in range(n)]
    elems = lambda rows: [w for el in rows for c in el for w in R.limbs4(R.to_mont(c))]
    g1 = elems(rows1) if mode & 1 else _soa(rows1)
    g2 = elems(rows2) if mode & 1 else _soa(rows2)
    kb = K4P.KernelBuilder(do_miller=True, do_fexp=True)
    lines = _concretize(kb.build()) + ["s_endpgm"]
    m = S.Machine()
    for base, words in ((G1B, g1), (G2B, g2)):
        for i, w in enumerate(words):
            m.gmem[base + 8 * i] = w & 0xFFFFFFFF
            m.gmem[base + 8 * i + 4] = (w >> 32) & 0xFFFFFFFF
    for name, val in (("s[2:3]", G1B), ("s[4:5]", G2B), ("s[6:7]", FINB), ("s[8:9]", OUTB), ("s10", n), ("s11", 1 | (mode << 28)), ("s[12:13]", SCR),
                      ("s14", 256 * K4.SLOT_BYTES), ("s[16:17]", STAT), ("s18", 0), ("s19", 1)):
        m.sset(name, val)
    m.v[255] = lane
    S.run(lines, m)
    want = HX(vec["pairing"][lane])
    got = []
    for c in range(12):
        v = 0
        for l in range(4):
            if mode & 2:
                j = c if not (mode & 4) else [jj for jj in range(12) if _ark_to_my(jj) == c][0]
                a = OUTB + (lane * 48 + j * 4 + l) * 8
            else:
                a = OUTB + ((c * 4 + l) * n + lane) * 8
            v |= (m.gmem[a] | (m.gmem[a + 4] << 32)) << (64 * l)
        got.append(R.from_mont(v))
    assert got == want
    # nothing was written outside the lane's own 384 bytes (element-major) / its own column (limb-major)
    touched = {a for a in m.gmem if OUTB <= a < OUTB + 8 * 48 * n}
    mine = {OUTB + (lane * 48 + w) * 8 + h for w in range(48) for h in (0, 4)} if mode & 2 else {OUTB + (w * n + lane) * 8 + h for w in range(48) for h in (0, 4)}
    assert touched == mine


def _ark_to_my(j):
    """MyFq12 coefficient index held at position j of ark's flat Fq12 (include/bn254_pairing.h: bn254_myfq12_to_ark_index)"""
    h, k, e = j // 6, (j % 6) // 2, j % 2
    return (2 * k + h) + 6 * e


@pytest.mark.parametrize("mode", [7])
def test_multi_pairing_kernel_element_major_io(vec, mode):
    """The k-pair kernels in element-major mode: group 1 of three two-pair groups (lane 1) reads its pairs at (g k + j) x 64 / 128 bytes
    and writes its Fq12 at g x 384 bytes (mode 7: in ark's coefficient order) -- the exact multi_miller_loop_native value of that group."""
    k, n, lane = 2, 3, 1
    gi = [[(g * k + j) % 12 for j in range(k)] for g in range(n)]
    rows1 = [HX(vec["g1"][i]) for g in gi for i in g]
    rows2 = [HX(vec["g2"][i]) for g in gi for i in g]
    elems = lambda rows: [w for el in rows for c in el for w in R.limbs4(R.to_mont(c))]
    want = None
    for i in gi[lane]:
        want = HX(vec["miller"][i]) if want is None else R.fq12_mul(want, HX(vec["miller"][i]))
    kb = K4P.KernelBuilder(do_miller=True, do_fexp=False, track=True, multi=True)
    lines = _concretize(kb.build()) + ["s_endpgm"]
    m = S.Machine()
    for base, words in ((G1B, elems(rows1)), (G2B, elems(rows2))):
        for i, w in enumerate(words):
            m.gmem[base + 8 * i] = w & 0xFFFFFFFF
            m.gmem[base + 8 * i + 4] = (w >> 32) & 0xFFFFFFFF
    for name, val in (("s[2:3]", G1B), ("s[4:5]", G2B), ("s[6:7]", FINB), ("s[8:9]", OUTB), ("s10", n), ("s11", k | (mode << 28)), ("s[12:13]", SCR),
                      ("s14", 256 * K4.SLOT_BYTES), ("s[16:17]", STAT), ("s18", 0), ("s19", 1)):
        m.sset(name, val)
    m.v[255] = lane
    S.run(lines, m)
    got = []
    for c in range(12):
        j = c if not (mode & 4) else [jj for jj in range(12) if _ark_to_my(jj) == c][0]
        v = 0
        for l in range(4):
            a = OUTB + (lane * 48 + j * 4 + l) * 8
            v |= (m.gmem[a] | (m.gmem[a + 4] << 32)) << (64 * l)
        got.append(R.from_mont(v))
    assert got == want
    assert {a for a in m.gmem if OUTB <= a < OUTB + 8 * 48 * n} == {OUTB + (lane * 48 + w) * 8 + h for w in range(48) for h in (0, 4)}


def test_fixed_g2_kernels(vec):
    """The line-table kernel (one fixed G2 point per lane: every step's line coefficients without an evaluation point) and the fixed-G2
    pairing kernel that consumes it: group = the group's own pair + two pairs whose G2 points are fixed for the batch.  Lane 1 of a
    two-group batch gives final_exp_native of the product of the three Miller values --
    multi_miller_loop_native's value (miller_loop_native.rs:192-282) after the final exponentiation, which does not see the chain."""
    kf, n, lane = 2, 2, 1
    fixed_idx = [4, 7]
    grp = [[1, 2, 3], [5, 6, 8]]                     # golden indices of (own pair, P of fixed pair 0, P of fixed pair 1)
    TABB = 0x700000
    kl = K4P.KernelBuilder(lines=True)
    lines_l = _concretize(kl.build()) + ["s_endpgm"]
    for n_ in ("L2_ldbl", "L2_ladd", "L2_ladd_last", "L2_linv", "L2_lnorm"):
        assert kl._check_routine(n_) <= K4P.V_CAP
    gmem = {}
    g2f = _soa([HX(vec["g2"][i]) for i in fixed_idx])
    for ln in range(kf):
        m = S.Machine()
        m.gmem.update(gmem)
        for i, w in enumerate(g2f):
            m.gmem[G2B + 8 * i] = w & 0xFFFFFFFF
            m.gmem[G2B + 8 * i + 4] = (w >> 32) & 0xFFFFFFFF
        for name, val in (("s[2:3]", G1B), ("s[4:5]", G2B), ("s[6:7]", FINB), ("s[8:9]", TABB), ("s10", kf), ("s11", 1), ("s[12:13]", SCR),
                          ("s14", 256 * K4.SLOT_BYTES), ("s[16:17]", STAT), ("s18", 0), ("s19", 1)):
            m.sset(name, val)
        m.v[255] = ln
        S.run(lines_l, m)
        gmem = {a: v for a, v in m.gmem.items() if TABB <= a < TABB + (1 << 20)}
    tab_bytes = kf * kl.n_fixed_lines * kl.FIX_LINE_SLOTS * K4.SLOT_BYTES
    assert len(gmem) == tab_bytes // 4 and kl.n_fixed_lines == 87
    kb = K4P.KernelBuilder(fixed=True)
    assert kb.n_fixed_lines == kl.n_fixed_lines and kb.naf == kl.naf
    lines_f = _concretize(kb.build()) + ["s_endpgm"]
    assert len(kb.fsp_variant) == 4
    for n_ in [f"L2_fix_{j}" for j in range(4)] + sorted(set(kb.fsp_variant)):
        assert kb._check_routine(n_) <= K4P.V_CAP
    assert kb.certify_values(1)["max_stored"] <= K4P.V_CAP          # (walks the fixed lines of four pairs behind every step: every variant's entry bounds)
    want = None
    for pi, qi in zip(grp[lane], [grp[lane][0]] + fixed_idx):
        # e(P_pi, Q_qi): the golden Miller values are for equal indices only -- compute through the big-int restatement
        mv = R.miller_loop_native((tuple(HX(vec["g2"][qi])[:2]), tuple(HX(vec["g2"][qi])[2:])), tuple(HX(vec["g1"][pi])))
        want = mv if want is None else R.fq12_mul(want, mv)
    want = R.final_exp_native(want)
    rows1 = [HX(vec["g1"][i]) for g in grp for i in g]
    rows2 = [HX(vec["g2"][g[0]]) for g in grp]
    elems = lambda rows: [w for el in rows for c in el for w in R.limbs4(R.to_mont(c))]
    for mode in (0,):                 # (element-major / ark order of this kernel: tests/test_gpu_fixed_g2.py; of its I/O code: test_pairing_kernel_element_major_io)
        m = S.Machine()
        m.gmem.update(gmem)
        for base, words in ((G1B, elems(rows1) if mode else _soa(rows1)), (G2B, elems(rows2) if mode else _soa(rows2))):
            for i, w in enumerate(words):
                m.gmem[base + 8 * i] = w & 0xFFFFFFFF
                m.gmem[base + 8 * i + 4] = (w >> 32) & 0xFFFFFFFF
        for name, val in (("s[2:3]", G1B), ("s[4:5]", G2B), ("s[6:7]", TABB), ("s[8:9]", OUTB), ("s10", n), ("s11", kf | (mode << 28)), ("s[12:13]", SCR),
                          ("s14", 256 * K4.SLOT_BYTES), ("s[16:17]", STAT), ("s18", 0), ("s19", 1)):
            m.sset(name, val)
        m.v[255] = lane
        S.run(lines_f, m)
        got = []
        for c in range(12):
            j = c if not (mode & 4) else [jj for jj in range(12) if _ark_to_my(jj) == c][0]
            v = 0
            for l in range(4):
                a = (OUTB + (lane * 48 + j * 4 + l) * 8) if mode & 2 else (OUTB + ((c * 4 + l) * n + lane) * 8)
                v |= (m.gmem[a] | (m.gmem[a + 4] << 32)) << (64 * l)
            got.append(R.from_mont(v))
        assert got == want, mode
        assert m.max_acc < (1 << 63) and STAT not in m.gmem
    print("fixed-G2 kernel:", m.count, "instructions for one group of 1 + 2 pairs")
    # groups WITHOUT a pair of their own (mode bit 3: a KZG-style check, every G2 point one of the table's): f = 1, squarings and table lines only
    assert kb.certify_values(1, own_pair=False)["max_stored"] <= K4P.V_CAP
    want = None
    for pi, qi in zip(grp[lane][1:], fixed_idx):
        mv = R.miller_loop_native((tuple(HX(vec["g2"][qi])[:2]), tuple(HX(vec["g2"][qi])[2:])), tuple(HX(vec["g1"][pi])))
        want = mv if want is None else R.fq12_mul(want, mv)
    want = R.final_exp_native(want)
    rows1 = [HX(vec["g1"][i]) for g in grp for i in g[1:]]
    m = S.Machine()
    m.gmem.update(gmem)
    for i, w in enumerate(_soa(rows1)):
        m.gmem[G1B + 8 * i] = w & 0xFFFFFFFF
        m.gmem[G1B + 8 * i + 4] = (w >> 32) & 0xFFFFFFFF
    for name, val in (("s[2:3]", G1B), ("s[4:5]", 0), ("s[6:7]", TABB), ("s[8:9]", OUTB), ("s10", n), ("s11", kf | (8 << 28)), ("s[12:13]", SCR),
                      ("s14", 256 * K4.SLOT_BYTES), ("s[16:17]", STAT), ("s18", 0), ("s19", 1)):
        m.sset(name, val)
    m.v[255] = lane
    S.run(lines_f, m)
    got = []
    for c in range(12):
        v = 0
        for l in range(4):
            a = OUTB + ((c * 4 + l) * n + lane) * 8
            v |= (m.gmem[a] | (m.gmem[a + 4] << 32)) << (64 * l)
        got.append(R.from_mont(v))
    assert got == want and m.max_acc < (1 << 63) and STAT not in m.gmem
    print("fixed-G2 kernel, no own pair:", m.count, "instructions for one group of 2 fixed pairs")
