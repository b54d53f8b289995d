"""The generated gfx950 kernels (tools/kgen.py, tools/kgen_prog.py) executed on ONE lane by the
instruction-level simulator tools/ksim.py, against big-int arithmetic and the golden fixtures.
This is the CPU-side check of the product's instruction stream (the GPU parity tests run the same
text on the hardware): results, 64-bit operand alignment, uninitialised-register reads and the
VALU-writes-SGPR -> VALU-reads hazard distance are all verified here."""
import os
import random
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kgen as K  # noqa: E402
import kgen_prog as KP  # noqa: E402
import ksim as S  # noqa: E402
import helpers as H  # noqa: E402
from helpers import R  # noqa: E402

P = K.P_INT
RI = pow(1 << 256, -1, P)
HX = lambda xs: [int(x, 16) for x in xs]  # noqa: E731


# ---------------------------------------------------------------- L1 routines
def _l1_machine(vals):
    m = S.Machine()
    for i in range(8):
        m.s[K.S_P + i] = K.P_LIMBS[i]
        m.v[K.PV0 + i] = K.P_LIMBS[i]
    m.s[K.S_N0] = K.N0
    for j, x in enumerate(vals):
        for i, w in enumerate(K.limbs8(x)):
            m.v[8 * j + i] = w
    return m


def _val(m, j):
    return sum(m.v[8 * j + i] << (32 * i) for i in range(8))


def _body(name):
    e = K.Emitter()
    getattr(K.L1(e), name)()
    return e.finalize()


def test_l1_routines():
    rng = random.Random(5)

    def rnd():
        return rng.choice([0, 1, P - 1, P - 2, 1 << 253]) if rng.random() < 0.3 else rng.randrange(P)

    names = ["r_mul", "r_sqr", "r_mulfq", "r_add", "r_sub", "r_rsub", "r_dbl", "r_neg", "r_negc1", "r_mulxi", "r_fqmul", "r_fqsqr"]
    B = {n: _body(n) for n in names}
    want = {
        "r_mul": lambda a0, a1, b0, b1: ((a0 * b0 - a1 * b1) * RI % P, (a0 * b1 + a1 * b0) * RI % P),
        "r_sqr": lambda a0, a1, b0, b1: ((a0 * a0 - a1 * a1) * RI % P, 2 * a0 * a1 * RI % P),
        "r_mulfq": lambda a0, a1, b0, b1: (a0 * b0 * RI % P, a1 * b0 * RI % P),
        "r_add": lambda a0, a1, b0, b1: ((a0 + b0) % P, (a1 + b1) % P),
        "r_sub": lambda a0, a1, b0, b1: ((a0 - b0) % P, (a1 - b1) % P),
        "r_rsub": lambda a0, a1, b0, b1: ((b0 - a0) % P, (b1 - a1) % P),
        "r_dbl": lambda a0, a1, b0, b1: (2 * a0 % P, 2 * a1 % P),
        "r_neg": lambda a0, a1, b0, b1: (-a0 % P, -a1 % P),
        "r_negc1": lambda a0, a1, b0, b1: (a0, -a1 % P),
        "r_mulxi": lambda a0, a1, b0, b1: ((9 * a0 - a1) % P, (a0 + 9 * a1) % P),
        "r_fqmul": lambda a0, a1, b0, b1: (a0 * b0 * RI % P, None),
        "r_fqsqr": lambda a0, a1, b0, b1: (a0 * a0 * RI % P, None),
    }
    for _ in range(25):
        vals = [rnd() for _ in range(4)]
        for n in names:
            m = _l1_machine(vals)
            S.run_block(B[n], m)
            w0, w1 = want[n](*vals)
            assert _val(m, 0) == w0, n
            if w1 is not None:
                assert _val(m, 1) == w1, n


# ---------------------------------------------------------------- whole kernels, one lane
OPS = {0: "s[2:3]", 1: "s[4:5]", 2: "s[6:7]", 3: "s[8:9]", 4: "s10", 5: "s11", 6: "s[12:13]", 7: "s14", 8: "s[16:17]", 9: "v255", 10: "s18",
       11: "s19"}
G1B, G2B, FINB, OUTB, SCR, STAT = 0x10000000, 0x20000000, 0x30000000, 0x40000000, 0x50000000, 0x60000000


def _concretize(lines):
    out = []
    for l in lines:
        l = l.replace("%=", "0")
        out.append(re.sub(r"%(\d+)", lambda mm: OPS[int(mm.group(1))], l))
    return out


def _run_kernel(kb, g1=None, g2=None, fin=None):
    lines = _concretize(kb.build()) + ["s_endpgm"]
    m = S.Machine()

    def put64(base, words):
        for i, w in enumerate(words):
            m.gmem[base + 8 * i] = w & 0xFFFFFFFF
            m.gmem[base + 8 * i + 4] = (w >> 32) & 0xFFFFFFFF

    for base, words in ((G1B, g1), (G2B, g2), (FINB, fin)):
        if words is not None:
            put64(base, words)
    for name, val in (("s[2:3]", G1B), ("s[4:5]", G2B), ("s[6:7]", FINB), ("s[8:9]", OUTB), ("s10", 1), ("s11", 1), ("s[12:13]", SCR),
                      ("s14", 256 * 64), ("s[16:17]", STAT), ("s18", 0), ("s19", 1)):
        m.sset(name, val)
    m.v[255] = 0
    S.run(lines, m)
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
    """k2_miller: projective steps + tracked scale must give miller_loop_native's exact value."""
    g1, g2 = _inputs(vec, 2)
    out, m = _run_kernel(KP.KernelBuilder(do_miller=True, do_fexp=False, track=True), g1, g2)
    assert out == HX(vec["miller"][2])
    assert STAT not in m.gmem            # no zero-divisor flag


def test_final_exp_kernel(vec):
    x = HX(vec["fq12_in"][1])
    fin = [w for c in x for w in R.limbs4(R.to_mont(c))]
    out, m = _run_kernel(KP.KernelBuilder(do_miller=False, do_fexp=True), fin=fin)
    assert out == HX(vec["final_exp"][1])


# ---------------------------------------------------------------- v3: signed radix-2^27 limbs (the default kernels)
import kgen3 as K3  # noqa: E402
import kgen3_prog as K3P  # noqa: E402


def _sval(limbs):
    v = 0
    for i, l in enumerate(limbs):
        l = l - (1 << 32) if l >> 31 else l
        v += l << (K3.LB * i)
    return v


def _redundant(x, rng, slack):
    """A signed redundant representation of x + k p with limb borrows of up to `slack` units."""
    y = x + (rng.randrange(-3, 4) if slack else 0) * P
    l = K3.to_limbs(abs(y))
    if y < 0:
        l = [-t for t in l]
    for i in range(K3.NL - 1):
        b = rng.randrange(-slack, slack + 1) if slack else 0
        l[i] += b << K3.LB
        l[i + 1] -= b
    return l


def _m3(vals, rng, slack):
    m = S.Machine()
    for i in range(K3.NL):
        m.s[K3.S_P + i] = K3.P_L[i]
    m.s[K3.S_N0] = K3.N0P
    for j, x in enumerate(vals):
        for i, w in enumerate(_redundant(x, rng, slack)):
            m.v[K3.NL * j + i] = w & 0xFFFFFFFF
    return m


def test_l1_v3_routines():
    """Redundant signed operands (negative limbs, limbs above 27 bits, value offsets by multiples of p): results
    are checked mod p; the simulator traps any signed 64-bit column overflow."""
    rng = random.Random(7)
    RPI = pow(K3.RP, -1, P)

    def body(n):
        e = K.Emitter()
        getattr(K3.L1v3(e), "r_" + n)()
        return e.finalize()

    B = {n: body(n) for n in K3.L1V3_NAMES}

    def rnd():
        return rng.choice([0, 1, P - 1, P - 2]) if rng.random() < 0.25 else rng.randrange(P)

    def val(m, j):
        return _sval([m.v[K3.NL * j + i] for i in range(K3.NL)])

    for t in range(30):
        a0, a1, b0, b1 = rnd(), rnd(), rnd(), rnd()
        sl = [0, 1, 3][t % 3]
        want = {"mul": ((a0 * b0 - a1 * b1) * RPI, (a0 * b1 + a1 * b0) * RPI), "sqr": ((a0 * a0 - a1 * a1) * RPI, 2 * a0 * a1 * RPI),
                "mulfq": (a0 * b0 * RPI, a1 * b0 * RPI), "fqmul": (a0 * b0 * RPI, None), "fqsqr": (a0 * a0 * RPI, None),
                "add": (a0 + b0, a1 + b1), "sub": (a0 - b0, a1 - b1), "rsub": (b0 - a0, b1 - a1), "dbl": (2 * a0, 2 * a1),
                "neg": (-a0, -a1), "negc1": (a0, -a1), "norm": (a0, a1), "redn": (a0, a1)}
        if sl == 0:
            want["mulxi"] = (9 * a0 - a1, a0 + 9 * a1)
        for name, (w0, w1) in want.items():
            m = _m3([a0, a1, b0, b1], rng, sl)
            S.run_block(B[name], m)
            assert (val(m, 0) - w0) % P == 0, name
            if w1 is not None:
                assert (val(m, 1) - w1) % P == 0, name
            if name in ("norm", "redn"):
                assert all(0 <= m.v[K3.NL * j + i] < (1 << K3.LB) for j in range(2) for i in range(K3.NL - 1))
            if name == "redn":
                assert all(-P // 64 < val(m, j) < P + P // 64 for j in range(2))
    xi = lambda x: ((9 * x[0] - x[1]) % P, (9 * x[1] + x[0]) % P)
    f2a = lambda x, y: ((x[0] + y[0]) % P, (x[1] + y[1]) % P)
    # sqr4 (fused Fq4 squaring of the cyclotomic squaring): normalised operands, also with negated limbs (conjugates)
    f2m = lambda x, y: ((x[0] * y[0] - x[1] * y[1]) % P, (x[0] * y[1] + x[1] * y[0]) % P)
    for t in range(24):
        a, b = (rnd(), rnd()), (rnd(), rnd())
        m = _m3([a[0], a[1], b[0], b[1]], rng, 0)
        if t % 3 == 1:                       # -a, -b as limb-wise negations
            for r in range(4 * K3.NL):
                m.v[r] = (-m.v[r]) & 0xFFFFFFFF
            a, b = ((-a[0]) % P, (-a[1]) % P), ((-b[0]) % P, (-b[1]) % P)
        for r in range(K3.HOME0, K3.HOME0 + 3 * K3.SLOT_DW):
            m.v[r] = rng.getrandbits(32)     # scratch blocks hold garbage
        S.run_block(B["sqr4"], m)
        b2 = f2m(b, b)
        xib2 = ((9 * b2[0] - b2[1]) % P, (9 * b2[1] + b2[0]) % P)
        a2 = f2m(a, a)
        ab = f2m(a, b)
        for j, w in enumerate(((a2[0] + xib2[0]) * RPI, (a2[1] + xib2[1]) * RPI, 2 * ab[0] * RPI, 2 * ab[1] * RPI)):
            assert (val(m, j) - w) % P == 0, ("sqr4", t, j)
        assert all(0 <= m.v[K3.NL * j + i] < (1 << K3.LB) for j in range(2) for i in range(K3.NL - 1))
        assert all(0 <= m.v[K3.NL * j + i] < (2 << K3.LB) for j in (2, 3) for i in range(K3.NL - 1))
    # sqr4c / sqr4cx: the same with the Granger-Scott recombination fused (zc, zd in home blocks 3, 4)
    for t in range(12):
        a, b, zc, zd = [(rnd(), rnd()) for _ in range(4)]
        for name in ("sqr4c", "sqr4cx"):
            m = _m3([a[0], a[1], b[0], b[1]], rng, 0)
            for r in range(K3.HOME0, K3.HOME0 + 3 * K3.SLOT_DW):
                m.v[r] = rng.getrandbits(32)
            for k, el in ((3, zc), (4, zd)):
                for h in range(2):
                    for i, w in enumerate(K3.to_limbs(el[h])):
                        m.v[K3.HOME0 + K3.SLOT_DW * k + K3.NL * h + i] = w
            S.run_block(B[name], m)
            b2 = f2m(b, b)
            a2 = f2m(a, a)
            r0 = f2a(a2, xi(b2))
            tt = f2m(a, b)
            if name == "sqr4cx":
                tt = xi(tt)
            want = [3 * r0[0] * RPI - 2 * zc[0], 3 * r0[1] * RPI - 2 * zc[1], 6 * tt[0] * RPI + 2 * zd[0], 6 * tt[1] * RPI + 2 * zd[1]]
            for j in range(4):
                assert (val(m, j) - want[j]) % P == 0, (name, t, j)
            assert all(0 <= m.v[K3.NL * j + i] < (1 << K3.LB) for j in range(4) for i in range(K3.NL - 1))
    # mul6 (fused Fq6 multiplication): operands in home blocks 0..5, limbs up to 2 units (sums of two normalised values)
    xi = lambda x: ((9 * x[0] - x[1]) % P, (9 * x[1] + x[0]) % P)
    f2a = lambda x, y: ((x[0] + y[0]) % P, (x[1] + y[1]) % P)
    for t in range(10):
        a = [(rnd(), rnd()) for _ in range(3)]
        b = [(rnd(), rnd()) for _ in range(3)]
        m = _m3([], rng, 0)
        for r in range(K3.HOME0 + 6 * K3.SLOT_DW, K3.HOME0 + 8 * K3.SLOT_DW):
            m.v[r] = rng.getrandbits(32)
        for r in range(0, 2 * K3.SLOT_DW):
            m.v[r] = rng.getrandbits(32)
        for k, el in enumerate(a + b):
            for h in range(2):
                if t % 2:                     # a sum of two normalised values, limb by limb
                    part = rng.randrange(P)
                    limbs = [x + y for x, y in zip(K3.to_limbs(part), K3.to_limbs((el[h] - part) % P))]
                else:
                    limbs = K3.to_limbs(el[h])
                for i, w in enumerate(limbs):
                    m.v[K3.HOME0 + K3.SLOT_DW * k + K3.NL * h + i] = w & 0xFFFFFFFF
        S.run_block(B["mul6"], m)
        v = [f2m(a[i], b[i]) for i in range(3)]
        cross = lambda i, j: f2m(f2a(a[i], a[j]), f2a(b[i], b[j]))
        sub = lambda x, y: ((x[0] - y[0]) % P, (x[1] - y[1]) % P)
        want = [f2a(v[0], xi(sub(sub(cross(1, 2), v[1]), v[2]))), f2a(sub(sub(cross(0, 1), v[0]), v[1]), xi(v[2])),
                f2a(sub(sub(cross(0, 2), v[0]), v[2]), v[1])]
        where = [K3.HOME0 + K3.SLOT_DW, K3.A0, K3.HOME0]            # c0 -> home 1, c1 -> A, c2 -> home 0
        for c in range(3):
            for h in range(2):
                regs = [m.v[where[c] + K3.NL * h + i] for i in range(K3.NL)]
                assert (_sval(regs) - want[c][h] * RPI) % P == 0, ("mul6", t, c, h)
                assert all(0 <= r < (1 << K3.LB) for r in regs[:-1])
    # extreme operands: every limb at the largest magnitude the routines accept (the simulator traps any signed 64-bit
    # overflow of a column accumulator and any int32 overflow shows up as a wrong residue)
    top = (1 << K3.LB) - 1
    for name, mag_, homes in (("mul6", 2, range(6)), ("sqr4c", 1, (3, 4)), ("sqr4cx", 1, (3, 4)), ("mul", 3, ()), ("mul3", 2.4, range(4))):       # mul3: the tracker bounds the SUM of the three operand-magnitude products by 18
        for pattern in (lambda i: 1, lambda i: -1, lambda i: 1 if i % 2 else -1, lambda i: 1 if (i // 2) % 2 else -1):
            m = _m3([], rng, 0)
            for r in range(0, K3.HOME0 + 8 * K3.SLOT_DW):
                m.v[r] = rng.getrandbits(32) if r >= 2 * K3.SLOT_DW else None
            blocks = [K3.A0, K3.B0] + [K3.HOME0 + K3.SLOT_DW * k for k in homes]
            for blk in blocks:
                for i in range(K3.SLOT_DW):
                    m.v[blk + i] = int(pattern(i) * mag_ * top) & 0xFFFFFFFF
            for r in range(K3.A0, K3.A0 + 2 * K3.SLOT_DW):
                if m.v[r] is None:
                    m.v[r] = 0
            S.run_block(B[name], m)
            assert m.max_acc < (1 << 63)
    # redn on large representatives (x + t p, |t| up to the certified cap) with unnormalised limbs
    for t in range(40):
        xs = [rnd() + rng.randrange(-60000, 60000) * P for _ in range(2)]
        m = _m3(xs, rng, [0, 1, 3][t % 3])
        before = [val(m, j) for j in range(2)]
        S.run_block(B["redn"], m)
        for j in range(2):
            assert (val(m, j) - xs[j]) % P == 0 and -P // 64 < val(m, j) < P + P // 64, (t, j, before[j] // P, val(m, j) // P)
            assert all(0 <= m.v[K3.NL * j + i] < (1 << K3.LB) for i in range(K3.NL - 1))
    # boundary conversions: ark 4 x u64 Montgomery (R = 2^256) <-> internal; cvtout is canonical
    for t in range(20):
        x = rnd()
        ext = (x << 256) % P
        m = _m3([], rng, 0)
        for i in range(8):
            m.v[i] = (ext >> (32 * i)) & 0xFFFFFFFF
        S.run_block(B["cvtin"], m)
        assert (val(m, 0) - x * K3.RP) % P == 0
        m2 = _m3([(x * K3.RP) % P], rng, 2)
        S.run_block(B["cvtout"], m2)
        assert sum(m2.v[i] << (32 * i) for i in range(8)) == ext


def _first_diff(a, b):
    for i, (x, y) in enumerate(zip(a, b)):
        if x != y:
            return f"call {i}: executed {a[max(0, i - 3):i + 3]} certified {b[max(0, i - 3):i + 3]}"
    return f"lengths {len(a)} vs {len(b)}"


def _run_kernel3(kb, g1=None, g2=None, fin=None, k=1, check_seq=True):
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
                      ("s14", 256 * 80), ("s[16:17]", STAT), ("s18", 0), ("s19", 1)):
        m.sset(name, val)
    m.v[255] = 0
    m.call_log = []
    S.run(lines, m)
    if check_seq:
        # the statically certified call sequence (value bounds, tools/kgen3_prog.py: certify_values) is the one executed
        rep = kb.certify_values(k_pairs=k)
        log = [re.sub(r"_\d+$", "", x) for x in m.call_log]
        assert log == rep["sequence"], _first_diff(log, rep["sequence"])
        assert rep["max_stored"] <= K3P.V_CAP
    out = []
    for c in range(12):
        v = 0
        for l in range(4):
            a = OUTB + (c * 4 + l) * 8
            v |= (m.gmem[a] | (m.gmem[a + 4] << 32)) << (64 * l)
        out.append(R.from_mont(v))
    return out, m


def test_v3_miller_kernel_exact(vec):
    g1, g2 = _inputs(vec, 3)
    out, m = _run_kernel3(K3P.KernelBuilder3(do_miller=True, do_fexp=False, track=True), g1, g2)
    assert out == HX(vec["miller"][3]) and STAT not in m.gmem
    assert m.max_acc < (1 << 62)


def test_v3_final_exp_kernel(vec):
    x = HX(vec["fq12_in"][2])
    fin = [w for c in x for w in R.limbs4(R.to_mont(c))]
    out, m = _run_kernel3(K3P.KernelBuilder3(do_miller=False, do_fexp=True), fin=fin)
    assert out == HX(vec["final_exp"][2])
    out, m = _run_kernel3(K3P.KernelBuilder3(do_miller=False, do_fexp=True), fin=[0] * 48)
    assert m.gmem.get(STAT) == 1            # zero input: the reference panics


def test_v3_pairing_kernel_generators(vec):
    """BASELINE.json configs[0]: e(G1gen, G2gen) through the fused default kernel."""
    g1, g2 = _inputs(vec, 0)
    out, m = _run_kernel3(K3P.KernelBuilder3(do_miller=True, do_fexp=True), g1, g2)
    assert out == HX(vec["pairing"][0])


def test_v3_multi_pairing_kernel(vec):
    """k = 2 shared-f kernel: exact multi_miller_loop_native value (tracked scale)."""
    g = vec["groups"][0]
    k, idx = g["k"], g["idx"]

    def soa(rows):
        n = len(rows)
        out = [0] * (len(rows[0]) * 4 * n)
        for i, el in enumerate(rows):
            for c, x in enumerate(el):
                for l, w in enumerate(R.limbs4(R.to_mont(x))):
                    out[(c * 4 + l) * n + i] = w
        return out

    g1, g2 = soa([HX(vec["g1"][i]) for i in idx]), soa([HX(vec["g2"][i]) for i in idx])
    out, m = _run_kernel3(K3P.KernelBuilder3(do_miller=True, do_fexp=False, track=True, multi=True), g1, g2, k=k)
    assert out == HX(g["miller"])


def test_v3_helper_kernel(vec):
    """k3_op: MyFq12 Mul, frobenius_map_native and pow_native on the v3 representation (general, non-unitary elements)."""
    xs = [HX(x) for x in vec["fq12_in"]]
    fq12_words = lambda x: [w for c in x for w in R.limbs4(R.to_mont(c))]
    kb = K3P.KernelBuilder3(helper=True)
    a, b = xs[1], xs[2]
    # Mul: a * b (golden fq12_mul[i] = fq12_in[i] * fq12_in[i + 1])
    out, m = _run_kernel3(kb, g1=fq12_words(b), fin=fq12_words(a), k=kb.OP_MUL, check_seq=False)
    assert out == HX(vec["fq12_mul"][1]) and STAT not in m.gmem
    # frobenius_map_native, powers 1, 2, 3, 6, 11
    for power in (1, 2, 3, 6, 11):
        out, m = _run_kernel3(kb, fin=fq12_words(a), k=kb.OP_FROB | power << 8, check_seq=False)
        assert out == HX(vec["frobenius"][str(power)][1]), f"frobenius power {power}"
    # pow_native(a, [BN_X]) with the reference's NAF (get_naf), -1 digits divide
    naf = R.get_naf([R.BN_X])
    while naf[-1] == 0:
        naf.pop()
    assert naf[-1] == 1
    packed = bytes((d & 0xFF) for d in naf) + b"\0" * 8
    words = [int.from_bytes(packed[8 * i: 8 * i + 8], "little") for i in range(len(packed) // 8)]
    out, m = _run_kernel3(kb, g2=words, fin=fq12_words(a), k=kb.OP_POW | 1 << 8 | len(naf) << 16, check_seq=False)
    assert out == HX(vec["pow_x"][1]) and STAT not in m.gmem
    # an exponent without -1 digits never divides: a = 0 is not an error (0^5 = 0)
    out, m = _run_kernel3(kb, g2=[0x0000000000010001], fin=[0] * 48, k=kb.OP_POW | 3 << 16, check_seq=False)
    assert out == [0] * 12 and STAT not in m.gmem
