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
    # zero input: the reference panics (division by zero) -> status word written
    out, m = _run_kernel(KP.KernelBuilder(do_miller=False, do_fexp=True), fin=[0] * 48)
    assert m.gmem.get(STAT) == 1


def test_pairing_kernel_generators(vec):
    """BASELINE.json configs[0]: e(G1gen, G2gen) through the fused kernel."""
    g1, g2 = _inputs(vec, 0)
    out, m = _run_kernel(KP.KernelBuilder(do_miller=True, do_fexp=True), g1, g2)
    assert out == HX(vec["pairing"][0])
