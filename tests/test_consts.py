"""Constants compiled into the product vs the oracle's values computed by the reference's own formulas:
  * host header (tools/gen_consts.py -> csrc/bn254_consts_gen.h): frob_coeffs table, SIX_U_PLUS_2_NAF, BN_X, MyFq12::one;
  * kernel constants (tools/kgen4_prog.py emits them as literal moves): twist / Frobenius coefficients, 3b', the x-power digits;
  * the fixed-base table of the input generator (tools/gen_tables.py -> csrc/gen_table_gen.h)."""
import os
import re
import sys

import helpers as H
from helpers import R

sys.path.insert(0, os.path.join(H.ROOT, "tools"))
import gen_consts  # noqa: E402
import gen_tables  # noqa: E402
import kgen4 as K4  # noqa: E402
import kgen4_prog as K4P  # noqa: E402

CSRC = os.path.join(H.ROOT, "plonky2-bn254-pairing_amd", "csrc")


def test_host_header():
    txt = open(os.path.join(CSRC, "bn254_consts_gen.h")).read()
    assert txt == gen_consts.render(), "bn254_consts_gen.h is stale: run python tools/gen_consts.py"
    rows = re.search(r"BN254_FROB_COEFFS_HOST\[12\]\[8\] = \{(.*?)\};", txt, flags=re.S).group(1)
    words = [int(w, 16) for w in re.findall(r"0x([0-9a-f]{16})ull", rows)]
    assert len(words) == 96
    for k in range(12):
        c = [R.from_mont(sum(words[8 * k + 4 * h + l] << (64 * l) for l in range(4))) for h in range(2)]
        assert tuple(c) == tuple(R.frob_coeffs(k))
    naf = re.search(r"BN254_SIX_U_PLUS_2_NAF\[65\] = \{(.*?)\};", txt).group(1)
    assert [int(x) for x in naf.split(",")] == R.SIX_U_PLUS_2_NAF
    assert int(re.search(r"BN254_BN_X (\d+)ull", txt).group(1)) == R.BN_X
    one = [int(w, 16) for w in re.findall(r"0x([0-9a-f]{16})ull", re.search(r"BN254_FQ_ONE_LIMBS \{(.*?)\}", txt).group(1))]
    assert R.from_mont(sum(w << (64 * l) for l, w in enumerate(one))) == 1


def test_kernel_constants():
    c2, c3 = R._end_constants()
    assert K4P.twist_consts() == (c2, c3)
    for k in range(12):
        for i in range(6):
            assert K4P.frob_const(k, i) == R.fq2_pow(R.frob_coeffs(k), i)
    assert (K4P.THREE_B.c0, K4P.THREE_B.c1) == R.fq2_mul((3, 0), R.TWIST_B)
    assert sum(d << i for i, d in enumerate(K4P.X_DIGITS)) == R.BN_X
    assert K4P.SIX_U_PLUS_2_NAF == R.SIX_U_PLUS_2_NAF
    # the chain of the Miller loops that end in the final exponentiation: the same number, 65 digits like the reference's table, and no
    # signed binary form of 65 digits has fewer non-zero ones (dynamic programme over the digits)
    import functools
    import asmcore
    short, n = asmcore.SIX_U_PLUS_2_SHORT, 6 * R.BN_X + 2
    assert len(short) == 65 and set(short) <= {-1, 0, 1} and sum(d << i for i, d in enumerate(short)) == n

    @functools.lru_cache(None)
    def least(v, digits):
        if digits == 0:
            return 0 if v == 0 else 1 << 20
        return min(least((v - d) // 2, digits - 1) + (d != 0) for d in (-1, 0, 1) if (v - d) % 2 == 0)
    assert least(n, 65) == sum(1 for d in short if d) == 22 and sum(1 for d in R.SIX_U_PLUS_2_NAF if d) == 26
    assert least(n, 64) >= 1 << 20 and least(n, 66) == 22          # 64 digits cannot hold it; a 66th digit buys nothing
    # limb form: balanced digits, Montgomery R' = 2^261
    assert K4.from_limbs(K4.P_L) == R.P and all(-K4.HALF <= l < K4.HALF for l in K4.P_L[:-1])
    assert (K4.N0P * R.P + 1) % (1 << K4.LB) == 0
    x = 0x1234567890ABCDEF1234567890ABCDEF1234567890ABCDEF
    assert K4.from_limbs(K4.bal_limbs(K4.mont4(x))) * pow(K4.RP, -1, R.P) % R.P == x


def test_generator_table():
    txt = open(os.path.join(CSRC, "gen_table_gen.h")).read()
    assert txt == gen_tables.render(), "gen_table_gen.h is stale: run python tools/gen_tables.py"
    w = gen_tables.all_words()
    rpi = pow(K4.RP, -1, R.P)

    def entry(curve, i, j):
        base = ((curve * 32 + i) * 15 + (j - 1)) * 36
        return [K4.from_limbs(w[base + 9 * c: base + 9 * c + 9]) * rpi % R.P for c in range(4)]
    for i, j in ((0, 1), (0, 15), (5, 7), (31, 15), (17, 2)):
        P_ = R.g1_mul(R.G1_GEN, j * 16 ** i)
        assert entry(0, i, j) == [P_[0], 0, P_[1], 0]
        Q_ = R.g2_mul(R.G2_GEN, j * 16 ** i)
        assert entry(1, i, j) == [Q_[0][0], Q_[0][1], Q_[1][0], Q_[1][1]]
