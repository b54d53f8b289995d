"""CPU tests of the oracle (oracle/bn254_oracle.c, oracle/bn254_pyref.py): golden fixtures,
the algebraic identities the reference's own tests assert (T1/T3/T4),
bilinearity, and the published constants of SURVEY.md Appendix A."""
import numpy as np
import pytest

import helpers as H
from helpers import R

HX = lambda xs: [int(x, 16) for x in xs]


@pytest.fixture(scope="module")
def vec():
    return H.load_golden("bn254_vectors.json")


def _pts(vec):
    P = [tuple(HX(p)) for p in vec["g1"]]
    Q = [((int(q[0], 16), int(q[1], 16)), (int(q[2], 16), int(q[3], 16))) for q in vec["g2"]]
    return P, Q


def test_field_constants():
    # SURVEY.md section 8 "common data facts"
    assert R.limbs4(R.P) == [0x3c208c16d87cfd47, 0x97816a916871ca8d, 0xb85045b68181585d, 0x30644e72e131a029]
    assert R.limbs4(R.MONT_R % R.P) == [0xd35d438dc58f0d9d, 0x0a78eb28f5c70b3d, 0x666ea36f7879462c, 0x0e0a77c19a07df2f]
    assert (-pow(R.P, -1, 1 << 64)) % (1 << 64) == 0x87d20782e4866389
    assert R.P % 4 == 3 and R.P % 6 == 1                      # final_exp_native.rs:20-21
    assert sum(d << i for i, d in enumerate(R.SIX_U_PLUS_2_NAF)) == 6 * R.BN_X + 2
    assert R.g1_on_curve(R.G1_GEN) and R.g2_on_curve(R.G2_GEN)
    assert R.g2_mul(R.G2_GEN, R.R_ORDER) is None and R.g1_mul(R.G1_GEN, R.R_ORDER) is None


def test_appendix_a_known_answer(vec):
    """e(G1gen, G2gen): SURVEY.md Appendix A (derived known answer; BASELINE.json configs[0])."""
    e = HX(vec["pairing"][0])
    assert e[0] == 8493334370784016972005089913588211327688223499729897951716206968320726508021
    assert e[11] == 7484542354754424633621663080190936924481536615300815203692506276894207018007
    assert R.pairing_myfq12(R.G1_GEN, R.G2_GEN) == e
    assert R.fq12_pow(e, R.R_ORDER) == R.fq12_one() and e != R.fq12_one()
    c2, c3 = R._end_constants()
    assert c2 == (21575463638280843010398324269430826099269044274347216827212613867836435027261,
                  10307601595873709700152284273816112264069230130616436755625194854815875713954)
    assert c3 == (2821565182194536844548159561693502659359617185244120367078079554186484126554,
                  3505843767911556378687030309984248845540243509899259641013678093033130930403)
    assert R.frob_coeffs(2) == (21888242871839275220042445260109153167277707414472061641714758635765020556617, 0)


def test_c_oracle_matches_golden(vec):
    P, Q = _pts(vec)
    n = len(P)
    g1, g2 = H.g1_aos(P), H.g2_aos(Q)
    assert H.fq12_from_aos(H.oracle_miller(g1, g2, n), n) == [HX(m) for m in vec["miller"]]
    assert H.fq12_from_aos(H.oracle_pairing(g1, g2, n), n) == [HX(m) for m in vec["pairing"]]
    assert H.fq12_from_aos(H.oracle_pairing(g1, g2, n, threads=4), n) == [HX(m) for m in vec["pairing"]]
    for g in vec["groups"]:
        idx = g["idx"]
        a, b = H.g1_aos([P[i] for i in idx]), H.g2_aos([Q[i] for i in idx])
        assert H.fq12_from_aos(H.oracle_multi_miller(a, b, 1, g["k"]), 1)[0] == HX(g["miller"])
        assert H.fq12_from_aos(H.oracle_multi_pairing(a, b, 1, g["k"]), 1)[0] == HX(g["pairing"])
    xs = [HX(x) for x in vec["fq12_in"]]
    a = H.fq12_aos(xs)
    rc, out = H.oracle_final_exp(a, len(xs))
    assert rc == 0 and H.fq12_from_aos(out, len(xs)) == [HX(x) for x in vec["final_exp"]]
    rc, out = H.oracle_pow_native(a, [R.BN_X], len(xs))
    assert rc == 0 and H.fq12_from_aos(out, len(xs)) == [HX(x) for x in vec["pow_x"]]
    for k, want in vec["frobenius"].items():
        assert H.fq12_from_aos(H.oracle_frobenius(a, int(k), len(xs)), len(xs)) == [HX(x) for x in want]
    b = H.fq12_aos(xs[1:] + xs[:1])
    assert H.fq12_from_aos(H.oracle_fq12_mul(a, b, len(xs)), len(xs)) == [HX(x) for x in vec["fq12_mul"]]


def test_T1_multi_equals_product():
    """test_multi_miller_loop_native (miller_loop_native.rs:336-348): exact MyFq12 equality."""
    P, Q = H.subgroup_points(2, seed=11)
    g1, g2 = H.g1_aos(P), H.g2_aos(Q)
    r = H.oracle_miller(g1, g2, 2)
    prod = H.oracle_fq12_mul(r[:48], r[48:], 1)
    multi = H.oracle_multi_miller(g1, g2, 1, 2)
    assert np.array_equal(multi, prod)


def test_T3_to_one(vec):
    """test_to_one (final_exp_native.rs:240-264) -- and the product really is one (e(5P,6Q) e(30P,-Q) = 1)."""
    t3 = vec["t3"]
    P3 = [tuple(HX(p)) for p in t3["g1"]]
    Q3 = [((int(q[0], 16), int(q[1], 16)), (int(q[2], 16), int(q[3], 16))) for q in t3["g2"]]
    g1, g2 = H.g1_aos(P3), H.g2_aos(Q3)
    m = H.oracle_multi_miller(g1, g2, 1, 2)
    ms = H.oracle_miller(g1, g2, 2)
    assert np.array_equal(m, H.oracle_fq12_mul(ms[:48], ms[48:], 1))                 # :258
    rc0, rs = H.oracle_final_exp(ms, 2)
    rc1, rm = H.oracle_final_exp(m, 1)
    assert rc0 == 0 and rc1 == 0
    assert np.array_equal(H.oracle_fq12_mul(rs[:48], rs[48:], 1), rm)                 # :259-263
    assert H.fq12_from_aos(rm, 1)[0] == R.fq12_one()


def test_T4_pow_and_exact_exponent():
    """test_pow (final_exp_native.rs:266-286): pow_native == pow; final_exp_native(x) == x^((p^12-1)/r)."""
    xs = H.rand_fq12(2, seed=4)
    a = H.fq12_aos(xs)
    rc, got = H.oracle_pow_native(a, [R.BN_X], 2)
    assert rc == 0 and np.array_equal(got, H.oracle_fq12_pow(a, [R.BN_X], 2))
    e = (R.P ** 12 - 1) // R.R_ORDER
    limbs = [(e >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range((e.bit_length() + 63) // 64)]
    rc, fe = H.oracle_final_exp(a, 2)
    assert rc == 0 and np.array_equal(fe, H.oracle_fq12_pow(a, limbs, 2))


def test_bilinearity_and_order():
    P, Q = H.subgroup_points(1, seed=5)
    a, b = 0x1234567, 0xABCDEF01
    e = H.fq12_from_aos(H.oracle_pairing(H.g1_aos(P), H.g2_aos(Q), 1), 1)[0]
    e_ab = H.fq12_from_aos(H.oracle_pairing(H.g1_aos([R.g1_mul(P[0], a)]), H.g2_aos([R.g2_mul(Q[0], b)]), 1), 1)[0]
    assert e_ab == R.fq12_pow(e, a * b % R.R_ORDER)
    assert R.fq12_pow(e, R.R_ORDER) == R.fq12_one()


def test_get_naf():
    n, naf = H.oracle_get_naf([R.BN_X])
    assert n == 64 and naf == R.get_naf([R.BN_X]) and sum(d << i for i, d in enumerate(naf)) == R.BN_X
    n, naf = H.oracle_get_naf([0xFFFFFFFFFFFFFFFF, 0x1234])
    assert naf == R.get_naf([0xFFFFFFFFFFFFFFFF, 0x1234])
    assert sum(d << i for i, d in enumerate(naf)) == 0xFFFFFFFFFFFFFFFF + (0x1234 << 64)
    # carry out of the top limb: the reference panics (final_exp_native.rs:123)
    n, _ = H.oracle_get_naf([0xFFFFFFFFFFFFFFFF])
    assert n == -1
    with pytest.raises(AssertionError):
        R.get_naf([0xFFFFFFFFFFFFFFFF])


def test_final_exp_zero_divisor_is_an_error():
    rc, _ = H.oracle_final_exp(H.fq12_aos([[0] * 12]), 1)
    assert rc != 0   # ark `/` panics on a zero divisor (final_exp_native.rs:200)


def test_ark_layout_roundtrip():
    x = list(range(100, 112))
    assert R.ark_to_myfq12(R.myfq12_to_ark(x)) == x
    # w^2 = v: MyFq12 coefficient 2 is ark c0.c1
    assert R.myfq12_to_ark(x)[2] == x[2] and R.myfq12_to_ark(x)[6] == x[1]
