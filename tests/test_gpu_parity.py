"""GPU parity tests (bit-exact, u64 Montgomery limbs) through the C ABI against the CPU oracle and
the committed golden fixtures.  Integer work: the bar is identical limbs, no tolerance."""
import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu

HX = lambda xs: [int(x, 16) for x in xs]


def _golden_points(vec):
    P = [tuple(HX(p)) for p in vec["g1"]]
    Q = [((int(q[0], 16), int(q[1], 16)), (int(q[2], 16), int(q[3], 16))) for q in vec["g2"]]
    return P, Q


@pytest.fixture(scope="module")
def vec():
    return H.load_golden("bn254_vectors.json")


def test_golden_pairing_and_miller(vec):
    """configs[0] (e(G1gen, G2gen)) and 12 random pairs: miller_loop_native, final_exp_native, pairing."""
    pk = H.pkg()
    P, Q = _golden_points(vec)
    n = len(P)
    g1, g2 = H.to_soa(H.g1_aos(P), 8), H.to_soa(H.g2_aos(Q), 16)
    got_m = H.fq12_from_aos(H.to_aos(pk.miller_loop_batch(g1, g2, n), 48), n)
    got_p = H.fq12_from_aos(H.to_aos(pk.pairing_batch(g1, g2, n), 48), n)
    for i in range(n):
        assert got_m[i] == HX(vec["miller"][i]), f"miller_loop_native mismatch at {i}"
        assert got_p[i] == HX(vec["pairing"][i]), f"pairing mismatch at {i}"
    # final_exp_native on the Miller values
    f = H.to_soa(H.fq12_aos([HX(m) for m in vec["miller"]]), 48)
    got_f = H.fq12_from_aos(H.to_aos(pk.final_exp_batch(f, n), 48), n)
    assert got_f == [HX(p) for p in vec["pairing"]]


def test_golden_final_exp_arbitrary_fq12(vec):
    """T4 shape (final_exp_native.rs:274-285): arbitrary, non-unitary Fq12 inputs (and the identity)."""
    pk = H.pkg()
    xs = [HX(x) for x in vec["fq12_in"]]
    n = len(xs)
    a = H.to_soa(H.fq12_aos(xs), 48)
    assert H.fq12_from_aos(H.to_aos(pk.final_exp_batch(a, n), 48), n) == [HX(x) for x in vec["final_exp"]]
    assert H.fq12_from_aos(H.to_aos(pk.pow_batch(a, [pk.BN_X], n), 48), n) == [HX(x) for x in vec["pow_x"]]
    for k, want in vec["frobenius"].items():
        got = H.fq12_from_aos(H.to_aos(pk.frobenius_map_batch(a, int(k), n), 48), n)
        assert got == [HX(x) for x in want], f"frobenius_map_native power {k}"
    b = H.to_soa(H.fq12_aos(xs[1:] + xs[:1]), 48)
    assert H.fq12_from_aos(H.to_aos(pk.fq12_mul_batch(a, b, n), 48), n) == [HX(x) for x in vec["fq12_mul"]]


def test_golden_multi_miller(vec):
    pk = H.pkg()
    P, Q = _golden_points(vec)
    for g in vec["groups"]:
        k, idx = g["k"], g["idx"]
        g1 = H.to_soa(H.g1_aos([P[i] for i in idx]), 8)
        g2 = H.to_soa(H.g2_aos([Q[i] for i in idx]), 16)
        m = H.fq12_from_aos(pk.multi_pairing_batch(g1, g2, 1, k, do_final_exp=False), 1)[0]
        assert m == HX(g["miller"]), f"multi_miller_loop_native k={k}"
        e = H.fq12_from_aos(pk.multi_pairing_batch(g1, g2, 1, k, do_final_exp=True), 1)[0]
        assert e == HX(g["pairing"])
    t3 = vec["t3"]
    P3 = [tuple(HX(p)) for p in t3["g1"]]
    Q3 = [((int(q[0], 16), int(q[1], 16)), (int(q[2], 16), int(q[3], 16))) for q in t3["g2"]]
    m = H.fq12_from_aos(pk.multi_pairing_batch(H.to_soa(H.g1_aos(P3), 8), H.to_soa(H.g2_aos(Q3), 16), 1, 2, do_final_exp=False), 1)[0]
    assert m == HX(t3["miller"])


def test_oracle_parity_ragged_batch():
    """HIP vs C oracle on seeded subgroup points, ragged size (not a multiple of the 256-lane work item)."""
    pk = H.pkg()
    n = 300
    base_P, base_Q = H.subgroup_points(8)
    P = [base_P[i % 8] for i in range(n)]
    Q = [base_Q[(i * 3 + i // 8) % 8] for i in range(n)]
    g1a, g2a = H.g1_aos(P), H.g2_aos(Q)
    want = H.oracle_pairing(g1a, g2a, n, threads=8)
    got = H.to_aos(pk.pairing_batch(H.to_soa(g1a, 8), H.to_soa(g2a, 16), n), 48)
    assert np.array_equal(got, want)
    want_m = H.oracle_miller(g1a[: 8 * 40], g2a[: 16 * 40], 40)
    got_m = H.to_aos(pk.miller_loop_batch(H.to_soa(g1a[: 8 * 40], 8), H.to_soa(g2a[: 16 * 40], 16), 40), 48)
    assert np.array_equal(got_m, want_m)
