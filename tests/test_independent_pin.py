"""A structurally independent pin of the pairing VALUE (tests/golden/independent_pairing.py: flat Fp12 = Fp[w]/(w^12 - 18 w^6 + 82),
untwisted points, generic chord / tangent lines, binary loop over 6u+2, Frobenius steps by honest p-th powers, one plain power
by (p^12 - 1)/r).  The optimal ate pairing is unique as a function, so it must equal `pairing()` of the restatement -- and with
it every golden `pairing` vector and SURVEY.md Appendix A -- EXACTLY.  This pins what bilinearity and the exponent identity
(T4) cannot: the twist map, the MyFq12 <-> w-basis coefficient order and the twisted-Frobenius constants."""
import os
import sys

import helpers as H
from helpers import R

sys.path.insert(0, os.path.join(H.ROOT, "tests", "golden"))
import independent_pairing as IP  # noqa: E402

HX = lambda xs: [int(x, 16) for x in xs]  # noqa: E731


def test_independent_optimal_ate_equals_pairing():
    vec = H.load_golden("bn254_vectors.json")
    for i in (0, 4, 9):                                   # e(G1gen, G2gen) and two random subgroup pairs
        g1 = HX(vec["g1"][i])
        q = HX(vec["g2"][i])
        got = IP.to_myfq12(IP.optimal_ate((g1[0], g1[1]), ((q[0], q[1]), (q[2], q[3]))))
        assert got == HX(vec["pairing"][i]), f"golden pairing vector {i}"
        assert got == R.pairing_myfq12((g1[0], g1[1]), ((q[0], q[1]), (q[2], q[3])))
    # c[0] of pairing(G1gen, G2gen)
    assert HX(vec["pairing"][0])[0] == 8493334370784016972005089913588211327688223499729897951716206968320726508021


def test_independent_statement_is_a_pairing():
    """sanity of the independent statement itself: bilinear and of order r"""
    a, b = 5, 7
    e = IP.optimal_ate(R.G1_GEN, R.G2_GEN)
    e_ab = IP.optimal_ate(R.g1_mul(R.G1_GEN, a), R.g2_mul(R.G2_GEN, b))
    assert e_ab == IP.ppow(e, a * b) and e != IP.ONE and IP.ppow(e, IP.R_ORDER) == IP.ONE
