"""Constants emitted by tools/gen_consts.py (compiled into the kernels) vs the oracle's values
computed by the reference's own formulas."""
import os
import re

import helpers as H
from helpers import R

HDR = os.path.join(H.ROOT, "plonky2-bn254-pairing_amd", "csrc", "bn254_consts_gen.h")


def _arrays(name):
    txt = open(HDR).read()
    m = re.search(name + r"[^=]*=\s*(\{.*?\});", txt, flags=re.S)
    assert m, name
    words = [int(w, 16) for w in re.findall(r"0x([0-9a-f]{8})u", m.group(1))]
    vals = []
    for i in range(0, len(words), 8):
        v = sum(w << (32 * j) for j, w in enumerate(words[i:i + 8]))
        vals.append(R.from_mont(v))
    return vals


def test_twist_and_frobenius_constants():
    c2, c3 = R._end_constants()
    assert tuple(_arrays("BN254_TWIST_C2")) == c2 and tuple(_arrays("BN254_TWIST_C3")) == c3
    fr = _arrays("BN254_FROB")
    for k in range(12):
        for i in range(6):
            g = R.fq2_pow(R.frob_coeffs(k), i)
            assert (fr[(k * 6 + i) * 2], fr[(k * 6 + i) * 2 + 1]) == g
    tb = _arrays("BN254_THREE_B")
    assert tuple(tb) == R.fq2_mul((3, 0), R.TWIST_B)
    assert tuple(_arrays("BN254_G2_GEN")) == (R.G2_GEN[0][0], R.G2_GEN[0][1], R.G2_GEN[1][0], R.G2_GEN[1][1])


def test_naf_tables():
    txt = open(HDR).read()
    m = re.search(r"BN254_X_NAF\[\d+\] = \{(.*?)\};", txt)
    naf = [int(x) for x in m.group(1).split(",")]
    want = R.get_naf([R.BN_X])
    while want[-1] == 0:
        want.pop()
    assert naf == want and naf[-1] == 1
    m = re.search(r"BN254_SIX_U_PLUS_2_NAF\[65\] = \{(.*?)\};", txt)
    assert [int(x) for x in m.group(1).split(",")] == R.SIX_U_PLUS_2_NAF
