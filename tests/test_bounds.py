"""Static bound certification of the shipped kernels (balanced radix-2^29 limbs, tools/kgen4*.py).

LIMB bounds are closed per routine: every multiplication asserts at generation time that its signed 64-bit column sums
cannot overflow (n_terms * |a| * |b| + 9 below 118 units of 2^56) and every store enforces its limb limit.  VALUE bounds
(which representative of a residue a slot holds -- with R'/p = 169.6 a Montgomery reduction contracts only small values)
cross routine boundaries through one contract: whatever a routine finds in a slot is below V_STORE p, and it leaves at
most V_STORE p in every slot that outlives it; its own temporaries stay below V_CAP p, which keeps the top limb (weight
2^232) inside one unit.  KernelBuilder.certify_values() walks the kernel's real, data-independent call sequence (NAF digits
of 6u+2, the x-power digits, the y-chain of hard_part_BN_native) and checks every routine on it against the contract;
tests/test_kgen4.py cross-checks that the walked sequence is exactly what the instruction simulator executes."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kgen4 as K4  # noqa: E402
import kgen4_prog as K4P  # noqa: E402
from asmcore import Emitter, P_INT  # noqa: E402
from gen_kernels import KERNELS  # noqa: E402


@pytest.mark.parametrize("name,kw", KERNELS, ids=[n for n, _ in KERNELS])
def test_value_contract_of_shipped_kernels(name, kw):
    kb = K4P.KernelBuilder(**kw)
    kb.build()
    if kw.get("helper"):
        rep = kb.certify_helper()
    else:
        for k_pairs in ((1, 2, 3, 8) if kw.get("multi") else (1,)):
            rep = kb.certify_values(k_pairs)
        if kw.get("fixed"):                   # ... and the groups without a pair of their own: squarings and table lines only (every variant's entry bounds)
            rep0 = kb.certify_values(1, own_pair=False)
            assert rep0["max_stored"] <= K4P.V_CAP and {"L2_fxred", "L2_sqr"} <= set(rep0["sequence"]) and "L2_dblmul" not in rep0["sequence"]
    assert rep["max_stored"] <= K4P.V_CAP
    # top limb (weight 2^232) of the largest temporary: within one unit (2^28), inside every limb interval
    assert rep["max_stored"] * P_INT / 2 ** 232 <= 2 ** 28
    for routine, exits in kb.l2_exit.items():
        assert all(v <= K4P.V_STORE for v in exits.values()), routine
    # cvtout (one more Montgomery multiplication) canonicalises any representative below 84 p: (84 p * p / R') / p + 1/2 < 1
    assert K4P.V_STORE * 1.0 / K4P.K_RP + 0.5 < 1.0


def test_constants_of_the_tracker():
    assert abs(K4P.K_RP - (1 << 261) / P_INT) < 1e-6 and 169 < K4P.K_RP < 170
    assert K4P.V_CAP <= K4P.K_TOP                          # a stored value's top limb stays within one unit
    # a column of the three-term multiplication: 54 products of normalised limbs + 9 reduction products, far below 2^63
    assert (54 + 9) * (1 << 56) < (1 << 63) and 2 * K4.NL * 3 <= K4P.COL_BUDGET
    assert (K4P.COL_BUDGET + K4.NL) * (1 << 56) + (1 << 35) < (1 << 63)       # products + reduction products + carry-in


def test_tracker_has_teeth():
    """The generator refuses what it cannot prove: a three-term multiplication of unnormalised operands, a multiplication
    whose column sums could overflow, a store beyond the value cap."""
    def prog():
        p = K4P.Prog(Emitter(), {n: f"L1_{n}" for n in K4.L1V4_NAMES})
        p.set_temps([K4P.HOME(i) for i in range(9)])
        return p
    p = prog()
    p.reserve_blocks()
    for k in range(4):
        p.ldH(k, K4P.AGPR(k))                                # unknown stored values: up to two units per limb
    with pytest.raises(AssertionError):
        p.A(K4P.AGPR(4)).mul3(K4P.AGPR(5))                   # 18 * (4 + 4 + 4) > 118
    p = prog()
    p.A(K4P.AGPR(0))
    p.rA = (-6.0, 6.0)
    p.mul(K4P.AGPR(1))                                       # the tracker inserts a normalisation by itself ...
    assert p.stats.get("norm", 0) + p.stats.get("redn", 0) == 1
    p = prog()
    p.slot_r[("agpr", 1)] = (-8.0, 8.0)
    with pytest.raises(AssertionError):
        p.A(K4P.AGPR(0)).mul(K4P.AGPR(1))                    # ... but cannot fix an operand that sits in a slot
    p = prog()
    p.A(K4P.AGPR(0))
    p.vA = 400.0
    with pytest.raises(AssertionError):
        p.store(K4.A0, K4P.AGPR(2)) or p._need(p.vA <= K4P.V_CAP, "value bound at store")


def test_trampolines_for_out_of_range_transfers():
    """asmcore.place_with_islands: a call whose target is farther than the branch reach goes through trampolines placed
    between the blocks (possibly more than one hop), everything else is untouched, and the shipped pairing kernel -- 330 KB of
    code against a reach of +-128 KB -- ends up with every transfer in range."""
    from asmcore import branch_table, max_branch_distance, place_with_islands
    filler = lambda n: ["v_mad_i64_i32 v[0:1], vcc, v2, v3, v[0:1]"] * n          # 8 bytes each
    blocks = [["L_a_%=:", "s_call_b64 s[54:55], L_far_%=", "s_branch L_near_%="] + filler(50), ["L_near_%=:"] + filler(900),
              ["L_mid_%=:"] + filler(900), ["L_far_%=:", "s_setpc_b64 s[54:55]"]]
    out, hops = place_with_islands(blocks, 8192, lambda n: n + "_%=")
    assert hops >= 1 and max_branch_distance(out) < 8192
    lab, ins = branch_table(out)
    # the near branch still goes straight to its label; the far call reaches L_far through L_hop labels only
    assert any(op == "s_branch" and t == "L_near_%=" for _, _, op, t in ins)
    tgt = [t for _, _, op, t in ins if op == "s_call_b64"][0]
    seen = 0
    while tgt.startswith("L_hop"):
        i = out.index(tgt + ":")
        tgt = out[i + 1].split()[-1]
        seen += 1
    assert tgt == "L_far_%=" and seen == hops
    kb = K4P.KernelBuilder(do_miller=True, do_fexp=True)
    assert max_branch_distance(kb.build()) < 131072 - 512 and kb.n_trampolines < 32
