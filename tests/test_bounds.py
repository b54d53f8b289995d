"""Static bound certification of the v3 kernels (signed radix-2^27 limbs, tools/kgen3*.py).

Limb bounds are closed per routine: every store enforces |limb| <= 3.05 units of 2^27 and every multiplication
asserts its signed 64-bit column sums when the code is generated.  VALUE bounds (which representative of a residue
a slot holds -- it lives in the top limb) cross routine boundaries: a cyclotomic squaring 3t - 2z roughly doubles
the representative and only a Montgomery multiplication contracts it again.  KernelBuilder3.certify_values()
replays the kernel's real, data-independent call sequence (NAF digits of 6u+2 and of BN_X, the y-chain of
hard_part_BN_native) through the generator's own transfer functions, re-generating every routine under the true
entry bounds: all generation-time checks must still pass and the emitted code must be identical to what ships.
tests/test_kgen.py cross-checks that the replayed sequence is exactly what the instruction simulator executes."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kgen3_prog as K3P  # noqa: E402
from gen_kernels import V3_KERNELS  # noqa: E402


@pytest.mark.parametrize("name,kw", V3_KERNELS, ids=[n for n, _ in V3_KERNELS])
def test_value_bounds_of_shipped_kernels(name, kw):
    kb = K3P.KernelBuilder3(**kw)
    kb.build()
    for k_pairs in ((1, 2, 3, 8) if kw.get("multi") else (1,)):
        rep = kb.certify_values(k_pairs)
        assert rep["max_stored"] <= K3P.V_CAP
        # top limb (weight 2^243) of the largest stored value: below one unit (2^27), inside every limb interval
        assert rep["max_stored"] * K3P.P_INT / 2 ** 243 < 2 ** 27
        if kw["do_miller"]:
            assert rep["miller_f_out"] < 2.0            # the Miller loop hands over a freshly reduced f
        if kw["do_fexp"]:
            assert rep["fexp_f_out"] < 4096.0          # cvtout (one more Montgomery multiplication) canonicalises any representative


def test_reduction_schedule_of_the_x_power_loop():
    """No more than RED_RUN cyclotomic squarings in a row without a multiplication or a representative reduction."""
    naf = K3P.X_DIGITS[:-1]                                # the shipped x-power schedule (digits in {0, +-1, +-5, +-9, +-13})
    red = K3P.x_red_mask(naf)
    run = longest = 0
    for j in range(len(naf) - 1, -1, -1):
        run += 1
        longest = max(longest, run)
        if naf[j] != 0 or red >> j & 1:
            run = 0
        assert not (naf[j] != 0 and red >> j & 1)
    assert longest == K3P.RED_RUN
    assert 0 < bin(red).count("1") <= 12                 # ~500 instructions each (L1 redn): < 0.5 % of the final exponentiation


def test_certification_has_teeth():
    """Unreduced squarings in a row make the worst-case representative diverge: the replay must reject them."""
    kb = K3P.KernelBuilder3(do_miller=False, do_fexp=True)
    kb.build()
    kb.certify_values()
    state = {}
    with pytest.raises(AssertionError):
        for _ in range(9):
            ex, _ = kb._eval("L2_cyc", state)
            state.update(ex)
