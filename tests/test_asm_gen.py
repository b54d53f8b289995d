"""The generated gfx950 Montgomery routines (tools/gen_fq_asm.py), run through the single-lane
instruction simulator (tools/asm_sim.py): results vs big-int arithmetic, even alignment of 64-bit
VGPR operands, and the VALU-writes-SGPR -> VALU-reads hazard distance the generator must keep."""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_fq_asm as G  # noqa: E402
from asm_sim import Sim  # noqa: E402

P = G.P_INT
RINV = pow(1 << 256, -1, P)


def limbs(a):
    return [(a >> (32 * i)) & 0xFFFFFFFF for i in range(8)]


def val(v, regs):
    return sum(v[r] << (32 * i) for i, r in enumerate(regs))


def operands(n_out):
    ops = {}
    S = G.sregs(n_out)
    for i, p in enumerate(S["p"]):
        ops[p] = G.P_LIMBS[i]
    ops[S["n0"]] = G.N0
    for c in G.carry_ops(n_out):
        ops[c] = 0
    return ops


def run(routine, n_out, inputs):
    lines = routine().finalize()
    sim = Sim(operands(n_out))
    for j, x in enumerate(inputs):
        for i, w in enumerate(limbs(x)):
            sim.v[8 * j + i] = w
    sim.run(lines)
    return sim.v


def samples(rng, k):
    edge = [0, 1, P - 1, P - 2, (1 << 253), 0xFFFFFFFF]
    return [rng.choice(edge) if rng.random() < 0.3 else rng.randrange(P) for _ in range(k)]


def test_fq_mul():
    rng = random.Random(1)
    for _ in range(60):
        a, b = samples(rng, 2)
        v = run(G.routine_fq_mul, 2, (a, b))
        assert val(v, range(8)) == a * b * RINV % P


def test_fq2_mul_sqr_mulfq():
    rng = random.Random(2)
    for _ in range(40):
        a0, a1, b0, b1 = samples(rng, 4)
        v = run(G.routine_fq2_mul, 4, (a0, a1, b0, b1))
        assert val(v, range(8)) == (a0 * b0 - a1 * b1) * RINV % P
        assert val(v, range(8, 16)) == (a0 * b1 + a1 * b0) * RINV % P
        v = run(G.routine_fq2_sqr, 2, (a0, a1))
        assert val(v, range(8)) == (a0 * a0 - a1 * a1) * RINV % P
        assert val(v, range(8, 16)) == 2 * a0 * a1 * RINV % P
        v = run(G.routine_fq2_mul_fq, 3, (a0, a1, b0))
        assert val(v, range(8)) == a0 * b0 * RINV % P and val(v, range(8, 16)) == a1 * b0 * RINV % P


def test_leaf_routines_only_clobber_caller_saved_vgprs():
    for routine in (G.routine_fq_mul, G.routine_fq2_mul, G.routine_fq2_sqr, G.routine_fq2_mul_fq):
        r = routine()
        r.finalize()
        for reg in r.used:
            assert not (reg >= 40 and ((reg - 40) // 8) % 2 == 0), f"v{reg} is callee-saved in the AMDGPU calling convention"
