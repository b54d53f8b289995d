"""The lane-cooperative pairing (tools/cvm.py, tools/cvm_kernel.py) on the CPU: the Fq2 graph, its lowering to Fq operations and
the scheduled sixteen-lane program against the Python restatement of the reference (oracle/bn254_pyref.py), and the interpreter
kernel's generated assembly on the instruction simulator (sixteen lanes, run round by round: tools/cvm_sim.py)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import bn254_pyref as R  # noqa: E402
import cvm  # noqa: E402
import cvm_kernel as CK  # noqa: E402
import cvm_sim as CS  # noqa: E402
import sched_model as SM  # noqa: E402

P_PT = R.g1_mul(R.G1_GEN, 12345)
Q_PT = R.g2_mul(R.G2_GEN, 67890)
FLAT = [P_PT[0], P_PT[1], Q_PT[0][0], Q_PT[0][1], Q_PT[1][0], Q_PT[1][1]]


def _graph(build, ins):
    g = cvm.Graph()
    g.const((0, 0))
    g.const((1, 0))
    iv = [g.inp(f"i{i}") for i in range(len(ins))]
    g.outputs = build(g, iv)
    return g.evaluate(ins)


def test_graph_builders_against_the_schedule_model():
    f = R.fq12_to_fp2s(R.miller_loop_native(Q_PT, P_PT))
    b = [R.fq2_mul(x, (3, 5)) for x in f[::-1]]
    assert _graph(lambda g, iv: g.fq12_mul(iv[:6], iv[6:]), f + b) == SM.fq12_mul(f, b)
    assert _graph(lambda g, iv: g.fq12_sqr(iv), f) == SM.fq12_sqr(f)
    assert _graph(lambda g, iv: g.fq12_inv(iv), f) == SM.fq12_inv(f)
    for k in (1, 2, 3):
        assert _graph(lambda g, iv: g.frobenius(iv, k), f) == SM.frobenius(f, k)
    m = R.fq12_to_fp2s(R.easy_part(R.fq12_from_fp2s(f)))
    assert _graph(lambda g, iv: g.cyc_sqr(iv), m) == SM.cyclotomic_sqr(m)
    assert _graph(lambda g, iv: g.pow_x(iv), m) == SM.pow_x_cyclotomic(m)
    L = (f[0], f[3], f[4])
    assert _graph(lambda g, iv: g.mul_by_034(iv[:6], iv[6:]), f + list(L)) == SM.mul_by_034(f, L)
    assert _graph(lambda g, iv: g.mul_by_235(iv[:6], iv[6:]), f + list(L)) == SM.mul_by_235(f, L)
    # final_exp_native itself (the reference's function, not the model)
    assert _graph(lambda g, iv: g.final_exp(iv), f) == R.fq12_to_fp2s(R.final_exp_native(R.fq12_from_fp2s(f)))


def test_pairing_program_equals_the_reference_pairing():
    """graph -> Fq operations -> sixteen-lane schedule with slot allocation: each level evaluates to pairing(P, Q)"""
    want = R.fq12_to_fp2s(R.pairing_myfq12(P_PT, Q_PT))
    g = cvm.build_pairing(pow_window="x19")          # (as shipped: tools/gen_kernels.py CVM_PROGRAMS)
    assert g.evaluate([(P_PT[0], 0), (P_PT[1], 0), Q_PT[0], Q_PT[1]]) == want
    low = cvm.Lowered(g)
    flat_want = [c for x in want for c in x]
    assert low.evaluate(FLAT) == flat_want
    assert max(v.bound for v in low.fv) <= cvm.Lowered.V_MAX
    pr = cvm.Program(low, nr=CK.NR)
    assert pr.run(FLAT) == flat_want
    st = pr.stats()
    assert st["rounds"] < 1400 and st["slots"] <= 277          # 277 slots of 48 bytes: three waves per CU; of 36 bytes: four
    # no slot is written in a round that still reads it (what lets the kernel do without barriers, and the simulator run lane by lane)
    for rnd, (kind, take) in enumerate(pr.rounds):
        written = {w.slot for v in take for w in (v, v.twin) if w is not None}
        read = {s.slot for v in take for s in v.srcs()}
        assert not (written & read), rnd


def test_miller_half_program_of_the_two_launch_pairing():
    """MILLER_U (round 5): the Miller loop of pairing() without the line scale, on the minimal-weight 65-digit form of 6 x + 2, scheduled for sixteen lanes in
    at most 142 slots (eight waves per CU in the 36-byte layout).  Its value differs from miller_loop_native's by a factor from Fq6 only --
    final_exp_native of it IS pairing(P, Q), limb for limb -- and the scheduled program obeys the no-write-while-read rule of every program."""
    g = cvm.build_miller_u()
    low = cvm.Lowered(g)
    pr = cvm.Program(low, nr=CK.NR)
    flat = pr.run(FLAT)
    f = [(flat[2 * i], flat[2 * i + 1]) for i in range(6)]
    assert R.fq12_to_fp2s(R.final_exp_native(R.fq12_from_fp2s(f))) == R.fq12_to_fp2s(R.pairing_myfq12(P_PT, Q_PT))
    # against miller_loop_native's own value: a factor from the subfield Fq6 (the lines' Fq2 scales, and -- the program walks the minimal-weight
    # 65-digit form of 6 x + 2, the reference its own digit table -- vertical lines): fixed by the p^6-Frobenius, which is the conjugation
    ratio = R.fq12_div(R.fq12_from_fp2s(f), R.miller_loop_native(Q_PT, P_PT))
    assert R.fq12_conjugate(ratio) == ratio and ratio != R.fq12_one()
    # ... and with the reference's table the factor is ONE Fq2 element for all six coefficients
    cvm.SHORT_CHAIN = False
    try:
        flat_r = cvm.Program(cvm.Lowered(cvm.build_miller_u()), nr=CK.NR).run(FLAT)
    finally:
        cvm.SHORT_CHAIN = True
    fr = [(flat_r[2 * i], flat_r[2 * i + 1]) for i in range(6)]
    exact = R.fq12_to_fp2s(R.miller_loop_native(Q_PT, P_PT))
    r2 = R.fq2_mul(fr[0], R.fq2_inv(exact[0]))
    assert all(R.fq2_mul(exact[i], r2) == fr[i] for i in range(6))
    st = pr.stats()
    assert st["slots"] <= 142 and st["rounds"] < 640, st
    for rnd, (kind, take) in enumerate(pr.rounds):
        written = {w.slot for v in take for w in (v, v.twin) if w is not None}
        read = {s.slot for v in take for s in v.srcs()}
        assert not (written & read), rnd


def test_multi_miller_half_programs():
    """MMILLER2_U / 3_U / 4_U: multi_miller_loop_native without the line scale -- final_exp_native of the scheduled program's value is the
    product of the pairings (the first launch of the mid-size form of the k-pair products)."""
    base = [(R.g1_mul(R.G1_GEN, 11 + 7 * j), R.g2_mul(R.G2_GEN, 5 + 3 * j)) for j in range(4)]
    for k in (2, 3, 4):
        pairs = base[:k]
        pr = cvm.Program(cvm.Lowered(cvm.build_multi_miller_u(k)), nr=CK.NR)
        flat_in = []
        for Pp, Qq in pairs:
            flat_in += [Pp[0], Pp[1], Qq[0][0], Qq[0][1], Qq[1][0], Qq[1][1]]
        w = pr.run(flat_in)
        f = [(w[2 * i], w[2 * i + 1]) for i in range(6)]
        want = R.final_exp_native(R.multi_miller_loop_native(pairs))
        assert R.final_exp_native(R.fq12_from_fp2s(f)) == want, k


def test_final_exp_pieces_compose_to_final_exp_native():
    """EASY, POWX (three times), YCH1, YCH2 (round 5: final_exp_native as six launches for mid-size batches): the scheduled sixteen-lane
    programs, run on integers one after the other with each piece's outputs as the next one's inputs, give final_exp_native(f) for a
    non-unitary f (a Miller value times a scale) -- and every piece fits seven or eight waves per CU (at most 146 slots)."""
    f = [R.fq2_mul(c, (7, 3)) for c in R.fq12_to_fp2s(R.miller_loop_native(Q_PT, P_PT))]
    want = R.fq12_to_fp2s(R.final_exp_native(R.fq12_from_fp2s(f)))
    flat = lambda x: [c for v in x for c in v]
    unflat = lambda w: [(w[2 * i], w[2 * i + 1]) for i in range(6)]
    progs = {}
    for piece in ("easy", "powx", "ych1", "ych2"):
        pr = cvm.Program(cvm.Lowered(cvm.build_fexp_piece(piece)), nr=CK.NR)
        assert pr.stats()["slots"] <= 146, (piece, pr.stats())
        for rnd, (kind, take) in enumerate(pr.rounds):
            written = {w.slot for v in take for w in (v, v.twin) if w is not None}
            assert not (written & {s.slot for v in take for s in v.srcs()}), (piece, rnd)
        progs[piece] = pr
    m = unflat(progs["easy"].run(flat(f)))
    assert m == R.fq12_to_fp2s(R.easy_part(R.fq12_from_fp2s(f)))
    mx = unflat(progs["powx"].run(flat(m)))
    assert R.fq12_from_fp2s(mx) == R.pow_native(R.fq12_from_fp2s(m), [R.BN_X])
    mx2 = unflat(progs["powx"].run(flat(mx)))
    mx3 = unflat(progs["powx"].run(flat(mx2)))
    t1 = unflat(progs["ych1"].run(flat(mx) + flat(mx2) + flat(mx3)))          # inputs in the order g1, g2, f_in
    assert unflat(progs["ych2"].run(flat(m) + flat(t1))) == want


def _mini():
    g = cvm.Graph()
    g.const((0, 0))
    g.const((1, 0))
    (px, py), (qx, qy) = g.g1_point(), g.g2_point()
    a = g.mul((qx, qy))
    b = g.mul((qx, qy), (a, qx), (qy, qy), add=a)
    c = g.lin((a, cvm.mxi()), (b, cvm.mk(-27)), (qx, cvm.CONJ), (qy, (3, -5, 7, 11)))
    d = g.fq2_inv(c)
    e_ = g.mul((d, px), (b, py))
    f = g.mul((e_, g.const((12345, 67890))))
    g.outputs = [a, b, c, d, e_, f]
    return g


def _sim(pr, want_flat, layout="aos48"):
    blob = CK.make_blob(pr.encode())
    lines = CK.VMKernel(layout, nr=pr.nr).build()
    g1 = [w for c in P_PT for w in R.limbs4(R.to_mont(c))]
    g2 = [w for c in (Q_PT[0][0], Q_PT[0][1], Q_PT[1][0], Q_PT[1][1]) for w in R.limbs4(R.to_mont(c))]
    gmem, ms, rounds = CS.simulate(lines, blob, g1, g2, tids=range(pr.nr))
    out = []
    for c in range(12):
        v = 0
        for l in range(4):
            a = CS.OUTB + (c * 4 + l) * 8
            v |= (gmem[a] | (gmem[a + 4] << 32)) << (64 * l)
        out.append(R.from_mont(v))
    assert out == [want_flat[2 * k] for k in range(6)] + [want_flat[2 * k + 1] for k in range(6)]
    assert CS.STAT not in gmem                       # no zero divisor
    assert max(m.max_acc for m in ms) < (1 << 63)
    return ms, rounds


@pytest.mark.parametrize("layout,nr", [("aos48", 16), ("split36", 16), ("aos48", 32), ("aos48", 64)])
def test_interpreter_kernel_every_round_kind_on_the_simulator(layout, nr):
    low = cvm.Lowered(_mini())
    pr = cvm.Program(low, nr=nr)
    kinds = {cvm.KIND_NAME[k] for k, _ in pr.rounds}
    assert {"m2", "m6", "l4", "l8", "inv"} <= kinds
    _sim(pr, low.evaluate(FLAT), layout)


def test_zero_divisor_raises_the_status_flag_on_the_simulator():
    g = cvm.Graph()
    g.const((0, 0))
    g.const((1, 0))
    (px, py), (qx, qy) = g.g1_point(), g.g2_point()
    z = g.lin((qx, cvm.ID), (qx, cvm.NEG))          # 0
    g.outputs = [g.fq2_inv(z)] + [qx] * 5
    pr = cvm.Program(cvm.Lowered(g), nr=CK.NR)
    blob = CK.make_blob(pr.encode())
    g1 = [w for c in P_PT for w in R.limbs4(R.to_mont(c))]
    g2 = [w for c in (Q_PT[0][0], Q_PT[0][1], Q_PT[1][0], Q_PT[1][1]) for w in R.limbs4(R.to_mont(c))]
    gmem, ms, rounds = CS.simulate(CK.VMKernel().build(), blob, g1, g2)
    assert gmem.get(CS.STAT) == 1


def test_input_descriptors_pairs_items_and_fq12_on_the_simulator():
    """k = 2 pairs per item, n = 3 items, the lanes of item 1; and an Fq12 input (f_in) with null G1 / G2 pointers"""
    pts = [(R.g1_mul(R.G1_GEN, 3 + i), R.g2_mul(R.G2_GEN, 5 + i)) for i in range(6)]
    g = cvm._graph()
    prs = [(g.g1_point(j), g.g2_point(j)) for j in range(2)]
    (p0, q0), (p1, q1) = prs
    g.outputs = [g.mul((q0[0], q1[1])), g.mul((q0[1], p1[0]), (q1[0], p0[1])), g.lin((q1[0], cvm.mxi()), (q0[1], cvm.CONJ)), q1[1], g.mul((p0[0], p1[1])),
                 g.mul((q1[0], q1[0]), add=q0[0])]
    low = cvm.Lowered(g)
    pr = cvm.Program(low, nr=CK.NR)
    item = 1
    flat = []
    for j in range(2):
        P, Q = pts[2 * item + j]
        flat += [P[0], P[1], Q[0][0], Q[0][1], Q[1][0], Q[1][1]]
    want = low.evaluate(flat)
    W = lambda x: R.limbs4(R.to_mont(x))
    g1 = CS.soa([[W(P[0]), W(P[1])] for P, _ in pts], 2)
    g2 = CS.soa([[W(Q[0][0]), W(Q[0][1]), W(Q[1][0]), W(Q[1][1])] for _, Q in pts], 4)
    gmem, ms, _ = CS.simulate(CK.VMKernel("split36").build(), CK.make_blob(pr.encode()), g1, g2, n=3, k=2, tids=range(16, 32))
    got = [R.from_mont(v) for v in CS.read_fq12(gmem, n=3, item=item)]
    assert got == [want[2 * i] for i in range(6)] + [want[2 * i + 1] for i in range(6)]
    assert all((CS.OUTB + 8 * other) not in gmem for other in (0, 2))          # only item 1's lanes ran: nothing else is written
    # Fq12 input
    g = cvm._graph()
    f = g.fq12_input()
    g.outputs = [g.mul((f[i], f[(i + 1) % 6])) for i in range(6)]
    low = cvm.Lowered(g)
    pr = cvm.Program(low, nr=CK.NR)
    x = [(7 + 3 * i) * 0x123456789ABCDEF123456789 % R.P for i in range(12)]              # MyFq12 coefficient order
    flat = [x[i + 6 * h] for i in range(6) for h in range(2)]
    want = low.evaluate(flat)
    gmem, ms, _ = CS.simulate(CK.VMKernel().build(), CK.make_blob(pr.encode()), fin_words=CS.soa([[W(c) for c in x]], 12))
    got = [R.from_mont(v) for v in CS.read_fq12(gmem)]
    assert got == [want[2 * i] for i in range(6)] + [want[2 * i + 1] for i in range(6)]


def test_other_programs_equal_the_reference_functions():
    """the exact Miller value (one pair and k = 2, 3 pairs with the shared f), the product of pairings, final_exp_native: graph, Fq
    lowering and schedule on big integers"""
    P = [R.g1_mul(R.G1_GEN, 11 + 7 * j) for j in range(3)]
    Q = [R.g2_mul(R.G2_GEN, 5 + 3 * j) for j in range(3)]

    def flat(k):
        return [c for j in range(k) for c in (P[j][0], P[j][1], Q[j][0][0], Q[j][0][1], Q[j][1][0], Q[j][1][1])]

    def check(g, ins, want12):
        low = cvm.Lowered(g)
        pr = cvm.Program(low, nr=CK.NR)
        want = [c for x in R.fq12_to_fp2s(want12) for c in x]
        assert low.evaluate(ins) == want and pr.run(ins) == want
        assert max(v.bound for v in low.fv) <= cvm.Lowered.V_MAX
        return pr
    check(cvm.build_miller(), flat(1), R.miller_loop_native(Q[0], P[0]))
    check(cvm.build_miller(run_ahead=2), flat(1), R.miller_loop_native(Q[0], P[0]))
    for k in (2, 3):
        m = R.multi_miller_loop_native([(P[j], Q[j]) for j in range(k)])
        check(cvm.build_multi(k, final_exp=False), flat(k), m)
        check(cvm.build_multi(k, final_exp=True, pow_window="x19"), flat(k), R.final_exp_native(m))
    f = R.miller_loop_native(Q[1], P[2])
    check(cvm.build_final_exp(pow_window="x19"), [c for x in R.fq12_to_fp2s(f) for c in x], R.final_exp_native(f))
    check(cvm.build_final_exp(), [c for x in R.fq12_to_fp2s(f) for c in x], R.final_exp_native(f))          # (the three-bit window: not shipped any more, still a valid schedule)


def test_wide_programs_equal_the_reference_functions():
    """the thirty-two-lane programs (monomial cyclotomic squarings, four-bit windows): pairing and the four-pair product"""
    want = [c for x in R.fq12_to_fp2s(R.pairing_myfq12(P_PT, Q_PT)) for c in x]
    low = cvm.Lowered(cvm.build_pairing(wide=True, pow_window="fixed"))
    pr = cvm.Program(low, nr=32)
    assert low.evaluate(FLAT) == want and pr.run(FLAT) == want
    assert len(pr.rounds) < 1000 and max(v.bound for v in low.fv) <= cvm.Lowered.V_MAX
    for rnd, (kind, take) in enumerate(pr.rounds):
        written = {w.slot for v in take for w in (v, v.twin) if w is not None}
        assert not (written & {s.slot for v in take for s in v.srcs()}), rnd
    P = [R.g1_mul(R.G1_GEN, 11 + 7 * j) for j in range(4)]
    Q = [R.g2_mul(R.G2_GEN, 5 + 3 * j) for j in range(4)]
    flat = [c for j in range(4) for c in (P[j][0], P[j][1], Q[j][0][0], Q[j][0][1], Q[j][1][0], Q[j][1][1])]
    m = R.final_exp_native(R.multi_miller_loop_native([(P[j], Q[j]) for j in range(4)]))
    low = cvm.Lowered(cvm.build_multi(4, True, pow_window="fixed", wide=True))
    pr = cvm.Program(low, nr=32)
    assert pr.run(flat) == [c for x in R.fq12_to_fp2s(m) for c in x]
    # sixty-four lanes: the lines of a step multiplied with each other first, one dense product per step for f (three and four pairs,
    # the product of pairings and the exact Miller value)
    for k in (3, 4):
        mm = R.multi_miller_loop_native([(P[j], Q[j]) for j in range(k)])
        for fe in (True, False):
            low = cvm.Lowered(cvm.build_multi(k, fe, pow_window="fixed", wide=True, line_tree=True))
            pr = cvm.Program(low, nr=64)
            want_k = [c for x in R.fq12_to_fp2s(R.final_exp_native(mm) if fe else mm) for c in x]
            assert low.evaluate(flat[:6 * k]) == want_k and pr.run(flat[:6 * k]) == want_k, (k, fe)
            assert max(v.bound for v in low.fv) <= cvm.Lowered.V_MAX


def test_sixty_four_lane_programs_equal_the_reference_functions():
    """the sixty-four-lane programs as shipped (tools/gen_kernels.py CVM_FULL_PROGRAMS: single products on f's chain, four-term
    combinations): graph, Fq lowering and schedule on big integers against the reference's functions; no slot is written in the round
    that reads it"""
    import gen_kernels as G
    P = [R.g1_mul(R.G1_GEN, 11 + 7 * j) for j in range(4)]
    Q = [R.g2_mul(R.G2_GEN, 5 + 3 * j) for j in range(4)]
    flat = [c for j in range(4) for c in (P[j][0], P[j][1], Q[j][0][0], Q[j][0][1], Q[j][1][0], Q[j][1][1])]
    mm = {k: R.multi_miller_loop_native([(P[j], Q[j]) for j in range(k)]) for k in (1, 2, 3, 4)}
    want = {"PAIRING_X": (1, R.final_exp_native(mm[1])), "MILLER_X": (1, mm[1])}
    for k in (2, 3, 4):
        want[f"MULTI{k}_X"] = (k, R.final_exp_native(mm[k]))
        want[f"MMILLER{k}_X"] = (k, mm[k])
    assert mm[1] == R.miller_loop_native(Q[0], P[0])
    for name, _, build in G.CVM_FULL_PROGRAMS:
        k, w12 = want[name]
        w = [c for x in R.fq12_to_fp2s(w12) for c in x]
        low = cvm.Lowered(build(cvm))
        pr = cvm.Program(low, nr=64, inv_weight=G.CVM_INV_WEIGHT.get(name), m_weight=G.CVM_M_WEIGHT.get(name, 1.0))
        assert low.evaluate(flat[:6 * k]) == w and pr.run(flat[:6 * k]) == w, name
        assert max(v.bound for v in low.fv) <= cvm.Lowered.V_MAX
        for rnd, (kind, take) in enumerate(pr.rounds):
            assert len(take) <= 64
            written = {x.slot for v in take for x in (v, v.twin) if x is not None}
            assert not (written & {s.slot for v in take for s in v.srcs()}), (name, rnd)
    name, _, build = G.CVM_EXTRA_PROGRAMS[-1]
    assert name == "FEXP_X"
    low = cvm.Lowered(build(cvm))
    pr = cvm.Program(low, nr=64)
    w = [c for x in R.fq12_to_fp2s(R.final_exp_native(mm[2])) for c in x]
    fin = [c for x in R.fq12_to_fp2s(mm[2]) for c in x]
    assert low.evaluate(fin) == w and pr.run(fin) == w
    # the pairing program is what the review's one-item target is about: single products -> no six-product round on f's chain
    pr = cvm.Program(cvm.Lowered(G.CVM_FULL_PROGRAMS[0][2](cvm)), nr=64)
    st = pr.stats()
    assert st["by_kind"].get("m6", 0) <= 4 and st["by_kind"].get("l8", 0) <= 2 and st["instr_est"] < 285_000


def test_whole_pairing_on_the_simulator():
    """the shipped program (csrc/cvm_asm_gen.h) on sixteen simulated lanes: pairing(P, Q), bit for bit"""
    low = cvm.Lowered(cvm.build_pairing(pow_window="x19"))
    pr = cvm.Program(low, nr=CK.NR)
    want = [c for x in R.fq12_to_fp2s(R.pairing_myfq12(P_PT, Q_PT)) for c in x]
    ms, rounds = _sim(pr, want)
    assert rounds == len(pr.rounds) + 2
    assert ms[0].count < 600_000                     # instructions per lane (the throughput kernel: 3.59 M)
