#!/usr/bin/env python3
"""cvm.py -- the lane-cooperative ("latency") path: ONE item (a pairing, a Miller loop, a final exponentiation, a product of up to four
pairings) on sixteen, thirty-two or sixty-four lanes instead of one.

The throughput kernels (tools/kgen4_prog.py) put one pairing on one lane: 3.55 M dependent instructions, 6.4 ms however few pairings
there are.  A single call of one of the reference's scalar functions wants the opposite trade: this file spreads ONE item over a group
of lanes and runs it as a table-driven sequence of ROUNDS.  In a round every lane of the group does one operation of the same family
on Fq values that live in an LDS slot array shared by the group:

    mul   dst <- up to six products x y (+ an addend)                    one Montgomery column pass (L1v4.fips)
    lin   dst <- up to eight small-integer multiples of sources          one reducing 64-bit chain (L1v4.lincomb)
    inv   dst <- 1 / src                                                 Bernstein-Yang divsteps (L1v4.fq_inv_safegcd)

Which slots and coefficients: a per-round, per-lane table row (32 bytes) in global memory.  The kernel is a small interpreter
(tools/cvm_kernel.py); the PROGRAMS -- the reference's pairing (src/pairing.rs:20-22), miller_loop_native / multi_miller_loop_native
(miller_loop_native.rs:320-326) with the exact value, final_exp_native (final_exp_native.rs:209-213), in the schedule of
tests/sched_model.py -- are data, produced here:

    Graph      SSA builder over Fq2 values: the field tower, G2 steps, sparse products, cyclotomic squarings, Frobenius, inversion, the
               Miller loop over k pairs with the shared f, the final exponentiation; `wide` / `line_tree` select the formulations that
               trade operations for depth when thirty-two / sixty-four lanes are there to take them
    Lowered    the same as Fq operations (one lane each): real values never multiply a zero component, negations are twins written by
               the producing lane, value bounds are checked
    Program    list scheduling into rounds (critical path first), LDS slots by liveness; `run` executes the scheduled program on big
               integers; `encode` packs the table the kernel reads

Nothing here is on the throughput path; tools/gen_kernels.py builds the programs and writes csrc/cvm_asm_gen.h.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from asmcore import P_INT, BN_X, SIX_U_PLUS_2_NAF, SIX_U_PLUS_2_SHORT  # noqa: E402

P = P_INT
XI = (9, 1)
ARR_G1, ARR_G2, ARR_F, ARR_NONE = 0, 1, 2, 3
BANK_AWARE = bool(int(os.environ.get("CVM_BANKS", "1")))       # LDS slots by liveness and bank class (Program._allocate_banked)
BANK_SLACK = int(os.environ.get("CVM_BANK_SLACK", "0"))        # slots the class-aware assignment may use beyond the liveness-only count
INPUT_SLOTS_EXPIRE = bool(int(os.environ.get("CVM_INPUT_EXPIRE", "1")))      # an input's slot is reused once its last reader has run (YCH1 takes 36 input values: 146 -> 141 slots)
N_TRASH = int(os.environ.get("CVM_TRASH", "1"))      # trash slots per item (Program.encode)
SHORT_CHAIN = bool(int(os.environ.get("CVM_SHORT_CHAIN", "1")))      # Miller loops that end in the final exponentiation walk the minimal-weight 65-digit form of 6 x + 2 (tools/asmcore.py)
GH_SHORT = bool(int(os.environ.get("CVM_GH_SHORT", "0")))      # add_step: G - H = 3 G - E - F beside H instead of behind it
X19_DIGITS = (-19, 0, 1, 0, 0, 0, 0, 0, 0, -19, 0, 0, 19, 0, 0, 0, 0, 0, 0, -19, 0, 0, 0, 0, -1, 0, 19, 0, 0, 0, 0, 0, 0, 0, -19, 0, 0, 0, 0, 0, 19, 0, 0,
              0, 0, 0, 0, 19, 0, 0, 0, 0, 0, 19, 0, -19, 0, 0, 19)      # BN_X over {0, +-1, +-19}, least significant first (pow_x "x19")
assert sum(d << i for i, d in enumerate(X19_DIGITS)) == BN_X
MAX_LIN_SRC = 6             # sources of an Fq2-level combination (the Fq operation it lowers to takes eight terms)


# ------------------------------------------------------------------ Fq2 on integers (the emulator's arithmetic)
def f2mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def f2add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def f2mat(m, s):
    a, b, c, d = m
    return ((a * s[0] + b * s[1]) % P, (c * s[0] + d * s[1]) % P)


def f2pow(a, e):
    r = (1, 0)
    while e:
        if e & 1:
            r = f2mul(r, a)
        a = f2mul(a, a)
        e >>= 1
    return r


ID, NEG, CONJ, NCONJ = (1, 0, 0, 1), (-1, 0, 0, -1), (1, 0, 0, -1), (-1, 0, 0, 1)
XI2 = (80, -18, 18, 80)          # matrix of multiplication by xi^2 = 80 + 18 u


def mxi(k=1):
    """matrix of multiplication by k (9 + u)"""
    return (9 * k, -k, k, 9 * k)


def mk(k):
    return (k, 0, 0, k)


class V:
    __slots__ = ("id", "kind", "args", "bound", "users", "name", "real", "src", "early", "after", "pt")

    def __init__(self, id_, kind, args, bound):
        self.id, self.kind, self.args, self.bound = id_, kind, args, bound
        self.users, self.name, self.real, self.src, self.early, self.after, self.pt = [], None, False, None, False, None, False

    def srcs(self):
        if self.kind in ("m1", "m3"):
            t, add = self.args
            return [v for xy in t for v in xy] + ([add] if add is not None else [])
        if self.kind == "lin":
            return [s for s, _ in self.args]
        if self.kind == "inv":
            return [self.args]
        return []


class Graph:
    """SSA builder over Fq2 values (lowered to Fq operations, where the value bounds are tracked, by `Lowered`)."""

    def __init__(self, run_ahead=None, pow_window=None, wide=False, flat_sqr=None, line_tree=False, full=False):
        # sixty-four lanes: f's chain as SINGLE products (two-product rounds instead of four- and six-product ones) whose sums are taken by the
        # combination that follows them -- a doubling iteration of one pair is [21 products of f's coefficients | f^2 | 18 products with the
        # line | f] = 385 + 225 + 385 + 225 instructions instead of [15 sums of two | f^2 | 6 sums of three] = 565 + 225 + 750; the dense
        # product 31 operations of at most two products
        self.full = full
        self.line_tree = line_tree        # several pairs: the lines of a step are multiplied with each other first (a tree, off f's chain), f takes ONE dense product
        self.wide = wide                  # the program is scheduled for thirty-two lanes: formulations that trade operations for depth
        self.flat_sqr = wide if flat_sqr is None else flat_sqr     # f^2 of the Miller loop as a schoolbook square (one pair: the lanes are there)
        self.pow_window = pow_window or self.POW_X_WINDOW          # signed window of the hard part's x-powers (pow_x)
        self.vals = []
        self.consts = {}
        self.inputs = []
        self.outputs = []
        self._in_point_step = False
        self._xi_of = {}                  # full: value id -> xi times the value, where a sparse multiplication produced it beside the value
        self.run_ahead = run_ahead        # iterations the Miller loop's point chain may run ahead of f's (None: as far as lanes are free)

    def _new(self, kind, args, bound):
        v = V(len(self.vals), kind, args, bound)
        v.pt = self._in_point_step              # an operation of a point step (Program: pt_weight)
        self.vals.append(v)
        for s in v.srcs():
            s.users.append(v)
        return v

    def inp(self, name, real=False, src=None):
        """src = (array, Fq number of c0, Fq number of c1 | None, pair): where the kernel finds the value -- array 0 = g1 (Fq 0..1 per
        point), 1 = g2 (0..3), 2 = f_in (0..11, MyFq12 order); pair = index inside the item's group of k pairs"""
        v = self._new("in", None, 1.01)
        v.name = name
        v.real = real
        v.src = src
        self.inputs.append(v)
        return v

    def g1_point(self, j=0):
        return self.inp(f"p{j}x", real=True, src=(ARR_G1, 0, None, j)), self.inp(f"p{j}y", real=True, src=(ARR_G1, 1, None, j))

    def g2_point(self, j=0):
        return self.inp(f"q{j}x", src=(ARR_G2, 0, 1, j)), self.inp(f"q{j}y", src=(ARR_G2, 2, 3, j))

    def fq12_input(self, arr=ARR_F):
        """an Fq12 (MyFq12 order) read from array `arr`: f_in, or -- the pieces of the final exponentiation that take two or three Fq12
        values -- the g1 / g2 arrays, which are then Fq12 batches too (the kernel reads any array the same way)"""
        return [self.inp(f"f{arr}_{i}", src=(arr, i, i + 6, 0)) for i in range(6)]

    def const(self, c):
        c = (c[0] % P, c[1] % P)
        if c not in self.consts:
            v = self._new("const", c, 0.51)
            self.consts[c] = v
        return self.consts[c]

    def mul(self, *terms, add=None, after=None):
        """after: a value this operation must not be scheduled before (no data flows; Graph.run_ahead: the Miller loop's point chain
        can be kept within a few iterations of f's chain, so that its lines do not pile up in LDS)"""
        assert 1 <= len(terms) <= 3
        v = self._new("m1" if len(terms) == 1 else "m3", (list(terms), add), 0)
        v.after = after
        return v

    def lin(self, *pairs):
        assert 1 <= len(pairs) <= MAX_LIN_SRC
        for s, m in pairs:
            assert all(-128 <= c <= 127 for c in m)
        return self._new("lin", list(pairs), 0)

    def inv(self, x):
        return self._new("inv", x, 4.2)

    # -------------------------------------------------------------- Fq2 helpers
    def xi(self, x, k=1):
        return self.lin((x, mxi(k)))

    def neg(self, x):
        return self.lin((x, NEG))

    def conj(self, x):
        return self.lin((x, CONJ))

    def fq2_inv(self, x):
        """1/x = conj(x) / (c0^2 + c1^2): the norm as the product x conj(x) (its c1 is 0), one Fq inversion"""
        xc = self.conj(x)
        n = self.mul((x, xc))
        return self.mul((xc, self.inv(n)))

    # -------------------------------------------------------------- Fq6 / Fq12 (w-basis: six Fq2 coefficients, w^6 = xi)
    def fq6_mul(self, a, b, ax=None):
        """(a0 + a1 v + a2 v^2)(b0 + b1 v + b2 v^2), v^3 = xi: three sums of three products (ax = (xi a1, xi a2))"""
        a1x, a2x = ax if ax is not None else (self.xi(a[1]), self.xi(a[2]))
        return (self.mul((a[0], b[0]), (a1x, b[2]), (a2x, b[1])),
                self.mul((a[0], b[1]), (a[1], b[0]), (a2x, b[2])),
                self.mul((a[0], b[2]), (a[1], b[1]), (a[2], b[0])))

    def fq12_mul(self, a, b, bx=None, after=None):
        """dense product: c_k = sum_i a_i B(k - i), B(j) = b_j (j >= 0) or xi b_(j+6); two chained three-term sums per coefficient.
        after: a value the product's first sums must not be scheduled before (no data flows: keeps a product that is only needed
        later from being computed early and waiting in LDS)"""
        if self.full:
            # sixty-four lanes: c_k = its six products in FOUR sources (two sums of two, two single products; the wrapped ones against
            # xi b_j, which the caller holds for a table entry or a combination off the chain makes): 24 operations of at most two
            # products and a four-term combination -- a four-product round and a cheap combination instead of a six-product round and
            # an eight-term one
            if bx is None:
                bx = [None] + [self.xi(b[j]) for j in range(1, 6)]
            out = []
            for k in range(6):
                t = [(a[i], b[k - i] if i <= k else bx[k - i + 6]) for i in range(6)]
                out.append(self.lin((self.mul(t[0], t[1], after=after), ID), (self.mul(t[2], t[3], after=after), ID),
                                    (self.mul(t[4], after=after), ID), (self.mul(t[5], after=after), ID)))
            return out
        if self.wide:
            # thirty-two lanes: the wrapped and the unwrapped part of every coefficient as sums of at most three products of their own --
            # sixteen sums, ONE round of thirty-two operations -- and c_k = lo_k + xi hi_k in the recombination: a product round and a
            # (five times cheaper) combination round instead of two product rounds, and no xi-multiples of b to keep
            out = []
            for k in range(6):
                lo = [(a[i], b[k - i]) for i in range(k + 1)]
                hi = [(a[i], b[k - i + 6]) for i in range(k + 1, 6)]
                srcs = [(self.mul(*lo[j:j + 3]), ID) for j in range(0, len(lo), 3)] + [(self.mul(*hi[j:j + 3]), mxi()) for j in range(0, len(hi), 3)]
                out.append(self.lin(*srcs))
            return out
        if bx is None:
            bx = [None] + [self.xi(b[j]) for j in range(1, 6)]

        def B(k, i):
            return b[k - i] if i <= k else bx[k - i + 6]
        out = []
        for k in range(6):
            t = self.mul(*[(a[i], B(k, i)) for i in range(3)], after=after)
            out.append(self.mul(*[(a[i], B(k, i)) for i in range(3, 6)], add=t))
        return out

    def fq12_mul_gen(self, a, b):
        """product of two Fq12 elements whose zero coefficients are None (lines, products of lines): the wide formulation of fq12_mul
        over the non-zero terms only"""
        out = []
        for k in range(6):
            lo = [(a[i], b[k - i]) for i in range(k + 1) if a[i] is not None and b[k - i] is not None]
            hi = [(a[i], b[k - i + 6]) for i in range(k + 1, 6) if a[i] is not None and b[k - i + 6] is not None]
            srcs = [(self.mul(*lo[j:j + 3]), ID) for j in range(0, len(lo), 3)] + [(self.mul(*hi[j:j + 3]), mxi()) for j in range(0, len(hi), 3)]
            out.append(self.lin(*srcs) if srcs else None)
        return out

    def fq12_mul_pre(self, b):
        if self.wide and not self.full:
            return None
        return [None] + [self.xi(b[j]) for j in range(1, 6)]

    def fq12_sqr(self, f):
        """complex squaring over Fq6 (tests/sched_model.py fq12_sqr): t = A0 A1, u = (A0 + A1)(A0 + v A1);
        f^2 = (u - t - v t) + 2 t w: six three-term sums between two layers of linear combinations"""
        if self.full:
            # sixty-four lanes: the 21 distinct products a_i a_j one by one (42 operations, a two-product round), the sums in the recombination
            a = f
            X2 = mk(2)
            if all(a[j].id in self._xi_of for j in (3, 4, 5)):
                # ... whose wrapped ones take xi a_3, xi a_4, xi a_5 (the sparse multiplication in front made them): four terms at most
                xa = {j: self._xi_of[a[j].id] for j in (3, 4, 5)}
                m = lambda i, j: self.mul((a[i], a[j] if i + j < 6 else xa[j]))
                return [self.lin((m(0, 0), ID), (m(3, 3), ID), (m(1, 5), X2), (m(2, 4), X2)),
                        self.lin((m(0, 1), X2), (m(2, 5), X2), (m(3, 4), X2)),
                        self.lin((m(1, 1), ID), (m(0, 2), X2), (m(4, 4), ID), (m(3, 5), X2)),
                        self.lin((m(0, 3), X2), (m(1, 2), X2), (m(4, 5), X2)),
                        self.lin((m(2, 2), ID), (m(0, 4), X2), (m(1, 3), X2), (m(5, 5), ID)),
                        self.lin((m(0, 5), X2), (m(1, 4), X2), (m(2, 3), X2))]
            m = lambda i, j: self.mul((a[i], a[j]))
            return [self.lin((m(0, 0), ID), (m(3, 3), mxi()), (m(1, 5), mxi(2)), (m(2, 4), mxi(2))),
                    self.lin((m(0, 1), X2), (m(2, 5), mxi(2)), (m(3, 4), mxi(2))),
                    self.lin((m(1, 1), ID), (m(0, 2), X2), (m(4, 4), mxi()), (m(3, 5), mxi(2))),
                    self.lin((m(0, 3), X2), (m(1, 2), X2), (m(4, 5), mxi(2))),
                    self.lin((m(2, 2), ID), (m(0, 4), X2), (m(1, 3), X2), (m(5, 5), mxi())),
                    self.lin((m(0, 5), X2), (m(1, 4), X2), (m(2, 3), X2))]
        if self.flat_sqr:
            # thirty-two lanes: the schoolbook square without sums in front -- the 21 distinct products a_i a_j in fifteen sums of at
            # most three (by weight and by wrap: c_k = lo_k + xi hi_k), ONE round of thirty operations, then the recombination:
            # two rounds on f's chain instead of three
            a = f
            m = self.mul
            X = {1: m((a[0], a[1])), 2: m((a[0], a[2])), 3: m((a[0], a[3]), (a[1], a[2])), 4: m((a[0], a[4]), (a[1], a[3])),
                 5: m((a[0], a[5]), (a[1], a[4])), 6: m((a[2], a[3]))}          # (X5 in two: no sum of the round has more than two
            Y = {0: m((a[1], a[5]), (a[2], a[4])), 1: m((a[2], a[5]), (a[3], a[4])), 2: m((a[3], a[5])), 3: m((a[4], a[5]))}   # products -- a four-product round)
            S = [m((a[i], a[i])) for i in range(6)]
            return [self.lin((S[0], ID), (S[3], mxi()), (Y[0], mxi(2))),
                    self.lin((X[1], mk(2)), (Y[1], mxi(2))),
                    self.lin((S[1], ID), (X[2], mk(2)), (S[4], mxi()), (Y[2], mxi(2))),
                    self.lin((X[3], mk(2)), (Y[3], mxi(2))),
                    self.lin((S[2], ID), (X[4], mk(2)), (S[5], mxi())),
                    self.lin((X[5], mk(2)), (X[6], mk(2)))]
        A0, A1 = (f[0], f[2], f[4]), (f[1], f[3], f[5])
        ax0 = (self.xi(A0[1]), self.xi(A0[2]))
        t = self.fq6_mul(A0, A1, ax0)
        s = tuple(self.lin((A0[i], ID), (A1[i], ID)) for i in range(3))                      # A0 + A1
        sx = (self.lin((A0[1], mxi()), (A1[1], mxi())), self.lin((A0[2], mxi()), (A1[2], mxi())))
        vA1 = (self.xi(A1[2]), A1[0], A1[1])                                                # v A1 = (xi a2, a0, a1)
        r = (self.lin((A0[0], ID), (A1[2], mxi())), self.lin((A0[1], ID), (A1[0], ID)), self.lin((A0[2], ID), (A1[1], ID)))
        del vA1
        u = self.fq6_mul(s, r, sx)
        # c0 part: u - t - v t ; v t = (xi t2, t0, t1)
        e0 = self.lin((u[0], ID), (t[0], NEG), (t[2], mxi(-1)))
        e1 = self.lin((u[1], ID), (t[1], NEG), (t[0], NEG))
        e2 = self.lin((u[2], ID), (t[2], NEG), (t[1], NEG))
        o = [self.lin((t[i], mk(2))) for i in range(3)]
        return [e0, o[0], e1, o[1], e2, o[2]]

    def fq12_conj(self, f):
        return [f[i] if i % 2 == 0 else self.neg(f[i]) for i in range(6)]

    def cyc_sqr(self, f):
        """Granger-Scott squaring in the cyclotomic subgroup, three units (a, b) = (f0, f3), (f1, f4), (f2, f5):
        P = (a + b)(a + xi b), t = a b;  a^2 + xi b^2 = P - (1 + xi) t;  outputs 3 (..) - 2 z and 6 t [xi] + 2 z"""
        out = [None] * 6
        if self.wide:
            # thirty-two lanes: no sums in front of the products -- a^2 + xi b^2 from the monomials a0^2 - a1^2, b0^2 - b1^2, a0 a1, b0 b1
            # (real parts as values of their own: aliases, no operation), 18 product operations in ONE round instead of 12 behind a
            # round of 12 sums: two rounds per squaring instead of three
            RE, IM = (1, 0, 0, 0), (0, 1, 0, 0)
            for (ia, ib, iza, izb, x) in ((0, 3, 0, 3, False), (1, 4, 2, 5, False), (2, 5, 4, 1, True)):
                a, b = f[ia], f[ib]
                a0, a1, b0, b1 = self.lin((a, RE)), self.lin((a, IM)), self.lin((b, RE)), self.lin((b, IM))
                sqa = self.mul((a0, a0), (self.neg(a1), a1))
                sqb = self.mul((b0, b0), (self.neg(b1), b1))
                pbb, paa = self.mul((b0, b1)), self.mul((a0, a1))
                t = self.mul((a, b))
                # a^2 + xi b^2 = (sqa + 9 sqb - 2 pbb) + (2 paa + 18 pbb + sqb) u
                out[iza] = self.lin((sqa, (3, 0, 0, 0)), (sqb, (27, 0, 3, 0)), (pbb, (-6, 0, 54, 0)), (paa, (0, 0, 6, 0)), (f[iza], mk(-2)))
                out[izb] = self.lin((t, mxi(6) if x else mk(6)), (f[izb], mk(2)))
            return out
        for (ia, ib, iza, izb, x) in ((0, 3, 0, 3, False), (1, 4, 2, 5, False), (2, 5, 4, 1, True)):
            a, b = f[ia], f[ib]
            u = self.lin((a, ID), (b, ID))
            s = self.lin((a, ID), (b, mxi()))
            Pv = self.mul((u, s))
            t = self.mul((a, b))
            out[iza] = self.lin((Pv, mk(3)), (t, (-30, 3, -3, -30)), (f[iza], mk(-2)))
            out[izb] = self.lin((t, mxi(6) if x else mk(6)), (f[izb], mk(2)))
        return out

    def frobenius(self, f, k):
        """frobenius_map_native (final_exp_native.rs:17-54): conj^k of each coefficient times frob_coeffs(k)^i"""
        g1 = f2pow(XI, (P ** k - 1) // 6)
        out = []
        for i in range(6):
            g = f2pow(g1, i)
            a = f[i]
            if g == (1, 0):
                out.append(self.conj(a) if k % 2 else a)
                continue
            if k % 2:
                # conj(a) g: fold the conjugation into the constant where the constant is real or purely imaginary; else one LIN first
                a = self.conj(a)
            out.append(self.mul((a, self.const(g))))
        return out

    def fq6_inv(self, a):
        a0, a1, a2 = a
        a1x, a2x = self.xi(a1), self.xi(a2)
        na0 = self.neg(a0)
        t0 = self.mul((a0, a0), (self.neg(a1x), a2))
        t1 = self.mul((a2x, a2), (na0, a1))
        t2 = self.mul((a1, a1), (na0, a2))
        n = self.mul((a0, t0), (a2x, t1), (a1x, t2))
        ni = self.fq2_inv(n)
        return (self.mul((t0, ni)), self.mul((t1, ni)), self.mul((t2, ni)))

    def fq12_inv(self, f):
        """ark Fq12 inverse through the Fq6 norm: 1 / (A0 + A1 w) = (A0 - A1 w) / (A0^2 - v A1^2)"""
        A0, A1 = (f[0], f[2], f[4]), (f[1], f[3], f[5])
        s0 = self.fq6_mul(A0, A0)
        s1 = self.fq6_mul(A1, A1)
        d = (self.lin((s0[0], ID), (s1[2], mxi(-1))), self.lin((s0[1], ID), (s1[0], NEG)), self.lin((s0[2], ID), (s1[1], NEG)))
        di = self.fq6_inv(d)
        r0 = self.fq6_mul(A0, di)
        nA1 = tuple(self.neg(x) for x in A1)
        r1 = self.fq6_mul(nA1, di)
        return [r0[0], r1[0], r0[1], r1[1], r0[2], r1[2]]

    # -------------------------------------------------------------- G2 steps and sparse products (tests/sched_model.py)
    def dbl_step(self, Rp, px, py, after=None):
        """homogeneous projective doubling, the point scaled by xi^2 (L1v4.r_dblstep); line (L0, L3, L4) (miller_loop_native.rs:30-44).
        px, py: the G1 point's coordinates as Fq2 values (c1 = 0)."""
        X, Y, Z = Rp
        self._in_point_step = True
        B = self.mul((Y, Y), after=after)
        C = self.mul((Z, Z), after=after)
        XX = self.mul((X, X), after=after)
        XY = self.mul((X, Y), after=after)
        YZ = self.mul((Y, Z), after=after)
        # N = 9 C, xB = xi B, T = xB - 3 N, S = xB + 3 N, H = 2 Y Z
        T = self.lin((B, mxi()), (C, mk(-27)))
        S = self.lin((B, mxi()), (C, mk(27)))
        N = self.lin((C, mk(9)))
        N12 = self.lin((C, mk(-108)))
        xB = self.xi(B)
        xH4 = self.lin((YZ, mxi(8)))                 # 4 xi H
        XYx2 = self.lin((XY, mxi(2)))
        X3 = self.mul((XYx2, T))
        # (sixty-four lanes: the two products of Y3 one by one, summed in the combination round that f's chain has anyway -- the round of
        # the second-level products stays a two-product round)
        Y3 = self.lin((self.mul((S, S)), ID), (self.mul((N, N12)), ID)) if self.full else self.mul((S, S), (N, N12))
        Z3 = self.mul((xB, xH4))
        L0 = self.lin((B, mxi()), (C, mk(-9)))
        H = self.lin((YZ, mk(2)))
        L3 = self.mul((H, py))
        XX3n = self.lin((XX, mk(-3)))
        L4 = self.mul((XX3n, px))
        self._in_point_step = False
        return (X3, Y3, Z3), (L0, L3, L4)

    def add_step(self, Rp, Q, px, py, nQy=None):
        """mixed addition R + Q, line (L2, L3, L5) (miller_loop_native.rs:10-28)"""
        X, Y, Z = Rp
        x2, y2 = Q
        self._in_point_step = True
        ny2 = nQy if nQy is not None else self.neg(y2)
        nx2 = self.neg(x2)
        theta = self.mul((ny2, Z), add=Y)
        mu = self.mul((nx2, Z), add=X)
        L5 = self.lin((self.mul((X, y2)), ID), (self.mul((nx2, Y)), ID)) if self.full else self.mul((X, y2), (nx2, Y))
        nmu = self.neg(mu)
        L2 = self.mul((nmu, py))
        L3 = self.mul((theta, px))
        C = self.mul((theta, theta))
        D = self.mul((mu, mu))
        E = self.mul((mu, D))
        F = self.mul((Z, C))
        G = self.mul((X, D))
        H = self.lin((E, ID), (F, ID), (G, mk(-2)))
        GH = self.lin((G, mk(3)), (E, NEG), (F, NEG)) if (self.full or GH_SHORT) else self.lin((G, ID), (H, NEG))       # (G - H beside H, not behind it)
        nE = self.neg(E)
        X3 = self.mul((mu, H))
        Y3 = self.mul((theta, GH), (nE, Y))
        Z3 = self.mul((Z, E))
        self._in_point_step = False
        return (X3, Y3, Z3), (L2, L3, L5)

    def _sparse_full(self, a, terms, want_x=False):
        """sixty-four lanes: the eighteen products of a sparse multiplication one by one, the sums in a combination of at most four
        terms (the line's coefficients times xi where the product wraps: on the line's path, which leads).  want_x: the result is squared
        next -- xi c_3, xi c_4, xi c_5 as nine more products with the coefficients times xi once more, so that the square's combinations
        keep to four terms too (Graph.fq12_sqr)"""
        bx = {}

        def X(b, n):
            if n and (b.id, n) not in bx:
                bx[b.id, n] = self.lin((b, mxi() if n == 1 else XI2))
            return bx[b.id, n] if n else b
        out = [self.lin(*[(self.mul((a[i], X(b, wrapped))), ID) for i, b, wrapped in row]) for row in terms]
        if want_x:
            for k in (3, 4, 5):
                self._xi_of[out[k].id] = self.lin(*[(self.mul((a[i], X(b, wrapped + 1))), ID) for i, b, wrapped in terms[k]])
        return out

    def mul_by_034(self, a, L, want_x=False):
        b0, b3, b4 = L
        if self.full:
            return self._sparse_full(a, want_x=want_x, terms=[[(0, b0, 0), (3, b3, 1), (2, b4, 1)], [(1, b0, 0), (4, b3, 1), (3, b4, 1)], [(2, b0, 0), (5, b3, 1), (4, b4, 1)],
                                         [(3, b0, 0), (0, b3, 0), (5, b4, 1)], [(4, b0, 0), (1, b3, 0), (0, b4, 0)], [(5, b0, 0), (2, b3, 0), (1, b4, 0)]])
        b3x, b4x = self.xi(b3), self.xi(b4)
        return [self.mul((a[0], b0), (a[3], b3x), (a[2], b4x)),
                self.mul((a[1], b0), (a[4], b3x), (a[3], b4x)),
                self.mul((a[2], b0), (a[5], b3x), (a[4], b4x)),
                self.mul((a[3], b0), (a[0], b3), (a[5], b4x)),
                self.mul((a[4], b0), (a[1], b3), (a[0], b4)),
                self.mul((a[5], b0), (a[2], b3), (a[1], b4))]

    def mul_by_235(self, a, L, want_x=False):
        b2, b3, b5 = L
        if self.full:
            return self._sparse_full(a, want_x=want_x, terms=[[(4, b2, 1), (3, b3, 1), (1, b5, 1)], [(5, b2, 1), (4, b3, 1), (2, b5, 1)], [(0, b2, 0), (5, b3, 1), (3, b5, 1)],
                                         [(1, b2, 0), (0, b3, 0), (4, b5, 1)], [(2, b2, 0), (1, b3, 0), (5, b5, 1)], [(3, b2, 0), (2, b3, 0), (0, b5, 0)]])
        b2x, b3x, b5x = self.xi(b2), self.xi(b3), self.xi(b5)
        return [self.mul((a[4], b2x), (a[3], b3x), (a[1], b5x)),
                self.mul((a[5], b2x), (a[4], b3x), (a[2], b5x)),
                self.mul((a[0], b2), (a[5], b3x), (a[3], b5x)),
                self.mul((a[1], b2), (a[0], b3), (a[4], b5x)),
                self.mul((a[2], b2), (a[1], b3), (a[5], b5x)),
                self.mul((a[3], b2), (a[2], b3), (a[0], b5))]

    # -------------------------------------------------------------- the reference's functions
    def multi_miller_loop(self, pairs, exact=False):
        """multi_miller_loop_native over pairs [((px, py), (qx, qy))] (miller_loop_native.rs:192-282; one pair: miller_loop_native,
        :320-322), shared f, projective lines: tests/sched_model.py miller_projective.  The lines carry an Fq2 factor each (lam = Z^2
        of a doubling, Z of an addition), which the final exponentiation removes; exact=True tracks their product and divides it
        out at the end -- the reference's value itself."""
        # exact: the reference's digit table (miller_loop_native.rs:314-318: 65 digits, 26 non-zero), digit for digit.  Otherwise the value
        # goes into the final exponentiation, which does not see the chain: the 65-digit form with 22 non-zero digits (the same doublings,
        # four additions less -- tools/asmcore.py)
        enc = SIX_U_PLUS_2_NAF if exact or not SHORT_CHAIN else SIX_U_PLUS_2_SHORT
        first = len(enc) - 2                          # the digit of the first doubling (the top digit is R = Q, f = 1)
        one, zero = self.const((1, 0)), self.const((0, 0))
        nQy = [self.neg(Q[1]) for _, Q in pairs]
        Rs = [(Q[0], Q[1], one) for _, Q in pairs]
        scale = None
        f = None
        hist = []                                     # f at the start of every iteration (run_ahead)

        pend = []                                     # line_tree: the lines of the current step that f has not taken yet

        def take(L6, last=False):
            """last: the iteration's last line -- f is squared next"""
            nonlocal f
            if self.line_tree and f is not None:
                pend.append(L6)
            elif f is None:
                f = [c if c is not None else zero for c in L6]
            elif L6[0] is not None:
                f = self.mul_by_034(f, (L6[0], L6[3], L6[4]), want_x=last)
            else:
                f = self.mul_by_235(f, (L6[2], L6[3], L6[5]), want_x=last)

        def flush():
            nonlocal f
            while len(pend) > 1:                      # pairwise products of the pending lines, level by level
                nxt = [self.fq12_mul_gen(pend[i], pend[i + 1]) for i in range(0, len(pend) - 1, 2)]
                if len(pend) % 2:
                    nxt.append(pend[-1])
                pend[:] = nxt
            if pend:
                T = pend.pop()
                f = self.fq12_mul(f, T) if self.full and all(c is not None for c in T) else self.fq12_mul_gen(f, T)

        def dbl(j, last=False):
            nonlocal f, scale
            (px, py), _ = pairs[j]
            lam = Rs[j][2]
            ra = self.run_ahead
            Rs[j], L = self.dbl_step(Rs[j], px, py, after=hist[-ra][0] if ra is not None and len(hist) >= ra else None)
            take([L[0], None, None, L[1], L[2], None], last)
            if exact and lam is not one:
                # the factor is lam = Z^2 and the scale is squared right after the iteration's doublings: (scale Z)^2 = scale^2 lam --
                # two dependent products per doubling instead of three, as many as f's own chain has rounds for
                scale = lam if scale is None else self.mul((scale, lam))

        def add(j, Qs, nQsy, last=False):
            nonlocal f, scale
            (px, py), _ = pairs[j]
            lam = Rs[j][2]
            Rs[j], L = self.add_step(Rs[j], Qs, px, py, nQy=nQsy)
            take([None, None, L[0], L[1], None, L[2]], last)
            if exact:
                scale = lam if scale is None else self.mul((scale, lam))

        for j in range(len(pairs)):
            dbl(j)
        flush()
        for i in range(first, -1, -1):
            if i != first:
                hist.append(f)
                f = self.fq12_sqr(f)
                for j in range(len(pairs)):
                    dbl(j, last=self.full and enc[i] == 0 and i > 0 and j == len(pairs) - 1)
                flush()
                if exact and scale is not None:           # scale <- (scale * prod Z_j)^2 = scale^2 * prod lam_j
                    scale = self.mul((scale, scale))
            if enc[i] != 0:
                for j, (_, Q) in enumerate(pairs):
                    last = self.full and i > 0 and j == len(pairs) - 1
                    if enc[i] == 1:
                        add(j, (Q[0], Q[1]), nQy[j], last)
                    else:
                        add(j, (Q[0], nQy[j]), Q[1], last)
                flush()
        c2, c3 = end_constants()
        for j, (_, Q) in enumerate(pairs):
            Q1 = (self.mul((self.conj(Q[0]), self.const(c2))), self.mul((self.conj(Q[1]), self.const(c3))))
            nQ2 = (self.mul((self.conj(Q1[0]), self.const(c2))), self.mul((self.lin((Q1[1], NCONJ)), self.const(c3))))
            add(j, Q1, None)
            add(j, nQ2, None)
        flush()
        if exact:
            si = self.fq2_inv(scale)
            f = [self.mul((c, si)) for c in f]
        return f

    def miller_loop(self, px, py, Q, exact=False):
        return self.multi_miller_loop([((px, py), Q)], exact=exact)

    POW_X_WINDOW = int(os.environ.get("CVM_POW_WINDOW", "3"))     # 4: 1.5 % faster for one pairing (0.992 ms), but 346 slots instead of 271: one wave less per CU

    def pow_x(self, a):
        """pow_native(a, [BN_X]) for a in the cyclotomic subgroup (the same element whatever the chain: final_exp_native.rs:56-84 walks
        the NAF with true divisions): cyclotomic squarings, the inverse as the conjugate, and a signed window of POW_X_WINDOW bits over
        the odd powers a, a^3, ... -- window 3: 17 multiplications and one for the table instead of the NAF's 23; a multiplication is
        two rounds on the critical path, a squaring three."""
        w = self.pow_window
        if w == "fixed":
            # the throughput kernels' signed fixed-set recoding (tools/kgen4_prog.py X_DIGITS, found by tools/exp/xchain2.py): digits in
            # {0, +-1, +-15, +-19}, 58 squarings + 11 multiplications in the loop, b^4, b^16 from 4 squarings, b^15 = b^16 conj(b), b^19 = b^15 b^4
            from kgen4_prog import X_DIGITS, X_POWERS
            assert X_POWERS == (1, 15, 19)
            digits = list(X_DIGITS)
            top = len(digits) - 1
            b4 = self.cyc_sqr(self.cyc_sqr(a))
            b16 = self.cyc_sqr(self.cyc_sqr(b4))
            odd = {1: a}
            odd[15] = self.fq12_mul(self.fq12_conj(a), b16, self.fq12_mul_pre(b16))
            odd[19] = self.fq12_mul(odd[15], b4, self.fq12_mul_pre(b4))
        elif w == "x19":
            # two table entries like the three-bit window ({1, 3}: 63 squarings + 18 multiplications), but the digit set {1, 19} of the search in
            # tools/exp/xchain2.py: 58 squarings + 12 multiplications in the loop, b^2 .. b^16 by four squarings, b^19 = b^16 b^2 b -- 62 S + 14 M
            # (the sixteen-lane programs, whose slot budget has no room for a third entry)
            digits = list(X19_DIGITS)
            top = len(digits) - 1
            b2 = self.cyc_sqr(a)
            b16 = self.cyc_sqr(self.cyc_sqr(self.cyc_sqr(b2)))
            b18 = self.fq12_mul(b16, b2, self.fq12_mul_pre(b2))
            odd = {1: a}
            odd[19] = self.fq12_mul(b18, a, self.fq12_mul_pre(a))
        else:
            digits = []
            e = BN_X
            while e:
                d = 0
                if e & 1:
                    d = e % (1 << w)
                    if d >= 1 << (w - 1):
                        d -= 1 << w
                    e -= d
                digits.append(d)
                e >>= 1
            top = len(digits) - 1
            assert digits[top] > 0
            odd = {1: a}
            need = max(abs(d) for d in digits)
            if need > 1:
                a2 = self.cyc_sqr(a)
                a2x = self.fq12_mul_pre(a2)
                for k in range(3, need + 1, 2):
                    odd[k] = self.fq12_mul(odd[k - 2], a2, a2x)
        pre = {}
        for k, v in odd.items():
            vc = self.fq12_conj(v)
            vx = self.fq12_mul_pre(v)
            pre[k] = (v, vx)
            pre[-k] = (vc, None if vx is None else [None] + [vx[j] if j % 2 == 0 else self.neg(vx[j]) for j in range(1, 6)])     # xi conj(v)_j = +- xi v_j: the twins
        res = odd[digits[top]]
        for i in range(top - 1, -1, -1):
            res = self.cyc_sqr(res)
            if digits[i]:
                b, bx = pre[digits[i]]
                res = self.fq12_mul(res, b, bx)
        return res

    def final_exp(self, f):
        """final_exp_native (final_exp_native.rs:209-213): easy part, then hard_part_BN_native (:130-169) in the kernels' schedule
        (tests/sched_model.py final_exp_gpu)"""
        m = self.easy_part(f)
        mx = self.pow_x(m)
        mx2 = self.pow_x(mx)
        mx3 = self.pow_x(mx2)
        return self.hard_tail_2(m, self.hard_tail_1(mx, mx2, mx3))

    # the pieces of final_exp_native, as the one-launch programs use them in sequence and the mid-size batches as separate launches
    # (EASY, POWX three times, YCH1, YCH2: each piece holds at most ~146 slots where the whole holds 277)
    def easy_part(self, f):
        """easy_part (final_exp_native.rs:195-206): f^((p^6 - 1)(p^2 + 1))"""
        f2 = self.fq12_mul(self.fq12_conj(f), self.fq12_inv(f))
        return self.fq12_mul(self.frobenius(f2, 2), f2)

    def hard_tail_1(self, mx, mx2, mx3, serial=False):
        """the part of hard_part_BN_native's y-chain (final_exp_native.rs:130-169) that needs m^x, m^(x^2), m^(x^3) only: ((y3 y5 T0)^2 (y2 T0))^2
        with T0 = y6^2 y4 y5"""
        mxp = self.frobenius(mx, 1)
        mx2p = self.frobenius(mx2, 1)
        y2 = self.frobenius(mx2, 2)
        y5 = self.fq12_conj(mx2)
        mx3p = self.frobenius(mx3, 1)
        y3 = self.fq12_conj(mxp)
        y4 = self.fq12_conj(self.fq12_mul(mx, mx2p))
        y6 = self.fq12_conj(self.fq12_mul(mx3, mx3p))
        T0 = self.cyc_sqr(y6)
        T0 = self.fq12_mul(T0, y4)
        T0 = self.fq12_mul(T0, y5)
        # serial (the stand-alone piece YCH1): y3 y5 is needed behind T0 only -- computed early it waits in twelve slots (+ twins)
        T1 = self.fq12_mul(y3, y5, after=T0[0] if serial else None)
        T1 = self.fq12_mul(T1, T0)
        T0 = self.fq12_mul(y2, T0)
        T1 = self.cyc_sqr(T1)
        T1 = self.fq12_mul(T1, T0)
        return self.cyc_sqr(T1)

    def hard_tail_2(self, m, T1):
        """... and the rest: (T1 y1)^2 (T1 y0) with y0 = m^p m^(p^2) m^(p^3), y1 = conj(m)"""
        mp, mp2, mp3 = self.frobenius(m, 1), self.frobenius(m, 2), self.frobenius(m, 3)
        y0 = self.fq12_mul(mp, self.fq12_mul(mp2, mp3))
        y1 = self.fq12_conj(m)
        T0 = self.fq12_mul(T1, y1)
        T1 = self.fq12_mul(T1, y0)
        T0 = self.cyc_sqr(T0)
        return self.fq12_mul(T0, T1)

    # -------------------------------------------------------------- evaluation on integers (no schedule: the DAG's meaning)
    def evaluate(self, inputs):
        val = {}
        for v, x in zip(self.inputs, inputs):
            val[v.id] = (x[0] % P, x[1] % P)
        for v in self.vals:
            if v.kind == "const":
                val[v.id] = v.args
            elif v.kind in ("m1", "m3"):
                t, add = v.args
                acc = val[add.id] if add is not None else (0, 0)
                for x, y in t:
                    acc = f2add(acc, f2mul(val[x.id], val[y.id]))
                val[v.id] = acc
            elif v.kind == "lin":
                acc = (0, 0)
                for s, m in v.args:
                    acc = f2add(acc, f2mat(m, val[s.id]))
                val[v.id] = acc
            elif v.kind == "inv":
                val[v.id] = (pow(val[v.args.id][0], P - 2, P), 0)
        return [val[o.id] for o in self.outputs]


def naf_of(e):
    out = []
    while e:
        if e & 1:
            z = 2 - (e % 4)
            e -= z
        else:
            z = 0
        out.append(z)
        e >>= 1
    return out


def end_constants():
    """(xi^((p-1)/3), xi^((p-1)/2)) (miller_loop_native.rs:176-181)"""
    return f2pow(XI, (P - 1) // 3), f2pow(XI, (P - 1) // 2)


# ------------------------------------------------------------------ lowering to Fq operations
class FV:
    """one Fq value: kind in / const / mul / lin / inv, or `negof` (the negated twin its producer's lane writes beside the result)"""
    __slots__ = ("id", "kind", "args", "bound", "users", "rnd", "slot", "height", "last", "twin", "cost", "after", "pt")

    def __init__(self, id_, kind, args, bound):
        self.id, self.kind, self.args, self.bound = id_, kind, args, bound
        self.users, self.rnd, self.slot, self.height, self.last, self.twin, self.cost, self.after, self.pt = [], None, None, 0, -1, None, None, None, False

    def srcs(self):
        if self.kind == "mul":
            t, add = self.args
            return [v for xy in t for v in xy] + ([add] if add is not None else [])
        if self.kind == "lin":
            return [s for s, _ in self.args]
        if self.kind in ("inv", "negof"):
            return [self.args]
        return []


class Lowered:
    """The Fq2 graph as Fq operations (one lane each):
        mul  dst <- sum of up to six products x y (+ addend)             (fips: one Montgomery column pass)
        lin  dst <- sum of up to eight coef * src, reduced to (-0.51 p, 0.51 p)
        inv  dst <- 1 / src
    An Fq2 value is (c0, c1) with c1 = None when it is known to be zero (G1 coordinates, inverses, real constants): products with
    a zero component are never formed.  The c0 part of a product needs -x1 y1: any value may have a NEGATED TWIN, written by the
    lane that produces it (nine subtractions) -- negations and conjugations of the graph cost nothing else.
    Bounds (units of p): mul sum|x||y| / 169.6 + 0.51 (+ addend), lin 0.51; checked against the column pass's and the reducing
    chain's limits."""
    MUL_BUDGET = 300.0
    LIN_BUDGET = 500.0
    V_MAX = 6.0

    def __init__(self, g):
        self.g = g
        self.fv = []
        self.fconst = {}
        self.inputs = []
        self.map = {}            # Fq2 value id -> (c0 FV, c1 FV | None)
        self.zero = self.const(0)
        for v in g.vals:
            n0 = len(self.fv)
            self.map[v.id] = self._lower(v)
            if v.pt:
                for r in self.fv[n0:]:
                    r.pt = True
        self.outputs = []
        for o in g.outputs:
            c0, c1 = self.map[o.id]
            self.outputs += [c0, c1 if c1 is not None else self.zero]

    def _new(self, kind, args, bound):
        v = FV(len(self.fv), kind, args, bound)
        assert bound <= self.V_MAX, bound
        self.fv.append(v)
        for s in dict.fromkeys(v.srcs()):
            s.users.append(v)
        return v

    def const(self, c):
        c %= P
        if c not in self.fconst:
            self.fconst[c] = self._new("const", c, 0.51)
        return self.fconst[c]

    def neg(self, x):
        if x.kind == "const":
            return self.const(-x.args)
        if x.kind == "negof":
            return x.args
        if x.kind == "in":                       # no lane produces an input: its negation is an operation of its own
            key = ("negin", x.id)
            if key not in self.fconst:
                self.fconst[key] = self._new("lin", [(x, -1)], 0.51)
            return self.fconst[key]
        if x.twin is None:
            x.twin = self._new("negof", x, x.bound)
        return x.twin

    def _has_neg(self, x):
        return x.kind in ("const", "negof") or x.twin is not None or ("negin", x.id) in self.fconst

    def _mul(self, prods, add):
        prods = [(x, y) for x, y in prods]
        if not prods:
            return add
        load = sum(x.bound * y.bound for x, y in prods)
        assert load <= self.MUL_BUDGET and len(prods) <= 6, (load, len(prods))
        return self._new("mul", (prods, add), load / 169.6 + 0.51 + (add.bound if add is not None else 0))

    def _lin(self, terms):
        terms = [(s, c) for s, c in terms if c != 0]
        if not terms:
            return None
        if len(terms) == 1 and terms[0][1] == 1:
            return terms[0][0]                                     # a copy: the same value
        if len(terms) == 1 and terms[0][1] == -1:
            return self.neg(terms[0][0])
        load = sum(s.bound * abs(c) for s, c in terms)
        assert load <= self.LIN_BUDGET and len(terms) <= 8, (load, len(terms))
        merged = {}
        for s, c in terms:
            merged[s.id] = (s, merged.get(s.id, (s, 0))[1] + c)
        return self._new("lin", [t for t in merged.values() if t[1] != 0], 0.51)

    def _lower(self, v):
        if v.kind == "in":
            arr, fq0, fq1, pair = v.src if v.src is not None else (ARR_NONE, 0, 0, 0)
            c0 = self._new("in", (v.name + ".c0", arr, fq0, pair), 1.01)
            self.inputs.append(c0)
            c1 = None
            if not v.real:
                c1 = self._new("in", (v.name + ".c1", arr, fq1, pair), 1.01)
                self.inputs.append(c1)
            return c0, c1
        if v.kind == "const":
            return self.const(v.args[0]), (self.const(v.args[1]) if v.args[1] else None)
        if v.kind in ("m1", "m3"):
            t, add = v.args
            p0, p1 = [], []
            for x, y in t:
                x0, x1 = self.map[x.id]
                y0, y1 = self.map[y.id]
                p0.append((x0, y0))
                if x1 is not None and y1 is not None:
                    if self._has_neg(y1) and not self._has_neg(x1):
                        p0.append((x1, self.neg(y1)))
                    else:
                        p0.append((self.neg(x1), y1))
                if y1 is not None:
                    p1.append((x0, y1))
                if x1 is not None:
                    p1.append((x1, y0))
            a0, a1 = self.map[add.id] if add is not None else (None, None)
            n0 = len(self.fv)
            r0, r1 = self._mul(p0, a0), self._mul(p1, a1)
            if v.after is not None:
                for r in self.fv[n0:]:
                    if r.kind == "mul":
                        r.after = self.map[v.after.id][0]
            return r0, r1
        if v.kind == "lin":
            out = []
            for h in range(2):
                terms = []
                for s, m in v.args:
                    s0, s1 = self.map[s.id]
                    terms.append((s0, m[2 * h]))
                    if s1 is not None:
                        terms.append((s1, m[2 * h + 1]))
                out.append(self._lin(terms))
            return (out[0] if out[0] is not None else self.zero), out[1]
        if v.kind == "inv":
            return self._new("inv", self.map[v.args.id][0], 1.01), None
        raise ValueError(v.kind)

    def evaluate(self, inputs):
        """the Fq graph on integers: inputs in the order of self.inputs"""
        val = {}
        for v, x in zip(self.inputs, inputs):
            val[v.id] = x % P
        for v in self.fv:
            if v.kind == "const":
                val[v.id] = v.args
            elif v.kind == "mul":
                t, add = v.args
                val[v.id] = (sum(val[x.id] * val[y.id] for x, y in t) + (val[add.id] if add is not None else 0)) % P
            elif v.kind == "lin":
                val[v.id] = sum(val[s.id] * c for s, c in v.args) % P
            elif v.kind == "inv":
                val[v.id] = pow(val[v.args.id], P - 2, P)
            elif v.kind == "negof":
                val[v.id] = -val[v.args.id] % P
        return [val[o.id] for o in self.outputs]


# ------------------------------------------------------------------ scheduling
K_END, K_M2, K_M4, K_M6, K_L4, K_L8, K_INV = 0, 1, 2, 3, 4, 5, 6
KIND_NAME = {K_M2: "m2", K_M4: "m4", K_M6: "m6", K_L4: "l4", K_L8: "l8", K_INV: "inv"}
COST = {K_M2: 385, K_M4: 565, K_M6: 750, K_L4: 160, K_L8: 225, K_INV: 17600}       # instructions per round incl. the loop's own (scheduling weights; tools/cvm_kernel.py prints the handlers' sizes)
FAMILY = {K_M2: "m", K_M4: "m", K_M6: "m", K_L4: "l", K_L8: "l", K_INV: "i"}


def op_kind(v):
    if v.kind == "mul":
        n = len(v.args[0])
        return K_M2 if n <= 2 else K_M4 if n <= 4 else K_M6
    if v.kind == "lin":
        return K_L4 if len(v.args) <= 4 else K_L8
    return K_INV


class Program:
    """rounds: [(kind, [FV] (one per lane, at most nr))] + slot numbers"""

    def __init__(self, low, nr=16, greedy_fill=True, inv_weight=None, m_weight=1.0, pt_weight=1.0):
        """inv_weight: the inversion's weight in the critical-path heights, if not its cost.  The exact Miller programs end with the
        inversion of the scale, which needs the whole point chain: a larger weight lets that chain (and its lines, in LDS) run further
        ahead of f -- fewer rounds, more slots."""
        self.low = low
        self.pt_weight = pt_weight        # factor on the weight of the point steps' operations in the heights (< 1: f's chain leads the choice of a round's family)
        self.m_weight = m_weight          # factor on the two- and four-product passes' weight in the heights (1.5: the sixty-four-lane multi-pair programs)
        self.inv_weight = inv_weight
        self.nr = nr
        self.greedy_fill = greedy_fill
        self._prune()
        self._heights()
        self._schedule()
        self._allocate()

    def _prune(self):
        low = self.low
        live = set()
        stack = list(low.outputs)
        while stack:
            v = stack.pop()
            if v.id in live:
                continue
            live.add(v.id)
            stack.extend(v.srcs())
        live |= {low.zero.id, low.const(1).id}         # what idle lanes read
        self.live = live
        for v in low.fv:
            v.users = [u for u in v.users if u.id in live]
            if v.twin is not None and v.twin.id not in live:
                v.twin = None
        self.ops = [v for v in low.fv if v.id in live and v.kind in ("mul", "lin", "inv")]
        for v in self.ops:
            v.cost = op_kind(v)

    @staticmethod
    def _consumers(v):
        """operations that wait for v's lane: its own users and its twin's"""
        out = list(v.users)
        if v.twin is not None:
            out += v.twin.users
        return [u for u in out if u.kind != "negof"]

    def _heights(self):
        for v in reversed(self.low.fv):
            if v.id not in self.live or v.kind not in ("mul", "lin", "inv"):
                continue
            own = self.inv_weight if self.inv_weight is not None and v.cost == K_INV else COST[v.cost] * (self.m_weight if v.cost in (K_M2, K_M4) else 1)
            if v.pt:
                own *= self.pt_weight
            v.height = own + max((u.height for u in self._consumers(v)), default=0)

    @staticmethod
    def _producer(s):
        return s.args if s.kind == "negof" else s

    def _schedule(self):
        pending = {}
        for v in self.ops:
            pr = {self._producer(s).id for s in v.srcs()}
            pending[v.id] = sum(1 for i in pr if self.low.fv[i].kind in ("mul", "lin", "inv"))
        waiters = {}
        for v in self.ops:
            t = self._producer(v.after) if v.after is not None else None
            if t is not None and t.kind in ("mul", "lin", "inv") and t.id in self.live:
                pending[v.id] += 1
                waiters.setdefault(t.id, []).append(v)
        ready = [v for v in self.ops if pending[v.id] == 0]
        rounds = []
        done = 0
        while done < len(self.ops):
            assert ready
            ready.sort(key=lambda v: -v.height)
            fam = FAMILY[ready[0].cost]
            cands = [v for v in ready if FAMILY[v.cost] == fam]
            if not self.greedy_fill:
                top = ready[0].cost
                cands = [v for v in cands if v.cost <= top]
            take = cands[:self.nr]
            kind = max(v.cost for v in take)
            rnd = len(rounds)
            rounds.append((kind, take))
            taken = {v.id for v in take}
            ready = [v for v in ready if v.id not in taken]
            for v in take:
                v.rnd = rnd
                done += 1
                seen = set()
                for u in self._consumers(v):
                    if u.id in seen:
                        continue
                    seen.add(u.id)
                    pending[u.id] -= 1
                    if pending[u.id] == 0:
                        ready.append(u)
                for u in waiters.get(v.id, ()):
                    pending[u.id] -= 1
                    if pending[u.id] == 0:
                        ready.append(u)
        self.rounds = rounds

    # -------------------------------------------------------------- LDS slots by liveness AND bank (round 5)
    # A round's operand fetch is one ds_read per operand part, executed by all lanes, each with its own slot address.  Measured
    # (tools/lds_bank_calib.hip, profiles/r05_lds_bank_calib.txt): with ONE wave on the CU a read whose lanes name random slots costs
    # 22.5 cycles against 19.5 for sixteen consecutive slots -- nothing to gain -- but the LDS pipe is shared by the waves of a CU, and
    # with FOUR waves (every launch of more than 768 waves) the same random pattern costs 44 cycles per read per wave against 21 for the
    # conflict-free one: the pipe, not the issue slots, then bounds the chain rounds (eight reads and six writes for 160 instructions).
    # So slots are assigned by liveness and by CLASS = slot mod 16: the values that the sixteen lanes of a group fetch in ONE read
    # (same round, same operand position) should sit in sixteen different classes -- 48-byte slots of different classes lie in
    # different 16-byte bank groups (3 is odd), 32-byte slots (the split layout) two per bank group, which is free.
    N_CLASS = 16

    def _operand_positions(self, v):
        """[(position in the row, FV)] of the slots operation v reads (Program.encode's layout)"""
        if v.kind == "mul":
            t, add = v.args
            out = []
            for i, (x, y) in enumerate(t):
                out += [(2 * i, x), (2 * i + 1, y)]
            if add is not None:
                out.append((12, add))
            return out
        if v.kind == "lin":
            return [(i, s_) for i, (s_, _) in enumerate(v.args)]
        return [(0, v.args)]

    def _access_sets(self):
        """{key: [FV]}: the values (distinct ones: lanes naming the same slot are served together) that the lanes of one sixteen-lane
        block touch with one LDS instruction -- an operand position of a round, or the round's result / twin stores"""
        sets = {}
        for rnd, (kind, take) in enumerate(self.rounds):
            for r, v in enumerate(take):
                blk = r // 16
                for pos, s_ in self._operand_positions(v):
                    sets.setdefault((rnd, pos, blk), {})[s_.id] = s_
                sets.setdefault((rnd, "w", blk), {})[v.id] = v
                if v.twin is not None:
                    sets.setdefault((rnd, "t", blk), {})[v.twin.id] = v.twin
        return {k: list(d.values()) for k, d in sets.items()}

    def conflict_stats(self):
        """(accesses, lanes beyond the first in a class) over all access sets: what the assignment leaves"""
        n = extra = 0
        for mem in self._access_sets().values():
            per = {}
            for v in mem:
                per[v.slot % self.N_CLASS] = per.get(v.slot % self.N_CLASS, 0) + 1
            n += len(mem)
            extra += sum(c - 1 for c in per.values())
        return n, extra

    def _allocate_banked(self, cap):
        """slots by liveness and class; at most `cap` slots (rows of sixteen are opened while the total stays below it)"""
        low, NC = self.low, self.N_CLASS
        n_rounds = len(self.rounds)
        sets = self._access_sets()
        keys_of = {}
        for k, mem in sets.items():
            for v in mem:
                keys_of.setdefault(v.id, []).append(k)
        cls = {}

        def cost(v, c):
            return sum(1 for k in keys_of.get(v.id, ()) for m in sets[k] if m.id != v.id and cls.get(m.id) == c)

        for v in low.fv:
            if v.id in self.live:
                v.last = max((u.rnd for u in v.users if u.kind != "negof"), default=-1)
        for o in low.outputs:
            o.last = n_rounds
        # constants: the kernel copies them to slots [0, n_const): a permutation inside that range, most-read first
        consts = [v for v in low.fv if v.kind == "const" and v.id in self.live]
        self.n_const = len(consts)
        free_c = list(range(self.n_const))
        for v in sorted(consts, key=lambda v: -len(keys_of.get(v.id, ()))):
            best = min(free_c, key=lambda s_: (cost(v, s_ % NC), s_))
            free_c.remove(best)
            v.slot = best
            cls[v.id] = best % NC
        nxt = self.n_const
        free = {c: [] for c in range(NC)}           # free slots by class
        top = [self.n_const]                        # slots [0, top) exist

        def take_slot(v):
            ranked = sorted(range(NC), key=lambda c: (cost(v, c), c))
            c0 = cost(v, ranked[0])
            for c in ranked:
                if free[c]:
                    # a free slot of a class that is as good as the best one, or (when the best classes are full) the best free one
                    if cost(v, c) == c0 or top[0] >= cap:
                        s_ = min(free[c])
                        free[c].remove(s_)
                        return s_
            # the best classes have no free slot and there is room: open slots up to the next one of the best class
            want = ranked[0]
            while top[0] < cap:
                s_ = top[0]
                top[0] += 1
                if s_ % NC == want:
                    return s_
                free[s_ % NC].append(s_)
            for c in ranked:
                if free[c]:
                    s_ = min(free[c])
                    free[c].remove(s_)
                    return s_
            s_ = top[0]                              # over the cap: the program needs more slots than that
            top[0] += 1
            return s_

        expire = {}
        for v in low.inputs:
            v.slot = take_slot(v)
            cls[v.id] = v.slot % NC
            if INPUT_SLOTS_EXPIRE and 0 <= v.last < n_rounds and not any(o is v for o in low.outputs):
                expire.setdefault(v.last + 1, []).append(v.slot)       # an input's slot serves other values once its last reader has run
        self.first_dyn = top[0]
        for rnd, (kind, take) in enumerate(self.rounds):
            for s_ in expire.pop(rnd, []):
                free[s_ % NC].append(s_)
            for v in take:
                for w in (v, v.twin):
                    if w is None:
                        continue
                    if w.last <= rnd and w.last < n_rounds:
                        assert w is v and v.twin is not None, (w.kind, w.id)
                    w.slot = take_slot(w)
                    cls[w.id] = w.slot % NC
                    if w.last < n_rounds:
                        expire.setdefault(max(w.last, rnd) + 1, []).append(w.slot)
        self.n_slots = top[0]

    def _allocate(self):
        if BANK_AWARE:
            self._allocate_plain()
            base = self.n_slots
            self._allocate_banked(base + BANK_SLACK)
            return
        self._allocate_plain()

    def _allocate_plain(self):
        low = self.low
        nxt = 0
        for v in low.fv:                        # constants first, then inputs: fixed slots
            if v.kind == "const" and v.id in self.live:
                v.slot = nxt
                nxt += 1
        self.n_const = nxt
        for v in low.inputs:
            v.slot = nxt
            nxt += 1
        self.first_dyn = nxt
        n_rounds = len(self.rounds)
        for v in low.fv:
            if v.id in self.live:
                v.last = max((u.rnd for u in v.users if u.kind != "negof"), default=-1)
        for o in low.outputs:
            o.last = n_rounds
        free, expire = [], {}
        if INPUT_SLOTS_EXPIRE:
            for v in low.inputs:
                if 0 <= v.last < n_rounds and not any(o is v for o in low.outputs):
                    expire.setdefault(v.last + 1, []).append(v.slot)
        for rnd, (kind, take) in enumerate(self.rounds):
            free += expire.pop(rnd, [])
            free.sort(reverse=True)
            for v in take:
                for w in (v, v.twin):
                    if w is None:
                        continue
                    if w.last <= rnd and w.last < n_rounds:
                        assert w is v and v.twin is not None, (w.kind, w.id)     # a value only its twin's users want: it still needs a slot to be written to
                    w.slot = free.pop() if free else nxt
                    if w.slot == nxt:
                        nxt += 1
                    if w.last < n_rounds:
                        expire.setdefault(max(w.last, rnd) + 1, []).append(w.slot)
        self.n_slots = nxt                       # + 1: the trash slot idle lanes write

    # -------------------------------------------------------------- execution of the scheduled program on integers
    def run(self, inputs):
        low = self.low
        slots = {}
        for v in low.fv:
            if v.kind == "const" and v.id in self.live:
                slots[v.slot] = v.args
        for v, x in zip(low.inputs, inputs):
            slots[v.slot] = x % P
        for kind, take in self.rounds:
            res = []
            for v in take:
                if v.kind == "mul":
                    t, add = v.args
                    acc = (sum(slots[x.slot] * slots[y.slot] for x, y in t) + (slots[add.slot] if add is not None else 0)) % P
                elif v.kind == "lin":
                    acc = sum(slots[s.slot] * c for s, c in v.args) % P
                else:
                    acc = pow(slots[v.args.slot], P - 2, P)
                res.append((v.slot, acc))
                if v.twin is not None:
                    res.append((v.twin.slot, -acc % P))
            for s, a in res:                     # a round reads everything before it writes anything (one wave, lockstep)
                slots[s] = a
        return [slots[o.slot] for o in low.outputs]

    # -------------------------------------------------------------- the table
    def encode(self):
        """-> (kinds: one int per round + K_END, rows: per round nr x 8 dwords, consts: list of Fq integers, inputs / outputs: slots)
        row layout (eight dwords per lane per round; slot numbers, 16 bits each):
          mul:  dw0..5 = x_i | y_i << 16 (six products), dw6 = addend | dst << 16, dw7 = twin | kind << 16 (every kind)
          lin:  dw0..3 = s_2i | s_(2i+1) << 16 (eight sources), dw4..5 = eight int8 coefficients, dw6 = _ | dst << 16, dw7 = twin
          inv:  dw0 = src, dw6 = _ | dst << 16, dw7 = twin
        Unused products / sources name the ZERO slot; lanes without work and results without a twin write the TRASH slot."""
        zero = self.low.zero.slot
        one = self.low.const(1).slot
        assert one is not None
        # lanes without work and results without a twin write a TRASH slot; N_TRASH of them, lane r takes number r mod N_TRASH (sixteen lanes
        # storing to ONE address are a sixteen-way conflict in the LDS: CVM_TRASH, measured in profiles/r05_latency_ab.txt)
        trash0 = self.n_slots
        kinds, rows = [], []
        for kind, take in self.rounds:
            kinds.append(kind)
            row = []
            for r in range(self.nr):
                dw = [0] * 8
                v = take[r] if r < len(take) else None
                trash = trash0 + r % N_TRASH
                dst = v.slot if v is not None else trash
                twin = v.twin.slot if v is not None and v.twin is not None else trash
                dw[7] = twin | kind << 16
                if FAMILY[kind] == "m":
                    t, add = v.args if v is not None else ([], None)
                    for i in range(6):
                        x, y = (t[i][0].slot, t[i][1].slot) if i < len(t) else (zero, zero)
                        dw[i] = x | y << 16
                    dw[6] = (add.slot if add is not None else zero) | dst << 16
                elif FAMILY[kind] == "l":
                    ss, co = [zero] * 8, [0] * 8
                    for i, (s, c) in enumerate(v.args if v is not None else []):
                        ss[i], co[i] = s.slot, c
                        assert -128 <= c <= 127
                    for i in range(4):
                        dw[i] = ss[2 * i] | ss[2 * i + 1] << 16
                    dw[4] = sum((co[j] & 0xFF) << (8 * j) for j in range(4))
                    dw[5] = sum((co[4 + j] & 0xFF) << (8 * j) for j in range(4))
                    dw[6] = dst << 16
                else:
                    # the kernel writes the zero-divisor status under the mask of its output stores (role < 12: cvm_kernel.epilogue);
                    # an inversion on a higher lane would lose its flag
                    assert v is None or r < 12, "inv operation scheduled on a lane that cannot report a zero divisor"
                    dw[0] = v.args.slot if v is not None else one
                    dw[6] = dst << 16
                row.append(dw)
            rows.append(row)
        kinds.append(K_END)
        consts = [None] * self.n_const
        for v in self.low.fv:
            if v.kind == "const" and v.id in self.live:
                consts[v.slot] = v.args
        return {"kinds": kinds, "rows": rows, "consts": consts, "inputs": [(v.slot,) + tuple(v.args[1:]) for v in self.low.inputs],
                "outputs": [v.slot for v in self.low.outputs], "n_slots": self.n_slots + N_TRASH, "trash": trash0, "nr": self.nr}

    def stats(self):
        from collections import Counter
        c = Counter(KIND_NAME[k] for k, _ in self.rounds)
        instr = sum(COST[k] for k, _ in self.rounds)
        fill = sum(len(t) for _, t in self.rounds) / (self.nr * len(self.rounds))
        return {"rounds": len(self.rounds), "by_kind": dict(c), "instr_est": instr, "fill": round(fill, 3), "slots": self.n_slots + N_TRASH,
                "consts": self.n_const, "ops": len(self.ops)}


# ------------------------------------------------------------------ the programs
def _graph(**kw):
    g = Graph(**kw)
    g.const((0, 0))
    g.const((1, 0))
    return g


def build_pairing(**kw):
    """pairing(p, q) (src/pairing.rs:20-22): the six Fq2 coefficients in MyFq12 order"""
    g = _graph(**kw)
    (px, py), Q = g.g1_point(), g.g2_point()
    g.outputs = g.final_exp(g.miller_loop(px, py, Q))
    return g


def build_miller(run_ahead=None, **kw):
    """miller_loop_native(q, p) (miller_loop_native.rs:320-322), the exact value"""
    g = _graph(run_ahead=run_ahead, **kw)
    (px, py), Q = g.g1_point(), g.g2_point()
    g.outputs = g.miller_loop(px, py, Q, exact=True)
    return g


def build_miller_u(**kw):
    """The Miller loop of pairing() WITHOUT the scale: miller_loop_native's value up to the Fq2 factor that the easy part of the final
    exponentiation kills -- what `pairing` = final_exp_native(miller_loop_native(..)) needs of it (src/pairing.rs:20-22).  The first half
    of the two-launch form of pairing for mid-size batches: 136 slots, so that EIGHT waves of it fit a CU (two per SIMD: one wave's
    operand fetch runs under the other's arithmetic), where the whole pairing program holds 271 slots and four."""
    g = _graph(**kw)
    (px, py), Q = g.g1_point(), g.g2_point()
    g.outputs = g.miller_loop(px, py, Q)
    return g


def build_multi_miller_u(k, **kw):
    """multi_miller_loop_native over k pairs WITHOUT the line scale (as build_miller_u): the first launch of the mid-size form of the k-pair
    products, whose final exponentiation then runs as the six pieces"""
    g = _graph(**kw)
    pairs = [(g.g1_point(j), g.g2_point(j)) for j in range(k)]
    g.outputs = g.multi_miller_loop(pairs, exact=False)
    return g


def build_fexp_piece(piece, **kw):
    """the launches of final_exp_native for mid-size batches: "easy" f_in -> m; "powx" f_in -> f_in^x (cyclotomic input); "ych1" (g1, g2, f_in) =
    (m^x, m^(x^2), m^(x^3)) -> T1; "ych2" (g1, f_in) = (m, T1) -> final_exp_native's value"""
    g = _graph(**kw)
    if piece == "easy":
        g.outputs = g.easy_part(g.fq12_input())
    elif piece == "powx":
        g.outputs = g.pow_x(g.fq12_input())
    elif piece == "ych1":
        g.outputs = g.hard_tail_1(g.fq12_input(ARR_G1), g.fq12_input(ARR_G2), g.fq12_input(), serial=True)
    else:
        assert piece == "ych2"
        g.outputs = g.hard_tail_2(g.fq12_input(ARR_G1), g.fq12_input())
    return g


def build_final_exp(**kw):
    """final_exp_native(f) (final_exp_native.rs:209-213)"""
    g = _graph(**kw)
    g.outputs = g.final_exp(g.fq12_input())
    return g


def build_multi(k, final_exp=True, run_ahead=None, pow_window=None, **kw):
    """multi_miller_loop_native over k pairs (miller_loop_native.rs:324-326), then final_exp_native (the Groth16-style product of
    pairings, final_exp_native.rs:245-263) or -- final_exp=False -- the exact Miller value"""
    full = kw.pop("full", False)            # "fexp": the sixty-four-lane formulations in the final exponentiation only (four pairs: the Miller loop's rounds are full of lanes as they are)
    g = _graph(run_ahead=run_ahead, pow_window=pow_window, full=full is True, **kw)
    pairs = [(g.g1_point(j), g.g2_point(j)) for j in range(k)]
    f = g.multi_miller_loop(pairs, exact=not final_exp)
    g.full = bool(full)
    g.outputs = g.final_exp(f) if final_exp else f
    return g


def build_synth(kind, count):
    """DIAGNOSTIC program (tools/exp/lat_variant.sh, CVM_SYNTH): `count` rounds in which all sixteen lanes do one operation of
    `kind` (m2 / m6 / l4 / l8) on the previous round's values -- the per-round cost of the interpreter, kind by kind"""
    g = _graph()
    (px, py), (qx, qy) = g.g1_point(), g.g2_point()
    v = [g.mul((qx, qy)), g.mul((qy, qy)), g.mul((qx, qx)), g.lin((qx, ID), (qy, mxi())), g.lin((qx, mxi()), (qy, ID)), g.lin((qx, CONJ), (qy, ID)),
         g.lin((qx, mk(2)), (qy, CONJ)), g.lin((qx, mk(3)), (qy, NEG))]
    for _ in range(count):
        nv = []
        for j in range(8):
            a = [v[(j + i) % 8] for i in range(6)]
            if kind == "m6":
                nv.append(g.mul((a[0], a[1]), (a[2], a[3]), (a[4], a[5])))
            elif kind == "m2":
                nv.append(g.mul((a[0], a[1])))
            elif kind == "l4":
                nv.append(g.lin((a[0], (1, 2, -1, 1)), (a[1], (2, -1, 1, 3))))
            else:
                nv.append(g.lin((a[0], (1, 2, -1, 1)), (a[1], (2, -1, 1, 3)), (a[2], (1, 1, -2, 1)), (a[3], (-1, 2, 1, 1))))
        v = nv
    g.outputs = v[:6]
    return g


if __name__ == "__main__":
    for name, g, nr in (("pairing", build_pairing(), 16), ("pairing, 32 lanes", build_pairing(wide=True, pow_window=4), 32),
                        ("4-pair product", build_multi(4, True, pow_window=2), 16), ("4-pair product, 32 lanes", build_multi(4, True, pow_window=4, wide=True), 32)):
        print(name, Program(Lowered(g), nr=nr).stats())
