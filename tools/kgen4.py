#!/usr/bin/env python3
"""kgen4.py -- L1 field routines of the kernels: carry-free BALANCED signed limbs, radix 2^29, nine limbs.

gfx950 issues one VALU instruction per 4 cycles for the single wave a SIMD can hold at this register / LDS footprint,
whatever the instruction (tools/valu_calib.hip, profiles/valu_calib_r01.txt), so a pairing costs its dynamic instruction
count; 69 % of the round-1 kernels' instructions were the limb products.  The representation is chosen to minimise those:

  * an Fq element is NL = 9 signed 32-bit limbs in radix 2^29 (value = sum l_i 2^(29 i)), Montgomery form with
    R' = 2^261: 81 limb products per Fq product instead of 100 (ten 27-bit limbs, round 1) or 64 + carry chains
    (eight 32-bit limbs).  A limb product is ONE instruction, v_mad_i64_i32 acc64 += a_i * b_j;
  * limbs are BALANCED: normalised limbs lie in [-2^28, 2^28) (the top limb is signed and carries the representative),
    so |a_i b_j| <= 2^56 and a column of 54 products plus the 9 reduction products (the three-term fused multiply, six
    Fq products per pass) stays below 2^63: 63 * 2^56 < 2^62; sums of two normalised values may enter a two-product pass
    unnormalised (18 * 4 * 2^56 + 9 * 2^56 < 2^63).  Unsigned 29-bit limbs would overflow the signed accumulator;
  * Montgomery reduction is fused column-wise (FIPS): m_k = balanced((lo(S) n0') mod 2^29), S += m_k p_0, S >>= 29; result
    limbs are extracted balanced (v_bfe_i32) and the extracted digit is subtracted from the accumulator before the shift;
  * R'/p = 2^7.4 only (round 1: 2^16.4): a reduction output is sum(a_i b_i)/R' +- p/2, so values must be kept small:
    `redn` (normalise + subtract round(top limb * 2^232 / p) * p, one 64-bit carry chain) brings a value back to
    (-0.51 p, 0.51 p); tools/kgen4_prog.py tracks value bounds and places it;
  * x(9+u) of a normalised value would overflow int32 limb-wise (10 * 2^28 > 2^31): linear combinations with large
    coefficients run on a 64-bit accumulator chain (`lincomb`: mads by inline constants, balanced digit extraction),
    which normalises -- and optionally reduces -- in the same pass.

Blocks: A = v[0:17] (c0 = v0..8, c1 = v9..17), B = v[18:35]; temporaries v[36:75].
At the kernel boundary values are converted from/to ark's 4 x u64 Montgomery (R = 2^256) form.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from asmcore import Emitter, Pool, P_INT  # noqa: E402

NL = 9
LB = 29
MASK = (1 << LB) - 1
HALF = 1 << (LB - 1)
TOP_W = LB * (NL - 1)                                   # weight of the top limb: 2^232
RP = 1 << (NL * LB)                                     # R' = 2^261
N0P = (-pow(P_INT, -1, 1 << LB)) % (1 << LB)
REDN_SHIFT = 52                                         # q = round(top limb * REDN_C / 2^52) ~ top limb * 2^232 / p
REDN_C = (1 << (REDN_SHIFT + TOP_W)) // P_INT           # < 2^31: fits a signed multiply-high
assert REDN_C < (1 << 31)

A0, B0 = 0, 18
TMP_FIRST, TMP_LAST = 36, 75
CYC_WIDE_M = bool(int(os.environ.get("KGEN_CYC_WIDE_M", "1")))                     # A/B switch: 32-bit Montgomery digits in the cyclotomic squaring's products
DBL_LAZY_Y3 = bool(int(os.environ.get("KGEN_DBL_LAZY_Y3", "1")))                   # A/B switch: Y3 of the doubling step with one reduction per component
ADD_INJECT = bool(int(os.environ.get("KGEN_ADD_INJECT", "1")))                     # A/B switch: theta / mu of the addition step as pass outputs
MUL3_KEEP_DY = bool(int(os.environ.get("KGEN_MUL3_KEEP_DY", "1")))                 # A/B switch: line-side Karatsuba differences of the sparse multiplications kept across passes
MUL6_KEEP_DIFFS = bool(int(os.environ.get("KGEN_MUL6_KEEP_DIFFS", "1")))      # A/B switch (tools/exp/build_variant.sh)
# A/B switch: the balanced digit of a column sum is taken off by ROUNDING -- (S - d) >> 29 == (S + 2^28) >> 29 -- i.e. a 64-bit add of a
# scalar constant (v_lshl_add_u64) instead of the multiply-add S += d * (-1): as many instructions, 126 k multiply-adds per pairing less.
# The constant sits in an SGPR pair (S_HALF; the split-loop experiment needs those registers itself).
DIGIT_ADD = bool(int(os.environ.get("KGEN_DIGIT_ADD", "1"))) and not bool(int(os.environ.get("KGEN_FISSION", "0")))
S_HALF = 52         # s[52:53] = 2^28 (kernels with another scalar register map: L1v4.half)
V_LDS = 76          # v76, v77: LDS byte address of the lane's 16-byte chunks (+0, +64 KiB) ; v78: of its 8-byte tail chunks
V_LTAIL = 78
V_GOFF = 79         # byte offset of the lane's 16-byte chunks inside a global scratch slot: wave * 4608 + lane * 16 (V_GOFF8 = 246: ... + lane * 8, the tail)
V_GOFF8 = 246
V_IDX8 = 80
V_IDX = 81
V_TID = 82
V_FLAG = 83
HOME0 = 84          # homes: v[84:245] = 9 x 18
N_HOME = 9
N_AGPR_SLOTS = 14   # a[0:251]
N_LDS_SLOTS = 8     # 8 x 72 B x 256 lanes = 144 KiB
SLOT_DW = 18
SLOT_BYTES = 4 * SLOT_DW
S_P = 36            # s36..s44: modulus limbs (balanced, radix 2^29)
S_N0 = 46
S_REDN = 47         # REDN_C
S_M30 = 45          # the constant -30 (not an inline constant; used by the cyclotomic recombination chains)
S_RET1 = "s[54:55]"
S_RET2 = "s[56:57]"
S_RET3 = "s[58:59]"


def bal_limbs(x):
    """Signed integer -> NL balanced limbs (limbs 0..NL-2 in [-2^28, 2^28), the top limb takes the rest)."""
    out = []
    for _ in range(NL - 1):
        d = x & MASK
        if d >= HALF:
            d -= 1 << LB
        out.append(d)
        x = (x - d) >> LB
    out.append(x)
    return out


def unsigned_limbs(x):
    """Canonical non-negative integer -> NL limbs in [0, 2^29) (top limb: the rest)."""
    return [(x >> (LB * i)) & MASK for i in range(NL - 1)] + [x >> (LB * (NL - 1))]


def from_limbs(l):
    return sum(int(v) << (LB * i) for i, v in enumerate(l))


def mont4(x):
    return x * RP % P_INT


def hx(v):
    """32-bit two's-complement literal of a signed limb."""
    return "0x%x" % (v & 0xFFFFFFFF)


P_L = bal_limbs(P_INT)
P_U = unsigned_limbs(P_INT)


class L1v4:
    def __init__(self, e):
        self.e = e
        self.pool = Pool(TMP_FIRST, TMP_LAST)
        self.p = [f"s{S_P + i}" for i in range(NL)]
        self.n0 = f"s{S_N0}"
        self.half = f"s[{S_HALF}:{S_HALF + 1}]" if DIGIT_ADD else None

    # ------------------------------------------------------------------ 64-bit accumulator helpers
    def _acc(self):
        a = self.pool.alloc_pair()
        return a, f"v[{a}:{a + 1}]"

    def _mad(self, acc, P, x, y, first):
        """acc64 (+)= x * y ; x, y: VGPR numbers (int) or operand text (SGPR name / inline constant as str)"""
        X = f"v{x}" if isinstance(x, int) else x
        Y = f"v{y}" if isinstance(y, int) else y
        self.e.emit(f"v_mad_i64_i32 {P}, vcc, {X}, {Y}, {0 if first else P}", w=["vcc"], vw=[acc, acc + 1])

    @staticmethod
    def _coef(c):
        """operand text of a small integer coefficient: inline constant (-16..64) or the SGPR that holds it"""
        if isinstance(c, tuple):            # ("v", register number): a per-lane coefficient (the lane-cooperative kernels' LIN rounds)
            return f"v{c[1]}"
        if -16 <= c <= 64:
            return str(c)
        return {-30: f"s{S_M30}"}[c]

    def _digit(self, acc, P, dst):
        """dst <- balanced low digit of the accumulator; accumulator <- (accumulator - digit) >> 29."""
        self.e.emit(f"v_bfe_i32 v{dst}, v{acc}, 0, {LB}", vw=[dst])
        if self.half is not None:
            self.e.emit(f"v_lshl_add_u64 {P}, {P}, 0, {self.half}", vw=[acc, acc + 1])
        else:
            self.e.emit(f"v_mad_i64_i32 {P}, vcc, v{dst}, -1, {P}", w=["vcc"], vw=[acc, acc + 1])
        self.e.emit(f"v_ashrrev_i64 {P}, {LB}, {P}", vw=[acc, acc + 1])

    # ------------------------------------------------------------------ fused Montgomery column pass
    def fips(self, prods, out, fillers=(), gap=8, balanced=True, wide_m=False, inject=()):
        """out[0..NL-1] <- (sum over (a, b) in prods of a*b) / R' mod p, balanced limbs, value in sum/R' +- p/2.
        a, b: lists of NL VGPR numbers.  Result limb j is written after column j + NL, when limb j of every operand is
        dead, so `out` may be one of the first operands a (in place) but must not overlap a second operand b.

        fillers: independent instructions [(text, vw, earliest column, latest column)] dropped into the multiply runs
        (work that has to be done anyway -- negations, copies -- and whose operands die / are born inside the pass).
        balanced=False: result limbs in [0, 2^29) instead (one instruction less per limb) -- for results that only feed the
        64-bit linear-combination chains, never a product.
        inject: [(limb list, inline-constant coefficient)]: coef * value is ADDED to the result (the limbs enter the upper half of
        the double-width sum, one multiply-add each: out = sum / R' + coef * value +- p/2), which comes out normalised like any
        pass result -- a subtraction / addition behind a product without a separate limb-wise pass and carry pass.
        wide_m: the Montgomery digits m_k = lo32(S n0') are used as the signed 32-bit values they are (n0' = -1/p mod 2^32) instead
        of being cut back to balanced 29-bit digits: any m = S n0' (mod 2^29) clears the low 29 bits of the column, so the bfe is
        dropped (nine instructions per pass).  The price: |m_k| < 2^31, i.e. the nine reduction products of a column may reach
        72 units of 2^56 (instead of 9) and the result is sum / R' +- 4 p (instead of +- p/2) -- only for passes whose own
        products stay below ~50 units per column and whose results go straight into a REDUCING chain."""
        acc, P = self._acc()
        m = [self.pool.alloc() for _ in range(NL)]
        first = True
        todo = list(fillers)
        run = 0
        col = 0

        def fill(force_before=None):
            nonlocal run
            for j, (text, vw, lo, hi) in enumerate(todo):
                if (force_before is None and lo <= col) or (force_before is not None and hi <= force_before):
                    self.e.emit(text, vw=vw)
                    del todo[j]
                    run = 0
                    return True
            return False

        def mad(x, y):
            nonlocal first, run
            self._mad(acc, P, x, y, first)
            first = False
            run += 1
            if run >= gap:
                fill()

        for k in range(2 * NL - 1):
            col = k
            while fill(force_before=k):
                pass
            lo_i, hi_i = max(0, k - (NL - 1)), min(NL - 1, k)
            for (a, b) in prods:
                for i in range(lo_i, hi_i + 1):
                    mad(a[i], b[k - i])
            if k < NL:
                for i in range(k):
                    mad(m[i], self.p[k - i])
                self.e.emit(f"v_mul_lo_u32 v{m[k]}, v{acc}, {self.n0}", vw=[m[k]])
                if not wide_m:
                    self.e.emit(f"v_bfe_i32 v{m[k]}, v{m[k]}, 0, {LB}", vw=[m[k]])
                run = 0
                mad(m[k], self.p[0])
                self.e.emit(f"v_ashrrev_i64 {P}, {LB}, {P}", vw=[acc, acc + 1])
            else:
                for i in range(k - (NL - 1), NL):
                    mad(m[i], self.p[k - i])
                for vec, coef in inject:
                    mad(vec[k - NL], self._coef(coef))
                if balanced:
                    self._digit(acc, P, out[k - NL])
                else:
                    self.e.emit(f"v_and_b32_e32 v{out[k - NL]}, 0x{MASK:x}, v{acc}", vw=[out[k - NL]])
                    self.e.emit(f"v_ashrrev_i64 {P}, {LB}, {P}", vw=[acc, acc + 1])
                run = 0
        for vec, coef in inject:
            mad(vec[NL - 1], self._coef(coef))
        self.e.emit(f"v_mov_b32_e32 v{out[NL - 1]}, v{acc}", vw=[out[NL - 1]])
        for (text, vw, lo, hi) in todo:
            self.e.emit(text, vw=vw)
        self.pool.free(acc, acc + 1, *m)

    def fips_sq(self, a, out, wide_m=False):
        """out <- a^2 / R' mod p for ONE Fq value (normalised limbs): the symmetric products once -- column k takes a_i (2 a_j) for
        i < j, i + j = k and a_(k/2)^2: 45 limb products instead of 81.  Same column sweep, reduction and result rules as fips
        (in place over a is fine).  Used by the Fermat inversion (254 squarings per pairing)."""
        acc, P = self._acc()
        m = [self.pool.alloc() for _ in range(NL)]
        d = [self.pool.alloc() for _ in range(NL)]
        for i in range(NL):
            self.e.emit(f"v_lshlrev_b32_e32 v{d[i]}, 1, v{a[i]}", vw=[d[i]])
        first = True
        for k in range(2 * NL - 1):
            for i in range(max(0, k - (NL - 1)), (k + 1) // 2):          # i < k - i
                self._mad(acc, P, a[i], d[k - i], first)
                first = False
            if k % 2 == 0:
                self._mad(acc, P, a[k // 2], a[k // 2], first)
                first = False
            if k < NL:
                for i in range(k):
                    self._mad(acc, P, m[i], self.p[k - i], False)
                self.e.emit(f"v_mul_lo_u32 v{m[k]}, v{acc}, {self.n0}", vw=[m[k]])
                if not wide_m:
                    self.e.emit(f"v_bfe_i32 v{m[k]}, v{m[k]}, 0, {LB}", vw=[m[k]])
                self._mad(acc, P, m[k], self.p[0], False)
                self.e.emit(f"v_ashrrev_i64 {P}, {LB}, {P}", vw=[acc, acc + 1])
            else:
                for i in range(k - (NL - 1), NL):
                    self._mad(acc, P, m[i], self.p[k - i], False)
                self._digit(acc, P, out[k - NL])
        self.e.emit(f"v_mov_b32_e32 v{out[NL - 1]}, v{acc}", vw=[out[NL - 1]])
        self.pool.free(acc, acc + 1, *m)
        self.pool.free(*d)

    def kfips(self, kterms, sterms, out_re, out_im, inject=None):
        """(out_re, out_im) <- both components of the sum of the Fq2 products x y over kterms + sterms, divided by R' (one
        Montgomery reduction per component).  x = (x0, x1), y = (y0, y1): limb lists.
        kterms use KARATSUBA per limb pair: per column
            U = sum x0[i] y0[j],   W = sum x1[i] y1[j]          (fresh 64-bit accumulators, each computed ONCE)
            re += U - W,           im += sum (x1 - x0)[i] (y0 - y1)[j] + U + W
        three multiply-adds per limb pair instead of four; the 64-bit combinations (five instructions per column) are shared
        by all the Karatsuba products of the pass.  sterms go the schoolbook way, four multiply-adds per limb pair straight
        into the two accumulators (they cost no extra registers beyond one negated vector).  x limbs of up to two units
        (an unnormalised sum), y normalised.
        Why: the kernels are power-bound and a multiply-add is the dearest instruction (1.84 nJ per wave-instruction against
        1.0 for a 32-bit add / subtract, profiles/r02_energy_calib.txt); fewer issue slots come on top.
        Magnitudes per column (units of 2^56; x of mx units, y of one): |U|, |W| <= 9 k mx, re <= 18 (k + s) mx: below 2^63
        like every column of the schoolbook passes.  The im accumulator takes the schoolbook products, then the DIFFERENCE
        products (up to 4 mx units each, 36 k mx in all), then U and W, which cancel most of them: with two-unit operands
        it may pass 2^63 in between.  That is harmless -- 64-bit sums are exact mod 2^64 and the value that is finally
        shifted out is the true one, <= 18 (k + s) mx -- and it is what the simulator's accumulator check verifies.
        Result limb j is written after column j + NL (in place over a FIRST operand x is fine; never over a y).
        inject = (re limbs, im limbs): a normalised value that is ADDED to the result (its limbs enter the upper half of the sums)."""
        e = self.e
        nk = len(kterms)
        # a term may bring its difference vectors along -- (x, y, x1 - x0, y0 - y1) -- when the caller keeps them across passes
        # (r_mul6: every operand takes part in three passes); those are neither computed nor freed here
        own, dx, dy = [], [], []
        for k, term in enumerate(kterms):
            x, y = term[0], term[1]
            gx, gy = (term[2], term[3]) if len(term) == 4 else (None, None)
            if gx is None:
                gx = [self.pool.alloc() for _ in range(NL)]
                own.append(gx)
                self.limbwise("v_sub_u32_e32", gx, x[1], x[0])
            if gy is None:
                gy = [self.pool.alloc() for _ in range(NL)]
                own.append(gy)
                self.limbwise("v_sub_u32_e32", gy, y[0], y[1])
            dx.append(gx)
            dy.append(gy)
        kterms = [(t[0], t[1]) for t in kterms]
        nx = []
        for x, y in sterms:
            n_ = [self.pool.alloc() for _ in range(NL)]
            self._neg_into(n_, x[1])
            nx.append(n_)
        (a0, P0), (a1, P1), (u, PU), (w, PW) = self._acc(), self._acc(), self._acc(), self._acc()
        m0 = [self.pool.alloc() for _ in range(NL)]
        m1 = [self.pool.alloc() for _ in range(NL)]
        for c in range(2 * NL - 1):
            lo_i, hi_i = max(0, c - (NL - 1)), min(NL - 1, c)
            rng_i = range(lo_i, hi_i + 1)
            f0 = f1 = c == 0
            for (x, y), n_ in zip(sterms, nx):                      # schoolbook terms: straight into the accumulators
                for i in rng_i:
                    self._mad(a0, P0, x[0][i], y[0][c - i], f0)
                    self._mad(a0, P0, n_[i], y[1][c - i], False)
                    self._mad(a1, P1, x[0][i], y[1][c - i], f1)
                    self._mad(a1, P1, x[1][i], y[0][c - i], False)
                    f0 = f1 = False
            fu = fw = True
            for x, y in kterms:
                for i in rng_i:
                    self._mad(u, PU, x[0][i], y[0][c - i], fu)
                    fu = False
            for x, y in kterms:
                for i in rng_i:
                    self._mad(w, PW, x[1][i], y[1][c - i], fw)
                    fw = False
            for k in range(nk):
                for i in rng_i:
                    self._mad(a1, P1, dx[k][i], dy[k][c - i], f1)
                    f1 = False
            # re += U - W, im += U + W.  The borrow of the 64-bit subtraction travels through VCC (two wait states between
            # its two halves): the two additions into im fill them.
            # (the VOP3 encodings: between 8-byte instructions a 4-byte one would cost an alignment s_nop)
            if f0:
                e.emit(f"v_sub_co_u32_e64 v{a0}, vcc, v{u}, v{w}", w=["vcc"], vw=[a0])
            else:
                e.emit(f"v_lshl_add_u64 {P0}, {PU}, 0, {P0}", vw=[a0, a0 + 1])
                e.emit(f"v_sub_co_u32_e64 v{a0}, vcc, v{a0}, v{w}", w=["vcc"], vw=[a0])
            e.emit(f"v_lshl_add_u64 {P1}, {PU}, 0, {P1}", vw=[a1, a1 + 1])
            e.emit(f"v_lshl_add_u64 {P1}, {PW}, 0, {P1}", vw=[a1, a1 + 1])
            if f0:
                e.emit(f"v_subb_co_u32_e64 v{a0 + 1}, vcc, v{u + 1}, v{w + 1}, vcc", r=["vcc"], w=["vcc"], vw=[a0 + 1])
            else:
                e.emit(f"v_subb_co_u32_e64 v{a0 + 1}, vcc, v{a0 + 1}, v{w + 1}, vcc", r=["vcc"], w=["vcc"], vw=[a0 + 1])
            for acc, P, m, out in ((a0, P0, m0, out_re), (a1, P1, m1, out_im)):
                if c < NL:
                    for i in range(c):
                        self._mad(acc, P, m[i], self.p[c - i], False)
                    e.emit(f"v_mul_lo_u32 v{m[c]}, v{acc}, {self.n0}", vw=[m[c]])
                    e.emit(f"v_bfe_i32 v{m[c]}, v{m[c]}, 0, {LB}", vw=[m[c]])
                    self._mad(acc, P, m[c], self.p[0], False)
                    e.emit(f"v_ashrrev_i64 {P}, {LB}, {P}", vw=[acc, acc + 1])
                else:
                    for i in range(c - (NL - 1), NL):
                        self._mad(acc, P, m[i], self.p[c - i], False)
                    if inject is not None:
                        self._mad(acc, P, inject[0 if acc == a0 else 1][c - NL], "1", False)
                    self._digit(acc, P, out[c - NL])
        if inject is not None:
            self._mad(a0, P0, inject[0][NL - 1], "1", False)
            self._mad(a1, P1, inject[1][NL - 1], "1", False)
        e.emit(f"v_mov_b32_e32 v{out_re[NL - 1]}, v{a0}", vw=[out_re[NL - 1]])
        e.emit(f"v_mov_b32_e32 v{out_im[NL - 1]}, v{a1}", vw=[out_im[NL - 1]])
        self.pool.free(a0, a0 + 1, a1, a1 + 1, u, u + 1, w, w + 1, *m0, *m1)
        for v_ in own + nx:
            self.pool.free(*v_)

    def lincomb(self, outs, termss, reduce=False, hooks=None):
        """outs[j] <- sum of coef * vec over termss[j] (lists of (inline-constant coefficient, limb list)), NORMALISED
        (balanced limbs; top limb: the rest), on one 64-bit carry chain per output, the chains interleaved limb by limb so
        that outputs may overwrite inputs (limb i of every input is dead once limb i of every output is written).
        reduce: also subtract q p with q = round(top limb of the combination * 2^232 / p): result in (-0.51 p, 0.51 p).
        hooks: {limb index: [instruction text]} emitted before that limb's work (the lane-cooperative kernel waits there for
        operand limbs that were still in flight)."""
        n = len(outs)
        accs = [self._acc() for _ in range(n)]
        qs = []
        if reduce:
            # quotient estimate from the top limbs alone: the lower limbs shift the combination's top limb by at most
            # sum|coef| / 2 units of 2^232, i.e. the quotient by < 1e-5
            for terms, (acc, P) in zip(termss, accs):
                if len(terms) == 1 and terms[0][0] == 1 and not isinstance(terms[0][0], tuple):
                    q = self.pool.alloc()
                    self.e.emit(f"v_mov_b32_e32 v{q}, v{terms[0][1][NL - 1]}", vw=[q])
                else:
                    for t_, (coef, vec) in enumerate(terms):
                        self._mad(acc, P, vec[NL - 1], self._coef(coef), t_ == 0)
                    q = self.pool.alloc()
                    self.e.emit(f"v_mov_b32_e32 v{q}, v{acc}", vw=[q])
                self._quot(q)
                qs.append(q)
        for i in range(NL):
            for text in (hooks or {}).get(i, ()):
                self.e.raw(text)
            for j, terms in enumerate(termss):
                acc, P = accs[j]
                first = i == 0
                for coef, vec in terms:
                    self._mad(acc, P, vec[i], self._coef(coef), first)
                    first = False
                if reduce:
                    self._mad(acc, P, qs[j], self.p[i], False)
            for j in range(n):
                acc, P = accs[j]
                if i < NL - 1:
                    self._digit(acc, P, outs[j][i])
                else:
                    self.e.emit(f"v_mov_b32_e32 v{outs[j][i]}, v{acc}", vw=[outs[j][i]])
        for acc, _ in accs:
            self.pool.free(acc, acc + 1)
        self.pool.free(*qs)

    def _quot(self, q):
        """q <- -round(q * 2^232 / p)  (q holds a top limb on entry)"""
        self.e.emit(f"v_mul_hi_i32 v{q}, v{q}, s{S_REDN}", vw=[q])
        self.e.emit(f"v_add_u32_e32 v{q}, 0x{1 << (REDN_SHIFT - 33):x}, v{q}", vw=[q])
        self.e.emit(f"v_ashrrev_i32_e32 v{q}, {REDN_SHIFT - 32}, v{q}", vw=[q])
        self.e.emit(f"v_sub_u32_e32 v{q}, 0, v{q}", vw=[q])

    # ------------------------------------------------------------------ inversion: Bernstein-Yang divsteps ("safegcd")
    SG_ITERS = 21            # outer iterations of LB = 29 divsteps: 609 >= the 590 that 0 <= g < f < 2^256 need (delta = 1/2 variant)

    def fq_inv_safegcd(self, g, f, d, e, out, s_cnt, label, bad=None):
        """out <- g^-1 * R'^2 mod p  (normalised; i.e. the Montgomery form of 1 / x when g is the Montgomery form of x), 0 for g == 0
        mod p.  g: nine limbs, normalised, any representative; f, d, e: three more blocks of nine registers (scratch); out may be any of
        the four blocks.  bad (a register, optional) <- nonzero iff g == 0 mod p (the zero divisor).
        The constant-time divsteps of Bernstein-Yang in the signed-digit formulation of libsecp256k1's modinv32, with 29-bit limbs:
        per outer iteration 29 divsteps on the low words of f and g build a transition matrix (u v; q r), |entries| <= 2^29, with
            (f, g) <- (u f + v g, q f + r g) / 2^29   exactly,      (d, e) <- (u d + v e, q d + r e) / 2^29  mod p
        (the division mod p as one Montgomery digit per chain: the same m = lo(S n0') that fips uses); d x == f, e x == g (mod p)
        throughout (x the input), so f = +-1 at the end leaves d = +- 1/x.  Both pairs are updated in place, the two chains of a pair
        interleaved limb by limb (limb i - 1 of the results is written after limb i of both inputs has been read).  g is made
        canonical first (the proven step bound is for 0 <= g < f); |d|, |e| grow by at most p/2 per iteration (< 12 p at the end),
        the final multiplication by R'^3 mod p brings the result back to +-0.55 p.
        17 k instructions against the Fermat chain's 61 k (253 squarings + 60 products), most of them 32-bit logic.
        s_cnt: an SGPR for the loop counter; label: prefix of the routine's (unique) labels."""
        e_ = self.e
        alloc = self.pool.alloc
        # ---- g <- canonical representative in [0, p), balanced limbs
        self.lincomb([g], [[(1, g)]], reduce=True)                       # (-0.51 p, 0.51 p)
        self.unorm_limbs(g)                                              # floor carries: the top limb has the sign of the value
        msk, t = alloc(), alloc()
        e_.emit(f"v_ashrrev_i32_e32 v{msk}, 31, v{g[NL - 1]}", vw=[msk])
        for i in range(NL):
            e_.emit(f"v_and_b32_e32 v{t}, 0x{P_U[i]:x}, v{msk}", vw=[t])
            e_.emit(f"v_add_u32_e32 v{g[i]}, v{g[i]}, v{t}", vw=[g[i]])
        self.norm_limbs(g)
        self.pool.free(msk, t)
        for i in range(NL):
            e_.emit(f"v_mov_b32_e32 v{f[i]}, {hx(P_L[i])}", vw=[f[i]])
            e_.emit(f"v_mov_b32_e32 v{d[i]}, 0", vw=[d[i]])
            e_.emit(f"v_mov_b32_e32 v{e[i]}, {1 if i == 0 else 0}", vw=[e[i]])
        zeta, fw, gw, u, v, q, r, c1, c2, n1, x, y, z = (alloc() for _ in range(13))
        e_.emit(f"v_mov_b32_e32 v{zeta}, -1", vw=[zeta])
        e_.salu(f"s_mov_b32 {s_cnt}, {self.SG_ITERS}")
        e_.label(f"{label}_loop")
        # ---- 29 divsteps on the low words
        e_.emit(f"v_mov_b32_e32 v{fw}, v{f[0]}", vw=[fw])
        e_.emit(f"v_mov_b32_e32 v{gw}, v{g[0]}", vw=[gw])
        for reg, val in ((u, 1), (v, 0), (q, 0), (r, 1)):
            e_.emit(f"v_mov_b32_e32 v{reg}, {val}", vw=[reg])
        for _ in range(LB):
            e_.emit(f"v_ashrrev_i32_e32 v{c1}, 31, v{zeta}", vw=[c1])            # zeta < 0
            e_.emit(f"v_bfe_i32 v{c2}, v{gw}, 0, 1", vw=[c2])                    # -(g & 1)
            e_.emit(f"v_sub_u32_e32 v{n1}, 0, v{c1}", vw=[n1])
            e_.emit(f"v_xad_u32 v{x}, v{fw}, v{c1}, v{n1}", vw=[x])              # +-f, +-u, +-v
            e_.emit(f"v_xad_u32 v{y}, v{u}, v{c1}, v{n1}", vw=[y])
            e_.emit(f"v_xad_u32 v{z}, v{v}, v{c1}, v{n1}", vw=[z])
            e_.emit(f"v_and_b32_e32 v{x}, v{x}, v{c2}", vw=[x])
            e_.emit(f"v_add_u32_e32 v{gw}, v{gw}, v{x}", vw=[gw])
            e_.emit(f"v_and_b32_e32 v{y}, v{y}, v{c2}", vw=[y])
            e_.emit(f"v_add_u32_e32 v{q}, v{q}, v{y}", vw=[q])
            e_.emit(f"v_and_b32_e32 v{z}, v{z}, v{c2}", vw=[z])
            e_.emit(f"v_add_u32_e32 v{r}, v{r}, v{z}", vw=[r])
            e_.emit(f"v_and_b32_e32 v{c1}, v{c1}, v{c2}", vw=[c1])
            e_.emit(f"v_xad_u32 v{zeta}, v{zeta}, v{c1}, -1", vw=[zeta])
            e_.emit(f"v_and_b32_e32 v{x}, v{gw}, v{c1}", vw=[x])
            e_.emit(f"v_add_u32_e32 v{fw}, v{fw}, v{x}", vw=[fw])
            e_.emit(f"v_and_b32_e32 v{y}, v{q}, v{c1}", vw=[y])
            e_.emit(f"v_add_lshl_u32 v{u}, v{u}, v{y}, 1", vw=[u])
            e_.emit(f"v_and_b32_e32 v{z}, v{r}, v{c1}", vw=[z])
            e_.emit(f"v_add_lshl_u32 v{v}, v{v}, v{z}, 1", vw=[v])
            e_.emit(f"v_lshrrev_b32_e32 v{gw}, 1, v{gw}", vw=[gw])
        # ---- (f, g) and (d, e) <- matrix * pair / 2^29, in place
        (a0, P0), (a1, P1) = self._acc(), self._acc()
        m0, m1 = alloc(), alloc()
        for (lo, hi, mont) in ((f, g, False), (d, e, True)):
            self._mad(a0, P0, lo[0], u, True)
            self._mad(a0, P0, hi[0], v, False)
            self._mad(a1, P1, lo[0], q, True)
            self._mad(a1, P1, hi[0], r, False)
            if mont:
                for acc, P, m in ((a0, P0, m0), (a1, P1, m1)):
                    e_.emit(f"v_mul_lo_u32 v{m}, v{acc}, {self.n0}", vw=[m])
                    e_.emit(f"v_bfe_i32 v{m}, v{m}, 0, {LB}", vw=[m])
                    self._mad(acc, P, m, self.p[0], False)
            e_.emit(f"v_ashrrev_i64 {P0}, {LB}, {P0}", vw=[a0, a0 + 1])
            e_.emit(f"v_ashrrev_i64 {P1}, {LB}, {P1}", vw=[a1, a1 + 1])
            for i in range(1, NL):
                self._mad(a0, P0, lo[i], u, False)
                self._mad(a0, P0, hi[i], v, False)
                self._mad(a1, P1, lo[i], q, False)
                self._mad(a1, P1, hi[i], r, False)
                if mont:
                    self._mad(a0, P0, m0, self.p[i], False)
                    self._mad(a1, P1, m1, self.p[i], False)
                self._digit(a0, P0, lo[i - 1])
                self._digit(a1, P1, hi[i - 1])
            e_.emit(f"v_mov_b32_e32 v{lo[NL - 1]}, v{a0}", vw=[lo[NL - 1]])
            e_.emit(f"v_mov_b32_e32 v{hi[NL - 1]}, v{a1}", vw=[hi[NL - 1]])
        self.pool.free(a0, a0 + 1, a1, a1 + 1, m0, m1)
        e_.salu(f"s_sub_u32 {s_cnt}, {s_cnt}, 1")
        e_.salu(f"s_cmp_lg_u32 {s_cnt}, 0")
        e_.salu(f"s_cbranch_scc1 {label}_loop")
        # ---- f = +-1 (or +-p for the zero divisor): d <- sign(f) d ; out <- d * R'^3 / R'
        e_.emit(f"v_ashrrev_i32_e32 v{c1}, 31, v{f[0]}", vw=[c1])
        e_.emit(f"v_sub_u32_e32 v{n1}, 0, v{c1}", vw=[n1])
        for i in range(NL):
            e_.emit(f"v_xad_u32 v{d[i]}, v{d[i]}, v{c1}, v{n1}", vw=[d[i]])
        if bad is not None:
            e_.emit(f"v_mul_lo_u32 v{x}, v{f[0]}, v{f[0]}", vw=[x])
            e_.emit(f"v_add_u32_e32 v{x}, -1, v{x}", vw=[x])
            e_.emit(f"v_or3_b32 v{x}, v{x}, v{f[1]}, v{f[2]}", vw=[x])
            e_.emit(f"v_or3_b32 v{x}, v{x}, v{f[3]}, v{f[4]}", vw=[x])
            e_.emit(f"v_or3_b32 v{x}, v{x}, v{f[5]}, v{f[6]}", vw=[x])
            e_.emit(f"v_or3_b32 v{bad}, v{x}, v{f[7]}, v{f[8]}", vw=[bad])
        self.pool.free(zeta, fw, gw, u, v, q, r, c1, c2, n1, x, y, z)
        cst = bal_limbs(pow(RP, 3, P_INT))
        for i in range(NL):                                  # (f is dead: its block takes the constant)
            e_.emit(f"v_mov_b32_e32 v{f[i]}, {hx(cst[i])}", vw=[f[i]])
        self.fips([(d, f)], out)

    # ------------------------------------------------------------------ blocks
    @staticmethod
    def blk(base, half):
        return list(range(base + NL * half, base + NL * half + NL))

    def fq2(self, base):
        return (self.blk(base, 0), self.blk(base, 1))

    def limbwise(self, op, dst, a, b):
        for i in range(NL):
            self.e.emit(f"{op} v{dst[i]}, v{a[i]}, v{b[i]}", vw=[dst[i]])

    def _lw(self, op, d, a, b):
        for h in range(2):
            self.limbwise(op, d[h], a[h], b[h])

    def _neg_into(self, dst, src):
        for i in range(NL):
            self.e.emit(f"v_sub_u32_e32 v{dst[i]}, 0, v{src[i]}", vw=[dst[i]])

    def norm_limbs(self, a):
        """One balanced carry pass: limbs 0..NL-2 -> [-2^28, 2^28), excess pushed up (the top limb keeps the rest).
        Limbs must be below 2^31 - 2^28 in magnitude."""
        c = self.pool.alloc()
        for i in range(NL - 1):
            self.e.emit(f"v_add_u32_e32 v{c}, 0x{HALF:x}, v{a[i]}", vw=[c])
            self.e.emit(f"v_ashrrev_i32_e32 v{c}, {LB}, v{c}", vw=[c])
            self.e.emit(f"v_bfe_i32 v{a[i]}, v{a[i]}, 0, {LB}", vw=[a[i]])
            self.e.emit(f"v_add_u32_e32 v{a[i + 1]}, v{a[i + 1]}, v{c}", vw=[a[i + 1]])
        self.pool.free(c)

    def unorm_limbs(self, a):
        """Unsigned carry pass: limbs 0..NL-2 -> [0, 2^29) (floor carries; the top limb carries the sign of the value)."""
        c = self.pool.alloc()
        for i in range(NL - 1):
            self.e.emit(f"v_ashrrev_i32_e32 v{c}, {LB}, v{a[i]}", vw=[c])
            self.e.emit(f"v_and_b32_e32 v{a[i]}, 0x{MASK:x}, v{a[i]}", vw=[a[i]])
            self.e.emit(f"v_add_u32_e32 v{a[i + 1]}, v{a[i + 1]}, v{c}", vw=[a[i + 1]])
        self.pool.free(c)

    # ------------------------------------------------------------------ routines: A <- op(A, B)
    def _fq2_mul(self, x, y, o, balanced=True, wide_m=False):
        """o <- x * y (Fq2; x, y, o = (c0 limbs, c1 limbs)); o may be x itself (in place) or a disjoint block."""
        (x0, x1), (y0, y1), (o0, o1) = x, y, o
        n = [self.pool.alloc() for _ in range(NL)]
        self._neg_into(n, x1)
        self.fips([(x1, y0), (x0, y1)], o1, balanced=balanced, wide_m=wide_m)
        self.fips([(x0, y0), (n, y1)], o0, balanced=balanced, wide_m=wide_m)
        self.pool.free(*n)

    def r_mul(self):
        """(a0 + a1 u)(b0 + b1 u): two fused two-product passes, both in place over A."""
        self._fq2_mul(self.fq2(A0), self.fq2(B0), self.fq2(A0))

    # Round 4: the y-side operands of the three-term multiply are the LINE coefficients (block B, home blocks 1 and 3), which stay
    # put over several of the six passes of a sparse multiplication: their Karatsuba differences y0 - y1 live in 27 registers of the
    # routine's own scratch (the top half of home block 7 and home block 8) and are formed by the CALLER when such an operand changes
    # (L1v4.mul3_dy; Prog.mul3 tracks it) -- 45 subtractions per sparse multiplication instead of 162.
    MUL3_DY = {"B": list(range(HOME0 + 7 * SLOT_DW + NL, HOME0 + 8 * SLOT_DW)),
               1: list(range(HOME0 + 8 * SLOT_DW, HOME0 + 8 * SLOT_DW + NL)),
               3: list(range(HOME0 + 8 * SLOT_DW + NL, HOME0 + 9 * SLOT_DW))}

    def mul3_dy(self, which):
        """difference vector of the y operand `which` ("B", 1 or 3) of r_mul3 <- y.0 - y.1"""
        y = self.fq2(B0 if which == "B" else HOME0 + SLOT_DW * which)
        self.limbwise("v_sub_u32_e32", self.MUL3_DY[which], y[0], y[1])

    def r_mul3(self):
        """A <- A*B + H0*H1 + H2*H3 (Fq2 products, H_k = home block k), ONE reduction per output component: one dual column
        pass with three Karatsuba products (kfips), in place over A.  The other five operands survive.  Operand limbs below 3.9
        units (the limb-wise differences stay within int32); the column budget is the caller's check (Prog.mul3).  Scratch: the
        pool and home blocks 6, 7, 8 (MUL3_KEEP_DY: of which the 27 registers MUL3_DY hold the y-side differences on entry)."""
        blocks = [(A0, B0, "B"), (HOME0, HOME0 + SLOT_DW, 1), (HOME0 + 2 * SLOT_DW, HOME0 + 3 * SLOT_DW, 3)]
        extra = list(range(HOME0 + 6 * SLOT_DW, HOME0 + 9 * SLOT_DW))
        if MUL3_KEEP_DY:
            terms = [(self.fq2(x), self.fq2(y), None, self.MUL3_DY[w]) for x, y, w in blocks]
            extra = [r for r in extra if not any(r in v for v in self.MUL3_DY.values())]
        else:
            terms = [(self.fq2(x), self.fq2(y)) for x, y, _ in blocks]
        self.pool.free_regs += extra
        a = self.fq2(A0)
        self.kfips(terms, [], a[0], a[1])
        for r in extra:
            self.pool.free_regs.remove(r)

    def r_mul2a(self):
        """A <- A + H0*H1 + H2*H3 (Fq2 products): the dual column pass of r_mul3 with two Karatsuba products, the value in A ADDED to the result
        (its limbs enter the upper half of the sums: one multiply-add per limb, the result comes out normalised like any pass result).  The
        sparse multiplication by a line whose constant coefficient is ONE (the table lines of the fixed-G2 kernels).  Same operand rules, scratch
        and kept difference vectors (of home blocks 1 and 3) as r_mul3; block B is not touched."""
        blocks = [(HOME0, HOME0 + SLOT_DW, 1), (HOME0 + 2 * SLOT_DW, HOME0 + 3 * SLOT_DW, 3)]
        extra = list(range(HOME0 + 6 * SLOT_DW, HOME0 + 9 * SLOT_DW))
        if MUL3_KEEP_DY:
            terms = [(self.fq2(x), self.fq2(y), None, self.MUL3_DY[w]) for x, y, w in blocks]
            extra = [r for r in extra if not any(r in v for v in self.MUL3_DY.values())]
        else:
            terms = [(self.fq2(x), self.fq2(y)) for x, y, _ in blocks]
        self.pool.free_regs += extra
        a = self.fq2(A0)
        self.kfips(terms, [], a[0], a[1], inject=(a[0], a[1]))
        for r in extra:
            self.pool.free_regs.remove(r)

    def r_sqr(self):
        """(a0 + a1 u)^2 = (a0+a1)(a0-a1) + 2 a0 a1 u ; both passes write in place.  A normalised."""
        a0, a1 = self.fq2(A0)
        t = [self.pool.alloc() for _ in range(NL)]
        u = [self.pool.alloc() for _ in range(NL)]
        self.limbwise("v_add_u32_e32", t, a0, a1)
        self.limbwise("v_sub_u32_e32", u, a0, a1)
        for r in a0:
            self.e.emit(f"v_lshlrev_b32_e32 v{r}, 1, v{r}", vw=[r])     # a0 <- 2 a0 (t, u already hold what c0 needs)
        self.fips([(a1, a0)], a1)                                       # c1 = a1 * 2a0, in place over a1
        self.fips([(t, u)], a0)                                         # c0 over (dead) a0; t, u are temporaries
        self.pool.free(*t)
        self.pool.free(*u)

    def sqr4c_core(self, a, b, zc, zd, oa, ob, t, u, s, xi=False):
        """One Fq4 squaring of the Granger-Scott cyclotomic squaring WITH its recombination, on explicit register blocks:
        (a + b y)^2, y^2 = xi (a, b, zc, zd normalised Fq2 blocks):
            oa <- 3 (a^2 + xi b^2) - 2 zc        ob <- 3 (2 a b) + 2 zd      (xi variant: ob <- 3 xi (2 a b) + 2 zd)
        both NORMALISED AND REDUCED (values in (-0.51 p, 0.51 p)).  oa / ob may be zc / zd or a / b themselves (a, b are dead
        once the second product is formed; the chains write limb i of both outputs after reading limb i of every input).
        t = a b ; S = xi b + a (normalised on a 64-bit chain) ; P = (a + b) S ; a^2 + xi b^2 = P - t - xi t.
        t, u, s: three scratch blocks; the pool."""
        # (round 4: both products with 32-bit Montgomery digits -- wide_m: their columns hold 18 / 36 units of limb products, their
        # results, now sum / R' +- 4 p, only feed the reducing chains below, which take up to 8 V_CAP)
        self._fq2_mul(a, b, t, balanced=False, wide_m=CYC_WIDE_M)       # t = a b (a, b stay intact); t and P only feed the chains
        # S = xi b + a = (9 b0 - b1 + a0, 9 b1 + b0 + a1), normalised
        self.lincomb([s[0], s[1]], [[(9, b[0]), (-1, b[1]), (1, a[0])], [(9, b[1]), (1, b[0]), (1, a[1])]])
        self._lw("v_add_u32_e32", u, a, b)                              # u = a + b (two units)
        self._fq2_mul(u, s, u, balanced=False, wide_m=CYC_WIDE_M)       # P = u S, in place
        # oa <- 3 (P - t - xi t) - 2 zc = 3 P - 30 t0 + 3 t1 - 2 zc | 3 P1 - 30 t1 - 3 t0 - 2 zc1
        self.lincomb([oa[0], oa[1]], [[(3, u[0]), (-30, t[0]), (3, t[1]), (-2, zc[0])],
                                      [(3, u[1]), (-30, t[1]), (-3, t[0]), (-2, zc[1])]], reduce=True)
        if not xi:                                                      # ob <- 6 t + 2 zd
            self.lincomb([ob[0], ob[1]], [[(6, t[0]), (2, zd[0])], [(6, t[1]), (2, zd[1])]], reduce=True)
        else:                                                           # ob <- 6 xi t + 2 zd = (54 t0 - 6 t1, 54 t1 + 6 t0) + 2 zd
            self.lincomb([ob[0], ob[1]], [[(54, t[0]), (-6, t[1]), (2, zd[0])], [(54, t[1]), (6, t[0]), (2, zd[1])]], reduce=True)

    def r_sqr4c(self, xi=False):
        """sqr4c_core on the accumulator-machine blocks: a in block A, b in block B, zc in home block 3, zd in home block 4;
        A <- out_a, B <- out_b.  Scratch: home blocks 0..2, the pool."""
        a, b = self.fq2(A0), self.fq2(B0)
        H = lambda k: self.fq2(HOME0 + SLOT_DW * k)
        self.sqr4c_core(a, b, H(3), H(4), a, b, H(0), H(1), H(2), xi=xi)

    def cyc3(self):
        """F <- F^2 for F in the cyclotomic subgroup (Granger-Scott), F0..F5 RESIDENT in home blocks 3..8 -- three Fq4 squarings
        with their recombination, results in place.  A run of squarings (the x-powers have runs of up to eight) then moves F between
        its slots and the registers once per run instead of eighteen slot accesses per squaring.
        The three squarings read each other's inputs in a cycle: (F1, F4) are copied to blocks A, B first.  Scratch: home blocks
        0..2, blocks A, B, the pool."""
        H = lambda k: self.fq2(HOME0 + SLOT_DW * k)
        F = [H(3 + i) for i in range(6)]
        t, u, s = H(0), H(1), H(2)
        A, B = self.fq2(A0), self.fq2(B0)
        for dst, src in ((A, F[1]), (B, F[4])):
            for h in range(2):
                for i in range(NL):
                    self.e.emit(f"v_mov_b32_e32 v{dst[h][i]}, v{src[h][i]}", vw=[dst[h][i]])
        self.sqr4c_core(F[2], F[5], F[4], F[1], F[4], F[1], t, u, s, xi=True)      # F4' = 3 t4 - 2 F4 ; F1' = 3 xi t5 + 2 F1
        self.sqr4c_core(A, B, F[2], F[5], F[2], F[5], t, u, s)                     # F2' = 3 t2 - 2 F2 ; F5' = 3 t3 + 2 F5  (a, b = old F1, F4)
        self.sqr4c_core(F[0], F[3], F[0], F[3], F[0], F[3], t, u, s)               # F0' = 3 t0 - 2 F0 ; F3' = 3 t1 + 2 F3

    def r_sqr4cx(self):
        self.r_sqr4c(xi=True)

    # ------------------------------------------------------------------ fused Fq6 multiplication
    def _pass3(self, out, terms, imag):
        """out <- one component of x0 y0 + x1 y1 + x2 y2 (three Fq2 products, terms = [(x, y)] with x, y = (c0 limbs, c1 limbs)),
        ONE reduction: the real component sum(x.0 y.0 - x.1 y.1) or the imaginary one sum(x.0 y.1 + x.1 y.0)."""
        if imag:
            prods = []
            for x, y in terms:
                prods += [(x[0], y[1]), (x[1], y[0])]
            self.fips(prods, out)
            return
        negs, prods = [], []
        for x, y in terms:
            n = [self.pool.alloc() for _ in range(NL)]
            self._neg_into(n, x[1])
            negs.append(n)
            prods += [(x[0], y[0]), (n, y[1])]
        self.fips(prods, out)
        for n in negs:
            self.pool.free(*n)

    def r_mul6(self):
        """Fq6 multiplication in Fq2[v]/(v^3 - xi), schoolbook over Fq2 with LAZY REDUCTION: nine Fq2 products, each of the
        three output coefficients ONE dual column pass (kfips: both components, one Montgomery reduction each):
            c0 = a0 b0 + (xi a1) b2 + (xi a2) b1     c1 = a0 b1 + a1 b0 + (xi a2) b2     c2 = a0 b2 + a1 b1 + a2 b0
        All nine are Karatsuba products.  a: limbs of up to two units -- unnormalised sums are fine --, b normalised.  (Karatsuba over Fq6 -- six Fq2 products -- needs twelve
        reductions and three recombination chains: 4 % more instructions.)
        a = (a0, a1, a2) in home blocks 0..2, b in home blocks 3..5.  Results (reduction outputs: normalised, values per
        kgen4_prog.Prog._mul6_regs):   c0 -> home block 6,  c1 -> home block 2,  c2 -> block A.
        xi a2 is formed after the c2 pass (home 7), xi a1 after the c1 pass (home 6), so that every pass finds its registers
        (two accumulator pairs + two fresh ones, two quotient vectors, six difference vectors: 80) in the pool, block B, the
        home blocks that are dead at that point and, for the last two passes, four parked bookkeeping registers (94 / 80 / 80).  a1, a2, block B and home blocks 6, 7 are destroyed; a0 and b survive."""
        H = lambda k: self.fq2(HOME0 + SLOT_DW * k)
        blk = lambda r0: list(range(r0, r0 + SLOT_DW))
        a, b = [H(0), H(1), H(2)], [H(3), H(4), H(5)]
        xa1, xa2 = H(6), H(7)
        A = self.fq2(A0)
        xi = lambda src, dst: self.lincomb([dst[0], dst[1]], [[(9, src[0]), (-1, src[1])], [(9, src[1]), (1, src[0])]])      # normalised

        def with_regs(extra, fn):
            self.pool.free_regs += extra
            self.pool.free_regs.sort()
            fn()
            for r in extra:
                self.pool.free_regs.remove(r)

        if not MUL6_KEEP_DIFFS:
            with_regs(blk(B0) + blk(HOME0 + 6 * SLOT_DW) + blk(HOME0 + 7 * SLOT_DW),
                      lambda: self.kfips([(a[0], b[2]), (a[1], b[1]), (a[2], b[0])], [], A[0], A[1]))              # c2 -> block A
            xi(a[2], xa2)
            # the other two passes find 76 registers where three Karatsuba products need 80: the four bookkeeping registers (batch
            # index, thread id, flags) wait in the four spare AGPRs meanwhile
            park = [V_IDX8, V_IDX, V_TID, V_FLAG]
            for i, r in enumerate(park):
                self.e.emit(f"v_accvgpr_write_b32 a{SLOT_DW * N_AGPR_SLOTS + i}, v{r}")
            with_regs(blk(B0) + blk(HOME0 + 6 * SLOT_DW) + park,
                      lambda: self.kfips([(a[0], b[1]), (a[1], b[0]), (xa2, b[2])], [], a[2][0], a[2][1]))        # c1 -> home 2 (a2 is dead)
            xi(a[1], xa1)
            with_regs(blk(B0) + blk(HOME0 + SLOT_DW) + park,
                      lambda: self.kfips([(a[0], b[0]), (xa1, b[2]), (xa2, b[1])], [], xa1[0], xa1[1]))           # c0 in place over xi a1
            for i, r in enumerate(park):
                self.e.emit(f"v_accvgpr_read_b32 v{r}, a{SLOT_DW * N_AGPR_SLOTS + i}", vw=[r])
            return
        # Round 4: the Karatsuba DIFFERENCE vectors are kept across the three passes.  Every operand takes part in three products
        # (a_i with b_0, b_1, b_2 and the other way round), so forming x1 - x0 / y0 - y1 per pass computed each of them three
        # times: 162 subtractions per multiplication.  Here the six vectors live in block B and the pool for the whole routine (54
        # of their 58 registers; the accumulators and quotient digits of a pass take the remaining four plus home blocks 6, 7 in
        # the first pass, home block 6 / 1 and the four parked bookkeeping registers in the other two: 26 each) and only the two
        # that change -- xi a2, xi a1 replace a2, a1 -- are formed anew: 72 subtractions, -90 instructions per multiplication.
        keep = blk(B0) + list(range(TMP_FIRST, TMP_LAST + 1))          # 58 registers that no pass hands out otherwise
        saved = self.pool.free_regs
        assert sorted(saved) == list(range(TMP_FIRST, TMP_LAST + 1)), "mul6 expects the whole pool"
        dregs = keep[:6 * NL]
        rest = keep[6 * NL:]
        da = [dregs[NL * k: NL * k + NL] for k in range(3)]
        db = [dregs[NL * (3 + k): NL * (3 + k) + NL] for k in range(3)]
        for k in range(3):
            self.limbwise("v_sub_u32_e32", da[k], a[k][1], a[k][0])
            self.limbwise("v_sub_u32_e32", db[k], b[k][0], b[k][1])
        self.pool.free_regs = sorted(rest)

        def run_pass(extra, terms, out):
            self.pool.free_regs = sorted(rest + extra)
            self.kfips(terms, [], out[0], out[1])
            assert sorted(self.pool.free_regs) == sorted(rest + extra)
            self.pool.free_regs = sorted(rest)

        def xi_diff(src, dst, d):
            """dst <- xi src (normalised), d <- dst.1 - dst.0; the chain's two accumulator pairs come from the pass workspace"""
            self.pool.free_regs = sorted(rest + blk(HOME0 + 6 * SLOT_DW)) if dst is xa2 else sorted(rest + park)
            xi(src, dst)
            self.pool.free_regs = sorted(rest)
            self.limbwise("v_sub_u32_e32", d, dst[1], dst[0])

        park = [V_IDX8, V_IDX, V_TID, V_FLAG]
        run_pass(blk(HOME0 + 6 * SLOT_DW) + blk(HOME0 + 7 * SLOT_DW),
                 [(a[0], b[2], da[0], db[2]), (a[1], b[1], da[1], db[1]), (a[2], b[0], da[2], db[0])], A)                # c2 -> block A
        xi_diff(a[2], xa2, da[2])                                                                                   # xi a2 -> home 7 (home 6 lends the chain its registers)
        for i, r in enumerate(park):
            self.e.emit(f"v_accvgpr_write_b32 a{SLOT_DW * N_AGPR_SLOTS + i}, v{r}")
        run_pass(blk(HOME0 + 6 * SLOT_DW) + park,
                 [(a[0], b[1], da[0], db[1]), (a[1], b[0], da[1], db[0]), (xa2, b[2], da[2], db[2])], a[2])               # c1 -> home 2 (a2 is dead)
        xi_diff(a[1], xa1, da[1])                                                                                   # xi a1 -> home 6 (a1 itself is dead now)
        run_pass(blk(HOME0 + SLOT_DW) + park,
                 [(a[0], b[0], da[0], db[0]), (xa1, b[2], da[1], db[2]), (xa2, b[1], da[2], db[1])], xa1)                 # c0 in place over xi a1
        for i, r in enumerate(park):
            self.e.emit(f"v_accvgpr_read_b32 v{r}, a{SLOT_DW * N_AGPR_SLOTS + i}", vw=[r])
        self.pool.free_regs = saved


    # ------------------------------------------------------------------ fused G2 steps of the Miller loop
    # A point step is ~20 Fq2 operations on a dozen values; as separate accumulator-machine calls every operand crosses a
    # slot <-> block boundary (18 moves each way) and every call costs two taken branches.  Fused, everything stays in named
    # register blocks: X, Y, Z in home blocks 0..2, the evaluation point in block B (Px limbs in B.c0, Py limbs in B.c1).
    def _fq2_sqr(self, x, o):
        """o <- x^2 (x normalised); o may be x itself or a disjoint block"""
        (x0, x1), (o0, o1) = x, o
        t = [self.pool.alloc() for _ in range(NL)]
        u = [self.pool.alloc() for _ in range(NL)]
        d = [self.pool.alloc() for _ in range(NL)]
        self.limbwise("v_add_u32_e32", t, x0, x1)
        self.limbwise("v_sub_u32_e32", u, x0, x1)
        for i in range(NL):
            self.e.emit(f"v_lshlrev_b32_e32 v{d[i]}, 1, v{x0[i]}", vw=[d[i]])
        self.fips([(x1, d)], o1)                       # 2 x0 x1
        self.fips([(t, u)], o0)                        # (x0 + x1)(x0 - x1)
        self.pool.free(*t)
        self.pool.free(*u)
        self.pool.free(*d)

    def _fq2_mulfq(self, x, k, o):
        """o <- (x.0 k, x.1 k) with k in Fq (one limb list)"""
        self.fips([(x[0], k)], o[0])
        self.fips([(x[1], k)], o[1])

    def _pass2(self, out, terms, imag):
        """out <- one component of sum of x y over terms = [(x, y, sign)] (Fq2 products, sign = +-1), ONE reduction."""
        prods, negs = [], []

        def neg(v):
            n = [self.pool.alloc() for _ in range(NL)]
            self._neg_into(n, v)
            negs.append(n)
            return n
        for x, y, sg in terms:
            if imag:       # x0 y1 + x1 y0
                prods += [(x[0], y[1]), (x[1], y[0])] if sg > 0 else [(neg(x[0]), y[1]), (neg(x[1]), y[0])]
            else:          # x0 y0 - x1 y1
                prods += [(x[0], y[0]), (neg(x[1]), y[1])] if sg > 0 else [(neg(x[0]), y[0]), (x[1], y[1])]
        self.fips(prods, out)
        for n in negs:
            self.pool.free(*n)

    def _load_const(self, blk, c0, c1):
        w = bal_limbs(mont4(c0)) + bal_limbs(mont4(c1))
        for i in range(SLOT_DW):
            self.e.emit(f"v_mov_b32_e32 v{blk + i}, {hx(w[i])}", vw=[blk + i])

    THREE_B = (81 * pow(82, -1, P_INT) % P_INT, (-9 * pow(82, -1, P_INT)) % P_INT)      # 3 b' = 9 / (9 + u)

    def r_dblstep(self):
        """R = (X, Y, Z) <- 2 R on y^2 = x^3 + 3/xi (homogeneous projective) and the tangent line at the old R evaluated at P:
        L0 = xi Y^2 - 9 Z^2,  L3 = 2 Y Z Py,  L4 = -3 X^2 Px   (the reference's sparse_line_function_equal_native value,
        miller_loop_native.rs:30-44, times Z^2).  In: X, Y, Z normalised in home blocks 0, 1, 2; Px in B.c0, Py in B.c1.
        Out: X3 -> home 0, Y3 -> home 1, Z3 -> home 2 (all three reduced), L0 -> home 7 (limbs of two units), L3 -> home 4,
        L4 -> home 5 (normalised).  Scratch: home blocks 3, 6, 8, block A, the pool.

        The doubling formulas carry the curve constant as E = 3 b' Z^2 = 9 Z^2 / xi: a full Fq2 multiplication by a constant.  The
        new point is returned scaled by xi^2 instead (any representative of the projective point serves the next step, and the
        line of the NEXT step is homogeneous in it -- the tracked scale of the exact Miller value is formed from the Z actually
        used), which leaves only multiplications by xi and small integers, all on the 64-bit chains:
            B = Y^2, N = 9 Z^2, H = 2 Y Z, T = xi B - 3 N, S = xi B + 3 N
            X3 = 2 xi (X Y) T        Y3 = S^2 - 12 N^2        Z3 = 4 (xi B)(xi H)        L0 = xi B - N
        (= xi^2 times the textbook 2 X Y (B - F), (B + F)^2 - 12 E^2, 4 B H with F = 3 E): one Fq2 product and two reductions
        fewer than with E, -342 multiply-adds per step."""
        H = lambda k: self.fq2(HOME0 + SLOT_DW * k)
        X, Y, Z, Bq, L3, L4, C, Hh, E = [H(k) for k in range(9)]
        A = self.fq2(A0)
        Px, Py = self.blk(B0, 0), self.blk(B0, 1)
        self._fq2_sqr(Y, Bq)
        self._fq2_sqr(Z, C)
        self._lw("v_add_u32_e32", Hh, Y, Z)
        self.norm_limbs(Hh[0])                          # (the squaring forms x0 + x1: two units in would be four)
        self.norm_limbs(Hh[1])
        self._fq2_sqr(Hh, Hh)
        self._lw("v_sub_u32_e32", Hh, Hh, Bq)
        self._lw("v_sub_u32_e32", Hh, Hh, C)            # H = 2 Y Z (three units)
        self._fq2_mulfq(Hh, Py, L3)                                                             # L3 = H Py
        xi = lambda src, dst: self.lincomb([dst[0], dst[1]], [[(9, src[0]), (-1, src[1])], [(9, src[1]), (1, src[0])]])     # normalised
        xH, xB, N = E, A, C
        xi(Hh, xH)                                                                              # xi H
        xi(Bq, xB)                                                                              # xi B
        self.lincomb([N[0], N[1]], [[(9, C[0])], [(9, C[1])]])                                  # N = 9 Z^2, in place
        L0 = Hh                                                                                 # (H is dead)
        self._lw("v_sub_u32_e32", L0, xB, N)                                                    # L0 = xi B - N (two units)
        self._fq2_mul(xB, xH, Z)                                                                # (Z is dead) xi^2 B H
        self.lincomb([Z[0], Z[1]], [[(4, Z[0])], [(4, Z[1])]], reduce=True)                     # Z3 = 4 xi^2 B H
        Xq = xH                                                                                 # (xi H is dead)
        self._fq2_sqr(X, Xq)                                                                    # X^2
        for h in range(2):
            for r in Xq[h]:
                self.e.emit(f"v_lshl_add_u32 v{r}, v{r}, 1, v{r}", vw=[r])                      # 3 X^2
        nPx = [self.pool.alloc() for _ in range(NL)]
        self._neg_into(nPx, Px)
        self._fq2_mulfq(Xq, nPx, L4)                                                            # L4 = -3 X^2 Px
        self.pool.free(*nPx)
        T = Bq                                                                                  # (B is dead)
        for h in range(2):
            for i in range(NL):
                t = self.pool.alloc()
                self.e.emit(f"v_lshl_add_u32 v{t}, v{N[h][i]}, 1, v{N[h][i]}", vw=[t])          # 3 N
                self.e.emit(f"v_sub_u32_e32 v{T[h][i]}, v{xB[h][i]}, v{t}", vw=[T[h][i]])       # T = xi B - 3 N (four units)
                self.e.emit(f"v_add_u32_e32 v{xB[h][i]}, v{xB[h][i]}, v{t}", vw=[xB[h][i]])     # S = xi B + 3 N, in place (block A)
                self.pool.free(t)
        self._fq2_mul(X, Y, X)                                                                  # X Y, in place over X
        self._fq2_mul(X, T, X)                                                                  # X Y T (one unit x four units)
        self.lincomb([X[0], X[1]], [[(18, X[0]), (-2, X[1])], [(18, X[1]), (2, X[0])]], reduce=True)   # X3 = 2 xi X Y T
        self.norm_limbs(A[0])
        self.norm_limbs(A[1])
        if not DBL_LAZY_Y3:
            self._fq2_sqr(A, Y)                                                                 # (Y is dead) S^2
            self._fq2_sqr(N, N)
            self.lincomb([Y[0], Y[1]], [[(1, Y[0]), (-12, N[0])], [(1, Y[1]), (-12, N[1])]], reduce=True)   # Y3 = S^2 - 12 N^2
            return
        # Round 4: Y3 = S^2 - 12 N^2 = S S + N (-12 N) with ONE reduction per component (two two-product passes) instead of two
        # squarings (four reductions) and a combining chain: M = -12 N comes normalised off one chain (into the dead block of T),
        #   re = (S0 + S1)(S0 - S1) + (N0 + N1)(M0 - M1)        im = S1 (2 S0) + N1 (2 M0)
        # then the result (below 15 p for reduced inputs) goes through the reducing chain like every coordinate.
        S_ = A
        M = Bq                                                                                  # (T is dead)
        self.lincomb([M[0], M[1]], [[(-12, N[0])], [(-12, N[1])]])
        tN, uM = E                                                                              # (block 8 is dead: X^2 went into L4)
        self.limbwise("v_add_u32_e32", tN, N[0], N[1])
        self.limbwise("v_sub_u32_e32", uM, M[0], M[1])
        tS = [self.pool.alloc() for _ in range(NL)]
        uS = [self.pool.alloc() for _ in range(NL)]
        self.limbwise("v_add_u32_e32", tS, S_[0], S_[1])
        self.limbwise("v_sub_u32_e32", uS, S_[0], S_[1])
        self.fips([(tS, uS), (tN, uM)], Y[0])
        self.pool.free(*tS)
        self.pool.free(*uS)
        dS, dM = tN, uM                                                                         # (dead now)
        for i in range(NL):
            self.e.emit(f"v_lshlrev_b32_e32 v{dS[i]}, 1, v{S_[0][i]}", vw=[dS[i]])
            self.e.emit(f"v_lshlrev_b32_e32 v{dM[i]}, 1, v{M[0][i]}", vw=[dM[i]])
        self.fips([(S_[1], dS), (N[1], dM)], Y[1])
        self.lincomb([Y[0], Y[1]], [[(1, Y[0])], [(1, Y[1])]], reduce=True)

    def r_addstep(self):
        """R <- R + Q (mixed addition, Q = (x2, y2) affine) and the chord through the old R and Q evaluated at P:
        L2 = -mu Py,  L3 = theta Px,  L5 = X y2 - x2 Y   with theta = Y - y2 Z, mu = X - x2 Z   (the reference's
        sparse_line_function_unequal_native value, miller_loop_native.rs:10-28, times Z).  In: X, Y, Z in home blocks 0, 1, 2,
        x2, y2 in home blocks 3, 4 (all normalised); Px in B.c0, Py in B.c1.
        Out: X3 -> home 6, Y3 -> home 4, Z3 -> home 2;  L2 -> home 7,  L3 -> home 8,  L5 -> block A; all normalised.
        Scratch: home blocks 0, 3, 5, block B, the pool.
        theta^2 = C, mu^2 = D, E = mu D, F = Z C, G = X D, H = E + F - 2 G:  X3 = mu H, Y3 = theta (G - H) - E Y, Z3 = Z E."""
        H = lambda k: self.fq2(HOME0 + SLOT_DW * k)
        X, Y, Z, x2, y2, th, mu, L2, L3 = [H(k) for k in range(9)]
        A, Bk = self.fq2(A0), self.fq2(B0)
        Px, Py = self.blk(B0, 0), self.blk(B0, 1)
        if not ADD_INJECT:
            self._fq2_mul(y2, Z, th)
            self._lw("v_sub_u32_e32", th, Y, th)                           # theta = Y - y2 Z
            self._fq2_mul(x2, Z, mu)
            self._lw("v_sub_u32_e32", mu, X, mu)                           # mu = X - x2 Z
            for v in (th, mu):
                self.norm_limbs(v[0])
                self.norm_limbs(v[1])
        else:
            # Round 4: theta = Y - y2 Z, mu = X - x2 Z straight out of the product passes: the products of the NEGATED affine
            # coordinate with Z, the projective coordinate injected into the upper half of the sum (fips inject) -- the results
            # are pass outputs, i.e. normalised: no limb-wise subtraction and no carry pass (-150 instructions per step)
            n0 = [self.pool.alloc() for _ in range(NL)]
            n1 = [self.pool.alloc() for _ in range(NL)]
            for q, dst, W in ((y2, th, Y), (x2, mu, X)):
                self._neg_into(n0, q[0])
                self._neg_into(n1, q[1])
                self.fips([(n1, Z[0]), (n0, Z[1])], dst[1], inject=[(W[1], 1)])      # W1 - (q1 Z0 + q0 Z1)
                self.fips([(n0, Z[0]), (q[1], Z[1])], dst[0], inject=[(W[0], 1)])    # W0 - (q0 Z0 - q1 Z1)
            self.pool.free(*n0)
            self.pool.free(*n1)
        nPy = [self.pool.alloc() for _ in range(NL)]
        self._neg_into(nPy, Py)
        self._fq2_mulfq(mu, nPy, L2)                                    # L2 = -mu Py
        self.pool.free(*nPy)
        self._fq2_mulfq(th, Px, L3)                                     # L3 = theta Px
        t5 = [(X, y2, 1), (x2, Y, -1)]                                  # L5 = X y2 - x2 Y: one reduction per component
        self._pass2(A[1], t5, imag=True)
        self._pass2(A[0], t5, imag=False)
        Cc, D, E = x2, y2, Bk                                           # (the affine point and P are dead)
        self._fq2_sqr(th, Cc)
        self._fq2_sqr(mu, D)
        self._fq2_mul(mu, D, E)
        self._fq2_mul(Cc, Z, Cc)                                        # F = Z C, in place over C
        self._fq2_mul(D, X, D)                                          # G = X D, in place over D
        Hh = X                                                          # (X is dead)
        for h in range(2):
            for i in range(NL):
                self.e.emit(f"v_add_u32_e32 v{Hh[h][i]}, v{E[h][i]}, v{Cc[h][i]}", vw=[Hh[h][i]])
                self.e.emit(f"v_sub_u32_e32 v{Hh[h][i]}, v{Hh[h][i]}, v{D[h][i]}", vw=[Hh[h][i]])
                self.e.emit(f"v_sub_u32_e32 v{Hh[h][i]}, v{Hh[h][i]}, v{D[h][i]}", vw=[Hh[h][i]])      # H = E + F - 2 G (four units)
        self._fq2_mul(mu, Hh, mu)                                       # X3 = mu H, in place over mu
        self._lw("v_sub_u32_e32", Hh, D, Hh)                            # W = G - H (five units)
        ty = [(th, Hh, 1), (E, Y, -1)]                                  # Y3 = theta W - E Y: one reduction per component
        self._pass2(D[1], ty, imag=True)                                # (G is dead)
        self._pass2(D[0], ty, imag=False)
        self._fq2_mul(Z, E, Z)                                          # Z3 = Z E, in place

    def r_mulfq(self):
        """A <- (A.c0 * B.c0, A.c1 * B.c0), in place"""
        a0, a1, k = self.blk(A0, 0), self.blk(A0, 1), self.blk(B0, 0)
        self.fips([(a0, k)], a0)
        self.fips([(a1, k)], a1)

    def r_fqmul(self):
        a0, k = self.blk(A0, 0), self.blk(B0, 0)
        self.fips([(a0, k)], a0)

    def r_fqsqr(self):
        a0 = self.blk(A0, 0)
        self.fips_sq(a0, a0)                 # result limb j lands after column j + NL, when a0[j] (and its doubled copy's use) is dead

    def r_add(self):
        self._lw("v_add_u32_e32", self.fq2(A0), self.fq2(A0), self.fq2(B0))

    def r_sub(self):
        self._lw("v_sub_u32_e32", self.fq2(A0), self.fq2(A0), self.fq2(B0))

    def r_rsub(self):
        self._lw("v_sub_u32_e32", self.fq2(A0), self.fq2(B0), self.fq2(A0))

    def home_variant(self, op, idx):
        """A <- A op HOME[idx] with the operand read straight from its home registers (no marshalling)."""
        h = self.fq2(HOME0 + SLOT_DW * idx)
        a = self.fq2(A0)
        if op == "add":
            self._lw("v_add_u32_e32", a, a, h)
        elif op == "sub":
            self._lw("v_sub_u32_e32", a, a, h)
        else:
            self._lw("v_sub_u32_e32", a, h, a)

    def r_dbl(self):
        for h in range(2):
            for r in self.blk(A0, h):
                self.e.emit(f"v_lshlrev_b32_e32 v{r}, 1, v{r}", vw=[r])

    def r_neg(self):
        for h in range(2):
            for r in self.blk(A0, h):
                self.e.emit(f"v_sub_u32_e32 v{r}, 0, v{r}", vw=[r])

    def r_negc1(self):
        for r in self.blk(A0, 1):
            self.e.emit(f"v_sub_u32_e32 v{r}, 0, v{r}", vw=[r])

    def r_mulxi(self):
        """A <- (9 a0 - a1, a0 + 9 a1), NORMALISED (64-bit chains: 10 * 2^28 does not fit an int32 limb)."""
        a0, a1 = self.fq2(A0)
        self.lincomb([a0, a1], [[(9, a0), (-1, a1)], [(9, a1), (1, a0)]])

    def r_mulxir(self):
        """as mulxi, also reduced"""
        a0, a1 = self.fq2(A0)
        self.lincomb([a0, a1], [[(9, a0), (-1, a1)], [(9, a1), (1, a0)]], reduce=True)

    def r_norm(self):
        self.norm_limbs(self.blk(A0, 0))
        self.norm_limbs(self.blk(A0, 1))

    def r_redn(self):
        """Normalise AND reduce the representative: A <- A - q p, q = round(top limb * 2^232 / p) per component: limbs
        0..NL-2 in [-2^28, 2^28), value in (-0.51 p, 0.51 p).  Limbs of any int32 magnitude (64-bit chain)."""
        a0, a1 = self.fq2(A0)
        self.lincomb([a0, a1], [[(1, a0)], [(1, a1)]], reduce=True)

    # ------------------------------------------------------------------ boundary conversions
    def r_cvtin(self):
        """A.c0 <- internal form of the packed external value in v[0:7] (8 x u32, canonical, Montgomery R = 2^256):
        unpack to 29-bit limbs, then one Montgomery multiplication by 2^266 mod p (x 2^256 * 2^266 / 2^261 = x 2^261)."""
        w = list(range(A0, A0 + 8))
        l = [self.pool.alloc() for _ in range(NL)]
        for i in range(NL):
            bit = LB * i
            j, s = bit // 32, bit % 32
            if i == NL - 1:
                self.e.emit(f"v_lshrrev_b32_e32 v{l[i]}, {s}, v{w[j]}", vw=[l[i]])
                continue
            if s + LB <= 32:
                self.e.emit(f"v_bfe_u32 v{l[i]}, v{w[j]}, {s}, {LB}", vw=[l[i]])
            else:
                self.e.emit(f"v_alignbit_b32 v{l[i]}, v{w[j + 1]}, v{w[j]}, {s}", vw=[l[i]])
                self.e.emit(f"v_and_b32_e32 v{l[i]}, 0x{MASK:x}, v{l[i]}", vw=[l[i]])
        c = [self.pool.alloc() for _ in range(NL)]
        cin = bal_limbs(pow(2, 2 * NL * LB - 256, P_INT))
        for i in range(NL):
            self.e.emit(f"v_mov_b32_e32 v{c[i]}, {hx(cin[i])}", vw=[c[i]])
        self.fips([(l, c)], self.blk(A0, 0))         # A.c0 region (v0..v8) overlaps w only after w is dead
        self.pool.free(*l)
        self.pool.free(*c)

    def r_cvtout(self):
        """v[0:7] <- canonical external form (8 x u32, Montgomery R = 2^256, in [0, p)) of internal A.c0 (|value| < 80 p)."""
        a0 = self.blk(A0, 0)
        c = [self.pool.alloc() for _ in range(NL)]
        cout = bal_limbs(pow(2, 256, P_INT))
        for i in range(NL):
            self.e.emit(f"v_mov_b32_e32 v{c[i]}, {hx(cout[i])}", vw=[c[i]])
        w = [self.pool.alloc() for _ in range(NL)]
        self.fips([(a0, c)], w)                        # w == y 2^256 mod p, in (-p, p)
        self.pool.free(*c)
        self.unorm_limbs(w)                            # floor carries: the top limb now has the sign of the value
        msk = self.pool.alloc()
        t = self.pool.alloc()
        self.e.emit(f"v_ashrrev_i32_e32 v{msk}, 31, v{w[NL - 1]}", vw=[msk])
        for i in range(NL):                            # w += p if negative
            self.e.emit(f"v_and_b32_e32 v{t}, 0x{P_U[i]:x}, v{msk}", vw=[t])
            self.e.emit(f"v_add_u32_e32 v{w[i]}, v{w[i]}, v{t}", vw=[w[i]])
        self.unorm_limbs(w)
        d = [self.pool.alloc() for _ in range(NL)]     # d = w - p ; take d if d >= 0 (w was in [0, 2p))
        for i in range(NL):
            self.e.emit(f"v_subrev_u32_e32 v{d[i]}, 0x{P_U[i]:x}, v{w[i]}", vw=[d[i]])
        self.unorm_limbs(d)
        self.e.emit(f"v_cmp_gt_i32_e32 vcc, 0, v{d[NL - 1]}", w=["vcc"])      # vcc = (d < 0)
        for i in range(NL):
            self.e.emit(f"v_cndmask_b32_e32 v{w[i]}, v{d[i]}, v{w[i]}, vcc", r=["vcc"], vw=[w[i]])
        # repack 9 x 29 -> 8 x 32
        for j in range(8):
            lo_bit = 32 * j
            i, s = lo_bit // LB, lo_bit % LB
            dst = A0 + j
            self.e.emit(f"v_lshrrev_b32_e32 v{dst}, {s}, v{w[i]}", vw=[dst])
            sh1 = LB - s
            if i + 1 < NL and sh1 < 32:
                self.e.emit(f"v_lshl_or_b32 v{dst}, v{w[i + 1]}, {sh1}, v{dst}", vw=[dst])
            sh2 = 2 * LB - s
            if i + 2 < NL and sh2 < 32:
                self.e.emit(f"v_lshl_or_b32 v{dst}, v{w[i + 2]}, {sh2}, v{dst}", vw=[dst])
        self.pool.free(msk, t, *w)
        self.pool.free(*d)


def routine_body(e, name):
    """Emits leaf routine `name` the way a stand-alone test has to run it: with the caller-side preparation the kernels do outside
    the routine (mul3: the kept y-side difference vectors, formed by Prog.mul3 when a line operand changes)."""
    g = L1v4(e)
    if name == "mul3" and MUL3_KEEP_DY:
        for w in ("B", 1, 3):
            g.mul3_dy(w)
    if name == "mul2a" and MUL3_KEEP_DY:
        for w in (1, 3):
            g.mul3_dy(w)
    getattr(g, "r_" + name)()
    return g


L1V4_NAMES = ["mul", "mul3", "mul2a", "mul6", "dblstep", "addstep", "sqr", "sqr4c", "sqr4cx", "mulfq", "add", "sub", "rsub", "dbl", "neg", "negc1", "mulxi", "mulxir", "norm",
              "redn", "fqmul", "fqsqr", "cvtin", "cvtout"]

if __name__ == "__main__":
    for n in L1V4_NAMES:
        e = Emitter()
        g = L1v4(e)
        getattr(g, "r_" + n)()
        lines = e.finalize()
        print(n, len(lines), "mads", sum(1 for l in lines if l.startswith("v_mad_i64")), "nops", sum(1 for l in lines if l.startswith("s_nop")),
              "max tmp", max(g.pool.used) if g.pool.used else None)
