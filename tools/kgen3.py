#!/usr/bin/env python3
"""kgen3.py -- L1 field routines of the v3 kernels: carry-free signed reduced-radix limbs.

gfx950 has no cheap carries (v_addc costs a full issue slot and a VALU may not read a carry a VALU
wrote < 2 instructions earlier), so v3 drops 32-bit limbs + carry chains:

  * an Fq element is NL = 10 signed 32-bit limbs in radix 2^27 (value = sum l_i 2^(27 i)), kept in
    Montgomery form with R' = 2^270; limbs and values are REDUNDANT (limbs may exceed 27 bits by a
    few bits and be negative, values are any representative), bounds are tracked by the generator;
  * a limb product is ONE instruction: v_mad_i64_i32 acc64 += a_i * b_j -- column sums of up to
    ~40 products of 28-bit limbs fit a signed 64-bit accumulator, so there is no carry handling inside
    a column; one arithmetic 64-bit shift per column moves on;
  * Montgomery reduction is fused column-wise (FIPS): m_k = (lo(S) * n0') mod 2^27, S += m_k p_0, S >>= 27;
  * an Fq2 product is two fused two-product columns passes (a0 b0 + (-a1) b1, a0 b1 + a1 b0): no Karatsuba
    recombination, no wide subtraction;
  * add / sub / neg are NL independent v_add_u32 / v_sub_u32; x(9+u) is v_lshl_add_u32 + add/sub;
    `norm` is one carry-propagation pass (shift, mask, add per limb).
  * R'/p = 2^16.4, so every reduction contracts values to < ~2p whatever the operands' slack.

Blocks: A = v[0:19] (c0 = v0..9, c1 = v10..19), B = v[20:39]; temporaries v[40:79].
At the kernel boundary values are converted from/to ark's 4 x u64 Montgomery (R = 2^256) form.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kgen import Emitter, Pool, P_INT  # noqa: E402

NL = 10
LB = 27
MASK = (1 << LB) - 1
REDN_SHIFT = 41                                         # q = (top limb * REDN_C) >> 41 ~ top limb * 2^243 / p
REDN_C = (1 << (REDN_SHIFT + 243)) // P_INT             # < 2^31: fits a signed multiply-high
RP = 1 << (NL * LB)                       # R' = 2^270
N0P = (-pow(P_INT, -1, 1 << LB)) % (1 << LB)
P_L = [(P_INT >> (LB * i)) & MASK for i in range(NL)]

A0, B0 = 0, 20
TMP_FIRST, TMP_LAST = 40, 79
V_LDS = 80          # v80, v81, v82: LDS byte address of the lane (+0, +64 KiB, +128 KiB)
V_GOFF = 83         # lane * 80 (byte offset inside a global scratch slot)
V_IDX8 = 84
V_IDX = 85
V_TID = 86
V_FLAG = 87
HOME0 = 88          # homes: v[88:247] = 8 x 20
N_HOME = 8
N_AGPR_SLOTS = 12   # a[0:239]
N_LDS_SLOTS = 8     # 8 x 80 B x 256 lanes = 160 KiB
SLOT_DW = 20
S_P = 36            # s36..s45: modulus limbs (radix 2^27)
S_N0 = 46
S_RET1 = "s[54:55]"
S_RET2 = "s[56:57]"
S_RET3 = "s[58:59]"


def to_limbs(x):
    """Canonical non-negative integer -> NL limbs."""
    return [(x >> (LB * i)) & MASK for i in range(NL - 1)] + [x >> (LB * (NL - 1))]


def from_limbs(l):
    return sum(int(v) << (LB * i) for i, v in enumerate(l))


def mont3(x):
    return x * RP % P_INT


class L1v3:
    def __init__(self, e):
        self.e = e
        self.pool = Pool(TMP_FIRST, TMP_LAST)
        self.p = [f"s{S_P + i}" for i in range(NL)]
        self.n0 = f"s{S_N0}"

    # ------------------------------------------------------------------ fused Montgomery column pass
    def fips(self, prods, out):
        """out[0..NL-1] <- (sum over (a, b) in prods of a*b) / R' mod p   (redundant result: limbs 0..NL-2 in
        [0, 2^27), top limb signed).  a, b: lists of NL VGPR numbers.  out registers may alias inputs only
        if those inputs are dead (they are written while later columns still read inputs -> use temps)."""
        acc = self.pool.alloc_pair()
        P = f"v[{acc}:{acc + 1}]"
        m = [self.pool.alloc() for _ in range(NL)]
        res = [self.pool.alloc() for _ in range(NL)]
        first = True

        def mad(x, y):
            nonlocal first
            X = f"v{x}" if isinstance(x, int) else x
            Y = f"v{y}" if isinstance(y, int) else y
            if first:
                self.e.emit(f"v_mad_i64_i32 {P}, vcc, {X}, {Y}, 0", w=["vcc"], vw=[acc, acc + 1])
                first = False
            else:
                self.e.emit(f"v_mad_i64_i32 {P}, vcc, {X}, {Y}, {P}", w=["vcc"], vw=[acc, acc + 1])

        for k in range(2 * NL - 1):
            lo_i, hi_i = max(0, k - (NL - 1)), min(NL - 1, k)
            for (a, b) in prods:
                for i in range(lo_i, hi_i + 1):
                    mad(a[i], b[k - i])
            if k < NL:
                for i in range(k):
                    mad(m[i], self.p[k - i])
                self.e.emit(f"v_mul_lo_u32 v{m[k]}, v{acc}, {self.n0}", vw=[m[k]])
                self.e.emit(f"v_and_b32_e32 v{m[k]}, 0x{MASK:x}, v{m[k]}", vw=[m[k]])
                mad(m[k], self.p[0])
                self.e.emit(f"v_ashrrev_i64 {P}, {LB}, {P}", vw=[acc, acc + 1])
            else:
                for i in range(k - (NL - 1), NL):
                    mad(m[i], self.p[k - i])
                self.e.emit(f"v_and_b32_e32 v{res[k - NL]}, 0x{MASK:x}, v{acc}", vw=[res[k - NL]])
                self.e.emit(f"v_ashrrev_i64 {P}, {LB}, {P}", vw=[acc, acc + 1])
        self.e.emit(f"v_mov_b32_e32 v{res[NL - 1]}, v{acc}", vw=[res[NL - 1]])
        for i in range(NL):
            self.e.emit(f"v_mov_b32_e32 v{out[i]}, v{res[i]}", vw=[out[i]])
        self.pool.free(acc, acc + 1, *m)
        self.pool.free(*res)

    def fips_direct(self, prods, out, fillers=(), gap=8):
        """Same, writing result limbs straight into `out` (out must not overlap any input).

        fillers: independent fast-class instructions [(text, vw, earliest column, latest column)] that are dropped into the
        multiply runs (one after every `gap` consecutive slow-class instructions).  A lone wave issues back-to-back
        v_mad_i64_i32 at 5 cycles each but at 4 when a fast-class instruction breaks the run (issue model), so
        work that has to be done anyway (negations, copies) is free there.  A filler is emitted no earlier than column
        `earliest` and before column `latest` starts; leftovers follow the pass."""
        acc = self.pool.alloc_pair()
        P = f"v[{acc}:{acc + 1}]"
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

        def slow(text, **kw):
            nonlocal run
            self.e.emit(text, **kw)
            run += 1
            if run >= gap:
                fill()

        def mad(x, y):
            nonlocal first
            X = f"v{x}" if isinstance(x, int) else x
            Y = f"v{y}" if isinstance(y, int) else y
            slow(f"v_mad_i64_i32 {P}, vcc, {X}, {Y}, {0 if first else P}", w=["vcc"], vw=[acc, acc + 1])
            first = False

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
                slow(f"v_mul_lo_u32 v{m[k]}, v{acc}, {self.n0}", vw=[m[k]])
                self.e.emit(f"v_and_b32_e32 v{m[k]}, 0x{MASK:x}, v{m[k]}", vw=[m[k]])
                run = 0
                mad(m[k], self.p[0])
                slow(f"v_ashrrev_i64 {P}, {LB}, {P}", vw=[acc, acc + 1])
            else:
                for i in range(k - (NL - 1), NL):
                    mad(m[i], self.p[k - i])
                self.e.emit(f"v_and_b32_e32 v{out[k - NL]}, 0x{MASK:x}, v{acc}", vw=[out[k - NL]])
                run = 0
                slow(f"v_ashrrev_i64 {P}, {LB}, {P}", vw=[acc, acc + 1])
        self.e.emit(f"v_mov_b32_e32 v{out[NL - 1]}, v{acc}", vw=[out[NL - 1]])
        for (text, vw, lo, hi) in todo:
            self.e.emit(text, vw=vw)
        self.pool.free(acc, acc + 1, *m)

    # ------------------------------------------------------------------ blocks
    @staticmethod
    def blk(base, half):
        return list(range(base + NL * half, base + NL * half + NL))

    def limbwise(self, op, dst, a, b):
        for i in range(NL):
            self.e.emit(f"{op} v{dst[i]}, v{a[i]}, v{b[i]}", vw=[dst[i]])

    # ------------------------------------------------------------------ routines: A <- op(A, B)
    def r_mul(self):
        """(a0 + a1 u)(b0 + b1 u): two fused two-product passes.  Result limb j of a pass is produced after
        column j + NL, when operand limb a_j is dead, so each pass writes its result in place: c1 over a1
        (pass 1: a0 b1 + a1 b0), then c0 over a0 (pass 2: a0 b0 + (-a1) b1 with -a1 kept in temporaries)."""
        a0, a1, b0, b1 = self.blk(A0, 0), self.blk(A0, 1), self.blk(B0, 0), self.blk(B0, 1)
        na1 = [self.pool.alloc() for _ in range(NL)]
        # (measured: spreading these negations into pass 1's multiply runs as fillers is 1-3 % SLOWER than doing them up front)
        for i in range(NL):
            self.e.emit(f"v_sub_u32_e32 v{na1[i]}, 0, v{a1[i]}", vw=[na1[i]])
        self.fips_direct([(a1, b0), (a0, b1)], a1)
        self.fips_direct([(a0, b0), (na1, b1)], a0)
        self.pool.free(*na1)

    def r_mul3(self):
        """A <- A*B + H0*H1 + H2*H3 (Fq2 products, H_k = home block k), ONE reduction per output component:
        two fused six-product column passes.  H0 and H2 are destroyed (their c1 halves get negated)."""
        blocks = [(A0, B0), (HOME0, HOME0 + SLOT_DW), (HOME0 + 2 * SLOT_DW, HOME0 + 3 * SLOT_DW)]
        xs = [(self.blk(x, 0), self.blk(x, 1)) for x, _ in blocks]
        ys = [(self.blk(y, 0), self.blk(y, 1)) for _, y in blocks]
        t = [self.pool.alloc() for _ in range(NL)]
        prods = []
        for (x0, x1), (y0, y1) in zip(xs, ys):
            prods += [(x0, y1), (x1, y0)]
        # x1[i] is dead in pass 1 after column i + NL - 1: its negation (for pass 2) rides in the upper columns
        neg = [(f"v_sub_u32_e32 v{x1[i]}, 0, v{x1[i]}", [x1[i]], NL + i, 99) for i in range(NL) for (x0, x1) in xs]
        self.fips_direct(prods, t, fillers=neg, gap=6)                # c1
        prods = []
        for (x0, x1), (y0, y1) in zip(xs, ys):
            prods += [(x0, y0), (x1, y1)]
        # c1 (in t) moves into A.c1 as soon as pass 2 has read A.c1[i] for the last time (column i + NL - 1)
        mov = [(f"v_mov_b32_e32 v{A0 + NL + i}, v{t[i]}", [A0 + NL + i], NL + i, 99) for i in range(NL)]
        self.fips_direct(prods, self.blk(A0, 0), fillers=mov)        # c0, in place over A.c0
        self.pool.free(*t)

    def r_sqr(self):
        """(a0 + a1 u)^2 = (a0+a1)(a0-a1) + 2 a0 a1 u ; both passes write in place."""
        a0, a1 = self.blk(A0, 0), self.blk(A0, 1)
        t = [self.pool.alloc() for _ in range(NL)]
        u = [self.pool.alloc() for _ in range(NL)]
        self.limbwise("v_add_u32_e32", t, a0, a1)
        self.limbwise("v_sub_u32_e32", u, a0, a1)
        for r in a0:
            self.e.emit(f"v_lshlrev_b32_e32 v{r}, 1, v{r}", vw=[r])     # a0 <- 2 a0 (t, u already hold what c0 needs)
        self.fips_direct([(a1, a0)], a1)                                # c1 = a1 * 2a0, in place over a1
        self.fips_direct([(t, u)], a0)                                  # c0 over (dead) a0; t, u are temporaries
        self.pool.free(*t)
        self.pool.free(*u)

    def r_sqr4(self, combine=None):
        """Fq4 squaring for the Granger-Scott cyclotomic squaring, fused: (a + b y)^2 with y^2 = xi, a in block A, b in block B
        (both NORMALISED: |limb| <= 2^27):  A <- a^2 + xi b^2 (normalised),  B <- 2 a b (limbs below 2^28).
        t = a b ; S = xi b + a ; P = (a + b) S ; r0 = P - t - xi t.  Scratch: home blocks 0..2 (the caller reserves them),
        the pool for the column passes.  Worst column: 20 products of |a + b| <= 2 units by |S| <= 11 units plus the reduction:
        450 * 2^54 < 2^63."""
        a0, a1, b0, b1 = self.blk(A0, 0), self.blk(A0, 1), self.blk(B0, 0), self.blk(B0, 1)
        t0, t1 = self.blk(HOME0, 0), self.blk(HOME0, 1)
        u0, u1 = self.blk(HOME0 + SLOT_DW, 0), self.blk(HOME0 + SLOT_DW, 1)
        s0, s1 = self.blk(HOME0 + 2 * SLOT_DW, 0), self.blk(HOME0 + 2 * SLOT_DW, 1)
        n = [self.pool.alloc() for _ in range(NL)]
        for i in range(NL):
            self.e.emit(f"v_sub_u32_e32 v{n[i]}, 0, v{a1[i]}", vw=[n[i]])
        self.fips_direct([(a1, b0), (a0, b1)], t1)                     # t = a b (a, b stay intact)
        self.fips_direct([(a0, b0), (n, b1)], t0)
        for i in range(NL):
            self.e.emit(f"v_add_u32_e32 v{u0[i]}, v{a0[i]}, v{b0[i]}", vw=[u0[i]])                  # u = a + b
            self.e.emit(f"v_add_u32_e32 v{u1[i]}, v{a1[i]}, v{b1[i]}", vw=[u1[i]])
            self.e.emit(f"v_sub_u32_e32 v{n[i]}, 0, v{u1[i]}", vw=[n[i]])                           # -u1 for the second pass
            self.e.emit(f"v_lshl_add_u32 v{s0[i]}, v{b0[i]}, 3, v{b0[i]}", vw=[s0[i]])              # S = xi b + a
            self.e.emit(f"v_sub_u32_e32 v{s0[i]}, v{s0[i]}, v{b1[i]}", vw=[s0[i]])
            self.e.emit(f"v_add_u32_e32 v{s0[i]}, v{s0[i]}, v{a0[i]}", vw=[s0[i]])
            self.e.emit(f"v_lshl_add_u32 v{s1[i]}, v{b1[i]}, 3, v{b1[i]}", vw=[s1[i]])
            self.e.emit(f"v_add_u32_e32 v{s1[i]}, v{s1[i]}, v{b0[i]}", vw=[s1[i]])
            self.e.emit(f"v_add_u32_e32 v{s1[i]}, v{s1[i]}, v{a1[i]}", vw=[s1[i]])
        self.fips_direct([(u1, s0), (u0, s1)], u1)                     # P = u S, in place over u
        self.fips_direct([(u0, s0), (n, s1)], u0)
        self.pool.free(*n)
        w = self.pool.alloc()
        for i in range(NL):
            # r0 = P - t - xi t = (P0 - 10 t0 + t1, P1 - 10 t1 - t0)
            self.e.emit(f"v_lshl_add_u32 v{w}, v{t0[i]}, 3, v{t0[i]}", vw=[w])
            self.e.emit(f"v_add_u32_e32 v{w}, v{w}, v{t0[i]}", vw=[w])
            self.e.emit(f"v_add_u32_e32 v{a0[i]}, v{u0[i]}, v{t1[i]}", vw=[a0[i]])
            self.e.emit(f"v_sub_u32_e32 v{a0[i]}, v{a0[i]}, v{w}", vw=[a0[i]])
            self.e.emit(f"v_lshl_add_u32 v{w}, v{t1[i]}, 3, v{t1[i]}", vw=[w])
            self.e.emit(f"v_add_u32_e32 v{w}, v{w}, v{t1[i]}", vw=[w])
            self.e.emit(f"v_sub_u32_e32 v{a1[i]}, v{u1[i]}, v{t0[i]}", vw=[a1[i]])
            self.e.emit(f"v_sub_u32_e32 v{a1[i]}, v{a1[i]}, v{w}", vw=[a1[i]])
            if combine is None:
                self.e.emit(f"v_lshlrev_b32_e32 v{b0[i]}, 1, v{t0[i]}", vw=[b0[i]])                 # r1 = 2 t
                self.e.emit(f"v_lshlrev_b32_e32 v{b1[i]}, 1, v{t1[i]}", vw=[b1[i]])
        self.norm_limbs(a0)
        self.norm_limbs(a1)
        if combine is not None:
            # Granger-Scott recombination with zc (home block 3) and zd (home block 4), both normalised:
            #   A <- 3 r0 - 2 zc ;  B <- 3 r1 + 2 zd = 6 t + 2 zd   (combine == "xi": B <- 3 xi r1 + 2 zd = 6 xi t + 2 zd)
            zc = (self.blk(HOME0 + 3 * SLOT_DW, 0), self.blk(HOME0 + 3 * SLOT_DW, 1))
            zd = (self.blk(HOME0 + 4 * SLOT_DW, 0), self.blk(HOME0 + 4 * SLOT_DW, 1))
            for h, ah in enumerate((a0, a1)):
                for i in range(NL):
                    self.e.emit(f"v_lshl_add_u32 v{w}, v{ah[i]}, 1, v{ah[i]}", vw=[w])
                    self.e.emit(f"v_sub_u32_e32 v{w}, v{w}, v{zc[h][i]}", vw=[w])
                    self.e.emit(f"v_sub_u32_e32 v{ah[i]}, v{w}, v{zc[h][i]}", vw=[ah[i]])
            src = (t0, t1)
            if combine == "xi":
                for i in range(NL):                                       # B <- xi t, then normalise (10 units)
                    self.e.emit(f"v_lshl_add_u32 v{b0[i]}, v{t0[i]}, 3, v{t0[i]}", vw=[b0[i]])
                    self.e.emit(f"v_sub_u32_e32 v{b0[i]}, v{b0[i]}, v{t1[i]}", vw=[b0[i]])
                    self.e.emit(f"v_lshl_add_u32 v{b1[i]}, v{t1[i]}, 3, v{t1[i]}", vw=[b1[i]])
                    self.e.emit(f"v_add_u32_e32 v{b1[i]}, v{b1[i]}, v{t0[i]}", vw=[b1[i]])
                self.norm_limbs(b0)
                self.norm_limbs(b1)
                src = (b0, b1)
            for h, bh in enumerate((b0, b1)):
                for i in range(NL):
                    self.e.emit(f"v_lshl_add_u32 v{w}, v{src[h][i]}, 1, v{src[h][i]}", vw=[w])      # 3 x
                    self.e.emit(f"v_add_u32_e32 v{w}, v{w}, v{zd[h][i]}", vw=[w])
                    self.e.emit(f"v_lshlrev_b32_e32 v{bh[i]}, 1, v{w}", vw=[bh[i]])                  # 2 (3 x + zd) = 6 x + 2 zd
            for blk in (a0, a1, b0, b1):
                self.norm_limbs(blk)
        self.pool.free(w)

    def r_sqr4c(self):
        self.r_sqr4(combine="plain")

    def r_sqr4cx(self):
        self.r_sqr4(combine="xi")

    # ------------------------------------------------------------------ fused Fq6 multiplication
    def _fq2_mul(self, x, y, o):
        """o <- x * y (Fq2; x, y, o = (c0 limbs, c1 limbs)); o may be x itself (in place, as r_mul) or a disjoint block."""
        (x0, x1), (y0, y1), (o0, o1) = x, y, o
        n = [self.pool.alloc() for _ in range(NL)]
        for i in range(NL):
            self.e.emit(f"v_sub_u32_e32 v{n[i]}, 0, v{x1[i]}", vw=[n[i]])
        self.fips_direct([(x1, y0), (x0, y1)], o1)
        self.fips_direct([(x0, y0), (n, y1)], o0)
        self.pool.free(*n)

    def _lw(self, op, d, a, b):
        for h in range(2):
            self.limbwise(op, d[h], a[h], b[h])

    def _add_xi(self, d, x):
        """d += xi * x  (xi = 9 + u): d0 += 9 x0 - x1 ; d1 += 9 x1 + x0"""
        t = self.pool.alloc()
        for i in range(NL):
            self.e.emit(f"v_lshl_add_u32 v{t}, v{x[0][i]}, 3, v{x[0][i]}", vw=[t])
            self.e.emit(f"v_sub_u32_e32 v{t}, v{t}, v{x[1][i]}", vw=[t])
            self.e.emit(f"v_add_u32_e32 v{d[0][i]}, v{d[0][i]}, v{t}", vw=[d[0][i]])
            self.e.emit(f"v_lshl_add_u32 v{t}, v{x[1][i]}, 3, v{x[1][i]}", vw=[t])
            self.e.emit(f"v_add_u32_e32 v{t}, v{t}, v{x[0][i]}", vw=[t])
            self.e.emit(f"v_add_u32_e32 v{d[1][i]}, v{d[1][i]}, v{t}", vw=[d[1][i]])
        self.pool.free(t)

    def _mulxi_inplace(self, x):
        t = self.pool.alloc()
        for i in range(NL):
            self.e.emit(f"v_lshl_add_u32 v{t}, v{x[0][i]}, 3, v{x[0][i]}", vw=[t])
            self.e.emit(f"v_sub_u32_e32 v{t}, v{t}, v{x[1][i]}", vw=[t])
            self.e.emit(f"v_lshl_add_u32 v{x[1][i]}, v{x[1][i]}, 3, v{x[1][i]}", vw=[x[1][i]])
            self.e.emit(f"v_add_u32_e32 v{x[1][i]}, v{x[1][i]}, v{x[0][i]}", vw=[x[1][i]])
            self.e.emit(f"v_mov_b32_e32 v{x[0][i]}, v{t}", vw=[x[0][i]])
        self.pool.free(t)

    def r_mul6(self):
        """Fq6 multiplication (Fq2[v]/(v^3 - xi), Karatsuba: six Fq2 multiplications), fused: a = (a0, a1, a2) in home blocks
        0..2, b in home blocks 3..5, limbs of magnitude <= 2 units (sums of two normalised values are fine).  Results, all
        NORMALISED:  c0 -> home block 1,  c1 -> block A,  c2 -> home block 0.  Scratch: home blocks 6, 7, blocks A, B, the pool.
        Every input block is destroyed.  Worst column: 20 products of 4 x 4 units plus the reduction: 330 * 2^54 < 2^63."""
        H = lambda k: (self.blk(HOME0 + SLOT_DW * k, 0), self.blk(HOME0 + SLOT_DW * k, 1))
        a, b = [H(0), H(1), H(2)], [H(3), H(4), H(5)]
        v0, v1 = H(6), H(7)
        A, B = (self.blk(A0, 0), self.blk(A0, 1)), (self.blk(B0, 0), self.blk(B0, 1))
        self._fq2_mul(a[0], b[0], v0)
        self._fq2_mul(a[1], b[1], v1)
        # c1 = (a0 + a1)(b0 + b1) - v0 - v1 + xi v2
        self._lw("v_add_u32_e32", A, a[0], a[1])
        self._lw("v_add_u32_e32", B, b[0], b[1])
        self._fq2_mul(A, B, A)
        self._lw("v_sub_u32_e32", A, A, v0)
        self._lw("v_sub_u32_e32", A, A, v1)
        v2 = B
        self._fq2_mul(a[2], b[2], v2)
        self._add_xi(A, v2)
        # c2 = (a0 + a2)(b0 + b2) - v0 - v2 + v1   (in place over a0 / b0: both are dead afterwards)
        self._lw("v_add_u32_e32", a[0], a[0], a[2])
        self._lw("v_add_u32_e32", b[0], b[0], b[2])
        self._fq2_mul(a[0], b[0], a[0])
        self._lw("v_sub_u32_e32", a[0], a[0], v0)
        self._lw("v_sub_u32_e32", a[0], a[0], v2)
        self._lw("v_add_u32_e32", a[0], a[0], v1)
        # c0 = v0 + xi ((a1 + a2)(b1 + b2) - v1 - v2)   (in place over a1 / b1)
        self._lw("v_add_u32_e32", a[1], a[1], a[2])
        self._lw("v_add_u32_e32", b[1], b[1], b[2])
        self._fq2_mul(a[1], b[1], a[1])
        self._lw("v_sub_u32_e32", a[1], a[1], v1)
        self._lw("v_sub_u32_e32", a[1], a[1], v2)
        for h in range(2):
            self.norm_limbs(a[1][h])                     # 3 units: x xi would overflow int32 otherwise
        self._mulxi_inplace(a[1])
        self._lw("v_add_u32_e32", a[1], a[1], v0)
        for blk in (a[1], A, a[0]):
            for h in range(2):
                self.norm_limbs(blk[h])

    def r_mulfq(self):
        """A <- (A.c0 * B.c0, A.c1 * B.c0), in place"""
        a0, a1, k = self.blk(A0, 0), self.blk(A0, 1), self.blk(B0, 0)
        self.fips_direct([(a0, k)], a0)
        self.fips_direct([(a1, k)], a1)

    def r_fqmul(self):
        a0, k = self.blk(A0, 0), self.blk(B0, 0)
        self.fips_direct([(a0, k)], a0)

    def r_fqsqr(self):
        a0 = self.blk(A0, 0)
        self.fips_direct([(a0, a0)], a0)            # result limb j lands after column j + NL, when a0[j] is dead

    def r_add(self):
        for h in range(2):
            self.limbwise("v_add_u32_e32", self.blk(A0, h), self.blk(A0, h), self.blk(B0, h))

    def r_sub(self):
        for h in range(2):
            self.limbwise("v_sub_u32_e32", self.blk(A0, h), self.blk(A0, h), self.blk(B0, h))

    def r_rsub(self):
        for h in range(2):
            self.limbwise("v_sub_u32_e32", self.blk(A0, h), self.blk(B0, h), self.blk(A0, h))

    def home_variant(self, op, idx):
        """A <- A op HOME[idx] with the operand read straight from its home registers (no marshalling)."""
        h0 = HOME0 + SLOT_DW * idx
        for half in range(2):
            a = self.blk(A0, half)
            h = list(range(h0 + NL * half, h0 + NL * half + NL))
            if op == "add":
                self.limbwise("v_add_u32_e32", a, a, h)
            elif op == "sub":
                self.limbwise("v_sub_u32_e32", a, a, h)
            else:
                self.limbwise("v_sub_u32_e32", a, h, a)

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
        """A <- (9 a0 - a1, a0 + 9 a1)"""
        a0, a1 = self.blk(A0, 0), self.blk(A0, 1)
        t = self.pool.alloc()
        for i in range(NL):
            self.e.emit(f"v_lshl_add_u32 v{t}, v{a0[i]}, 3, v{a0[i]}", vw=[t])             # 9 a0
            self.e.emit(f"v_sub_u32_e32 v{t}, v{t}, v{a1[i]}", vw=[t])                      # 9 a0 - a1
            self.e.emit(f"v_lshl_add_u32 v{a1[i]}, v{a1[i]}, 3, v{a1[i]}", vw=[a1[i]])      # 9 a1
            self.e.emit(f"v_add_u32_e32 v{a1[i]}, v{a1[i]}, v{a0[i]}", vw=[a1[i]])          # 9 a1 + a0
            self.e.emit(f"v_mov_b32_e32 v{a0[i]}, v{t}", vw=[a0[i]])
        self.pool.free(t)

    def norm_limbs(self, a):
        """One carry pass: limbs 0..NL-2 -> [0, 2^27), excess pushed up (top limb keeps the sign)."""
        c = self.pool.alloc()
        for i in range(NL - 1):
            self.e.emit(f"v_ashrrev_i32_e32 v{c}, {LB}, v{a[i]}", vw=[c])
            self.e.emit(f"v_and_b32_e32 v{a[i]}, 0x{MASK:x}, v{a[i]}", vw=[a[i]])
            self.e.emit(f"v_add_u32_e32 v{a[i + 1]}, v{a[i + 1]}, v{c}", vw=[a[i + 1]])
        self.pool.free(c)

    def r_norm(self):
        self.norm_limbs(self.blk(A0, 0))
        self.norm_limbs(self.blk(A0, 1))

    def redn_limbs(self, a):
        """Normalise AND reduce the representative: a <- a - q p with q = floor(top limb * 2^243 / p) (the top limb, weight
        2^243, carries the representative; the lower limbs change the quotient by < 0.01).  Result: limbs 0..NL-2 in
        [0, 2^27), value in (-0.01 p, 1.01 p).  One signed 64-bit carry chain: 4 instructions per limb."""
        q, t = self.pool.alloc(), self.pool.alloc()
        acc = self.pool.alloc_pair()
        P = f"v[{acc}:{acc + 1}]"
        self.e.emit(f"v_mov_b32_e32 v{t}, 0x{REDN_C:x}", vw=[t])
        self.e.emit(f"v_mul_hi_i32 v{q}, v{a[NL - 1]}, v{t}", vw=[q])
        self.e.emit(f"v_ashrrev_i32_e32 v{q}, {REDN_SHIFT - 32}, v{q}", vw=[q])
        self.e.emit(f"v_sub_u32_e32 v{q}, 0, v{q}", vw=[q])                       # -q
        for i in range(NL):
            self.e.emit(f"v_mad_i64_i32 {P}, vcc, v{q}, {self.p[i]}, {P if i else 0}", w=["vcc"], vw=[acc, acc + 1])
            self.e.emit(f"v_mad_i64_i32 {P}, vcc, v{a[i]}, 1, {P}", w=["vcc"], vw=[acc, acc + 1])
            if i < NL - 1:
                self.e.emit(f"v_and_b32_e32 v{a[i]}, 0x{MASK:x}, v{acc}", vw=[a[i]])
                self.e.emit(f"v_ashrrev_i64 {P}, {LB}, {P}", vw=[acc, acc + 1])
            else:
                self.e.emit(f"v_mov_b32_e32 v{a[i]}, v{acc}", vw=[a[i]])
        self.pool.free(q, t, acc, acc + 1)

    def r_redn(self):
        self.redn_limbs(self.blk(A0, 0))
        self.redn_limbs(self.blk(A0, 1))

    # ------------------------------------------------------------------ boundary conversions
    def r_cvtin(self):
        """A.c0 <- internal form of the packed external value in v[0:7] (8 x u32, canonical, Montgomery R = 2^256).
        unpack to 27-bit limbs, then one Montgomery multiplication by 2^284 mod p (x 2^256 * 2^284 / 2^270 = x 2^270)."""
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
        cin = to_limbs(pow(2, 284, P_INT))
        for i in range(NL):
            self.e.emit(f"v_mov_b32_e32 v{c[i]}, 0x{cin[i]:x}", vw=[c[i]])
        self.fips_direct([(l, c)], self.blk(A0, 0))      # A.c0 region (v0..v9) overlaps w only after w is dead
        self.pool.free(*l)
        self.pool.free(*c)

    def r_cvtout(self):
        """v[0:7] <- canonical external form (8 x u32, Montgomery R = 2^256, in [0,p)) of internal A.c0."""
        a0 = self.blk(A0, 0)
        c = [self.pool.alloc() for _ in range(NL)]
        cout = to_limbs(pow(2, 256, P_INT))
        for i in range(NL):
            self.e.emit(f"v_mov_b32_e32 v{c[i]}, 0x{cout[i]:x}", vw=[c[i]])
        w = [self.pool.alloc() for _ in range(NL)]
        self.fips_direct([(a0, c)], w)                    # w == y 2^256 mod p, in (-p, 2p), limbs 0..8 normalised
        self.pool.free(*c)
        # w += p if negative (top limb < 0)
        msk = self.pool.alloc()
        t = self.pool.alloc()
        self.e.emit(f"v_ashrrev_i32_e32 v{msk}, 31, v{w[NL - 1]}", vw=[msk])
        for i in range(NL):
            self.e.emit(f"v_and_b32_e32 v{t}, {self.p[i]}, v{msk}", vw=[t])
            self.e.emit(f"v_add_u32_e32 v{w[i]}, v{w[i]}, v{t}", vw=[w[i]])
        self.norm_limbs(w)
        # d = w - p ; take d if d >= 0
        d = [self.pool.alloc() for _ in range(NL)]
        for i in range(NL):
            self.e.emit(f"v_subrev_u32_e32 v{d[i]}, {self.p[i]}, v{w[i]}", vw=[d[i]])
        self.norm_limbs(d)
        self.e.emit(f"v_cmp_gt_i32_e32 vcc, 0, v{d[NL - 1]}", w=["vcc"])      # vcc = (d < 0)
        for i in range(NL):
            self.e.emit(f"v_cndmask_b32_e32 v{w[i]}, v{d[i]}, v{w[i]}, vcc", r=["vcc"], vw=[w[i]])
        # the same once more (w was < 2p + p)
        for i in range(NL):
            self.e.emit(f"v_subrev_u32_e32 v{d[i]}, {self.p[i]}, v{w[i]}", vw=[d[i]])
        self.norm_limbs(d)
        self.e.emit(f"v_cmp_gt_i32_e32 vcc, 0, v{d[NL - 1]}", w=["vcc"])
        for i in range(NL):
            self.e.emit(f"v_cndmask_b32_e32 v{w[i]}, v{d[i]}, v{w[i]}, vcc", r=["vcc"], vw=[w[i]])
        # repack 10 x 27 -> 8 x 32
        for j in range(8):
            lo_bit = 32 * j
            i, s = lo_bit // LB, lo_bit % LB
            dst = A0 + j
            # word j = (l_i >> s) | (l_{i+1} << (27 - s)) | (l_{i+2} << (54 - s))
            self.e.emit(f"v_lshrrev_b32_e32 v{dst}, {s}, v{w[i]}", vw=[dst])
            sh1 = LB - s
            if i + 1 < NL and sh1 < 32:
                self.e.emit(f"v_lshl_or_b32 v{dst}, v{w[i + 1]}, {sh1}, v{dst}", vw=[dst])
            sh2 = 2 * LB - s
            if i + 2 < NL and sh2 < 32:
                self.e.emit(f"v_lshl_or_b32 v{dst}, v{w[i + 2]}, {sh2}, v{dst}", vw=[dst])
        self.pool.free(msk, t, *w)
        self.pool.free(*d)


L1V3_NAMES = ["mul", "mul3", "mul6", "sqr", "sqr4", "sqr4c", "sqr4cx", "mulfq", "add", "sub", "rsub", "dbl", "neg", "negc1", "mulxi", "norm", "redn", "fqmul", "fqsqr", "cvtin", "cvtout"]

if __name__ == "__main__":
    for n in L1V3_NAMES:
        e = Emitter()
        g = L1v3(e)
        getattr(g, "r_" + n)()
        lines = e.finalize()
        print(n, len(lines), "nops", sum(1 for l in lines if l.startswith("s_nop")), "max tmp", max(g.pool.used) if g.pool.used else None)
