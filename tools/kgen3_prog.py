#!/usr/bin/env python3
"""kgen3_prog.py -- L2/L3 of the v3 kernels (signed radix-2^27 limbs, see tools/kgen3.py).

Re-uses the accumulator-machine algorithms of tools/kgen_prog.py (Fq6/Fq12 arithmetic, line
multiplications, G2 steps, Miller loop, final exponentiation) on the v3 register map and adds
  * 20-dword slots (LDS 8, VGPR homes 8, AGPR 12, global scratch),
  * a static BOUND TRACKER: every value carries log2 of its largest limb; multiplications assert that
    their signed 64-bit column sums cannot overflow and `norm` (one carry pass) is inserted exactly
    where a bound would be exceeded (before x(9+u), before stores of values with > 28-bit limbs, ...),
  * conversion from/to ark's 4 x u64 Montgomery (R = 2^256) limbs at the kernel boundary (bit-exact
    canonical output).
"""
import math
import re
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kgen_prog as KP  # noqa: E402
from kgen import Emitter, P_INT, SIX_U_PLUS_2_NAF, align_code, max_branch_distance  # noqa: E402
from kgen3 import (A0, B0, HOME0, L1V3_NAMES, L1v3, LB, MASK, N0P, N_AGPR_SLOTS, N_HOME, N_LDS_SLOTS, NL, P_L, S_N0, S_P, S_RET1, S_RET2,  # noqa: E402
                   S_RET3, SLOT_DW, V_FLAG, V_GOFF, V_IDX, V_IDX8, V_LDS, V_TID, mont3, to_limbs)
from kgen_prog import (AGPR, GLOB, HOME, LDS, Const, GlobDyn, Slot, S_FIN, S_G1, S_G2, S_GADDR, S_GBASE, S_GRID, S_GSTRIDE, S_I, S_IOADDR,  # noqa: E402
                       S_ITEM, S_J, S_K, S_N, S_NAF_NEG, S_NAF_NZ, S_NITEMS, S_NSTRIDE, S_OUT, S_SAVE_EXEC, S_SCRATCH, S_STATUS, S_TMP0, S_TMP1,
                       S_XNAF_NEG, S_XNAF_NZ, S_XNAF_RED, f2mul, f2pow, naf_masks, x_naf)

# ---- static bound tracking: every value carries an interval [lo, hi] (in units of 2^27) that contains all of
# its limbs 0..NL-2 (both Fq2 components); the top limb is small by construction (values stay below ~2^262).
# ---- x-power schedule of the final exponentiation (F <- F^BN_X for cyclotomic F, three times per pairing).
# pow_native (final_exp_native.rs:56-84) walks the NAF of BN_X: 62 squarings + 23 multiplications.  The value F^x does
# not depend on the chain, so the kernels use a signed fixed-set recoding instead: digits in {0, +-1, +-5, +-9, +-13}
# (found by exhaustive search over digit sets, tools/exp/xchain.py): 59 squarings + 12 multiplications in the loop and
# b^4, b^5, b^9, b^13 from 2 squarings + 3 multiplications: 61 S + 15 M instead of 62 S + 23 M (a squaring costs a
# third of a multiplication).  Negative digits multiply by the conjugate (= inverse of a unitary element).
X_POWERS = (1, 5, 9, 13)
X_DIGITS = (-13, -1, 0, 0, 0, 0, 0, 0, 0, 5, 0, 0, 0, 0, 0, 0, 9, 0, 0, 0, 0, -13, 0, 0, 0, 0, -13, 0, 0, 0, 0, 9, 0, 0, 0, 0, -5, 0, 0, 0, -13,
            0, 0, 0, 0, 13, 0, 0, 0, 0, 0, 5, 0, 0, -13, 0, 0, 0, 0, 9)            # least significant first
assert sum(d << i for i, d in enumerate(X_DIGITS)) == 4965661367192848881 and X_DIGITS[-1] in X_POWERS
assert all(d == 0 or abs(d) in X_POWERS for d in X_DIGITS)
G_POW = {5: 8, 9: 9, 13: 10}     # scratch Fq12 registers of b^5, b^9, b^13 (b itself: the caller's register)
G_B4 = 11                        # b^4 (only while the powers are built)
N_GREG = 12                      # Fq12 scratch registers G0..G11 = slots 0..71
GLOB_TMP0 = 6 * N_GREG           # eight overflow temporaries: slots 72..79
S_PB = 47                        # byte offset of the base's scratch register during the x-power routine
S_XIDX0, S_XIDX1 = "s[50:51]", "s[52:53]"      # which power a non-zero digit selects (index into X_POWERS)

USE_MUL6 = bool(int(os.environ.get("KGEN3_MUL6", "1")))          # fused Fq6 multiplication (L1 mul6) in fq12_mul / fq12_sqr
ALIGN_CODE = bool(int(os.environ.get("KGEN3_ALIGN", "1")))     # keep 8-byte instructions 8-byte aligned (kgen.align_code)
RED_POWERS = True            # reduce the representatives of b^5, b^9, b^13 before they are stored
RED_RUN = 4                  # cyclotomic squarings in a row before the x-power loop reduces the representative


def x_red_mask(digits):
    """Bit j set: digit j of the x-power loop (walked from the top, as L3_powx does) is zero and closes a run of
    RED_RUN squarings without a multiplication -> L2_redF is called there."""
    mask, run = 0, 0
    for j in range(len(digits) - 1, -1, -1):
        run += 1
        if digits[j] != 0:
            run = 0
        elif run == RED_RUN:
            mask |= 1 << j
            run = 0
    return mask


R_NORM = (0.0, 1.0)          # Montgomery-reduction outputs / normalised values / constants
STORE_MAG = 3.05             # stored values may keep limbs up to 3 units (e.g. 3t - 2z of normalised t, z)
LIMB_MAG = 15.9              # int32 limbs: |limb| < 2^31 = 16 units
COL_LIMIT = 62.5             # log2 bound of a signed 64-bit column sum (0.5 bit of margin)
N_CHUNK = SLOT_DW // 4       # 16-byte chunks per slot
E_NORM = 27.0
E_STORE_MAX = 27.0 + math.log2(STORE_MAG)


# Value bounds (in units of p): a Montgomery reduction maps a sum of products of values a_i b_i to
# sum(a_i b_i) / R' + (< p), and R'/p = 2^16.4, so reductions contract; additions and x(9+u) grow values.
# Every stored value must stay below V_STORE p, which keeps the top limb (weight 2^243) below 2^17.
K_RP = float((1 << (NL * LB)) // P_INT)
V_STORE = 64.0          # assumed bound (in p) of a value another routine left in a slot
V_CAP = 65536.0         # nothing larger is ever stored: the top limb (weight 2^243) then stays below 2^27 = one unit


def mag(r):
    return max(abs(r[0]), abs(r[1]))


def lg(r):
    """log2 of the largest limb magnitude of a value with range r."""
    return 27.0 + math.log2(max(mag(r), 1e-9))


def r_add(a, b):
    return (a[0] + b[0], a[1] + b[1])


def r_sub(a, b):
    return (a[0] - b[1], a[1] - b[0])


def r_neg(a):
    return (-a[1], -a[0])


def r_hull(a, b):
    return (min(a[0], b[0]), max(a[1], b[1]))


def r_mulxi(a):
    l, h = a
    c0 = (9 * l - h, 9 * h - l)
    c1 = (10 * min(l, 0) if False else min(10 * l, 10 * h), max(10 * l, 10 * h))
    return r_hull(c0, c1)


class Prog3(KP.Prog):
    def __init__(self, e, l1_labels):
        super().__init__(e, l1_labels)
        self.rA = None
        self.slot_r = {}
        self.vA = V_STORE
        self.slot_v = {}
        self.max_v = 0.0
        self.norm_keys = frozenset()
        self.entry_v = {}           # certification: bounds of the values other routines left in the slots
        self.default_v = V_STORE
        self.read_keys = {}         # slot keys whose entry bound was used -> the bound

    UNKNOWN = (-STORE_MAG, STORE_MAG)      # contract for values stored by other routines
    # slots in `norm_keys` hold NORMALISED values on every routine boundary (the Fq12 accumulator and the operand
    # copies of the final exponentiation): limbs 0..NL-2 in [0, 2^27), top limb within the value cap.  Their
    # consumers start from a 3x tighter interval, which removes most normalisations inside the routines.
    STORED_NORM = (-1.0, 1.0)              # (negations / conjugates of normalised values included)

    def v_of(self, slot):
        if slot.kind == "const":
            return 1.0
        k = self.key(slot)
        if k in self.slot_v:
            return self.slot_v[k]
        v = self.entry_v.get(k, self.default_v)
        self.read_keys[k] = v
        return v

    # compatibility shims (exponent view of the interval)
    @property
    def eA(self):
        return None if self.rA is None else lg(self.rA)

    @eA.setter
    def eA(self, e):
        self.rA = None if e is None else ((0.0, 2.0 ** (e - 27.0)) if e <= 27.0 else (-(2.0 ** (e - 27.0)), 2.0 ** (e - 27.0)))

    @property
    def slot_e(self):
        outer = self

        class _View(dict):
            def __setitem__(s2, k, e):
                outer.slot_r[k] = (0.0, 1.0) if e <= 27.0 else (-(2.0 ** (e - 27.0)), 2.0 ** (e - 27.0))
        return _View()

    # ---------------------------------------------------------------- bounds
    def r_norm(self):
        """Limb interval of a normalised value (a reduction or norm output): limbs 0..NL-2 lie in [0, 2^27); the
        top limb (weight 2^243) is signed and carries the representative: |top| <= vA p / 2^243 = vA / K_RP units."""
        t = self.vA / K_RP
        return (-t, max(1.0, t))

    @staticmethod
    def key(slot):
        if slot.kind == "globdyn":
            return ("globdyn", slot.k)
        if slot.kind == "const":
            return ("const", slot.name)
        return (slot.kind, slot.idx)

    def r_of(self, slot):
        if slot.kind == "const":
            return R_NORM
        k = self.key(slot)
        return self.slot_r.get(k, self.STORED_NORM if k in self.norm_keys else self.UNKNOWN)

    def e_of(self, slot):
        return lg(self.r_of(slot))

    def reset_tags(self):
        super().reset_tags()
        self.eA = None

    # ---------------------------------------------------------------- data movement (20-dword slots)
    def _lds_addr(self, slot, c):
        ci = slot.idx * N_CHUNK + c
        return V_LDS + ci // 16, (ci % 16) * 4096

    def _glob_base(self, slot):
        if slot.kind == "glob":
            self.e.salu(f"s_mul_i32 s{S_TMP0}, s{S_GSTRIDE}, {slot.idx}")
        else:
            self.e.salu(f"s_mul_i32 s{S_TMP0}, s{S_GSTRIDE}, {slot.k}")
            self.e.salu(f"s_add_u32 s{S_TMP0}, s{S_TMP0}, s{S_GBASE}")
        self.e.salu(f"s_add_u32 s62, s64, s{S_TMP0}")
        self.e.salu("s_addc_u32 s63, s65, 0")

    def load(self, blk, slot):
        if slot.kind == "lds":
            for c in range(N_CHUNK):
                base, off = self._lds_addr(slot, c)
                self.e.emit(f"ds_read_b128 v[{blk + 4 * c}:{blk + 4 * c + 3}], v{base} offset:{off}", kind="lds", vw=range(blk + 4 * c, blk + 4 * c + 4))
            self.lds_pending = True
        elif slot.kind == "home":
            r0 = HOME0 + SLOT_DW * slot.idx
            for i in range(SLOT_DW):
                self.e.emit(f"v_mov_b32_e32 v{blk + i}, v{r0 + i}", vw=[blk + i])
        elif slot.kind == "agpr":
            for i in range(SLOT_DW):
                self.e.emit(f"v_accvgpr_read_b32 v{blk + i}, a{SLOT_DW * slot.idx + i}", vw=[blk + i])
        elif slot.kind == "const":
            w = to_limbs(mont3(slot.c0)) + to_limbs(mont3(slot.c1))
            for i in range(SLOT_DW):
                self.e.emit(f"v_mov_b32_e32 v{blk + i}, 0x{w[i]:x}", vw=[blk + i])
        elif slot.kind in ("glob", "globdyn"):
            self._glob_base(slot)
            for c in range(N_CHUNK):
                self.e.emit(f"global_load_dwordx4 v[{blk + 4 * c}:{blk + 4 * c + 3}], v{V_GOFF}, {S_GADDR} offset:{16 * c}", kind="vmem",
                            vw=range(blk + 4 * c, blk + 4 * c + 4))
            self.vm_pending = True
        else:
            raise ValueError(slot.kind)
        self._count("ld_" + slot.kind)

    def store(self, blk, slot):
        if slot.kind == "lds":
            for c in range(N_CHUNK):
                base, off = self._lds_addr(slot, c)
                self.e.emit(f"ds_write_b128 v{base}, v[{blk + 4 * c}:{blk + 4 * c + 3}] offset:{off}", kind="lds")
        elif slot.kind == "home":
            r0 = HOME0 + SLOT_DW * slot.idx
            for i in range(SLOT_DW):
                self.e.emit(f"v_mov_b32_e32 v{r0 + i}, v{blk + i}", vw=[r0 + i])
        elif slot.kind == "agpr":
            for i in range(SLOT_DW):
                self.e.emit(f"v_accvgpr_write_b32 a{SLOT_DW * slot.idx + i}, v{blk + i}")
        elif slot.kind in ("glob", "globdyn"):
            self._glob_base(slot)
            for c in range(N_CHUNK):
                self.e.emit(f"global_store_dwordx4 v{V_GOFF}, v[{blk + 4 * c}:{blk + 4 * c + 3}], {S_GADDR} offset:{16 * c}", kind="vmem",
                            store=range(blk + 4 * c, blk + 4 * c + 4))
            self.e.raw("s_nop 1")
        else:
            raise ValueError(slot.kind)
        self._count("st_" + slot.kind)

    # ---------------------------------------------------------------- accumulator machine with bound tracking
    def A(self, x):
        if self.tagA is not x:
            self.load(A0, x)
            self.tagA = x
            self.rA = self.r_of(x)
            self.vA = self.v_of(x)
        return self

    def _B(self, y):
        if self.tagB is not y:
            self.load(B0, y)
            self.tagB = y

    # ---- operand blocks H0..H3 of the three-term multiply (home registers 0..3 used as raw blocks)
    def reserve_blocks(self, count=4):
        self._saved_tmp = self.free_tmp
        self._n_reserved = count
        self.free_tmp = [t for t in self.free_tmp if not (t.kind == "home" and t.idx < count)]
        assert len(self._saved_tmp) - len(self.free_tmp) == count, f"home blocks 0..{count - 1} must be free"
        self.tagH = [None] * 4
        self.eH = [None] * 4
        self.vH = [V_STORE] * 4
        self._blocks_reserved = True

    def release_blocks(self):
        held = [t for t in self._saved_tmp if t.kind == "home" and t.idx < self._n_reserved]
        self.free_tmp = held + self.free_tmp
        self._blocks_reserved = False

    def ldH(self, k, slot):
        """home block k <- slot (straight ds_read for LDS slots)."""
        if self.tagH[k] is slot:
            return self
        blk = HOME0 + SLOT_DW * k
        self.load(blk, slot)
        self.tagH[k] = slot
        self.eH[k] = self.r_of(slot)
        self.vH[k] = self.v_of(slot)
        return self

    def mul3(self, y):
        """A <- A*y + H0*H1 + H2*H3"""
        self._B(y)
        rA = self.rA if self.rA is not None else self.UNKNOWN
        worst = mag(rA) * mag(self.r_of(y)) + mag(self.eH[0]) * mag(self.eH[1]) + mag(self.eH[2]) * mag(self.eH[3])
        self._need(math.log2(2 * NL) + 54.0 + math.log2(worst) <= COL_LIMIT, f"mul3 {worst}")
        self.vA = 2 * (self.vA * self.v_of(y) + self.vH[0] * self.vH[1] + self.vH[2] * self.vH[3]) / K_RP + 1
        self._raw_call("mul3")
        self.rA = self.r_norm()
        self.tagH[0] = self.tagH[2] = None          # destroyed
        return self

    homes_free = False

    def _mul6_regs(self, a, b, a_plus=None, b_plus=None):
        """Fused Fq6 multiplication (L1 mul6) of (a + a_plus) by (b + b_plus), coefficient-wise sums formed while the operands
        are loaded into the home blocks.  Returns the three result 'slots' [c0, c1, c2]: c0 = HOME(1), c1 = block A (None),
        c2 = HOME(0), all normalised, with their bounds recorded; every home block is clobbered."""
        va = max(self.v_of(s_) for s_ in a) + (max(self.v_of(s_) for s_ in a_plus if s_ is not None) if a_plus else 0.0)
        vb = max(self.v_of(s_) for s_ in b) + (max(self.v_of(s_) for s_ in b_plus if s_ is not None) if b_plus else 0.0)
        for base, slots, plus in ((0, a, a_plus), (3, b, b_plus)):
            for k, s_ in enumerate(slots):
                blk = HOME0 + SLOT_DW * (base + k)
                s2 = plus[k] if plus else None
                m_ = mag(self.r_of(s_)) + (mag(self.r_of(s2)) if s2 is not None else 0.0)
                if m_ > 2.0:                                  # mul6 takes limbs of up to two units
                    self.A(s_)
                    if s2 is not None:
                        self.add(s2)
                    self.norm()
                    self.wait()
                    for i in range(SLOT_DW):
                        self.e.emit(f"v_mov_b32_e32 v{blk + i}, v{A0 + i}", vw=[blk + i])
                else:
                    self.load(blk, s_)
                    if s2 is not None:
                        self.load(A0, s2)
                        self.tagA = None
                        self.wait()
                        for i in range(SLOT_DW):
                            self.e.emit(f"v_add_u32_e32 v{blk + i}, v{blk + i}, v{A0 + i}", vw=[blk + i])
        self._raw_call("mul6")
        self.tagB = None
        v_prod = 2 * va * vb / K_RP + 1
        v_sum = 8 * va * vb / K_RP + 1
        res = [HOME(1, "mul6.c0"), None, HOME(0, "mul6.c2")]
        for slot, v in ((res[0], 10 * (v_sum + 2 * v_prod) + v_prod), (res[2], v_sum + 3 * v_prod)):
            self._need(v <= V_CAP, f"mul6 result value {v}")
            self.slot_r[self.key(slot)] = (-v / K_RP, max(1.0, v / K_RP))
            self.slot_v[self.key(slot)] = v
            self.max_v = max(self.max_v, v)
        self.vA = v_sum + 12 * v_prod
        self._need(self.vA <= V_CAP, f"mul6 result value {self.vA}")
        self.rA = (-self.vA / K_RP, max(1.0, self.vA / K_RP))
        self.tagA = None
        return res

    def fq6_mul(self, a, b, out, a_plus=None, b_plus=None):
        """(a0, a1, a2)(b0, b1, b2) in Fq2[v]/(v^3 - xi): ONE fused L1 routine (mul6: operands in the home blocks, results
        normalised) when the routine keeps no temporary in the home registers; six calls plus glue otherwise."""
        if not self.homes_free:
            assert a_plus is None and b_plus is None
            return super().fq6_mul(a, b, out)
        res = self._mul6_regs(a, b, a_plus, b_plus)
        self.to(out[1])                                       # c1 sits in block A
        self.A(res[0]).to(out[0])
        self.A(res[2]).to(out[2])

    def fq12_mul(self, F, Bs, conj_b=False):
        """F <- F * B on the fused Fq6 multiplication: the sums of the third product are formed while its operands are
        loaded and its results are combined straight from the registers (B is not modified)."""
        if not self.homes_free:
            return super().fq12_mul(F, Bs, conj_b)
        if conj_b:
            for k in (1, 3, 5):
                self.A(Bs[k]).neg().to(Bs[k])
        A_0, A_1 = [F[0], F[2], F[4]], [F[1], F[3], F[5]]
        B_0, B_1 = [Bs[0], Bs[2], Bs[4]], [Bs[1], Bs[3], Bs[5]]
        T0 = [self.tmp() for _ in range(3)]
        T1 = [self.tmp() for _ in range(3)]
        self.fq6_mul(A_0, B_0, T0)
        self.fq6_mul(A_1, B_1, T1)
        M = self._mul6_regs(A_0, B_0, A_1, B_1)               # (A0 + A1)(B0 + B1)
        self.sub(T0[1]).sub(T1[1]).to(F[3])                   # c1 is in block A
        self.A(M[0]).sub(T0[0]).sub(T1[0]).to(F[1])
        self.A(M[2]).sub(T0[2]).sub(T1[2]).to(F[5])
        self.A(T1[2]).mulxi().add(T0[0]).to(F[0])
        self.A(T0[1]).add(T1[0]).to(F[2])
        self.A(T0[2]).add(T1[1]).to(F[4])
        self.rel(*T0)
        self.rel(*T1)

    def fq12_sqr(self, F):
        """F <- F^2 (complex squaring over Fq6: t = A0 A1, u = (A0 + A1)(A0 + v A1)) on the fused Fq6 multiplication."""
        if not self.homes_free:
            return super().fq12_sqr(F)
        A_0, A_1 = [F[0], F[2], F[4]], [F[1], F[3], F[5]]
        T = [self.tmp() for _ in range(3)]
        S0 = self.tmp()
        self.fq6_mul(A_0, A_1, T)
        self.A(F[5]).mulxi().add(F[0]).to(S0)                 # first coefficient of A0 + v A1
        U = self._mul6_regs(A_0, [S0, F[2], F[4]], A_1, [None, F[1], F[3]])
        self.sub(T[1]).sub(T[0]).to(F[2])                     # u1 - t1 - t0   (u1 is in block A)
        X = S0
        self.A(T[2]).mulxi().to(X)
        self.A(U[0]).sub(T[0]).sub(X).to(F[0])
        self.A(U[2]).sub(T[2]).sub(T[1]).to(F[4])
        self.A(T[0]).dbl().to(F[1])
        self.A(T[1]).dbl().to(F[3])
        self.A(T[2]).dbl().to(F[5])
        self.rel(S0, *T)

    USE_SQR4 = bool(int(os.environ.get("KGEN3_SQR4", "1")))

    def fq4_sqr(self, a, b, r0, r1):
        """(a + b y)^2, y^2 = xi: r0 = a^2 + xi b^2, r1 = 2 a b -- one fused L1 routine (sqr4) when the home blocks are
        reserved and both operands are normalised; the generic two-multiplication form otherwise."""
        ra, rb = self.r_of(a), self.r_of(b)
        if not (self.USE_SQR4 and getattr(self, "_blocks_reserved", False) and mag(ra) <= 1.0 and mag(rb) <= 1.0):
            return super().fq4_sqr(a, b, r0, r1)
        self.A(a)
        self._B(b)
        va, vb = self.vA, self.v_of(b)
        v_t = 2 * va * vb / K_RP + 1
        v_p = 2 * (va + vb) * (va + 10 * vb) / K_RP + 1
        self._raw_call("sqr4")
        self.tagH = [None] * 4                      # home blocks 0..2 are scratch of the routine
        self.vA = v_p + 11 * v_t
        self.rA = self.r_norm()
        self.to(r0)
        self.wait()
        self.store(B0, r1)                          # block B holds r1 = 2 t
        k1 = self.key(r1)
        self.slot_v[k1] = 2 * v_t
        self.max_v = max(self.max_v, 2 * v_t)
        self._need(2 * v_t <= V_CAP, "sqr4 r1 value")
        self.slot_r[k1] = (-2 * v_t / K_RP, max(2.0, 2 * v_t / K_RP))
        self._need(k1 not in self.norm_keys, "sqr4: r1 is not normalised")
        self.tagB = r1

    def _sqr4c(self, a, b, zc, zd, out_a, out_b, xi=False):
        """out_a <- 3 (a^2 + xi b^2) - 2 zc ; out_b <- 3 (2 a b) + 2 zd  (xi: 3 xi (2 a b) + 2 zd): one Fq4 squaring of the
        Granger-Scott cyclotomic squaring with its recombination, in ONE L1 routine.  All four operands normalised."""
        for s_ in (a, b, zc, zd):
            self._need(mag(self.r_of(s_)) <= 1.0, f"sqr4c operand {s_} is not normalised")
        self.load(HOME0 + 3 * SLOT_DW, zc)
        self.load(HOME0 + 4 * SLOT_DW, zd)
        self.A(a)
        self._B(b)
        va, vb, vc, vd = self.vA, self.v_of(b), self.v_of(zc), self.v_of(zd)
        v_t = 2 * va * vb / K_RP + 1
        v_p = 2 * (va + vb) * (va + 10 * vb) / K_RP + 1
        self._raw_call("sqr4cx" if xi else "sqr4c")
        self.tagH = [None] * 4
        self.vA = 3 * (v_p + 11 * v_t) + 2 * vc
        self.rA = self.r_norm()
        self.to(out_a)
        v_b = (60 if xi else 6) * v_t + 2 * vd
        self._need(v_b <= V_CAP, "sqr4c value")
        self.wait()
        self.store(B0, out_b)
        k1 = self.key(out_b)
        self.slot_v[k1] = v_b
        self.max_v = max(self.max_v, v_b)
        self.slot_r[k1] = (-v_b / K_RP, max(1.0, v_b / K_RP))
        self.tagB = out_b

    def fq12_cyc_sqr(self, F):
        """Granger-Scott squaring (F in the cyclotomic subgroup), in place: three fused Fq4 squarings with recombination."""
        if not (self.USE_SQR4 and all(mag(self.r_of(s_)) <= 1.0 for s_ in F)):
            self.reserve_blocks()
            super().fq12_cyc_sqr(F)
            self.release_blocks()
            return
        self.reserve_blocks(5)
        t2, t5 = self.tmp(), self.tmp()
        self._sqr4c(F[1], F[4], F[2], F[5], t2, t5)            # F2' = 3 t2 - 2 F2 ; F5' = 3 t3 + 2 F5  (kept aside: F2, F5 are read below)
        self._sqr4c(F[0], F[3], F[0], F[3], F[0], F[3])        # F0' = 3 t0 - 2 F0 ; F3' = 3 t1 + 2 F3
        self._sqr4c(F[2], F[5], F[4], F[1], F[4], F[1], xi=True)   # F4' = 3 t4 - 2 F4 ; F1' = 3 xi t5 + 2 F1
        self.mov(F[2], t2)
        self.mov(F[5], t5)
        self.rel(t2, t5)
        self.release_blocks()

    def mul_by_034(self, F, L0, L3, L4):
        """f *= L0 + L3 w^3 + L4 w^4 with one reduction per output coefficient (xi folded into the line)."""
        self.reserve_blocks()
        L3x, L4x = self.tmp(), self.tmp()
        self.A(L3).mulxi().to(L3x)
        self.A(L4).mulxi().to(L4x)
        c = [self.tmp() for _ in range(3)]
        # c0 = a0 L0 + a3 xiL3 + a2 xiL4 ; c1 = a1 L0 + a4 xiL3 + a3 xiL4 ; c2 = a2 L0 + a5 xiL3 + a4 xiL4
        for k, (i0, i3, i4) in enumerate(((0, 3, 2), (1, 4, 3), (2, 5, 4))):
            self.ldH(1, L3x).ldH(3, L4x).ldH(0, F[i3]).ldH(2, F[i4])
            self.A(F[i0]).mul3(L0).to(c[k])
        # c3 = a3 L0 + a0 L3 + a5 xiL4   (a3 is not read again: the result goes straight to its place)
        self.ldH(1, L3).ldH(3, L4x).ldH(0, F[0]).ldH(2, F[5])
        self.A(F[3]).mul3(L0).to(F[3])
        # c4 = a4 L0 + a1 L3 + a0 L4 ; c5 = a5 L0 + a2 L3 + a1 L4
        self.ldH(1, L3).ldH(3, L4).ldH(0, F[1]).ldH(2, F[0])
        self.A(F[4]).mul3(L0).to(F[4])
        self.ldH(0, F[2]).ldH(2, F[1])
        self.A(F[5]).mul3(L0).to(F[5])
        for k in range(3):
            self.mov(F[k], c[k])
        self.rel(L3x, L4x, *c)
        self.release_blocks()

    def mul_by_235(self, F, L2, L3, L5):
        """f *= L2 w^2 + L3 w^3 + L5 w^5, same scheme."""
        self.reserve_blocks()
        L2x, L3x, L5x = self.tmp(), self.tmp(), self.tmp()
        self.A(L2).mulxi().to(L2x)
        self.A(L3).mulxi().to(L3x)
        self.A(L5).mulxi().to(L5x)
        c = [self.tmp() for _ in range(4)]
        # c0 = xi (a4 b2 + a3 b3 + a1 b5) ; c1 = xi (a5 b2 + a4 b3 + a2 b5)
        for k, (i2, i3, i5) in enumerate(((4, 3, 1), (5, 4, 2))):
            self.ldH(1, L3x).ldH(3, L5x).ldH(0, F[i3]).ldH(2, F[i5])
            self.A(F[i2]).mul3(L2x).to(c[k])
        # c2 = a0 b2 + xi (a5 b3 + a3 b5)
        self.ldH(1, L3x).ldH(3, L5x).ldH(0, F[5]).ldH(2, F[3])
        self.A(F[0]).mul3(L2).to(c[2])
        # c3 = a1 b2 + a0 b3 + xi a4 b5 ; c4 = a2 b2 + a1 b3 + xi a5 b5
        for k, (i2, i3, i5) in ((3, (1, 0, 4)), (4, (2, 1, 5))):
            self.ldH(1, L3).ldH(3, L5x).ldH(0, F[i3]).ldH(2, F[i5])
            self.A(F[i2]).mul3(L2).to(c[k] if k == 3 else F[4])           # a4 is not read after c3
        # c5 = a3 b2 + a2 b3 + a0 b5
        self.ldH(1, L3).ldH(3, L5).ldH(0, F[2]).ldH(2, F[0])
        self.A(F[3]).mul3(L2).to(F[5])
        for k in range(4):
            self.mov(F[k], c[k])
        self.rel(L2x, L3x, L5x, *c)
        self.release_blocks()

    def set_A_fresh(self, e=E_NORM):
        """A was filled by hand-written code with a normalised, reduced value."""
        self.tagA = None
        self.rA = R_NORM
        self.vA = 2.0

    INLINE_SMALL = bool(int(os.environ.get("KGEN3_INLINE_SMALL", "1")))
    # a call/return pair costs a lone wave ~70 cycles (two taken branches, each refilling the instruction buffer): the
    # 20-instruction routines are inlined; so are norm (54) and mulxi (50): +0.6 % while every Fq2 operation was its own
    # call, +0.4 % (same box, A/B) with the fused leaf routines, for 6 % more code.
    INLINE_SET = ("add", "sub", "rsub", "dbl", "neg", "negc1") + tuple([x for x in os.environ.get("KGEN3_INLINE_MORE", "norm,mulxi").split(",") if x])

    def _raw_call(self, name):
        self.wait()
        base = name.split("_h")[0]
        if self.INLINE_SMALL and base in self.INLINE_SET:
            g = L1v3(self.e)
            if "_h" in name:
                g.home_variant(base, int(name.split("_h")[1]))
            else:
                getattr(g, "r_" + name)()
        else:
            self.e.salu(f"s_call_b64 {S_RET1}, {self.l1[name]}")
        self.tagA = None
        self._count(name)

    def norm(self):
        self._raw_call("norm")
        self.rA = self.r_norm()
        return self

    def redn(self):
        """Normalise and bring the representative back to (-0.01 p, 1.01 p) (L1 redn: quotient from the top limb)."""
        return self.call("redn")

    def _need(self, ok, what):
        if not ok:
            raise AssertionError("bound violated: " + what)

    def call(self, name, rB=None, direct=None, vB=1.0):
        vA = self.vA
        v_out = {"mul": 2 * vA * vB / K_RP + 1, "mulfq": vA * vB / K_RP + 1, "fqmul": vA * vB / K_RP + 1, "sqr": 4 * vA * vA / K_RP + 1,
                   "fqsqr": vA * vA / K_RP + 1, "add": vA + vB, "sub": vA + vB, "rsub": vA + vB, "dbl": 2 * vA, "neg": vA, "negc1": vA,
                   "mulxi": 10 * vA, "norm": vA, "redn": 1.01}[name]
        reduced = (-v_out / K_RP, max(1.0, v_out / K_RP))       # limb interval of a reduction output (see r_norm)
        rA = self.rA if self.rA is not None else self.UNKNOWN
        if isinstance(rB, float):
            rB = (-(2.0 ** (rB - 27.0)), 2.0 ** (rB - 27.0))

        def col(n_terms, x, y):
            return math.log2(n_terms) + 54.0 + math.log2(max(mag(x), 1e-9)) + math.log2(max(mag(y), 1e-9))

        if name == "mul":
            if col(2 * NL, rA, rB) > COL_LIMIT:
                self.norm()
                rA = self.rA
            self._need(col(2 * NL, rA, rB) <= COL_LIMIT, f"mul {rA} {rB}")
            out = reduced
        elif name in ("mulfq", "fqmul"):
            if col(NL, rA, rB) > COL_LIMIT:
                self.norm()
                rA = self.rA
            self._need(col(NL, rA, rB) <= COL_LIMIT, f"{name} {rA} {rB}")
            out = reduced
        elif name in ("sqr", "fqsqr"):
            t, u, d = r_add(rA, rA), r_sub(rA, rA), (2 * rA[0], 2 * rA[1])
            if max(col(NL, t, u), col(NL, rA, d)) > COL_LIMIT or mag(t) > LIMB_MAG:
                self.norm()
                rA = self.rA
            out = reduced
        elif name in ("add", "sub", "rsub"):
            f = {"add": r_add, "sub": r_sub, "rsub": lambda x, y: r_sub(y, x)}[name]
            if mag(f(rA, rB)) > LIMB_MAG:
                self.norm()
                rA = self.rA
            out = f(rA, rB)
            self._need(mag(out) <= LIMB_MAG, f"{name} {rA} {rB}")
        elif name == "dbl":
            if 2 * mag(rA) > LIMB_MAG:
                self.norm()
                rA = self.rA
            out = (2 * rA[0], 2 * rA[1])
        elif name == "neg":
            out = r_neg(rA)
        elif name == "negc1":
            out = r_hull(rA, r_neg(rA))
        elif name == "mulxi":
            if mag(r_mulxi(rA)) > LIMB_MAG:
                self.norm()
                rA = self.rA
            out = r_mulxi(rA)
        elif name == "norm":
            out = reduced
        elif name == "redn":
            # the quotient estimate reads the top limb: it must hold the representative (|top| < 2^31) and nothing else
            self._need(vA <= V_CAP and mag(rA) <= LIMB_MAG, f"redn {vA} {rA}")
            out = reduced
        else:
            raise ValueError(name)
        self._raw_call(direct or name)
        self.vA = v_out
        self.rA = out
        return self

    def _bin(self, name, y):
        if y.kind == "home" and name in ("add", "sub", "rsub"):
            return self.call(name, self.r_of(y), direct=f"{name}_h{y.idx}", vB=self.v_of(y))
        self._B(y)
        return self.call(name, self.r_of(y), vB=self.v_of(y))

    def to(self, dst):
        rA = self.rA if self.rA is not None else self.UNKNOWN
        if self.key(dst) in self.norm_keys:
            if rA[0] < self.STORED_NORM[0] or rA[1] > self.STORED_NORM[1]:
                self.norm()
        elif mag(rA) > STORE_MAG:
            self.norm()
        self.wait()
        self.store(A0, dst)
        self.slot_r[self.key(dst)] = self.rA if self.rA is not None else self.UNKNOWN
        self._need(self.vA <= V_CAP, f"value bound {self.vA} p at store")
        self.slot_v[self.key(dst)] = self.vA
        self.max_v = max(self.max_v, self.vA)
        self.tagA = dst
        if self.tagB is dst:
            self.tagB = None
        return self


class _PhaseList(list):
    """The builder's section list: remembers in which phase (Miller loop / final exponentiation) a section was added."""

    def __init__(self, kb):
        super().__init__()
        self.kb = kb

    def append(self, e):
        self.kb.section_phase[id(e)] = self.kb._phase
        super().append(e)


class KernelBuilder3(KP.KernelBuilder):
    F = [LDS(i, f"F{i}") for i in range(6)]
    R = [LDS(6, "RX"), LDS(7, "RY"), AGPR(9, "RZ")]          # RZ shares AGPR 9 with the Fq-inversion base (R is dead by then)
    SCALE = AGPR(11, "scale")
    QX, QY, PX, PY = AGPR(0, "QX"), AGPR(1, "QY"), AGPR(2, "PX"), AGPR(3, "PY")
    SX, SY = AGPR(4, "SX"), AGPR(5, "SY")
    LINE = [AGPR(6, "La"), AGPR(7, "Lb"), AGPR(8, "Lc")]
    FQINV_BASE = AGPR(9, "fqinv_base")
    BOP = [AGPR(i, f"B{i}") for i in (0, 1, 2, 3, 4, 5)]      # fq12_mul operand copy (final exponentiation only)

    PAIR_SLOT0 = GLOB_TMP0 + 8          # scratch slots of pair j: PAIR_SLOT0 + 7 j + {PX, PY, QX, QY, RX, RY, RZ}

    def __init__(self, do_miller=True, do_fexp=True, track=False, multi=False, helper=False):
        """helper: the batched public helpers of the reference on Fq12 batches -- MyFq12 `Mul`, frobenius_map_native
        (final_exp_native.rs:17-54), pow_native (:56-84) -- selected at run time by the kernel's `k` argument."""
        if helper:
            do_miller, do_fexp, track, multi = False, True, False, False
        super().__init__(do_miller, do_fexp, track)
        self.multi = multi
        self.helper = helper
        self.labels = {n: f"L1_{n}_%=" for n in L1V3_NAMES}
        for op in ("add", "sub", "rsub"):
            for i in range(N_HOME):
                self.labels[f"{op}_h{i}"] = f"L1_{op}_h{i}_%="

    _phase = "miller"

    def new_prog(self, temps, phase=None):
        if not hasattr(self, "l2_bodies"):
            self.l2_bodies, self.l2_exit, self.l2_maxv, self.l2_phase = {}, {}, {}, {}
        e = Emitter()
        p = Prog3(e, self.labels)
        p.set_temps(temps)
        p.norm_keys = self.norm_keys(phase or self._phase)
        p.homes_free = USE_MUL6 and not any(t.kind == "home" for t in temps)
        if self._cold:
            p.INLINE_SET = Prog3.INLINE_SET[:6]          # routines that run once per pairing call norm / mulxi (code size)
        return e, p

    NORM_CONTRACT = bool(int(os.environ.get("KGEN3_NORM_CONTRACT", "1")))

    def norm_keys(self, phase):
        """Slots that hold normalised values on every routine boundary of `phase` (Prog3.norm_keys)."""
        if not self.NORM_CONTRACT:
            return frozenset()
        keys = [Prog3.key(s_) for s_ in self.F]
        if phase == "fexp":
            keys += [Prog3.key(s_) for s_ in self.BOP] + [("globdyn", i) for i in range(6)]
        return frozenset(keys)

    COLD = ("L2_inv", "L2_frob1", "L2_frob2", "L2_frob3", "L2_dblfirst", "L2_addmul_last", "L2_descale", "L2_fqinv")
    _cold = False

    def l2_routine(self, name, body, temps):
        """Also records the value bounds (units of p) the routine leaves in every non-temporary slot, given that all
        its inputs were below V_STORE p: the basis of the inductive certification in certify_values()."""
        self._cold = name in self.COLD
        p = super().l2_routine(name, body, temps)
        self._cold = False
        tk = {Prog3.key(t) for t in temps}
        self.l2_bodies[name] = (body, temps)
        self.l2_phase[name] = self._phase
        self.l2_exit[name] = {k: v for k, v in p.slot_v.items() if k not in tk}
        self.l2_maxv[name] = p.max_v
        return p

    def miller_temps(self, extra=(), no_homes=False):
        """Fast temporaries of the Miller-loop routines: homes 0..6, AGPR 10 (11 when no scale is tracked),
        plus routine-specific dead slots; global scratch slots only as overflow."""
        if no_homes:                     # routines built on the fused Fq6 multiplication (all eight home blocks are its workspace)
            return [AGPR(10)] + ([] if self.track else [AGPR(11)]) + [AGPR(i) for i in extra] + [GLOB(GLOB_TMP0 + i) for i in range(8)]
        return ([HOME(i) for i in range(8)] + [AGPR(10)] + ([] if self.track else [AGPR(11)]) + [AGPR(i) for i in extra]
                + [GLOB(GLOB_TMP0 + i) for i in range(8)])

    def fexp_temps(self, no_homes=False):
        # LDS 6,7 ; homes ; AGPR 6..11 (0..5 hold the multiplication operand)
        # fastest first: home registers, then AGPR slots (80 cycles either way), then LDS (a slot store costs 130-270 cycles)
        return ([] if no_homes else [HOME(i) for i in range(8)]) + [AGPR(i) for i in (6, 7, 8, 10, 11)] + [LDS(6), LDS(7)] + [GLOB(GLOB_TMP0 + i) for i in range(8)]

    # ---------------------------------------------------------------------------------------------
    def build(self):
        main = Emitter()
        self.prologue(main)
        l1 = {}                                  # one section per leaf routine: their order is chosen below
        for n in L1V3_NAMES:
            l1[n] = Emitter()
            l1[n].label(self.labels[n])
            getattr(L1v3(l1[n]), "r_" + n)()
            l1[n].salu(f"s_setpc_b64 {S_RET1}")
        for op in ("add", "sub", "rsub"):
            for i in range(N_HOME):
                n = f"{op}_h{i}"
                l1[n] = Emitter()
                l1[n].label(self.labels[n])
                L1v3(l1[n]).home_variant(op, i)
                l1[n].salu(f"s_setpc_b64 {S_RET1}")
        self.sections = _PhaseList(self)
        self.section_phase = {}
        self.control_sections = []
        self._phase = "miller"
        if self.do_miller:
            sc = self.SCALE if self.track else None
            # during f^2 the line (AGPR 6..8) and the addition point (AGPR 4, 5) are dead
            self.l2_routine("L2_sqr", lambda p: p.fq12_sqr(self.F), self.miller_temps(extra=(4, 5, 6, 7, 8), no_homes=USE_MUL6))
            self.l2_routine("L2_dblmul", lambda p: (p.dbl_step(self.R, (self.PX, self.PY), self.LINE, scale=sc),
                                                    p.mul_by_034(self.F, *self.LINE)), self.miller_temps(extra=(4, 5)))
            self.l2_routine("L2_dblfirst", lambda p: self._dbl_first(p), self.miller_temps())
            def addmul(p, update):
                p.add_step(self.R, (self.SX, self.SY), (self.PX, self.PY), self.LINE, scale=sc, update=update)
                glob = [t for t in p.free_tmp if t.kind == "glob"]
                p.free_tmp = [t for t in p.free_tmp if t.kind != "glob"] + [self.SX, self.SY] + glob    # S is dead now
                p.mul_by_235(self.F, *self.LINE)

            self.l2_routine("L2_addmul", lambda p: addmul(p, True), self.miller_temps())
            self.l2_routine("L2_addmul_last", lambda p: addmul(p, False), self.miller_temps())
            if self.track:
                self.l2_routine("L2_fqinv", self._fq_inv, self.miller_temps())
                self.l2_routine("L2_descale", self._descale, self.miller_temps())
                self.l2_routine("L2_sqscale", lambda p: p.A(self.SCALE).sqr().to(self.SCALE), self.miller_temps())
        self._phase = "fexp"
        if self.do_fexp:
            if not (self.do_miller and self.track):
                self.l2_routine("L2_fqinv", self._fq_inv, self.fexp_temps())
            self.l2_routine("L2_cyc", lambda p: p.fq12_cyc_sqr(self.F), self.fexp_temps())
            self.l2_routine("L2_redF", self._reduce_f, self.fexp_temps())
            self._mulG_routines()
            for k in (1, 2, 3):
                self.l2_routine(f"L2_frob{k}", lambda p, k=k: self._frobenius(p, k), self.fexp_temps())
            self.l2_routine("L2_inv", self._fq12_inv, self.fexp_temps())
            self.l2_routine("L2_stG", lambda p: [p.A(self.F[i]).to(GlobDyn(i)) for i in range(6)], self.fexp_temps())
            self.l2_routine("L2_ldG", lambda p: self.batch_load_globdyn(p.e, p, range(6), self.F), self.fexp_temps())
            self.l2_routine("L2_ldGc", lambda p: (self.batch_load_globdyn(p.e, p, range(6), self.F),
                                                  [p.A(self.F[i]).neg().to(self.F[i]) for i in (1, 3, 5)]), self.fexp_temps())
            self.l2_routine("L2_conjF", lambda p: [p.A(self.F[i]).neg().to(self.F[i]) for i in (1, 3, 5)], self.fexp_temps())
            self._powx_routine()
            if self.helper:
                for k in range(4, 12):
                    self.l2_routine(f"L2_frob{k}", lambda p, k=k: self._frobenius(p, k), self.fexp_temps())
                self.l2_routine("L2_sqrF", lambda p: p.fq12_sqr(self.F), self.fexp_temps(no_homes=USE_MUL6))
        self._phase = "miller"              # the main program only touches F
        self.main_body(main)
        # Layout: s_call_b64 / s_branch reach +-128 KB.  The leaf routines (called from everywhere) and the main
        # control code sit in the middle, the L2 routines are split around them by size.
        # Miller-loop routines in front of the block, final-exponentiation routines behind it, the small ones (called from
        # the main program) nearest to the middle, the big cold ones (Fq12 inversion) at the far ends
        size = {id(e): len(e.finalize()) for e in self.sections}
        first = sorted([e for e in self.sections if self.section_phase[id(e)] == "miller"], key=lambda e: -size[id(e)])
        second = sorted([e for e in self.sections if self.section_phase[id(e)] == "fexp"], key=lambda e: size[id(e)])
        tot = lambda lst: sum(size[id(e)] for e in lst)
        while second and tot(second) - tot(first) > size[id(second[-1])]:      # one-phase kernels: balance the two sides
            first.insert(0, second.pop())
        while first and tot(first) - tot(second) > size[id(first[0])]:
            second.append(first.pop(0))
        main.salu(f"s_branch {self.lab('L_exit_hop')}")   # main is not the last section; the end is out of reach in one hop
        hop = Emitter()
        hop.label(self.lab("L_exit_hop"))
        hop.salu(f"s_branch {self.lab('L_exit')}")
        tail = Emitter()
        tail.label(self.lab("L_exit"))
        # leaf routines nobody calls are dropped; the others are ordered by where their callers sit: routines called only
        # from the sections in front of the block come first, those called only from behind last
        def calls(secs_, lbl):
            return sum(1 for e_ in secs_ for it in e_.ins if it["text"].startswith("s_call_b64") and it["text"].endswith(lbl))
        front, back = first + [main] + self.control_sections, second
        order = []
        for n, e_ in l1.items():
            nf, nb = calls(front, self.labels[n]), calls(back, self.labels[n])
            if nf + nb:
                order.append((nb / (nf + nb), -len(e_.ins) if nb <= nf else len(e_.ins), n))
        order.sort()
        out = []
        for e in [self._pro] + first + [main] + self.control_sections + [l1[n] for _, _, n in order] + [hop] + second + [tail]:
            out.extend(e.finalize())
        out = [".p2align 3"] + align_code(out) if ALIGN_CODE else out
        worst = max_branch_distance(out)
        assert worst < 131072 - 512, f"branch of {worst} bytes: s_call_b64 / s_branch reach +-128 KB (re-balance the layout)"
        return out

    # ------------------------------------------------------------------ value-bound certification
    # Limb bounds are closed per routine (every store enforces STORE_MAG, every multiplication its column sums).
    # VALUE bounds cross routine boundaries: each routine is generated assuming its inputs are below V_STORE p and
    # the tracker only needs "value <= K_RP * limb interval" (the top limb).  certify_values() replays the real
    # call sequences (data independent: NAF digits only) with the routines' own transfer functions and checks
    # that every routine still generates IDENTICAL code and passes all its checks under the true entry bounds.
    @staticmethod
    def _grid(v):
        """Round a bound up to a coarse grid (sound, and makes the memo hit)."""
        if v <= 1.0:
            return 1.0
        s_ = 2.0 ** (math.floor(math.log2(v)) - 3)
        return math.ceil(v / s_) * s_

    @staticmethod
    def _norm_text(lines):
        return [re.sub(r"_\d+_\d+_%=", "_%=", l_) for l_ in lines]      # per-instance loop label suffixes

    def _eval(self, name, state, body=None, temps=None):
        """Abstract run of L2 routine `name` from `state` (slot key -> bound): returns (exit bounds, max stored)."""
        if body is None:
            body, temps = self.l2_bodies[name]
        memo = self._memo.setdefault(name, [])
        for reads, ex, mv in memo:
            if all(self._grid(state.get(k, 2.0)) == v for k, v in reads.items()):
                return ex, mv
        self._cold = name in self.COLD
        e, p = self.new_prog(temps, phase=self.l2_phase.get(name, "fexp"))
        self._cold = False
        p.entry_v = {k: self._grid(v) for k, v in state.items()}
        p.default_v = 2.0                       # never-written slots hold converted inputs
        body(p)
        ref = self._ref_text.get(name)
        if ref is not None:
            assert self._norm_text(e.finalize()) == ref, f"{name}: code depends on the value bounds"
        tk = {Prog3.key(t) for t in temps}
        ex = {k: v for k, v in p.slot_v.items() if k not in tk}
        memo.append((dict(p.read_keys), ex, p.max_v))
        return ex, p.max_v

    def certify_values(self, k_pairs=1):
        """Replays the kernel's L2 call sequence on value bounds.  Returns a report dict; raises on any violation."""
        self._memo, self._ref_text = {}, {}
        for name, (body, temps) in self.l2_bodies.items():       # the shipped code of each routine body
            self._cold = name in self.COLD
            e, p = self.new_prog(temps, phase=self.l2_phase[name])
            self._cold = False
            body(p)
            self._ref_text[name] = self._norm_text(e.finalize())
        st, worst, calls = {}, 0.0, 0
        fk = [Prog3.key(s_) for s_ in self.F]

        seq = []                          # labels in call order: cross-checked against the simulator's call log

        def run(name, label=None, **kw):
            nonlocal worst, calls
            ex, mv = self._eval(name, st, **kw)
            st.update(ex)
            worst = max(worst, mv)
            calls += 1
            seq.append(label or name)
            if name in ("L2_descale", "L2_inv"):
                seq.append("L2_fqinv")           # nested: the Fq inversion (fixed exponent, outputs below 2 p)

        report = {}
        if self.do_miller:
            # converted inputs (< 2 p): P, Q, R = (Q, 1), scale = 1; S = +-Q on the non-zero digits
            run("L2_dblfirst")
            for _ in range(k_pairs - 1):
                run("L2_dblmul")
            for i in range(63, -1, -1):
                if i != 63:
                    run("L2_sqr")
                    if self.track:
                        run("L2_sqscale")
                    for _ in range(k_pairs):
                        run("L2_dblmul")
                if SIX_U_PLUS_2_NAF[i] != 0:
                    for _ in range(k_pairs):
                        st[Prog3.key(self.SX)] = st[Prog3.key(self.SY)] = 2.0
                        run("L2_addmul")
            for _ in range(k_pairs):                                    # per pair: + Q1, then - Q2
                for s_ in (self.SX, self.SY, self.QX, self.QY):         # Frobenius images: reduction outputs
                    st[Prog3.key(s_)] = 2.0
                run("L2_addmul")
                run("L2_addmul_last")
            if self.track:
                run("L2_descale")
            report["miller_f_out"] = max(st[k] for k in fk)
        if self.do_fexp:
            G = {}

            def mul_body(p):
                p.fq12_mul(self.F, self.BOP)

            for op in self.fexp_trace:
                if op[0] == "st":
                    G[op[1]] = [st.get(k, 2.0) for k in fk]
                    seq.append("L2_stG")
                elif op[0] == "ld":
                    for k, v in zip(fk, G[op[1]]):
                        st[k] = v
                    seq.append("L2_ldGc" if op[2] else "L2_ldG")
                elif op[0] == "mul":
                    for b, v in zip(self.BOP, G[op[1]]):
                        st[Prog3.key(b)] = v
                    run("L2_mul_body", label="L2_mulGc" if op[2] else "L2_mulG", body=mul_body, temps=self.fexp_temps(no_homes=USE_MUL6))
                elif op[0] == "powx":
                    j = op[1]

                    def mul_by(reg, conj=False):
                        for b_, v in zip(self.BOP, G[reg]):
                            st[Prog3.key(b_)] = v
                        run("L2_mul_body", label="L2_mulGc" if conj else "L2_mulG", body=mul_body, temps=self.fexp_temps(no_homes=USE_MUL6))

                    def store(reg):
                        G[reg] = [st[k] for k in fk]
                        seq.append("L2_stG")

                    run("L2_redF"); store(j)
                    run("L2_cyc"); run("L2_cyc"); store(G_B4)
                    for reg, src in ((G_POW[5], j), (G_POW[9], G_B4), (G_POW[13], G_B4)):
                        mul_by(src)
                        if RED_POWERS:
                            run("L2_redF")
                        store(reg)
                    top = X_DIGITS[-1]
                    if top != 13:
                        for k, v in zip(fk, G[j if top == 1 else G_POW[top]]):
                            st[k] = v
                        seq.append("L2_ldG")
                    xd = X_DIGITS[:-1]
                    red = x_red_mask(xd)
                    for d in range(len(xd) - 1, -1, -1):
                        run("L2_cyc")
                        if xd[d] != 0:
                            mul_by(j if abs(xd[d]) == 1 else G_POW[abs(xd[d])], conj=xd[d] < 0)
                        elif red >> d & 1:
                            run("L2_redF")
                else:
                    run(op[1])
            report["fexp_f_out"] = max(st[k] for k in fk)
        report["max_stored"] = worst
        report["calls"] = calls
        report["sequence"] = seq
        assert worst <= V_CAP
        return report

    def certify_helper(self):
        """Value bounds of the helper kernel: its loops are data dependent (NAF of the caller's exponent), so the bounds
        must hold for ANY order of squarings and multiplications: iterate the routines' transfer functions from the
        converted inputs to a fixed point and check it (and every intermediate store) against the cap."""
        self._memo, self._ref_text = {}, {}
        fk = [Prog3.key(s_) for s_ in self.F]
        st, worst = {}, 0.0

        def run(name, **kw):
            nonlocal worst
            ex, mv = self._eval(name, st, **kw)
            st.update(ex)
            worst = max(worst, mv)

        def mul_body(p):
            p.fq12_mul(self.F, self.BOP)

        def mul_by(g):
            for b_, v in zip(self.BOP, g):
                st[Prog3.key(b_)] = v
            run("L2_mul_body", body=mul_body, temps=self.fexp_temps(no_homes=USE_MUL6))

        g0 = [2.0] * 6                                    # a (converted input)
        run("L2_inv")
        run("L2_redF")
        g1 = [st[k] for k in fk]                          # 1 / a, reduced
        for k in range(1, 12):
            for k_ in fk:
                st[k_] = 2.0
            run(f"L2_frob{k}")
        for k_ in fk:
            st[k_] = 2.0
        mul_by(g0)                                        # Mul: one multiplication of converted inputs
        hi = 2.0
        for _ in range(8):                                # pow loop: (square, reduce, [multiply by a or 1/a, reduce]) in any order
            for k_ in fk:
                st[k_] = hi
            run("L2_sqrF")
            run("L2_redF")
            after = max(st[k] for k in fk)
            for g in (g0, g1):
                for k_ in fk:
                    st[k_] = max(hi, after)
                mul_by(g)
                run("L2_redF")
                after = max(after, max(st[k] for k in fk))
            if after <= hi:
                break
            hi = after
        else:
            raise AssertionError("helper kernel: value bounds do not reach a fixed point")
        assert worst <= V_CAP
        return {"fixed_point": hi, "max_stored": worst}

    def _reduce_f(self, p):
        """F <- the same residues with representatives back in (-0.01 p, 1.01 p) (L1 redn).  Every cyclotomic squaring
        roughly doubles the representative (3t - 2z), so the x-power loop calls this after RED_RUN squarings in a row
        without a multiplication (see x_red_mask), and on the base and its stored powers."""
        for k in range(6):
            p.A(self.F[k]).redn().to(self.F[k])

    def _powx_routine(self):
        """F <- F^BN_X for cyclotomic F (base b = F on entry, S_GBASE = its scratch register): the X_DIGITS schedule.
        Same value as pow_native(a, [BN_X]) (final_exp_native.rs:56-84) for unitary a."""
        e = Emitter()
        L = self.lab

        def c2(name):
            e.salu(f"s_call_b64 {S_RET2}, {L(name)}")

        def greg(j):
            e.salu(f"s_mul_i32 s{S_GBASE}, s{S_GSTRIDE}, {6 * j}")

        e.label(L("L3_powx"))
        e.salu(f"s_mov_b32 s{S_PB}, s{S_GBASE}")
        # reduced copies of the base and its powers: every multiplication by one of them contracts the accumulator
        c2("L2_redF"); c2("L2_stG")                                   # G[base] = b
        c2("L2_cyc"); c2("L2_cyc"); greg(G_B4); c2("L2_stG")          # b^4
        e.salu(f"s_mov_b32 s{S_GBASE}, s{S_PB}")
        red = (lambda: c2("L2_redF")) if RED_POWERS else (lambda: None)
        c2("L2_mulG"); red(); greg(G_POW[5]); c2("L2_stG")            # b^5 = b^4 b
        greg(G_B4); c2("L2_mulG"); red(); greg(G_POW[9]); c2("L2_stG")             # b^9 = b^5 b^4
        greg(G_B4); c2("L2_mulG"); red(); greg(G_POW[13]); c2("L2_stG")            # b^13 = b^9 b^4
        top = X_DIGITS[-1]
        if top == 1:
            e.salu(f"s_mov_b32 s{S_GBASE}, s{S_PB}")
        else:
            greg(G_POW[top])
        if top != 13:
            c2("L2_ldG")
        e.salu(f"s_mov_b32 s{S_J}, {self.x_top - 1}")
        e.label(L("L3_powx_loop"))
        c2("L2_cyc")
        e.salu(f"s_bitcmp1_b64 {S_XNAF_NZ}, s{S_J}")
        e.salu(f"s_cbranch_scc0 {L('L3_powx_zero')}")
        # select the power: S_GBASE <- register of b^(X_POWERS[idx])
        e.salu(f"s_mov_b32 s{S_GBASE}, s{S_PB}")
        for bit, mask in ((0, S_XIDX0), (1, S_XIDX1)):
            e.salu(f"s_bitcmp1_b64 {mask}, s{S_J}")
            e.salu(f"s_cselect_b32 s{S_TMP0}, {1 << bit}, 0")
            e.salu(f"s_{'mov' if bit == 0 else 'or'}_b32 s{S_TMP1}, s{S_TMP0}" + (f", s{S_TMP1}" if bit else ""))
        e.salu(f"s_cmp_eq_u32 s{S_TMP1}, 0")
        e.salu(f"s_cbranch_scc1 {L('L3_powx_sel')}")
        # registers of b^5, b^9, b^13 are consecutive: G_POW[5] + (idx - 1)
        e.salu(f"s_add_u32 s{S_TMP1}, s{S_TMP1}, {G_POW[5] - 1}")
        e.salu(f"s_mul_i32 s{S_TMP1}, s{S_TMP1}, 6")
        e.salu(f"s_mul_i32 s{S_GBASE}, s{S_GSTRIDE}, s{S_TMP1}")
        e.label(L("L3_powx_sel"))
        e.salu(f"s_bitcmp1_b64 {S_XNAF_NEG}, s{S_J}")
        e.salu(f"s_cbranch_scc1 {L('L3_powx_neg')}")
        c2("L2_mulG")
        e.salu(f"s_branch {L('L3_powx_next')}")
        e.label(L("L3_powx_neg"))
        c2("L2_mulGc")
        e.salu(f"s_branch {L('L3_powx_next')}")
        e.label(L("L3_powx_zero"))
        e.salu(f"s_bitcmp1_b64 {S_XNAF_RED}, s{S_J}")
        e.salu(f"s_cbranch_scc0 {L('L3_powx_next')}")
        c2("L2_redF")
        e.label(L("L3_powx_next"))
        e.salu(f"s_sub_u32 s{S_J}, s{S_J}, 1")
        e.salu(f"s_cbranch_scc0 {L('L3_powx_loop')}")
        e.salu(f"s_setpc_b64 {S_RET3}")
        self.control_sections.append(e)           # control code: placed next to the main program (it calls L2 routines of both halves)

    def batch_load_globdyn(self, e, p, ks, dests):
        """dests[i] <- scratch slot (S_GBASE + ks[i]): all global loads issued back to back into landing registers
        (blocks A, B and the home registers -- every temporary is dead at a routine boundary), ONE wait, then the
        stores.  A dependent load->wait->store round trip per slot costs ~2 us each."""
        land = [A0, B0] + [HOME0 + SLOT_DW * i for i in range(N_HOME)]
        assert len(ks) <= len(land)
        p.reset_tags()
        p.wait()
        for n, k in enumerate(ks):
            e.salu(f"s_mul_i32 s{S_TMP0}, s{S_GSTRIDE}, {k}")
            e.salu(f"s_add_u32 s{S_TMP0}, s{S_TMP0}, s{S_GBASE}")
            e.salu(f"s_add_u32 s62, s64, s{S_TMP0}")
            e.salu("s_addc_u32 s63, s65, 0")
            for c in range(N_CHUNK):
                r = land[n] + 4 * c
                e.emit(f"global_load_dwordx4 v[{r}:{r + 3}], v{V_GOFF}, {S_GADDR} offset:{16 * c}", kind="vmem", vw=range(r, r + 4))
        e.raw("s_waitcnt vmcnt(0)")
        for n, d in enumerate(dests):
            r0 = land[n]
            if d.kind == "agpr":
                for i in range(SLOT_DW):
                    e.emit(f"v_accvgpr_write_b32 a{SLOT_DW * d.idx + i}, v{r0 + i}")
            elif d.kind == "lds":
                for c in range(N_CHUNK):
                    base, off = p._lds_addr(d, c)
                    e.emit(f"ds_write_b128 v{base}, v[{r0 + 4 * c}:{r0 + 4 * c + 3}] offset:{off}", kind="lds")
            elif d.kind == "home":
                h0 = HOME0 + SLOT_DW * d.idx
                if h0 != r0:
                    for i in range(SLOT_DW):
                        e.emit(f"v_mov_b32_e32 v{h0 + i}, v{r0 + i}", vw=[h0 + i])
            else:
                raise ValueError(d.kind)
            p.slot_r.pop(p.key(d), None)
        p.reset_tags()

    def _mulG_routines(self):
        e, p = self.new_prog(self.fexp_temps(no_homes=USE_MUL6))
        e.label(self.lab("L2_mulGc"))
        self.batch_load_globdyn(e, p, range(6), self.BOP)
        for i in (1, 3, 5):
            p.A(self.BOP[i]).neg().to(self.BOP[i])
        p.wait()
        e.salu(f"s_branch {self.lab('L2_mul_body')}")
        e.label(self.lab("L2_mulG"))
        p.reset_tags()
        self.batch_load_globdyn(e, p, range(6), self.BOP)
        e.label(self.lab("L2_mul_body"))
        p.reset_tags()
        p.fq12_mul(self.F, self.BOP)
        p.wait()
        e.salu(f"s_setpc_b64 {S_RET2}")
        self.sections.append(e)

    # ---------------------------------------------------------------------------------------------
    def prologue(self, main):
        e = Emitter()
        self._pro = e
        e.salu(f"s_mov_b64 {S_G1}, %0")
        e.salu(f"s_mov_b64 {S_G2}, %1")
        e.salu(f"s_mov_b64 {S_FIN}, %2")
        e.salu(f"s_mov_b64 {S_OUT}, %3")
        e.salu(f"s_mov_b32 s{S_N}, %4")
        e.salu(f"s_mov_b32 s{S_K}, %5")
        e.salu(f"s_mov_b32 s{S_GSTRIDE}, %7")
        e.salu(f"s_mov_b64 {S_STATUS}, %8")
        e.salu(f"s_mov_b32 s{S_ITEM}, %10")
        e.salu(f"s_mov_b32 s{S_GRID}, %11")
        # scratch base of this workgroup: scratch + block * 256 * 80 ; lane offset = tid * 80
        e.salu(f"s_mul_i32 s{S_TMP0}, %10, {256 * 4 * SLOT_DW}")
        e.salu(f"s_mov_b64 {S_SCRATCH}, %6")
        e.salu(f"s_add_u32 s64, s64, s{S_TMP0}")
        e.salu("s_addc_u32 s65, s65, 0")
        e.emit(f"v_mul_u32_u24_e32 v{V_GOFF}, {4 * SLOT_DW}, %9", vw=[V_GOFF])
        e.emit(f"v_lshlrev_b32_e32 v{V_LDS}, 4, %9", vw=[V_LDS])
        e.emit(f"v_add_u32_e32 v{V_LDS + 1}, 0x10000, v{V_LDS}", vw=[V_LDS + 1])
        e.emit(f"v_add_u32_e32 v{V_LDS + 2}, 0x20000, v{V_LDS}", vw=[V_LDS + 2])
        e.emit(f"v_mov_b32_e32 v{V_TID}, %9", vw=[V_TID])
        for i in range(NL):
            e.salu(f"s_mov_b32 s{S_P + i}, 0x{P_L[i]:x}")
        e.salu(f"s_mov_b32 s{S_N0}, 0x{N0P:x}")
        nz, neg = naf_masks(SIX_U_PLUS_2_NAF[:64])
        e.salu(f"s_mov_b32 s68, 0x{nz & 0xFFFFFFFF:x}")
        e.salu(f"s_mov_b32 s69, 0x{nz >> 32:x}")
        e.salu(f"s_mov_b32 s70, 0x{neg & 0xFFFFFFFF:x}")
        e.salu(f"s_mov_b32 s71, 0x{neg >> 32:x}")
        xd = list(X_DIGITS[:-1])                       # the top digit is the initial value of the accumulator
        nz, neg = naf_masks([(d > 0) - (d < 0) for d in xd])
        self.x_top = len(xd)
        idx = [X_POWERS.index(abs(d)) if d else 0 for d in xd]
        red = x_red_mask(xd)
        for reg, val in ((72, nz), (74, neg), (48, red), (50, sum((i & 1) << j for j, i in enumerate(idx))),
                         (52, sum((i >> 1) << j for j, i in enumerate(idx)))):
            e.salu(f"s_mov_b32 s{reg}, 0x{val & 0xFFFFFFFF:x}")
            e.salu(f"s_mov_b32 s{reg + 1}, 0x{val >> 32:x}")
        e.salu(f"s_lshl_b32 s{S_NSTRIDE}, s{S_N}, 3")
        e.salu(f"s_add_u32 s{S_NITEMS}, s{S_N}, 255")
        e.salu(f"s_lshr_b32 s{S_NITEMS}, s{S_NITEMS}, 8")
        e.emit(f"v_mov_b32_e32 v{V_FLAG}, 0", vw=[V_FLAG])
        e.salu(f"s_branch {self.lab('L_main')}")

    # ---------------------------------------------------------------------------------------------
    def io_load_fq(self, e, reg0):
        for l in range(4):
            e.emit(f"global_load_dwordx2 v[{reg0 + 2 * l}:{reg0 + 2 * l + 1}], v{V_IDX8}, {S_IOADDR}", kind="vmem", vw=[reg0 + 2 * l, reg0 + 2 * l + 1])
            self.io_walk_next(e)

    def io_store_fq(self, e, reg0):
        for l in range(4):
            e.emit(f"global_store_dwordx2 v{V_IDX8}, v[{reg0 + 2 * l}:{reg0 + 2 * l + 1}], {S_IOADDR}", kind="vmem")
            self.io_walk_next(e)

    def zero_block(self, e, blk, n=SLOT_DW):
        for i in range(n):
            e.emit(f"v_mov_b32_e32 v{blk + i}, 0", vw=[blk + i])

    def one_into_A(self, e):
        w = to_limbs(mont3(1))
        for i in range(NL):
            e.emit(f"v_mov_b32_e32 v{A0 + i}, 0x{w[i]:x}", vw=[A0 + i])
        self.zero_block(e, A0 + NL, NL)

    def cvt_call(self, e, name):
        e.salu(f"s_call_b64 {S_RET1}, {self.labels[name]}")

    def io_load_fq2_into_A(self, e, p, c1_present=True):
        """Loads c0 (and c1) of the SoA batch at the walking address, converts to internal form -> block A."""
        if c1_present:
            self.io_load_fq(e, 20)                 # c0 packed -> v[20:27] (block B as staging)
            self.io_load_fq(e, A0)                 # c1 packed -> v[0:7]
            e.raw("s_waitcnt vmcnt(0)")
            self.cvt_call(e, "cvtin")              # A.c0 <- internal(c1)
            for i in range(NL):
                e.emit(f"v_mov_b32_e32 v{A0 + NL + i}, v{A0 + i}", vw=[A0 + NL + i])
            for i in range(8):
                e.emit(f"v_mov_b32_e32 v{A0 + i}, v{20 + i}", vw=[A0 + i])
            self.cvt_call(e, "cvtin")
        else:
            self.io_load_fq(e, A0)
            e.raw("s_waitcnt vmcnt(0)")
            self.cvt_call(e, "cvtin")
            self.zero_block(e, A0 + NL, NL)
        p.set_A_fresh()
        p.tagB = None

    def _dbl_first(self, p):
        p.dbl_step(self.R, (self.PX, self.PY), self.LINE, scale=None)
        p.mov(self.F[0], self.LINE[0])
        p.mov(self.F[3], self.LINE[1])
        p.mov(self.F[4], self.LINE[2])
        p.wait()
        self.zero_block(p.e, A0)
        p.set_A_fresh()
        for k in (1, 2, 5):
            p.store(A0, self.F[k])
            p.slot_r[p.key(self.F[k])] = R_NORM

    def _fq_inv(self, p):
        """A.c0 <- A.c0^(p-2) (Fermat; fixed exponent).  Input/outputs normalised."""
        e = p.e
        base = self.FQINV_BASE
        p.wait()
        p.store(A0, base)
        p.load(B0, base)
        p.tagA = p.tagB = None
        # exponent p - 2, canonical 32-bit words as literals in SGPR s61 per word
        ex = P_INT - 2
        words = [(ex >> (32 * i)) & 0xFFFFFFFF for i in range(8)]
        self._fqinv_uid = getattr(self, "_fqinv_uid", 0) + 1          # deterministic per-instance label suffix
        uid = self._fqinv_uid
        for limb in range(7, -1, -1):
            top = 28 if limb == 7 else 31
            lbl = self.lab(f"L_fqinv_{limb}_{uid}")
            skip = self.lab(f"L_fqinv_skip_{limb}_{uid}")
            e.salu(f"s_mov_b32 s{S_TMP1}, 0x{words[limb]:x}")
            e.salu(f"s_mov_b32 s{S_TMP0}, {top}")
            e.label(lbl)
            e.salu(f"s_call_b64 {S_RET1}, {self.labels['fqsqr']}")
            e.salu(f"s_bitcmp1_b32 s{S_TMP1}, s{S_TMP0}")
            e.salu(f"s_cbranch_scc0 {skip}")
            e.salu(f"s_call_b64 {S_RET1}, {self.labels['fqmul']}")
            e.label(skip)
            e.salu(f"s_sub_u32 s{S_TMP0}, s{S_TMP0}, 1")
            e.salu(f"s_cbranch_scc0 {lbl}")
        p.set_A_fresh()

    def _fq2_inv_inline(self, p, src, dst):
        e = p.e
        n0, tmp = p.tmp(), p.tmp()
        p.A(src)
        if mag(p.rA) > 2.0:
            p.norm()
        p._raw_call("fqsqr")
        p.set_A_fresh()
        p.to(n0)
        p.A(src)
        p.wait()
        for i in range(NL):
            e.emit(f"v_mov_b32_e32 v{A0 + i}, v{A0 + NL + i}", vw=[A0 + i])
        rS = p.rA
        p.tagA = None
        p.rA = rS
        if mag(p.rA) > 2.0:
            p.norm()
        p._raw_call("fqsqr")
        p.set_A_fresh()
        p.add(n0).norm()                                         # A.c0 = c0^2 + c1^2
        # zero test needs the canonical representative: convert a copy out (value zero <-> all words zero)
        p.to(tmp)
        p.wait()
        self.cvt_call(e, "cvtout")
        e.emit(f"v_or3_b32 v{V_TID}, v{A0}, v{A0 + 1}, v{A0 + 2}", vw=[V_TID])
        e.emit(f"v_or3_b32 v{V_TID}, v{V_TID}, v{A0 + 3}, v{A0 + 4}", vw=[V_TID])
        e.emit(f"v_or3_b32 v{V_TID}, v{V_TID}, v{A0 + 5}, v{A0 + 6}", vw=[V_TID])
        e.emit(f"v_or_b32_e32 v{V_TID}, v{V_TID}, v{A0 + 7}", vw=[V_TID])
        e.emit(f"v_cmp_eq_u32_e32 vcc, 0, v{V_TID}", w=["vcc"])
        e.emit(f"v_cndmask_b32_e64 v{V_TID}, 0, 1, vcc", r=["vcc"], vw=[V_TID])
        e.emit(f"v_or_b32_e32 v{V_FLAG}, v{V_FLAG}, v{V_TID}", vw=[V_FLAG])
        p.tagA = None
        p.A(tmp)
        p.wait()
        e.salu(f"s_mov_b64 {S_RET3}, {S_RET2}")
        e.salu(f"s_call_b64 {S_RET2}, {self.lab('L2_fqinv')}")
        e.salu(f"s_mov_b64 {S_RET2}, {S_RET3}")
        p.set_A_fresh()
        p.tagB = None
        p.to(tmp)
        p.A(src).mulfq(tmp).conj().to(dst)
        p.rel(n0, tmp)

    # ---------------------------------------------------------------------------------------------
    def main_body(self, e):
        L = self.lab
        e.label(L("L_main"))
        e.label(L("L_item"))
        e.salu(f"s_cmp_ge_u32 s{S_ITEM}, s{S_NITEMS}")
        e.salu(f"s_cbranch_scc1 {L('L_done')}")
        e.salu(f"s_lshl_b32 s{S_TMP0}, s{S_ITEM}, 8")
        e.emit(f"v_add_u32_e32 v{V_IDX}, s{S_TMP0}, v{V_TID}", vw=[V_IDX])
        e.salu(f"s_sub_u32 s{S_TMP1}, s{S_N}, 1")
        e.emit(f"v_min_u32_e32 v{V_IDX8}, s{S_TMP1}, v{V_IDX}", vw=[V_IDX8])
        e.emit(f"v_lshlrev_b32_e32 v{V_IDX8}, 3, v{V_IDX8}", vw=[V_IDX8])
        e.emit(f"v_mov_b32_e32 v{V_FLAG}, 0", vw=[V_FLAG])
        p = Prog3(e, self.labels)
        p.set_temps(self.miller_temps())
        p.norm_keys = self.norm_keys("miller")
        if self.do_miller and self.multi:
            self.miller_main_multi(e, p)
        elif self.do_miller:
            self.miller_main(e, p)
        else:
            self.load_fq12_into_F(e, p, S_FIN)
        if self.helper:
            self.helper_main(e, p)
        elif self.do_fexp:
            self.fexp_main(e, p)
        self.store_out(e, p)
        e.salu(f"s_add_u32 s{S_ITEM}, s{S_ITEM}, s{S_GRID}")
        e.salu(f"s_branch {L('L_item')}")
        e.label(L("L_done"))

    def load_fq12_into_F(self, e, p, ptr):
        """F <- the lane's MyFq12 of the SoA batch at `ptr` (components 0..5 are the c0 parts of w^0..w^5, 6..11 the c1
        parts): two passes over the planes, c0 parts first into AGPR staging (the operand slots, free at that point)."""
        self.io_walk_begin(e, ptr)
        for k in range(6):
            self.io_load_fq(e, A0)
            e.raw("s_waitcnt vmcnt(0)")
            self.cvt_call(e, "cvtin")
            for i in range(NL):
                e.emit(f"v_accvgpr_write_b32 a{NL * k + i}, v{A0 + i}")       # c0 of coefficient k
        for k in range(6):
            self.io_load_fq(e, A0)
            e.raw("s_waitcnt vmcnt(0)")
            self.cvt_call(e, "cvtin")
            for i in range(NL):
                e.emit(f"v_mov_b32_e32 v{A0 + NL + i}, v{A0 + i}", vw=[A0 + NL + i])
            for i in range(NL):
                e.emit(f"v_accvgpr_read_b32 v{A0 + i}, a{NL * k + i}", vw=[A0 + i])
            p.set_A_fresh()
            p.to(self.F[k])
        p.reset_tags()

    # ---------------------------------------------------------------------------------------------
    # batched helpers: k argument = op | power << 8 | naf_len << 16
    OP_MUL, OP_FROB, OP_POW = 0, 1, 2

    def helper_main(self, e, p):
        L = self.lab

        def gsel(j):
            e.salu(f"s_mul_i32 s{S_GBASE}, s{S_GSTRIDE}, {6 * j}")

        def c2(name):
            self.call2(e, name)

        e.salu(f"s_and_b32 s{S_TMP0}, s{S_K}, 0xff")
        e.salu(f"s_cmp_eq_u32 s{S_TMP0}, {self.OP_FROB}")
        e.salu(f"s_cbranch_scc1 {L('L_h_frob')}")
        e.salu(f"s_cmp_eq_u32 s{S_TMP0}, {self.OP_POW}")
        e.salu(f"s_cbranch_scc1 {L('L_h_pow')}")
        # ---- MyFq12 Mul: F holds a; b comes from the g1 pointer
        gsel(0); c2("L2_stG")
        self.load_fq12_into_F(e, p, S_G1)
        gsel(0); c2("L2_mulG")
        e.salu(f"s_branch {L('L_h_done')}")
        # ---- frobenius_map_native(a, power), power = 0..11
        e.label(L("L_h_frob"))
        e.salu(f"s_lshr_b32 s{S_TMP0}, s{S_K}, 8")
        e.salu(f"s_and_b32 s{S_TMP0}, s{S_TMP0}, 0xf")
        for k in range(1, 12):
            e.salu(f"s_cmp_eq_u32 s{S_TMP0}, {k}")
            e.salu(f"s_cbranch_scc0 {L(f'L_h_nf{k}')}")
            c2(f"L2_frob{k}")
            e.salu(f"s_branch {L('L_h_done')}")
            e.label(L(f"L_h_nf{k}"))
        e.salu(f"s_branch {L('L_h_done')}")                      # power 0: identity
        # ---- pow_native(a, exp): NAF digits (int8, least significant first) at the g2 pointer, top digit = +1
        e.label(L("L_h_pow"))
        gsel(0); c2("L2_stG")                                     # G0 = a
        e.salu(f"s_bitcmp1_b32 s{S_K}, 8")                        # bit 8: the NAF has a -1 digit (only then is 1/a formed: the
        e.salu(f"s_cbranch_scc0 {L('L_h_noinv')}")                 # reference divides -- and panics on a = 0 -- only on such a digit)
        c2("L2_inv"); c2("L2_redF")
        gsel(1); c2("L2_stG")                                     # G1 = 1/a   (`res / a` on a -1 digit, final_exp_native.rs:72-75)
        gsel(0); c2("L2_ldG")                                     # res = a (the top digit)
        e.label(L("L_h_noinv"))
        e.salu(f"s_lshr_b32 s{S_J}, s{S_K}, 16")
        e.salu(f"s_sub_u32 s{S_J}, s{S_J}, 2")
        e.salu(f"s_cbranch_scc1 {L('L_h_done')}")                 # a single digit: a^1
        e.label(L("L_h_ploop"))
        c2("L2_sqrF"); c2("L2_redF")            # the loop length is the caller's: representatives are reduced every step
        e.salu(f"s_and_b32 s{S_TMP0}, s{S_J}, 0xfffffffc")
        e.salu(f"s_load_dword s{S_TMP1}, {S_G2}, s{S_TMP0}")
        e.salu(f"s_and_b32 s{S_TMP0}, s{S_J}, 3")
        e.salu(f"s_lshl_b32 s{S_TMP0}, s{S_TMP0}, 3")
        e.raw("s_waitcnt lgkmcnt(0)")
        e.salu(f"s_lshr_b32 s{S_TMP1}, s{S_TMP1}, s{S_TMP0}")
        e.salu(f"s_sext_i32_i8 s{S_TMP1}, s{S_TMP1}")
        e.salu(f"s_cmp_eq_i32 s{S_TMP1}, 0")
        e.salu(f"s_cbranch_scc1 {L('L_h_pnext')}")
        e.salu(f"s_cmp_gt_i32 s{S_TMP1}, 0")
        e.salu(f"s_cselect_b32 s{S_TMP1}, 0, 1")                   # register 0 (a) for +1, 1 (1/a) for -1
        e.salu(f"s_mul_i32 s{S_TMP1}, s{S_TMP1}, 6")
        e.salu(f"s_mul_i32 s{S_GBASE}, s{S_GSTRIDE}, s{S_TMP1}")
        c2("L2_mulG"); c2("L2_redF")
        e.label(L("L_h_pnext"))
        e.salu(f"s_sub_u32 s{S_J}, s{S_J}, 1")
        e.salu(f"s_cbranch_scc0 {L('L_h_ploop')}")
        e.label(L("L_h_done"))
        p.reset_tags()

    def miller_main(self, e, p):
        L = self.lab
        self.io_walk_begin(e, S_G1)
        self.io_load_fq2_into_A(e, p, c1_present=False)          # Px
        p.to(self.PX)
        self.io_load_fq2_into_A(e, p, c1_present=False)          # Py
        p.to(self.PY)
        self.io_walk_begin(e, S_G2)
        self.io_load_fq2_into_A(e, p)                            # Q.x
        p.to(self.QX)
        p.store(A0, self.R[0])
        p.slot_r[p.key(self.R[0])] = R_NORM
        self.io_load_fq2_into_A(e, p)                            # Q.y
        p.to(self.QY)
        p.store(A0, self.R[1])
        p.slot_r[p.key(self.R[1])] = R_NORM
        self.one_into_A(e)
        p.set_A_fresh()
        p.to(self.R[2])
        if self.track:
            p.store(A0, self.SCALE)
        p.reset_tags()
        self.call2(e, "L2_dblfirst")
        e.salu(f"s_mov_b32 s{S_I}, 63")
        e.label(L("L_mloop"))
        e.salu(f"s_cmp_eq_u32 s{S_I}, 63")
        e.salu(f"s_cbranch_scc1 {L('L_mskip')}")
        self.call2(e, "L2_sqr")
        if self.track:
            self.call2(e, "L2_sqscale")
        self.call2(e, "L2_dblmul")
        e.label(L("L_mskip"))
        e.salu(f"s_bitcmp1_b64 {S_NAF_NZ}, s{S_I}")
        e.salu(f"s_cbranch_scc0 {L('L_mnoadd')}")
        p.reset_tags()
        p.mov(self.SX, self.QX)
        p.A(self.QY)
        p.wait()
        e.salu(f"s_bitcmp1_b64 {S_NAF_NEG}, s{S_I}")
        e.salu(f"s_cbranch_scc0 {L('L_mpos')}")
        e.salu(f"s_call_b64 {S_RET1}, {self.labels['neg']}")
        e.label(L("L_mpos"))
        p.tagA = None
        p.to(self.SY)
        self.call2(e, "L2_addmul")
        e.label(L("L_mnoadd"))
        e.salu(f"s_sub_u32 s{S_I}, s{S_I}, 1")
        e.salu(f"s_cbranch_scc0 {L('L_mloop')}")
        xi = (9, 1)
        c = f2pow(xi, (P_INT - 1) // 6)
        c2 = f2mul(c, c)
        c3 = f2mul(c2, c)
        C2, C3 = Const(c2[0], c2[1], "c2"), Const(c3[0], c3[1], "c3")
        # Q1 -> S ; -Q2 (derived from Q1) -> the Q slots, which are dead from here on.  (The sparse multiplication
        # inside L2_addmul uses the S slots as temporaries, so -Q2 must exist before the first call.)
        p.reset_tags()
        p.A(self.QX).conj().mul(C2).to(self.SX)
        p.A(self.QY).conj().mul(C3).to(self.SY)
        p.A(self.SX).conj().mul(C2).to(self.QX)
        p.A(self.SY).conj().neg().mul(C3).to(self.QY)
        self.call2(e, "L2_addmul")
        p.reset_tags()
        p.mov(self.SX, self.QX)
        p.mov(self.SY, self.QY)
        self.call2(e, "L2_addmul_last")
        if self.track:
            self.call2(e, "L2_descale")
        p.reset_tags()

    # ---------------------------------------------------------------------------------------------
    # multi-pairing: shared f, k pairs per lane (multi_miller_loop_native, miller_loop_native.rs:192-282).
    # Pair state (P, Q converted; R projective) lives in scratch and is swapped through the resident slots.
    S_JP = 97                  # pair counter
    S_PHASE = 91               # scratch: loop bookkeeping

    def pair_select(self, e):
        """S_GBASE <- byte offset of pair S_JP's scratch block."""
        e.salu(f"s_mul_i32 s{S_TMP0}, s{self.S_JP}, 7")
        e.salu(f"s_add_u32 s{S_TMP0}, s{S_TMP0}, {self.PAIR_SLOT0}")
        e.salu(f"s_mul_i32 s{S_GBASE}, s{S_TMP0}, s{S_GSTRIDE}")

    def pair_in(self, e, p, with_q):
        """Resident slots <- scratch block of the selected pair: all loads issued back to back, one wait."""
        dests = [self.PX, self.PY] + ([self.QX, self.QY] if with_q else []) + list(self.R)
        srcs = [0, 1] + ([2, 3] if with_q else []) + [4, 5, 6]
        self.batch_load_globdyn(e, p, srcs, dests)

    def pair_out(self, e, p):
        """scratch block of the selected pair <- R (the only state a step changes)."""
        p.reset_tags()
        for k, src in zip((4, 5, 6), self.R):
            p.A(src).to(GlobDyn(k))
        p.reset_tags()

    def pair_loop(self, e, name, body):
        """for S_JP in 0..k-1: body()"""
        L = self.lab
        e.salu(f"s_mov_b32 s{self.S_JP}, 0")
        e.label(L(f"L_pl_{name}"))
        self.pair_select(e)
        body()
        e.salu(f"s_add_u32 s{self.S_JP}, s{self.S_JP}, 1")
        e.salu(f"s_cmp_lt_u32 s{self.S_JP}, s{S_K}")
        e.salu(f"s_cbranch_scc1 {L(f'L_pl_{name}')}")

    def miller_main_multi(self, e, p):
        L = self.lab
        # ---- init: convert P_j, Q_j into scratch, R_j = (Q_j, 1)
        e.salu(f"s_mul_i32 s{S_NSTRIDE}, s{S_N}, s{S_K}")
        e.salu(f"s_lshl_b32 s{S_NSTRIDE}, s{S_NSTRIDE}, 3")          # bytes between limb planes of the PAIR batches
        e.emit(f"v_lshrrev_b32_e32 v{V_IDX8}, 3, v{V_IDX8}", vw=[V_IDX8])   # clamped group index
        e.emit(f"v_mul_lo_u32 v{V_IDX8}, v{V_IDX8}, s{S_K}", vw=[V_IDX8])
        e.emit(f"v_lshlrev_b32_e32 v{V_IDX8}, 3, v{V_IDX8}", vw=[V_IDX8])   # byte offset of the group's first pair

        def init_pair():
            # element offset of pair j = (group*k + j) * 8
            e.salu(f"s_lshl_b32 s{S_TMP1}, s{self.S_JP}, 3")
            e.emit(f"v_add_u32_e32 v{V_IDX8}, s{S_TMP1}, v{V_IDX8}", vw=[V_IDX8])
            self.io_walk_begin(e, S_G1)
            self.io_load_fq2_into_A(e, p, c1_present=False)
            p.to(GlobDyn(0))
            self.io_load_fq2_into_A(e, p, c1_present=False)
            p.to(GlobDyn(1))
            self.io_walk_begin(e, S_G2)
            self.io_load_fq2_into_A(e, p)
            p.to(GlobDyn(2))
            p.store(A0, GlobDyn(4))
            self.io_load_fq2_into_A(e, p)
            p.to(GlobDyn(3))
            p.store(A0, GlobDyn(5))
            self.one_into_A(e)
            p.set_A_fresh()
            p.to(GlobDyn(6))
            p.wait()
            e.salu(f"s_lshl_b32 s{S_TMP1}, s{self.S_JP}, 3")
            e.emit(f"v_subrev_u32_e32 v{V_IDX8}, s{S_TMP1}, v{V_IDX8}", vw=[V_IDX8])
            p.reset_tags()

        self.pair_loop(e, "init", init_pair)
        if self.track:
            self.one_into_A(e)
            p.set_A_fresh()
            p.to(self.SCALE)
            p.reset_tags()

        # ---- top digit: f = product of the tangent lines at Q_j
        def first_step():
            self.pair_in(e, p, with_q=False)
            e.salu(f"s_cmp_eq_u32 s{self.S_JP}, 0")
            e.salu(f"s_cbranch_scc0 {L('L_mf_rest')}")
            self.call2(e, "L2_dblfirst")
            e.salu(f"s_branch {L('L_mf_done')}")
            e.label(L("L_mf_rest"))
            self.call2(e, "L2_dblmul")
            e.label(L("L_mf_done"))
            self.pair_out(e, p)

        self.pair_loop(e, "first", first_step)
        e.salu(f"s_mov_b32 s{S_I}, 63")
        e.label(L("L_mloop"))
        e.salu(f"s_cmp_eq_u32 s{S_I}, 63")
        e.salu(f"s_cbranch_scc1 {L('L_mskip')}")
        self.call2(e, "L2_sqr")
        if self.track:
            self.call2(e, "L2_sqscale")

        def dbl_pair():
            self.pair_in(e, p, with_q=False)
            self.call2(e, "L2_dblmul")
            self.pair_out(e, p)

        self.pair_loop(e, "dbl", dbl_pair)
        e.label(L("L_mskip"))
        e.salu(f"s_bitcmp1_b64 {S_NAF_NZ}, s{S_I}")
        e.salu(f"s_cbranch_scc0 {L('L_mnoadd')}")

        def add_pair():
            self.pair_in(e, p, with_q=True)
            p.mov(self.SX, self.QX)
            p.A(self.QY)
            p.wait()
            e.salu(f"s_bitcmp1_b64 {S_NAF_NEG}, s{S_I}")
            e.salu(f"s_cbranch_scc0 {L('L_mpos')}")
            e.salu(f"s_call_b64 {S_RET1}, {self.labels['neg']}")
            e.label(L("L_mpos"))
            p.tagA = None
            p.to(self.SY)
            self.call2(e, "L2_addmul")
            self.pair_out(e, p)

        self.pair_loop(e, "add", add_pair)
        e.label(L("L_mnoadd"))
        e.salu(f"s_sub_u32 s{S_I}, s{S_I}, 1")
        e.salu(f"s_cbranch_scc0 {L('L_mloop')}")
        xi = (9, 1)
        c = f2pow(xi, (P_INT - 1) // 6)
        c2 = f2mul(c, c)
        c3 = f2mul(c2, c)
        C2, C3 = Const(c2[0], c2[1], "c2"), Const(c3[0], c3[1], "c3")

        def end_pair():
            self.pair_in(e, p, with_q=True)
            p.A(self.QX).conj().mul(C2).to(self.SX)
            p.A(self.QY).conj().mul(C3).to(self.SY)
            p.A(self.SX).conj().mul(C2).to(self.QX)
            p.A(self.SY).conj().neg().mul(C3).to(self.QY)
            self.call2(e, "L2_addmul")
            p.reset_tags()
            p.mov(self.SX, self.QX)
            p.mov(self.SY, self.QY)
            self.call2(e, "L2_addmul_last")
            p.reset_tags()

        self.pair_loop(e, "end", end_pair)
        if self.track:
            self.call2(e, "L2_descale")
        p.reset_tags()
        # output indexing is per group again
        e.salu(f"s_lshl_b32 s{S_NSTRIDE}, s{S_N}, 3")
        e.salu(f"s_sub_u32 s{S_TMP1}, s{S_N}, 1")
        e.emit(f"v_min_u32_e32 v{V_IDX8}, s{S_TMP1}, v{V_IDX}", vw=[V_IDX8])
        e.emit(f"v_lshlrev_b32_e32 v{V_IDX8}, 3, v{V_IDX8}", vw=[V_IDX8])

    def store_out(self, e, p):
        p.reset_tags()
        e.emit(f"v_cmp_gt_u32_e32 vcc, s{S_N}, v{V_IDX}", w=["vcc"])
        e.raw("s_nop 1")
        e.salu(f"s_and_saveexec_b64 {S_SAVE_EXEC}, vcc")
        self.io_walk_begin(e, S_OUT)
        for half in range(2):
            for k in range(6):
                p.load(A0, self.F[k])
                p.wait()
                if half == 1:
                    for i in range(NL):
                        e.emit(f"v_mov_b32_e32 v{A0 + i}, v{A0 + NL + i}", vw=[A0 + i])
                self.cvt_call(e, "cvtout")
                self.io_store_fq(e, A0)
                e.raw("s_nop 1")
        e.emit(f"v_cmp_ne_u32_e32 vcc, 0, v{V_FLAG}", w=["vcc"])
        e.raw("s_nop 1")
        e.salu("s_and_saveexec_b64 s[60:61], vcc")
        e.emit(f"v_mov_b32_e32 v{V_IDX8}, 1", vw=[V_IDX8])
        e.emit("v_mov_b32_e32 v40, 0", vw=[40])
        e.emit(f"global_store_dword v40, v{V_IDX8}, {S_STATUS}", kind="vmem")
        e.salu(f"s_mov_b64 exec, {S_SAVE_EXEC}")
        e.raw("s_waitcnt vmcnt(0)")
        e.emit(f"v_lshrrev_b32_e32 v{V_TID}, 4, v{V_LDS}", vw=[V_TID])
