#!/usr/bin/env python3
"""kgen_prog.py -- L2 (Fq12 / curve steps) and L3 (kernels) of the generated gfx950 assembly.

Accumulator-machine programs over Fq2 slots (see tools/kgen.py for the machine and L1):
    p.A(x).mul(y).sub(v1).sub(v2).mulxi().add(v0).to(c0)
loads x into block A, y into block B, calls the L1 multiply, ... and stores A into slot c0.
The algebra is the GPU schedule stated and checked in tests/sched_model.py.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kgen import (A0, B0, BLOCK, BN_X, Emitter, HOME0, L1, N_AGPR_SLOTS, N_HOME, N_LDS_SLOTS, N0, P_INT, P_LIMBS, PV0, S_CARRY,  # noqa: E402
                  S_N0, S_P, S_RET1, S_RET2, S_RET3, SIX_U_PLUS_2_NAF, V_GOFF, V_IDX, V_IDX8, V_LDS, limbs8, mont)

# ---- scalar registers used by L2/L3 (all inside the clobbered range s36..s101) ----------------
S_TMP0, S_TMP1 = 60, 61
S_GADDR = "s[62:63]"      # address of the global slot being accessed
S_SCRATCH = "s[64:65]"    # scratch base of this workgroup
S_GSTRIDE = 66            # bytes between consecutive global slots
S_I = 67                  # Miller-loop digit index
S_NAF_NZ = "s[68:69]"     # 6u+2 NAF: non-zero mask, negative mask (digits 0..63)
S_NAF_NEG = "s[70:71]"
S_XNAF_NZ = "s[72:73]"    # BN_X NAF masks
S_XNAF_NEG = "s[74:75]"
S_XNAF_RED = "s[48:49]"   # v3 only: digits after which the accumulator's representative is reduced (L2_redF)
S_J = 76                  # pow_x digit index
S_GBASE = 77              # global Fq12 register operand of fq12_mul (slot number * stride, low 32 bits)
S_ITEM = 78
S_G1 = "s[80:81]"
S_G2 = "s[82:83]"
S_OUT = "s[84:85]"
S_N = 86                  # batch size (u32)
S_IOADDR = "s[88:89]"
S_NSTRIDE = 90            # n * 8 (bytes between limbs in the SoA batch)
S_STATUS = "s[92:93]"
S_FIN = "s[94:95]"
S_K = 96
S_SAVE_EXEC = "s[98:99]"
S_NITEMS = 79
S_GRID = 87


class Slot:
    def __init__(self, kind, idx, name=""):
        self.kind, self.idx, self.name = kind, idx, name

    def __repr__(self):
        return f"{self.kind}{self.idx}" + (f"({self.name})" if self.name else "")


def LDS(i, name=""):
    assert 0 <= i < N_LDS_SLOTS
    return Slot("lds", i, name)


def HOME(i, name=""):
    assert 0 <= i < N_HOME
    return Slot("home", i, name)


def AGPR(i, name=""):
    assert 0 <= i < N_AGPR_SLOTS
    return Slot("agpr", i, name)


def GLOB(i, name=""):
    return Slot("glob", i, name)


class GlobDyn:
    """Global slot whose number is base register S_GBASE (bytes) + k * stride (runtime Fq12 operand)."""

    def __init__(self, k):
        self.kind, self.k = "globdyn", k


class Const:
    """Fq2 constant (canonical integers), materialised with literal moves."""

    def __init__(self, c0, c1, name=""):
        self.kind, self.c0, self.c1, self.name = "const", c0, c1, name


def marsh_label(key):
    op, kind, idx = key
    return f"LM_{op}_{kind}_{idx}_%="


# Inlining the 16-move marshalling sequences doubles the code size but measured 4.5 % faster than calling
# shared move routines (a call/return pair costs ~14 cycles of fetch redirect on gfx950).
INLINE_MARSH = bool(int(os.environ.get("KGEN_INLINE_MARSH", "1")))


def emit_marsh_routine(e, key, slot):
    """Leaf routine: 16 register moves between block A/B and a home/AGPR slot (or a literal constant)."""
    e.label(marsh_label(key))
    emit_marsh_body(e, key, slot)
    e.salu(f"s_setpc_b64 {S_RET1}")


def emit_marsh_body(e, key, slot):
    op, kind, _ = key
    blk = B0 if op == "ldB" else A0
    if kind == "const":
        w = limbs8(mont(slot.c0)) + limbs8(mont(slot.c1))
        for i in range(16):
            e.emit(f"v_mov_b32_e32 v{blk + i}, 0x{w[i]:x}", vw=[blk + i])
    elif kind == "home":
        r0 = HOME0 + 16 * slot.idx
        for i in range(16):
            if op == "stA":
                e.emit(f"v_mov_b32_e32 v{r0 + i}, v{blk + i}", vw=[r0 + i])
            else:
                e.emit(f"v_mov_b32_e32 v{blk + i}, v{r0 + i}", vw=[blk + i])
    else:
        for i in range(16):
            if op == "stA":
                e.emit(f"v_accvgpr_write_b32 a{16 * slot.idx + i}, v{blk + i}")
            else:
                e.emit(f"v_accvgpr_read_b32 v{blk + i}, a{16 * slot.idx + i}", vw=[blk + i])


class Prog:
    def __init__(self, e, l1_labels):
        self.e = e
        self.l1 = l1_labels
        self.tagA = None
        self.tagB = None
        self.free_tmp = []
        self.lds_pending = False
        self.vm_pending = False
        self.stats = {}
        self.marsh = {}        # marshalling routines needed: key -> slot

    # ---------------------------------------------------------------- data movement
    def _lds_addr(self, slot, q):
        qi = slot.idx * 4 + q
        return V_LDS + qi // 16, (qi % 16) * 4096

    def _glob_base(self, slot):
        """Sets S_GADDR to the address of a global slot (this lane's 64 bytes are at + V_GOFF)."""
        if slot.kind == "glob":
            self.e.salu(f"s_mul_i32 s{S_TMP0}, s{S_GSTRIDE}, {slot.idx}")
        else:
            self.e.salu(f"s_mul_i32 s{S_TMP0}, s{S_GSTRIDE}, {slot.k}")
            self.e.salu(f"s_add_u32 s{S_TMP0}, s{S_TMP0}, s{S_GBASE}")
        lo, hi = 62, 63
        self.e.salu(f"s_add_u32 s{lo}, s64, s{S_TMP0}")
        self.e.salu(f"s_addc_u32 s{hi}, s65, 0")

    def load(self, blk, slot):
        bn = "A" if blk == A0 else "B"
        if slot.kind == "lds":
            for q in range(4):
                base, off = self._lds_addr(slot, q)
                self.e.emit(f"ds_read_b128 v[{blk + 4 * q}:{blk + 4 * q + 3}], v{base} offset:{off}", kind="lds",
                            vw=range(blk + 4 * q, blk + 4 * q + 4))
            self.lds_pending = True
        elif slot.kind in ("home", "agpr", "const"):
            # register-to-register marshalling lives in shared leaf routines (16 moves + return): a call
            # site is one 4-byte s_call instead of 128 bytes of moves -- the hot loop must fit the I-cache
            key = (("ld" + bn), slot.kind, slot.idx if slot.kind != "const" else slot.name)
            if INLINE_MARSH:
                emit_marsh_body(self.e, key, slot)
            else:
                self.marsh[key] = slot
                self.e.salu(f"s_call_b64 {S_RET1}, {marsh_label(key)}")
        elif slot.kind in ("glob", "globdyn"):
            self._glob_base(slot)
            for q in range(4):
                self.e.emit(f"global_load_dwordx4 v[{blk + 4 * q}:{blk + 4 * q + 3}], v{V_GOFF}, {S_GADDR} offset:{16 * q}", kind="vmem",
                            vw=range(blk + 4 * q, blk + 4 * q + 4))
            self.vm_pending = True
        else:
            raise ValueError(slot.kind)
        self._count("ld_" + slot.kind)

    def store(self, blk, slot):
        assert blk == A0
        if slot.kind == "lds":
            for q in range(4):
                base, off = self._lds_addr(slot, q)
                self.e.emit(f"ds_write_b128 v{base}, v[{blk + 4 * q}:{blk + 4 * q + 3}] offset:{off}", kind="lds")
        elif slot.kind in ("home", "agpr"):
            key = ("stA", slot.kind, slot.idx)
            if INLINE_MARSH:
                emit_marsh_body(self.e, key, slot)
            else:
                self.marsh[key] = slot
                self.e.salu(f"s_call_b64 {S_RET1}, {marsh_label(key)}")
        elif slot.kind in ("glob", "globdyn"):
            self._glob_base(slot)
            for q in range(4):
                self.e.emit(f"global_store_dwordx4 v{V_GOFF}, v[{blk + 4 * q}:{blk + 4 * q + 3}], {S_GADDR} offset:{16 * q}", kind="vmem",
                            store=range(blk + 4 * q, blk + 4 * q + 4))
            self.e.raw("s_nop 1")   # wide-store data hazard: the next VALU write of block A may sit behind a call
        else:
            raise ValueError(slot.kind)
        self._count("st_" + slot.kind)

    def _count(self, k):
        self.stats[k] = self.stats.get(k, 0) + 1

    def wait(self):
        if self.lds_pending and self.vm_pending:
            self.e.raw("s_waitcnt vmcnt(0) lgkmcnt(0)")
        elif self.lds_pending:
            self.e.raw("s_waitcnt lgkmcnt(0)")
        elif self.vm_pending:
            self.e.raw("s_waitcnt vmcnt(0)")
        self.lds_pending = self.vm_pending = False

    # ---------------------------------------------------------------- accumulator machine
    def reset_tags(self):
        self.tagA = self.tagB = None

    def A(self, x):
        if self.tagA is not x:
            self.load(A0, x)
            self.tagA = x
        return self

    def _B(self, y):
        if self.tagB is not y:
            self.load(B0, y)
            self.tagB = y

    def call(self, name):
        self.wait()
        self.e.salu(f"s_call_b64 {S_RET1}, {self.l1[name]}")
        self.tagA = None
        self._count(name)
        return self

    def _bin(self, name, y):
        self._B(y)
        return self.call(name)

    def mul(self, y): return self._bin("mul", y)
    def add(self, y): return self._bin("add", y)
    def sub(self, y): return self._bin("sub", y)
    def rsub(self, y): return self._bin("rsub", y)
    def mulfq(self, y): return self._bin("mulfq", y)      # A * (Fq in y.c0)
    def sqr(self): return self.call("sqr")
    def dbl(self): return self.call("dbl")
    def neg(self): return self.call("neg")
    def conj(self): return self.call("negc1")
    def mulxi(self): return self.call("mulxi")

    def to(self, dst):
        self.wait()
        self.store(A0, dst)
        self.tagA = dst
        if self.tagB is dst:
            self.tagB = None
        return self

    def mov(self, dst, src):
        self.A(src).to(dst)

    # ---------------------------------------------------------------- temp slots
    def set_temps(self, slots):
        self.free_tmp = list(slots)

    def tmp(self):
        return self.free_tmp.pop(0)

    def rel(self, *slots):
        for s in slots:
            assert s not in self.free_tmp
            self.free_tmp.insert(0, s)

    # ================================================================ L2 algorithms
    def fq6_mul(self, a, b, out):
        """out[i] <- (a0,a1,a2)*(b0,b1,b2) in Fq2[v]/(v^3 - xi); a, b, out: lists of 3 slots, out disjoint."""
        V0, V1, V2, S = self.tmp(), self.tmp(), self.tmp(), self.tmp()
        self.A(a[0]).mul(b[0]).to(V0)
        self.A(a[1]).mul(b[1]).to(V1)
        self.A(a[2]).mul(b[2]).to(V2)
        self.A(a[1]).add(a[2]).to(S)
        self.A(b[1]).add(b[2]).mul(S).sub(V1).sub(V2).mulxi().add(V0).to(out[0])
        self.A(a[0]).add(a[1]).to(S)
        self.A(b[0]).add(b[1]).mul(S).sub(V0).sub(V1).to(S)
        self.A(V2).mulxi().add(S).to(out[1])
        self.A(a[0]).add(a[2]).to(S)
        self.A(b[0]).add(b[2]).mul(S).sub(V0).sub(V2).add(V1).to(out[2])
        self.rel(V0, V1, V2, S)

    def fq12_sqr(self, F):
        """F <- F^2 in place (complex squaring over Fq6: t = A0 A1, u = (A0 + A1)(A0 + v A1))."""
        T = [self.tmp() for _ in range(3)]
        SB = [self.tmp() for _ in range(3)]
        A_0, A_1 = [F[0], F[2], F[4]], [F[1], F[3], F[5]]
        self.fq6_mul(A_0, A_1, T)
        self.A(F[5]).mulxi().add(F[0]).to(SB[0])
        self.A(F[2]).add(F[1]).to(SB[1])
        self.A(F[4]).add(F[3]).to(SB[2])
        for i in range(3):
            self.A(A_0[i]).add(A_1[i]).to(A_0[i])          # SA in place of A0 (A0 is dead from here)
        U = A_1                                            # A1 is dead too: u lands in the odd slots
        self.fq6_mul(A_0, SB, U)
        for s in SB:
            self.rel(s)
        X = self.tmp()
        self.A(T[2]).mulxi().to(X)
        self.A(U[0]).sub(T[0]).sub(X).to(F[0])
        self.A(U[1]).sub(T[1]).sub(T[0]).to(F[2])
        self.A(U[2]).sub(T[2]).sub(T[1]).to(F[4])
        self.A(T[0]).dbl().to(F[1])
        self.A(T[1]).dbl().to(F[3])
        self.A(T[2]).dbl().to(F[5])
        self.rel(X, *T)

    def fq12_mul(self, F, Bs, conj_b=False):
        """F <- F * B.  B (six slots) is a private copy and is DESTROYED (its even slots end up holding B0 + B1).
        conj_b: use conjugate_fp12(B) (odd coefficients negated in place first)."""
        b = list(Bs)
        if conj_b:
            for k in (1, 3, 5):
                self.A(Bs[k]).neg().to(Bs[k])
        A_0, A_1 = [F[0], F[2], F[4]], [F[1], F[3], F[5]]
        B_0, B_1 = [b[0], b[2], b[4]], [b[1], b[3], b[5]]
        T0 = [self.tmp() for _ in range(3)]
        T1 = [self.tmp() for _ in range(3)]
        self.fq6_mul(A_0, B_0, T0)
        self.fq6_mul(A_1, B_1, T1)
        for i in range(3):
            self.A(A_0[i]).add(A_1[i]).to(A_0[i])          # SA in place of A0
            self.A(B_0[i]).add(B_1[i]).to(B_0[i])          # SB in place of B0
        M = A_1
        self.fq6_mul(A_0, B_0, M)
        for i in range(3):
            self.A(M[i]).sub(T0[i]).sub(T1[i]).to(M[i])
        self.A(T1[2]).mulxi().add(T0[0]).to(F[0])
        self.A(T0[1]).add(T1[0]).to(F[2])
        self.A(T0[2]).add(T1[1]).to(F[4])
        self.rel(*T0)
        self.rel(*T1)

    def fq4_sqr(self, a, b, r0, r1):
        """(a + b y)^2, y^2 = xi -> r0 = a^2 + xi b^2, r1 = 2ab (r0, r1 temp slots distinct from a, b)."""
        S = self.tmp()
        self.A(a).mul(b).to(r1)
        self.A(b).mulxi().add(a).to(S)
        self.A(a).add(b).mul(S).sub(r1).to(r0)
        self.A(r1).mulxi().rsub(r0).to(r0)          # r0 = r0 - xi*t
        self.A(r1).dbl().to(r1)
        self.rel(S)

    def fq12_cyc_sqr(self, F):
        """Granger-Scott squaring (F in the cyclotomic subgroup), in place."""
        t = [self.tmp() for _ in range(6)]
        self.fq4_sqr(F[0], F[3], t[0], t[1])
        self.fq4_sqr(F[1], F[4], t[2], t[3])
        self.fq4_sqr(F[2], F[5], t[4], t[5])
        self.A(t[0]).sub(F[0]).dbl().add(t[0]).to(F[0])
        self.A(t[1]).add(F[3]).dbl().add(t[1]).to(F[3])
        X = self.tmp()
        self.A(t[5]).mulxi().to(X)
        self.A(X).add(F[1]).dbl().add(X).to(F[1])
        self.A(t[4]).sub(F[4]).dbl().add(t[4]).to(F[4])
        self.A(t[2]).sub(F[2]).dbl().add(t[2]).to(F[2])
        self.A(t[3]).add(F[5]).dbl().add(t[3]).to(F[5])
        self.rel(X, *t)

    def mul_by_034(self, F, L0, L3, L4):
        c = [self.tmp() for _ in range(3)]
        S = self.tmp()
        # c0 = a0 b0 + xi (a3 b3 + a2 b4) ; c1 = a1 b0 + xi (a4 b3 + a3 b4) ; c2 = a2 b0 + xi (a5 b3 + a4 b4)
        for k, (i0, i3, i4) in enumerate(((0, 3, 2), (1, 4, 3), (2, 5, 4))):
            self.A(F[i3]).mul(L3).to(S)
            self.A(F[i4]).mul(L4).add(S).mulxi().to(S)
            self.A(F[i0]).mul(L0).add(S).to(c[k])
        # c3 = a3 b0 + a0 b3 + xi a5 b4 ; c4 = a4 b0 + a1 b3 + a0 b4 ; c5 = a5 b0 + a2 b3 + a1 b4
        d = [self.tmp() for _ in range(2)]
        self.A(F[5]).mul(L4).mulxi().to(S)
        self.A(F[0]).mul(L3).add(S).to(S)
        self.A(F[3]).mul(L0).add(S).to(d[0])
        self.A(F[1]).mul(L3).to(S)
        self.A(F[0]).mul(L4).add(S).to(S)
        self.A(F[4]).mul(L0).add(S).to(d[1])
        self.A(F[2]).mul(L3).to(S)
        self.A(F[1]).mul(L4).add(S).to(S)
        self.A(F[5]).mul(L0).add(S).to(F[5])
        self.mov(F[3], d[0])
        self.mov(F[4], d[1])
        for k in range(3):
            self.mov(F[k], c[k])
        self.rel(S, *c)
        self.rel(*d)

    def mul_by_235(self, F, L2, L3, L5):
        c = [self.tmp() for _ in range(3)]
        S = self.tmp()
        d = [self.tmp() for _ in range(2)]
        # c0 = xi (a4 b2 + a3 b3 + a1 b5) ; c1 = xi (a5 b2 + a4 b3 + a2 b5) ; c2 = a0 b2 + xi (a5 b3 + a3 b5)
        self.A(F[4]).mul(L2).to(S)
        self.A(F[3]).mul(L3).add(S).to(S)
        self.A(F[1]).mul(L5).add(S).mulxi().to(c[0])
        self.A(F[5]).mul(L2).to(S)
        self.A(F[4]).mul(L3).add(S).to(S)
        self.A(F[2]).mul(L5).add(S).mulxi().to(c[1])
        self.A(F[5]).mul(L3).to(S)
        self.A(F[3]).mul(L5).add(S).mulxi().to(S)
        self.A(F[0]).mul(L2).add(S).to(c[2])
        # c3 = a1 b2 + a0 b3 + xi a4 b5 ; c4 = a2 b2 + a1 b3 + xi a5 b5 ; c5 = a3 b2 + a2 b3 + a0 b5
        self.A(F[4]).mul(L5).mulxi().to(S)
        self.A(F[1]).mul(L2).add(S).to(S)
        self.A(F[0]).mul(L3).add(S).to(d[0])
        self.A(F[5]).mul(L5).mulxi().to(S)
        self.A(F[2]).mul(L2).add(S).to(S)
        self.A(F[1]).mul(L3).add(S).to(d[1])
        self.A(F[3]).mul(L2).to(S)
        self.A(F[2]).mul(L3).add(S).to(S)
        self.A(F[0]).mul(L5).add(S).to(F[5])
        self.mov(F[3], d[0])
        self.mov(F[4], d[1])
        for k in range(3):
            self.mov(F[k], c[k])
        self.rel(S, *c)
        self.rel(*d)

    def dbl_step(self, R, Pt, line, scale=None, sq_scale=False):
        """R=(X,Y,Z) <- 2R ; line = (L0, L3, L4) of the tangent at the old R evaluated at P (Pt = (PX, PY) slots, scalar in c0).
        scale: slot of the running line scale s <- [s^2] * Z^2."""
        X, Y, Z = R
        L0, L3, L4 = line
        Bq, C, E, Fv, H, T = [self.tmp() for _ in range(6)]
        self.A(Y).sqr().to(Bq)
        self.A(Z).sqr().to(C)
        if scale is not None:
            if sq_scale:
                self.A(scale).sqr().mul(C).to(scale)
            else:
                self.A(scale).mul(C).to(scale)
        self.A(C).mul(THREE_B).to(E)
        self.A(E).dbl().add(E).to(Fv)
        self.A(Y).add(Z).sqr().sub(Bq).sub(C).to(H)            # H = 2 Y Z = (Y + Z)^2 - Y^2 - Z^2: a squaring instead of a multiplication
        # line
        self.A(C).dbl().dbl().dbl().add(C).to(T)            # 9 C
        self.A(Bq).mulxi().sub(T).to(L0)
        self.A(H).mulfq(Pt[1]).to(L3)                         # H * Py
        self.A(X).sqr().to(T)
        self.A(T).dbl().add(T).mulfq(Pt[0]).neg().to(L4)      # -3 X^2 * Px
        # point
        self.A(Bq).sub(Fv).to(T)
        self.A(X).mul(Y).dbl().mul(T).to(X)
        self.A(E).sqr().to(T)
        self.A(T).dbl().add(T).dbl().dbl().to(T)              # 12 E^2
        self.A(Bq).add(Fv).sqr().sub(T).to(Y)
        self.A(Bq).mul(H).dbl().dbl().to(Z)
        self.rel(Bq, C, E, Fv, H, T)

    def add_step(self, R, Q, Pt, line, scale=None, update=True):
        """R <- R + Q (Q = (x2, y2) affine slots); line = (L2, L3, L5) of the chord through old R and Q at P."""
        X, Y, Z = R
        x2, y2 = Q
        L2, L3, L5 = line
        th, mu, T, U = [self.tmp() for _ in range(4)]
        if scale is not None:
            self.A(scale).mul(Z).to(scale)
        self.A(y2).mul(Z).rsub(Y).to(th)                      # theta = Y - y2 Z
        self.A(x2).mul(Z).rsub(X).to(mu)                      # mu = X - x2 Z
        self.A(mu).mulfq(Pt[1]).neg().to(L2)                  # -mu * Py
        self.A(th).mulfq(Pt[0]).to(L3)                        # theta * Px
        self.A(x2).mul(Y).to(T)
        self.A(X).mul(y2).sub(T).to(L5)                       # X y2 - x2 Y
        if update:
            Cc, D, E = self.tmp(), self.tmp(), self.tmp()
            self.A(th).sqr().to(Cc)
            self.A(mu).sqr().to(D)
            self.A(mu).mul(D).to(E)
            self.A(Z).mul(Cc).to(Cc)                          # F = Z * C
            self.A(X).mul(D).to(D)                            # G = X * D
            self.A(D).dbl().to(T)
            self.A(E).add(Cc).sub(T).to(T)                    # H = E + F - 2G
            self.A(mu).mul(T).to(X)                           # X3 = mu H
            self.A(E).mul(Y).to(U)                            # E * Y
            self.A(D).sub(T).mul(th).sub(U).to(Y)             # Y3 = theta (G - H) - E Y
            self.A(Z).mul(E).to(Z)                            # Z3 = Z E
            self.rel(Cc, D, E)
        self.rel(th, mu, T, U)



def _three_b():
    xi_inv_n = pow(82, -1, P_INT)          # 1/(9+u) = (9 - u)/82
    return Const(81 * xi_inv_n % P_INT, (-9 * xi_inv_n) % P_INT, "threeb")


THREE_B = _three_b()


def f2mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P_INT, (a[0] * b[1] + a[1] * b[0]) % P_INT)


def f2pow(a, e):
    r = (1, 0)
    for bit in bin(e)[2:]:
        r = f2mul(r, r)
        if bit == "1":
            r = f2mul(r, a)
    return r


def naf_masks(naf):
    nz = sum(1 << i for i, d in enumerate(naf) if d != 0)
    neg = sum(1 << i for i, d in enumerate(naf) if d < 0)
    return nz, neg


def x_naf():
    e, out = BN_X, []
    for _ in range(64):
        if e & 1:
            z = 2 - (e % 4)
            e //= 2
            if z == -1:
                e += 1
            out.append(z)
        else:
            out.append(0)
            e //= 2
    while out[-1] == 0:
        out.pop()
    return out


L1_NAMES = ["mul", "sqr", "mulfq", "add", "sub", "rsub", "dbl", "neg", "negc1", "mulxi", "fqmul", "fqsqr"]


class KernelBuilder:
    """Assembles one kernel blob.  Operand order of the asm statement (all inputs):
       %0 g1 (s64)  %1 g2 (s64)  %2 f_in (s64)  %3 out (s64)  %4 n (s32)  %5 k (s32)  %6 scratch (s64)
       %7 gslot_stride_bytes (s32)  %8 status (s64)  %9 tid (v32)  %10 block id (s32)  %11 grid size (s32)"""

    # slot map -------------------------------------------------------------------------------------
    F = [LDS(i, f"F{i}") for i in range(6)]
    R = [LDS(6, "RX"), LDS(7, "RY"), LDS(8, "RZ")]
    SCALE = LDS(9, "scale")
    QX, QY, PX, PY = AGPR(0, "QX"), AGPR(1, "QY"), AGPR(2, "PX"), AGPR(3, "PY")
    SX, SY = AGPR(4, "SX"), AGPR(5, "SY")          # the affine point of the current addition step
    LINE = [AGPR(6, "La"), AGPR(7, "Lb"), AGPR(8, "Lc")]
    BOP = [AGPR(10 + i, f"B{i}") for i in range(6)]  # fq12_mul operand copied in from scratch (final exp)
    G_FQ12 = lambda self, j: [GLOB(6 * j + i) for i in range(6)]  # noqa: E731
    N_G_FQ12 = 8

    def __init__(self, do_miller=True, do_fexp=True, track=False):
        self.do_miller, self.do_fexp, self.track = do_miller, do_fexp, track
        self.labels = {n: f"L1_{n}_%=" for n in L1_NAMES}
        self.sections = []
        self.marsh = {}

    def lab(self, name):
        return f"{name}_%="

    # ---------------------------------------------------------------------------------------------
    def new_prog(self, temps):
        e = Emitter()
        p = Prog(e, self.labels)
        p.marsh = self.marsh
        p.set_temps(temps)
        return e, p

    def miller_temps(self):
        # homes first (cheapest moves), then the AGPR slots not used by persistent values
        return [HOME(i) for i in range(8)] + [AGPR(i) for i in (10, 11, 12, 13, 14, 15)]      # AGPR 9: fq_inv's own slot

    def fexp_temps(self):
        return [HOME(i) for i in range(8)] + [AGPR(i) for i in (0, 1, 2, 3, 4, 5, 6, 7, 8)] + [LDS(i) for i in (6, 7, 8, 9)]

    def l2_routine(self, name, body, temps):
        e, p = self.new_prog(temps)
        e.label(self.lab(name))
        body(p)
        p.wait()
        e.salu(f"s_setpc_b64 {S_RET2}")
        self.sections.append(e)
        return p

    # ---------------------------------------------------------------------------------------------
    def build(self):
        main = Emitter()
        self.prologue(main)
        subs_marker = len(self.sections)
        l1e = Emitter()
        for n in L1_NAMES:
            g = L1(l1e)
            l1e.label(self.labels[n])
            getattr(g, "r_" + n)()
            l1e.salu(f"s_setpc_b64 {S_RET1}")
        # L2 routines
        if self.do_miller:
            sc = self.SCALE if self.track else None
            self.l2_routine("L2_sqr", lambda p: p.fq12_sqr(self.F), self.miller_temps())
            self.l2_routine("L2_dblmul", lambda p: (p.dbl_step(self.R, (self.PX, self.PY), self.LINE, scale=sc),
                                                    p.mul_by_034(self.F, *self.LINE)), self.miller_temps())
            self.l2_routine("L2_dblfirst", lambda p: self._dbl_first(p), self.miller_temps())
            self.l2_routine("L2_addmul", lambda p: (p.add_step(self.R, (self.SX, self.SY), (self.PX, self.PY), self.LINE, scale=sc, update=True),
                                                    p.mul_by_235(self.F, *self.LINE)), self.miller_temps())
            self.l2_routine("L2_addmul_last", lambda p: (p.add_step(self.R, (self.SX, self.SY), (self.PX, self.PY), self.LINE, scale=sc, update=False),
                                                         p.mul_by_235(self.F, *self.LINE)), self.miller_temps())
            if self.track:
                self.l2_routine("L2_fqinv", self._fq_inv, self.miller_temps())
                self.l2_routine("L2_descale", self._descale, self.miller_temps())
                self.l2_routine("L2_sqscale", lambda p: p.A(self.SCALE).sqr().to(self.SCALE), self.miller_temps())
        if self.do_fexp:
            self.l2_routine("L2_fqinv", self._fq_inv, self.fexp_temps()) if not (self.do_miller and self.track) else None
            self.l2_routine("L2_cyc", lambda p: p.fq12_cyc_sqr(self.F), self.fexp_temps())
            self._mulG_routines()
            for k in (1, 2, 3):
                self.l2_routine(f"L2_frob{k}", lambda p, k=k: self._frobenius(p, k), self.fexp_temps())
            self.l2_routine("L2_inv", self._fq12_inv, self.fexp_temps())
            self.l2_routine("L2_stG", lambda p: [p.A(self.F[i]).to(GlobDyn(i)) for i in range(6)], self.fexp_temps())
            self.l2_routine("L2_ldG", lambda p: [p.A(GlobDyn(i)).to(self.F[i]) for i in range(6)], self.fexp_temps())
            self.l2_routine("L2_ldGc", lambda p: [(p.A(GlobDyn(i)).neg().to(self.F[i]) if i % 2 else p.A(GlobDyn(i)).to(self.F[i])) for i in range(6)],
                            self.fexp_temps())
            self.l2_routine("L2_conjF", lambda p: [p.A(self.F[i]).neg().to(self.F[i]) for i in (1, 3, 5)], self.fexp_temps())
            self._powx_routine()
        self.main_body(main)
        # marshalling routines discovered while emitting L2/L3
        me = Emitter()
        for key, slot in sorted(self.marsh.items(), key=lambda kv: str(kv[0])):
            emit_marsh_routine(me, key, slot)
        # final order: prologue+main are in `main` (prologue jumps over the subroutines)
        out = []
        for e in [self._pro] + [l1e, me] + self.sections[subs_marker:] + [main]:
            out.extend(e.finalize())
        return out

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
        # scratch base of this workgroup: scratch + block * 256 * 64 ; lane offset = tid * 64
        e.salu(f"s_lshl_b32 s{S_TMP0}, %10, 14")
        e.salu(f"s_mov_b64 {S_SCRATCH}, %6")
        e.salu(f"s_add_u32 s64, s64, s{S_TMP0}")
        e.salu("s_addc_u32 s65, s65, 0")
        e.emit(f"v_lshlrev_b32_e32 v{V_GOFF}, 6, %9", vw=[V_GOFF])
        e.emit(f"v_lshlrev_b32_e32 v{V_LDS}, 4, %9", vw=[V_LDS])
        e.emit(f"v_add_u32_e32 v{V_LDS + 1}, 0x10000, v{V_LDS}", vw=[V_LDS + 1])
        e.emit(f"v_add_u32_e32 v{V_LDS + 2}, 0x20000, v{V_LDS}", vw=[V_LDS + 2])
        e.emit(f"v_mov_b32_e32 v118, %9", vw=[118])      # tid
        for i in range(8):
            e.salu(f"s_mov_b32 s{S_P + i}, 0x{P_LIMBS[i]:x}")
            e.emit(f"v_mov_b32_e32 v{PV0 + i}, 0x{P_LIMBS[i]:x}", vw=[PV0 + i])
        e.salu(f"s_mov_b32 s{S_N0}, 0x{N0:x}")
        nz, neg = naf_masks(SIX_U_PLUS_2_NAF[:64])
        e.salu(f"s_mov_b32 s68, 0x{nz & 0xFFFFFFFF:x}")
        e.salu(f"s_mov_b32 s69, 0x{nz >> 32:x}")
        e.salu(f"s_mov_b32 s70, 0x{neg & 0xFFFFFFFF:x}")
        e.salu(f"s_mov_b32 s71, 0x{neg >> 32:x}")
        xn = x_naf()
        nz, neg = naf_masks(xn[:-1])
        self.x_top = len(xn) - 1
        e.salu(f"s_mov_b32 s72, 0x{nz & 0xFFFFFFFF:x}")
        e.salu(f"s_mov_b32 s73, 0x{nz >> 32:x}")
        e.salu(f"s_mov_b32 s74, 0x{neg & 0xFFFFFFFF:x}")
        e.salu(f"s_mov_b32 s75, 0x{neg >> 32:x}")
        e.salu(f"s_lshl_b32 s{S_NSTRIDE}, s{S_N}, 3")
        e.salu(f"s_add_u32 s{S_NITEMS}, s{S_N}, 255")
        e.salu(f"s_lshr_b32 s{S_NITEMS}, s{S_NITEMS}, 8")
        e.emit("v_mov_b32_e32 v119, 0", vw=[119])         # zero-divisor flag
        e.salu(f"s_branch {self.lab('L_main')}")

    # ---------------------------------------------------------------------------------------------
    def io_walk_begin(self, e, base):
        e.salu(f"s_mov_b64 {S_IOADDR}, {base}")

    def io_walk_next(self, e):
        e.salu(f"s_add_u32 s88, s88, s{S_NSTRIDE}")
        e.salu("s_addc_u32 s89, s89, 0")

    def io_load_fq(self, e, reg0):
        """Loads one Fq (4 u64 limbs of the SoA batch at the walking address) into v[reg0:reg0+7]."""
        for l in range(4):
            e.emit(f"global_load_dwordx2 v[{reg0 + 2 * l}:{reg0 + 2 * l + 1}], v{V_IDX8}, {S_IOADDR}", kind="vmem", vw=[reg0 + 2 * l, reg0 + 2 * l + 1])
            self.io_walk_next(e)

    def io_store_fq(self, e, reg0):
        for l in range(4):
            e.emit(f"global_store_dwordx2 v{V_IDX8}, v[{reg0 + 2 * l}:{reg0 + 2 * l + 1}], {S_IOADDR}", kind="vmem")
            self.io_walk_next(e)

    def zero_block(self, e, blk, n=16):
        for i in range(n):
            e.emit(f"v_mov_b32_e32 v{blk + i}, 0", vw=[blk + i])

    def one_into_A(self, e):
        w = limbs8(mont(1))
        for i in range(8):
            e.emit(f"v_mov_b32_e32 v{A0 + i}, 0x{w[i]:x}", vw=[A0 + i])
        self.zero_block(e, A0 + 8, 8)

    # ---------------------------------------------------------------------------------------------
    def _dbl_first(self, p):
        """i = 63: R = Q -> 2Q, f = dense(tangent line) (miller_loop_native.rs:127-149); scale stays 1."""
        p.dbl_step(self.R, (self.PX, self.PY), self.LINE, scale=None)
        p.mov(self.F[0], self.LINE[0])
        p.mov(self.F[3], self.LINE[1])
        p.mov(self.F[4], self.LINE[2])
        p.wait()
        self.zero_block(p.e, A0)
        p.tagA = None
        for k in (1, 2, 5):
            p.store(A0, self.F[k])

    def _fq_inv(self, p):
        """A.c0 <- A.c0^(p-2) (Fermat; fixed exponent, uniform control flow).  Clobbers B and HOME temps.
        Called with S_RET2; uses S_RET1 for the multiplies and s[60:61] as scratch."""
        e = p.e
        base = AGPR(9, "fqinv_base")      # dedicated: this routine is called from inside other L2 routines
        p.wait()
        p.store(A0, base)                 # keep a (c1 part is whatever it was)
        p.load(B0, base)                  # B.c0 = a
        p.tagA = p.tagB = None
        # exponent p-2: limb 0 differs from p by 2, the others are the modulus limbs already in SGPRs
        e.salu(f"s_sub_u32 s{S_TMP1}, s{S_P}, 2")
        self._fqinv_uid = getattr(self, "_fqinv_uid", 0) + 1          # deterministic per-instance label suffix
        uid = self._fqinv_uid
        for limb in range(7, -1, -1):
            top = 29 if limb == 7 else 31          # p < 2^254: bit 253 is the top bit; it is consumed by r = a
            if limb == 7:
                top = 28                           # start below bit 253 (= bit 29 of limb 7)
            reg = f"s{S_TMP1}" if limb == 0 else f"s{S_P + limb}"
            lbl = self.lab(f"L_fqinv_{limb}_{uid}")
            skip = self.lab(f"L_fqinv_skip_{limb}_{uid}")
            e.salu(f"s_mov_b32 s{S_TMP0}, {top}")
            e.label(lbl)
            e.salu(f"s_call_b64 {S_RET1}, {self.labels['fqsqr']}")
            e.salu(f"s_bitcmp1_b32 {reg}, s{S_TMP0}")
            e.salu(f"s_cbranch_scc0 {skip}")
            e.salu(f"s_call_b64 {S_RET1}, {self.labels['fqmul']}")
            e.label(skip)
            e.salu(f"s_sub_u32 s{S_TMP0}, s{S_TMP0}, 1")
            e.salu(f"s_cbranch_scc0 {lbl}")

    def _fq2_inv_inline(self, p, src, dst):
        """dst <- 1/src (Fq2): conj(src) / (c0^2 + c1^2); sets the zero-divisor flag v119 when the norm is 0."""
        e = p.e
        n0, tmp = p.tmp(), p.tmp()
        p.A(src)
        p.call("fqsqr").to(n0)                                  # n0.c0 = c0^2
        p.A(src)
        p.wait()
        for i in range(8):
            e.emit(f"v_mov_b32_e32 v{A0 + i}, v{A0 + 8 + i}", vw=[A0 + i])
        p.tagA = None
        p.call("fqsqr").add(n0)                                 # A.c0 = c0^2 + c1^2 (c1 lane is don't-care)
        # zero check on the norm
        e.emit(f"v_or3_b32 v118, v{A0}, v{A0 + 1}, v{A0 + 2}", vw=[118])
        e.emit(f"v_or3_b32 v118, v118, v{A0 + 3}, v{A0 + 4}", vw=[118])
        e.emit(f"v_or3_b32 v118, v118, v{A0 + 5}, v{A0 + 6}", vw=[118])
        e.emit(f"v_or_b32_e32 v118, v118, v{A0 + 7}", vw=[118])
        e.emit("v_cmp_eq_u32_e32 vcc, 0, v118", w=["vcc"])
        e.emit("v_cndmask_b32_e64 v118, 0, 1, vcc", r=["vcc"], vw=[118])
        e.emit("v_or_b32_e32 v119, v119, v118", vw=[119])
        # inverse of the norm
        p.wait()
        e.salu(f"s_mov_b64 {S_RET3}, {S_RET2}")
        e.salu(f"s_call_b64 {S_RET2}, {self.lab('L2_fqinv')}")
        e.salu(f"s_mov_b64 {S_RET2}, {S_RET3}")
        p.tagA = p.tagB = None
        p.to(tmp)                                               # tmp.c0 = 1/norm
        p.A(src).mulfq(tmp).conj().to(dst)
        p.rel(n0, tmp)

    def _descale(self, p):
        """F <- F / scale  (exact miller_loop_native value)."""
        inv = p.tmp()
        self._fq2_inv_inline(p, self.SCALE, inv)
        for i in range(6):
            p.A(self.F[i]).mul(inv).to(self.F[i])
        p.rel(inv)

    def _frobenius(self, p, k):
        """F <- frobenius_map_native(F, k) (final_exp_native.rs:17-54): conj^k on each coefficient, times frob_coeffs(k)^i."""
        xi = (9, 1)
        fc = f2pow(xi, (P_INT ** k - 1) // 6)
        for i in range(6):
            g = f2pow(fc, i)
            p.A(self.F[i])
            if k % 2:
                p.conj()
            if g == (1, 0):
                pass
            elif g[1] == 0:
                p.mulfq(Const(g[0], 0, f"frob{k}_{i}"))
            else:
                p.mul(Const(g[0], g[1], f"frob{k}_{i}"))
            p.to(self.F[i])

    def _mulG_routines(self):
        """F <- F * G (G = Fq12 in scratch at S_GBASE); L2_mulGc multiplies by conjugate_fp12(G)."""
        e, p = self.new_prog(self.fexp_temps())
        # the operand copy lives in AGPR slots 10..15: take them out of the temp pool
        p.free_tmp = [t for t in p.free_tmp if not (t.kind == "agpr" and t.idx >= 10)]
        e.label(self.lab("L2_mulGc"))
        for i in range(6):
            if i % 2:
                p.A(GlobDyn(i)).neg().to(self.BOP[i])
            else:
                p.A(GlobDyn(i)).to(self.BOP[i])
        p.wait()
        e.salu(f"s_branch {self.lab('L2_mul_body')}")
        e.label(self.lab("L2_mulG"))
        p.reset_tags()
        for i in range(6):
            p.A(GlobDyn(i)).to(self.BOP[i])
        p.wait()
        e.label(self.lab("L2_mul_body"))
        p.reset_tags()
        p.fq12_mul(self.F, self.BOP)
        p.wait()
        e.salu(f"s_setpc_b64 {S_RET2}")
        self.sections.append(e)

    def _fq12_inv(self, p):
        """F <- 1/F (ark Fq12 inverse, through Fq6 and Fq2 norms)."""
        F = self.F
        A_0, A_1 = [F[0], F[2], F[4]], [F[1], F[3], F[5]]
        S0 = [p.tmp() for _ in range(3)]
        S1 = [p.tmp() for _ in range(3)]
        p.fq6_mul(A_0, A_0, S0)
        p.fq6_mul(A_1, A_1, S1)
        # d = S0 - v*S1 = (s00 - xi s12, s01 - s10, s02 - s11)
        p.A(S1[2]).mulxi().rsub(S0[0]).to(S0[0])
        p.A(S0[1]).sub(S1[0]).to(S0[1])
        p.A(S0[2]).sub(S1[1]).to(S0[2])
        d = S0
        t = S1
        # Fq6 inverse of d
        X = p.tmp()
        p.A(d[1]).mul(d[2]).mulxi().to(X)
        p.A(d[0]).sqr().sub(X).to(t[0])                       # t0 = d0^2 - xi d1 d2
        p.A(d[0]).mul(d[1]).to(X)
        p.A(d[2]).sqr().mulxi().sub(X).to(t[1])               # t1 = xi d2^2 - d0 d1
        p.A(d[0]).mul(d[2]).to(X)
        p.A(d[1]).sqr().sub(X).to(t[2])                       # t2 = d1^2 - d0 d2
        Y = p.tmp()
        p.A(d[2]).mul(t[1]).to(X)
        p.A(d[1]).mul(t[2]).add(X).mulxi().to(X)
        p.A(d[0]).mul(t[0]).add(X).to(Y)                      # norm in Fq2
        self._fq2_inv_inline(p, Y, X)
        for i in range(3):
            p.A(t[i]).mul(X).to(d[i])                         # d <- d^-1 (Fq6)
        p.rel(X, Y)
        # result = (A0 * dinv) - (A1 * dinv) w : even coefficients <- r0, odd <- -r1
        r0, r1 = t, [p.tmp() for _ in range(3)]
        p.fq6_mul(A_0, d, r0)
        p.fq6_mul(A_1, d, r1)
        for i in range(3):
            p.mov(F[2 * i], r0[i])
            p.A(r1[i]).neg().to(F[2 * i + 1])
        p.rel(*S0)
        p.rel(*S1)
        p.rel(*r1)

    def _powx_routine(self):
        """F <- F^x (x = BN_X) for cyclotomic F; the base sits in scratch at S_GBASE.  Same digits as pow_native
        (final_exp_native.rs:56-84); division by a unitary element = multiplication by its conjugate."""
        e = Emitter()
        e.label(self.lab("L3_powx"))
        self._powx_entry(e)
        e.salu(f"s_mov_b32 s{S_J}, {self.x_top - 1}")
        e.label(self.lab("L3_powx_loop"))
        e.salu(f"s_call_b64 {S_RET2}, {self.lab('L2_cyc')}")
        e.salu(f"s_bitcmp1_b64 {S_XNAF_NZ}, s{S_J}")
        e.salu(f"s_cbranch_scc0 {self.lab('L3_powx_zero')}")
        e.salu(f"s_bitcmp1_b64 {S_XNAF_NEG}, s{S_J}")
        e.salu(f"s_cbranch_scc1 {self.lab('L3_powx_neg')}")
        e.salu(f"s_call_b64 {S_RET2}, {self.lab('L2_mulG')}")
        e.salu(f"s_branch {self.lab('L3_powx_next')}")
        e.label(self.lab("L3_powx_neg"))
        e.salu(f"s_call_b64 {S_RET2}, {self.lab('L2_mulGc')}")
        e.salu(f"s_branch {self.lab('L3_powx_next')}")
        e.label(self.lab("L3_powx_zero"))
        self._powx_zero_digit(e)
        e.label(self.lab("L3_powx_next"))
        e.salu(f"s_sub_u32 s{S_J}, s{S_J}, 1")
        e.salu(f"s_cbranch_scc0 {self.lab('L3_powx_loop')}")
        e.salu(f"s_setpc_b64 {S_RET3}")
        self.sections.append(e)

    def _powx_entry(self, e):
        """Hook at the entry of the x-power routine (F = base, S_GBASE selects the base's scratch register)."""

    def _powx_zero_digit(self, e):
        """Hook for a zero digit of the x-power loop (the v3 builder reduces the accumulator here)."""

    # ---------------------------------------------------------------------------------------------
    def gsel(self, e, j):
        """S_GBASE <- byte offset of Fq12 scratch register j."""
        e.salu(f"s_mul_i32 s{S_GBASE}, s{S_GSTRIDE}, {6 * j}")

    def call2(self, e, name):
        e.salu(f"s_call_b64 {S_RET2}, {self.lab(name)}")

    def main_body(self, e):
        L = self.lab
        e.label(L("L_main"))
        e.label(L("L_item"))
        e.salu(f"s_cmp_ge_u32 s{S_ITEM}, s{S_NITEMS}")
        e.salu(f"s_cbranch_scc1 {L('L_done')}")
        # element index of this lane, clamped
        e.salu(f"s_lshl_b32 s{S_TMP0}, s{S_ITEM}, 8")
        e.emit(f"v_add_u32_e32 v{V_IDX}, s{S_TMP0}, v118", vw=[V_IDX])
        e.salu(f"s_sub_u32 s{S_TMP1}, s{S_N}, 1")
        e.emit(f"v_min_u32_e32 v{V_IDX8}, s{S_TMP1}, v{V_IDX}", vw=[V_IDX8])
        e.emit(f"v_lshlrev_b32_e32 v{V_IDX8}, 3, v{V_IDX8}", vw=[V_IDX8])
        e.emit("v_mov_b32_e32 v119, 0", vw=[119])
        p = Prog(e, self.labels)
        p.marsh = self.marsh
        p.set_temps(self.miller_temps())
        if self.do_miller:
            self.miller_main(e, p)
        else:
            # f_in (MyFq12, SoA): components 0..5 are the c0 parts of w^0..w^5, 6..11 the c1 parts
            self.io_walk_begin(e, S_FIN)
            for k in range(6):
                self.io_load_fq(e, 32 + 8 * k)                 # c0 parts staged in v[32:79]
            for k in range(6):
                self.io_load_fq(e, A0 + 8)                     # c1 part of coefficient k
                e.raw("s_waitcnt vmcnt(0)")
                for i in range(8):
                    e.emit(f"v_mov_b32_e32 v{A0 + i}, v{32 + 8 * k + i}", vw=[A0 + i])
                p.store(A0, self.F[k])
            p.reset_tags()
        if self.do_fexp:
            self.fexp_main(e, p)
        self.store_out(e, p)
        e.salu(f"s_add_u32 s{S_ITEM}, s{S_ITEM}, s{S_GRID}")
        e.salu(f"s_branch {L('L_item')}")
        e.label(L("L_done"))

    def miller_main(self, e, p):
        L = self.lab
        # ---- inputs: P = (x, y) -> PX.c0, PY.c0 ; Q = (x.c0, x.c1, y.c0, y.c1) -> QX, QY ; R = (Q, 1)
        self.io_walk_begin(e, S_G1)
        self.io_load_fq(e, A0)              # Px -> A.c0
        self.io_load_fq(e, B0)              # Py -> B.c0
        e.raw("s_waitcnt vmcnt(0)")
        self.zero_block(e, A0 + 8, 8)
        p.tagA = p.tagB = None
        p.store(A0, self.PX)
        for i in range(8):
            e.emit(f"v_mov_b32_e32 v{A0 + i}, v{B0 + i}", vw=[A0 + i])
        p.store(A0, self.PY)
        self.io_walk_begin(e, S_G2)
        self.io_load_fq(e, A0)
        self.io_load_fq(e, A0 + 8)
        self.io_load_fq(e, B0)
        self.io_load_fq(e, B0 + 8)
        e.raw("s_waitcnt vmcnt(0)")
        p.store(A0, self.QX)
        p.store(A0, self.R[0])
        for i in range(16):
            e.emit(f"v_mov_b32_e32 v{A0 + i}, v{B0 + i}", vw=[A0 + i])
        p.store(A0, self.QY)
        p.store(A0, self.R[1])
        self.one_into_A(e)
        p.store(A0, self.R[2])
        if self.track:
            p.store(A0, self.SCALE)
        p.tagA = p.tagB = None
        # ---- top digit (+1): f = tangent at Q, R = 2Q
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
        # S = +-Q
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
        # ---- Q1 = pi(Q) = (c2 conj(x), c3 conj(y)) ; -Q2 = (c2 conj(Q1.x), c3 neg_conj(Q1.y))   (:298-312)
        xi = (9, 1)
        c = f2pow(xi, (P_INT - 1) // 6)
        c2 = f2mul(c, c)
        c3 = f2mul(c2, c)
        C2, C3 = Const(c2[0], c2[1], "c2"), Const(c3[0], c3[1], "c3")
        p.reset_tags()
        p.A(self.QX).conj().mul(C2).to(self.SX)
        p.A(self.QY).conj().mul(C3).to(self.SY)
        self.call2(e, "L2_addmul")
        p.reset_tags()
        p.A(self.SX).conj().mul(C2).to(self.SX)
        p.A(self.SY).conj().neg().mul(C3).to(self.SY)          # neg_conjugate_fp2(y) = -conj(y)
        self.call2(e, "L2_addmul_last")
        if self.track:
            self.call2(e, "L2_descale")
        p.reset_tags()

    def fexp_main(self, e, p):
        """final_exp_native on F (LDS), F-centric schedule (tests/sched_model.py: final_exp_gpu)."""
        G0, GM, G2, G3, G4, G5, G6, G7 = range(8)
        tr = self.fexp_trace = []         # the straight-line call sequence, replayed by the v3 bound certification

        def st(j):
            tr.append(("st", j))
            self.gsel(e, j)
            self.call2(e, "L2_stG")

        def ld(j, conj=False):
            tr.append(("ld", j, conj))
            self.gsel(e, j)
            self.call2(e, "L2_ldGc" if conj else "L2_ldG")

        def mul(j, conj=False):
            tr.append(("mul", j, conj))
            self.gsel(e, j)
            self.call2(e, "L2_mulGc" if conj else "L2_mulG")

        def powx(j):
            tr.append(("powx", j))
            self.gsel(e, j)
            e.salu(f"s_call_b64 {S_RET3}, {self.lab('L3_powx')}")

        def c2(n):
            tr.append(("call", n))
            self.call2(e, n)
        # easy part (:195-206)
        st(G0); c2("L2_inv"); mul(G0, conj=True); st(G0); c2("L2_frob2"); mul(G0)
        # hard part (:130-169)
        st(GM)
        c2("L2_frob1"); st(G2)
        ld(GM); c2("L2_frob2"); st(G3)
        ld(GM); c2("L2_frob3"); mul(G3); mul(G2); st(G2)              # y0
        ld(GM); powx(GM); st(G3)                                      # mx
        powx(G3); st(G4)                                              # mx2
        powx(G4); st(G5)                                              # mx3
        ld(G3); c2("L2_frob1"); st(G6)                                # mxp
        ld(G4); c2("L2_frob1"); mul(G3); st(G7)                       # mx * mx2p
        ld(G4); c2("L2_frob2"); st(G3)                                # y2
        ld(G5); c2("L2_frob1"); mul(G5); c2("L2_conjF")               # y6
        c2("L2_cyc")                                                  # T0 = y6^2
        mul(G7, conj=True)                                            # * y4
        mul(G4, conj=True)                                            # * y5
        st(G0)
        ld(G6, conj=True); mul(G4, conj=True)                         # T1 = y3 * y5
        mul(G0); st(G5)                                               # T1 *= T0
        ld(G0); mul(G3); st(G0)                                       # T0 = y2 * T0
        ld(G5); c2("L2_cyc"); mul(G0); c2("L2_cyc"); st(G5)           # T1 = (T1^2 * T0)^2
        mul(GM, conj=True); st(G0)                                    # T0 = T1 * y1
        ld(G5); mul(G2); st(G5)                                       # T1 = T1 * y0
        ld(G0); c2("L2_cyc"); mul(G5)                                 # T0 = T0^2 * T1

    def store_out(self, e, p):
        L = self.lab
        p.reset_tags()
        # lanes past the end of the batch do not store
        e.emit(f"v_cmp_gt_u32_e32 vcc, s{S_N}, v{V_IDX}", w=["vcc"])
        e.raw("s_nop 1")
        e.salu(f"s_and_saveexec_b64 {S_SAVE_EXEC}, vcc")
        self.io_walk_begin(e, S_OUT)
        for half in range(2):
            for k in range(6):
                p.load(A0, self.F[k])
                p.wait()
                self.io_store_fq(e, A0 + 8 * half)
        # zero-divisor flag -> status word
        e.emit("v_cmp_ne_u32_e32 vcc, 0, v119", w=["vcc"])
        e.raw("s_nop 1")
        e.salu(f"s_and_saveexec_b64 s[60:61], vcc")
        e.emit("v_mov_b32_e32 v118, 1", vw=[118])
        e.emit("v_mov_b32_e32 v32, 0", vw=[32])
        e.emit(f"global_store_dword v32, v118, {S_STATUS}", kind="vmem")
        e.salu(f"s_mov_b64 exec, {S_SAVE_EXEC}")
        e.raw("s_waitcnt vmcnt(0)")
        # restore tid in v118 (used by the item loop)
        e.emit(f"v_lshrrev_b32_e32 v118, 4, v{V_LDS}", vw=[118])
