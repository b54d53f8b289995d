#!/usr/bin/env python3
"""kgen.py -- generator for the gfx950 whole-kernel assembly of the BN254 pairing engine (v2).

Everything a lane computes is emitted from here as ONE inline-asm blob per kernel: the Montgomery
leaf routines (L1), the Fq12 / curve-step routines built from them (L2) and the Miller-loop /
final-exponentiation control flow (L3).  hipcc only provides the kernel descriptor and hands the
kernel arguments over in SGPRs.  Why: on gfx950 nearly every integer VALU instruction costs one
~4.5-cycle issue slot for the single wave a SIMD can hold at this register/LDS footprint
(profiles/valu_calib_r01.txt), so throughput is instruction count -- and compiler-generated code
around hand-written multiplies spends a third of its slots on hazard s_nops, 64-bit address
arithmetic and ABI marshalling.

Model
  * "accumulator machine" on Fq2 values: block A = v[0:15], block B = v[16:31]; every L1 routine
    computes A <- op(A, B).  Values live in SLOTS: LDS (10/lane, ds_read/write_b128), VGPR homes
    (v[144:255], 7 slots), AGPRs (a[0:255], 16 slots) and, for cold Fq12 temporaries, global
    scratch.  L2 code is a sequence of  ldA / ldB / call / stA.
  * 32x32 MAC unit:  v_mad_u64_u32 acc, carry -> SGPR pair ; v_addc_co_u32 third word  (the addc is
    delayed by two instructions and the carry registers rotate: a VALU may not read an SGPR/VCC a
    VALU wrote < 2 instructions earlier on gfx940/gfx950).
  * straight carry chains (add/sub) are interleaved 4-way with distinct carry registers.
The same instruction stream is executed by tools/ksim.py (single-lane simulator) in the CPU tests.
"""
import os
import re
import sys

P_INT = 21888242871839275222246405745257275088696311157297823662689037894645226208583
P_LIMBS = [(P_INT >> (32 * i)) & 0xFFFFFFFF for i in range(8)]
N0 = (-pow(P_INT, -1, 1 << 32)) % (1 << 32)
R_INT = 1 << 256
BN_X = 4965661367192848881
SIX_U_PLUS_2_NAF = [
    0, 0, 0, 1, 0, 1, 0, -1, 0, 0, 1, -1, 0, 0, 1, 0, 0, 1, 1, 0, -1, 0, 0, 1, 0, -1, 0, 0, 0, 0,
    1, 1, 1, 0, 0, -1, 0, 0, 1, 0, 0, 0, 0, 0, -1, 0, 0, 1, 1, 0, 0, -1, 0, 0, 0, 1, 1, 0, -1, 0,
    0, 1, 0, 1, 1,
]

# ------------------------------------------------------------------------------------------ register map
A0 = 0          # block A: v[0:15]
B0 = 16         # block B: v[16:31]
TMP_FIRST, TMP_LAST = 32, 103
PV0 = 104       # v[104:111]: modulus limbs in VGPRs (carry chains cannot take SGPR + carry-in)
V_LDS = 112     # v112/113/114: LDS byte address of this lane (+0, +64 KiB, +128 KiB)
V_GOFF = 115    # v115: lane's byte offset inside a global scratch slot (lane * 64)
V_IDX8 = 116    # v116: element index * 8 (SoA batch I/O)
V_IDX = 117     # v117: element index (unclamped) ; v118, v119: scratch for the control code
HOME0 = 120     # VGPR home slots: v[120:247] = 8 x 16 ; v[248:255] spare
N_HOME = 8
N_AGPR_SLOTS = 16
N_LDS_SLOTS = 10
BLOCK = 256
# SGPRs (fixed, clobbered): 36..43 p, 44 n0', carries 46..53, return addresses 54..59, scalars 60..
S_P = 36
S_N0 = 44
S_CARRY = ["s[46:47]", "s[48:49]", "s[50:51]", "s[52:53]"]
S_RET1 = "s[54:55]"
S_RET2 = "s[56:57]"
S_RET3 = "s[58:59]"


def mont(x):
    return x * R_INT % P_INT


def limbs8(x):
    return [(x >> (32 * i)) & 0xFFFFFFFF for i in range(8)]


# ------------------------------------------------------------------------------------------ emitter
class Emitter:
    """Instruction list with SGPR read/write annotations + hazard post-pass.

    Hazards handled (gfx940/gfx950, LLVM GCNHazardRecognizer):
      * VALU writes SGPR/VCC -> VALU reads it: 2 wait states
      * global store of > 64 bits -> VALU overwrites its data registers: 2 wait states
    Labels and control-flow instructions reset the tracking conservatively."""

    def __init__(self):
        self.ins = []   # dicts: text, r (set), w (set), kind, store_regs

    def emit(self, text, r=(), w=(), kind="valu", vw=(), store=()):
        self.ins.append(dict(text=text, r=frozenset(r), w=frozenset(w), kind=kind, vw=frozenset(vw), store=frozenset(store)))

    def label(self, name):
        self.ins.append(dict(text=name + ":", r=frozenset(), w=frozenset(), kind="label", vw=frozenset(), store=frozenset()))

    def salu(self, text):
        self.emit(text, kind="salu")

    def raw(self, text, kind="other"):
        self.emit(text, kind=kind)

    def finalize(self):
        out = []
        last_w = {}          # carry reg -> index in out
        last_store = {}      # vgpr -> index of the wide store that reads it
        for it in self.ins:
            if it["kind"] == "label":
                # unknown predecessors: be conservative
                out.append(it["text"])
                idx = len(out)
                for k in list(last_w):
                    last_w[k] = idx - 1
                continue
            need = 0
            if it["kind"] == "valu":
                for reg in it["r"]:
                    if reg in last_w:
                        gap = len(out) - last_w[reg] - 1
                        need = max(need, 2 - gap)
                for reg in it["vw"]:
                    if reg in last_store:
                        gap = len(out) - last_store[reg] - 1
                        need = max(need, 2 - gap)
            if need > 0:
                out.append("s_nop %d" % (need - 1))
            if it["kind"] == "valu":
                for reg in it["w"]:
                    last_w[reg] = len(out)
            for reg in it["store"]:
                last_store[reg] = len(out)
            out.append(it["text"])
        return out


# ------------------------------------------------------------------------------------------ L1: field routines on fixed blocks
class Pool:
    def __init__(self, first, last):
        self.free_regs = list(range(first, last + 1))
        self.used = set()

    def alloc(self):
        fs = set(self.free_regs)
        pick = None
        for r in self.free_regs:
            if (r ^ 1) not in fs:
                pick = r
                break
        if pick is None:
            pick = self.free_regs[0]
        self.free_regs.remove(pick)
        self.used.add(pick)
        return pick

    def find_orphan(self):
        fs = set(self.free_regs)
        for r in self.free_regs:
            if (r ^ 1) not in fs:
                self.free_regs.remove(r)
                self.used.add(r)
                return r
        return None

    def alloc_pair(self):
        for r in self.free_regs:
            if r % 2 == 0 and (r + 1) in self.free_regs:
                self.free_regs.remove(r)
                self.free_regs.remove(r + 1)
                self.used.update((r, r + 1))
                return r
        raise RuntimeError("out of VGPR pairs")

    def free(self, *regs):
        for r in regs:
            assert r not in self.free_regs, r
            self.free_regs.append(r)
        self.free_regs.sort()


class L1:
    """Generates the leaf routines.  `e` is the Emitter; carries rotate over S_CARRY."""

    def __init__(self, e):
        self.e = e
        self.pool = Pool(TMP_FIRST, TMP_LAST)
        self.carry_next = 0
        self.p = [f"s{S_P + i}" for i in range(8)]
        self.n0 = f"s{S_N0}"
        self.pv = list(range(PV0, PV0 + 8))

    def next_carry(self):
        c = S_CARRY[self.carry_next % len(S_CARRY)]
        self.carry_next += 1
        return c

    # ---- 96-bit column accumulator -----------------------------------------------------
    class Col:
        def __init__(self, g):
            self.g = g
            self.cur = g.pool.alloc_pair()
            self.oth = g.pool.alloc_pair()
            self.top_init = False
            self.empty = True
            self.pending = []

        def _top(self):
            return self.oth + 1

        def _addc(self, c):
            t = self._top()
            if self.top_init:
                self.g.e.emit(f"v_addc_co_u32_e64 v{t}, {c}, 0, v{t}, {c}", r=[c], w=[c], vw=[t])
            else:
                self.g.e.emit(f"v_addc_co_u32_e64 v{t}, {c}, 0, 0, {c}", r=[c], w=[c], vw=[t])
                self.top_init = True

        def _mad(self, A, B):
            P = f"v[{self.cur}:{self.cur + 1}]"
            c = self.g.next_carry()
            if self.empty:
                self.g.e.emit(f"v_mad_u64_u32 {P}, {c}, {A}, {B}, 0", w=[c], vw=[self.cur, self.cur + 1])
                self.empty = False
                return
            self.g.e.emit(f"v_mad_u64_u32 {P}, {c}, {A}, {B}, {P}", w=[c], vw=[self.cur, self.cur + 1])
            self.pending.append(c)
            while len(self.pending) > 2:
                self._addc(self.pending.pop(0))

        def mac(self, a, b):
            self._mad(f"v{a}" if isinstance(a, int) else a, f"v{b}" if isinstance(b, int) else b)

        def add_word(self, w):
            if self.empty:
                self.g.e.emit(f"v_mov_b32_e32 v{self.cur}, v{w}", vw=[self.cur])
                self.g.e.emit(f"v_mov_b32_e32 v{self.cur + 1}, 0", vw=[self.cur + 1])
                self.empty = False
                return
            self._mad(f"v{w}", "1")

        def flush(self):
            while self.pending:
                self._addc(self.pending.pop(0))

        def low(self):
            return self.cur

        def shift(self, keep_low):
            assert not self.empty
            lo, mid = self.cur, self.cur + 1
            nlo, ntop_old = self.oth, self.oth + 1
            self.g.e.emit(f"v_mov_b32_e32 v{nlo}, v{mid}", vw=[nlo])
            self.flush()
            if not self.top_init:
                self.g.e.emit(f"v_mov_b32_e32 v{ntop_old}, 0", vw=[ntop_old])
            kept = None
            if keep_low:
                orphan = self.g.pool.find_orphan()
                if orphan is not None:
                    self.g.e.emit(f"v_mov_b32_e32 v{orphan}, v{lo}", vw=[orphan])
                    kept = orphan
                    self.cur, self.oth = self.oth, self.cur
                else:
                    kept = lo
                    self.g.pool.free(mid)
                    newp = self.g.pool.alloc_pair()
                    self.cur, self.oth = self.oth, newp
            else:
                self.cur, self.oth = self.oth, self.cur
            self.top_init = False
            return kept

        def finish(self):
            self.flush()
            return self.cur

        def release(self, keep=()):
            for p in (self.cur, self.oth):
                for r in (p, p + 1):
                    if r not in keep:
                        self.g.pool.free(r)

    # ---- building blocks ---------------------------------------------------------------
    def product(self, a, b):
        col = L1.Col(self)
        out = []
        for k in range(15):
            for i in range(max(0, k - 7), min(7, k) + 1):
                col.mac(a[i], b[k - i])
            out.append(col.shift(keep_low=True))
        lo = col.finish()
        out.append(lo)
        col.release(keep=(lo,))
        return out

    def redc(self, t, out_regs):
        col = L1.Col(self)
        m = []
        for k in range(8):
            col.add_word(t[k])
            for i in range(k):
                col.mac(m[i], self.p[k - i])
            mk = self.pool.alloc()
            self.e.emit(f"v_mul_lo_u32 v{mk}, v{col.low()}, {self.n0}", vw=[mk])
            m.append(mk)
            col.mac(mk, self.p[0])
            col.shift(keep_low=False)
        r = []
        for k in range(8, 16):
            col.add_word(t[k])
            for i in range(k - 7, 8):
                col.mac(m[i], self.p[k - i])
            if k < 15:
                r.append(col.shift(keep_low=True))
        lo = col.finish()
        r.append(lo)
        col.release(keep=(lo,))
        self.pool.free(*m)
        return self.cond_sub_p(r, out_regs)

    def fips(self, a, b, out_regs):
        col = L1.Col(self)
        m = []
        for k in range(8):
            for i in range(k + 1):
                col.mac(a[i], b[k - i])
            for i in range(k):
                col.mac(m[i], self.p[k - i])
            mk = self.pool.alloc()
            self.e.emit(f"v_mul_lo_u32 v{mk}, v{col.low()}, {self.n0}", vw=[mk])
            m.append(mk)
            col.mac(mk, self.p[0])
            col.shift(keep_low=False)
        r = []
        for k in range(8, 15):
            for i in range(k - 7, 8):
                col.mac(a[i], b[k - i])
            for i in range(k - 7, 8):
                col.mac(m[i], self.p[k - i])
            r.append(col.shift(keep_low=True))
        lo = col.finish()
        r.append(lo)
        col.release(keep=(lo,))
        self.pool.free(*m)
        return self.cond_sub_p(r, out_regs)

    def cond_sub_p(self, r, out_regs):
        """r (< 2p) -> r mod p in out_regs; frees r."""
        d = [self.pool.alloc() for _ in range(8)]
        for i in range(8):
            if i == 0:
                self.e.emit(f"v_sub_co_u32_e32 v{d[i]}, vcc, v{r[i]}, v{self.pv[i]}", w=["vcc"], vw=[d[i]])
            else:
                self.e.emit(f"v_subb_co_u32_e32 v{d[i]}, vcc, v{r[i]}, v{self.pv[i]}, vcc", r=["vcc"], w=["vcc"], vw=[d[i]])
        for i in range(8):
            self.e.emit(f"v_cndmask_b32_e32 v{out_regs[i]}, v{d[i]}, v{r[i]}, vcc", r=["vcc"], vw=[out_regs[i]])
        self.pool.free(*d)
        for x in r:
            if x not in out_regs:
                self.pool.free(x)
        return out_regs

    # ---- interleaved carry chains --------------------------------------------------------
    CARRIES = ["vcc"] + S_CARRY

    def chains(self, specs, lags=None):
        """specs: list of carry chains; a chain = list of links (op, dst, a, b), op in add/sub, operands
        VGPR numbers or operand strings.  Chain k uses carry register CARRIES[k].  Links are emitted
        round-robin; chain k starts `lags[k]` rounds late (needed when it consumes, limb by limb, what an
        earlier chain produces).  With >= 3 chains in flight no s_nop is needed; the hazard post-pass
        pads otherwise."""
        assert len(specs) <= len(self.CARRIES)
        lags = lags or [0] * len(specs)
        rounds = max(len(ch) + lg for ch, lg in zip(specs, lags))
        for step in range(rounds):
            for ci, ch in enumerate(specs):
                i = step - lags[ci]
                if i < 0 or i >= len(ch):
                    continue
                op, dst, a, b = ch[i]
                c = self.CARRIES[ci]
                first = (i == 0)
                A = f"v{a}" if isinstance(a, int) else a
                Bv = f"v{b}" if isinstance(b, int) else b
                base = "v_add" if op == "add" else "v_sub"
                suf = "c" if op == "add" else "b"
                if c == "vcc":
                    txt = f"{base}_co_u32_e32 v{dst}, vcc, {A}, {Bv}" if first else f"{base}{suf}_co_u32_e32 v{dst}, vcc, {A}, {Bv}, vcc"
                else:
                    txt = f"{base}_co_u32_e64 v{dst}, {c}, {A}, {Bv}" if first else f"{base}{suf}_co_u32_e64 v{dst}, {c}, {A}, {Bv}, {c}"
                self.e.emit(txt, r=([] if first else [c]), w=[c], vw=[dst])
        return self.CARRIES[:len(specs)]

    def select(self, dst, if0, if1, c):
        """dst[i] = c ? if1[i] : if0[i]"""
        for i in range(len(dst)):
            if c == "vcc":
                self.e.emit(f"v_cndmask_b32_e32 v{dst[i]}, v{if0[i]}, v{if1[i]}, vcc", r=["vcc"], vw=[dst[i]])
            else:
                self.e.emit(f"v_cndmask_b32_e64 v{dst[i]}, v{if0[i]}, v{if1[i]}, {c}", r=[c], vw=[dst[i]])

    def fq2_addsub(self, op, swap=False):
        """A <- A + B | A - B | B - A (mod p), both components, four interleaved chains."""
        a = [list(range(A0, A0 + 8)), list(range(A0 + 8, A0 + 16))]
        b = [list(range(B0, B0 + 8)), list(range(B0 + 8, B0 + 16))]
        if swap:
            a, b = b, a
        t = [[self.pool.alloc() for _ in range(8)] for _ in range(2)]
        d = [[self.pool.alloc() for _ in range(8)] for _ in range(2)]
        if op == "add":
            # t = a + b ; d = t - p ; result = borrow ? t : d
            # the correction chain of limb i needs t[i]: run it one link behind
            specs = []
            for h in range(2):
                specs.append([("add", t[h][i], a[h][i], b[h][i]) for i in range(8)])
            for h in range(2):
                specs.append([("sub", d[h][i], t[h][i], self.pv[i]) for i in range(8)])
            # lag the correction chains by one step: emit first link of the add chains alone
            cs = self.chains(specs, lags=[0, 0, 1, 1])
            for h in range(2):
                out = list(range(A0 + 8 * h, A0 + 8 * h + 8))
                self.select(out, d[h], t[h], cs[2 + h])
        else:
            # t = a - b ; d = t + p ; result = borrow ? d : t
            specs = []
            for h in range(2):
                specs.append([("sub", t[h][i], a[h][i], b[h][i]) for i in range(8)])
            for h in range(2):
                specs.append([("add", d[h][i], t[h][i], self.pv[i]) for i in range(8)])
            cs = self.chains(specs, lags=[0, 0, 1, 1])
            for h in range(2):
                out = list(range(A0 + 8 * h, A0 + 8 * h + 8))
                self.select(out, t[h], d[h], cs[h])
        for h in range(2):
            self.pool.free(*t[h])
            self.pool.free(*d[h])

    # ---- the routines (bodies only; kernels wrap them with label + s_setpc) --------------
    def r_mul(self):
        a0, a1 = list(range(0, 8)), list(range(8, 16))
        b0, b1 = list(range(16, 24)), list(range(24, 32))
        sa = [self.pool.alloc() for _ in range(8)]
        sb = [self.pool.alloc() for _ in range(8)]
        self.chains([[("add", sa[i], a0[i], a1[i]) for i in range(8)], [("add", sb[i], b0[i], b1[i]) for i in range(8)]])
        v0 = self.product(a0, b0)
        v1 = self.product(a1, b1)
        v2 = self.product(sa, sb)
        self.pool.free(*sa)
        self.pool.free(*sb)
        # c1 = v2 - v0 - v1 ; c0 = v0 - v1 (+ p*2^256 when negative): three chains, the second c1 chain lags
        c0 = [self.pool.alloc() for _ in range(16)]
        self._three_way(v2, v0, v1, c0)
        self.pool.free(*v1)
        self.pool.free(*v0)
        self.redc(c0, list(range(0, 8)))
        for r in c0:
            if r in self.pool.free_regs:
                continue
            self.pool.free(r)
        self.redc(v2, list(range(8, 16)))
        for r in v2:
            if r not in self.pool.free_regs:
                self.pool.free(r)

    def _three_way(self, v2, v0, v1, c0):
        """v2 <- v2 - v0 - v1 (in place), c0 <- v0 - v1, then c0[8..15] += p if c0 went negative."""
        cA, cB, cC = S_CARRY[0], S_CARRY[1], S_CARRY[2]
        n = 16
        for step in range(n + 1):
            i = step
            if i < n:
                f = (i == 0)
                self.e.emit((f"v_sub_co_u32_e64 v{v2[i]}, {cA}, v{v2[i]}, v{v0[i]}" if f else
                             f"v_subb_co_u32_e64 v{v2[i]}, {cA}, v{v2[i]}, v{v0[i]}, {cA}"), r=([] if f else [cA]), w=[cA], vw=[v2[i]])
                self.e.emit((f"v_sub_co_u32_e64 v{c0[i]}, {cC}, v{v0[i]}, v{v1[i]}" if f else
                             f"v_subb_co_u32_e64 v{c0[i]}, {cC}, v{v0[i]}, v{v1[i]}, {cC}"), r=([] if f else [cC]), w=[cC], vw=[c0[i]])
            j = step - 1
            if 0 <= j < n:
                f = (j == 0)
                self.e.emit((f"v_sub_co_u32_e64 v{v2[j]}, {cB}, v{v2[j]}, v{v1[j]}" if f else
                             f"v_subb_co_u32_e64 v{v2[j]}, {cB}, v{v2[j]}, v{v1[j]}, {cB}"), r=([] if f else [cB]), w=[cB], vw=[v2[j]])
        # conditional + p on the high half of c0 (borrow in cC)
        msk = self.pool.alloc()
        self.e.emit(f"v_cndmask_b32_e64 v{msk}, 0, -1, {cC}", r=[cC], vw=[msk])
        tmp = [self.pool.alloc() for _ in range(2)]
        for i in range(8):
            t = tmp[i % 2]
            self.e.emit(f"v_and_b32_e32 v{t}, {self.p[i]}, v{msk}", vw=[t])
            if i == 0:
                self.e.emit(f"v_add_co_u32_e32 v{c0[8 + i]}, vcc, v{c0[8 + i]}, v{t}", w=["vcc"], vw=[c0[8 + i]])
            else:
                self.e.emit(f"v_addc_co_u32_e32 v{c0[8 + i]}, vcc, v{c0[8 + i]}, v{t}, vcc", r=["vcc"], w=["vcc"], vw=[c0[8 + i]])
        self.pool.free(msk, *tmp)

    def r_sqr(self):
        a0, a1 = list(range(0, 8)), list(range(8, 16))
        s = [self.pool.alloc() for _ in range(8)]
        t = [self.pool.alloc() for _ in range(8)]
        dd = [self.pool.alloc() for _ in range(8)]
        d = [self.pool.alloc() for _ in range(8)]
        # s = a0 + a1 (unreduced, < 2p) ; d = (a0 - a1) mod p
        specs = [[("add", s[i], a0[i], a1[i]) for i in range(8)], [("sub", t[i], a0[i], a1[i]) for i in range(8)],
                 [("add", dd[i], t[i], self.pv[i]) for i in range(8)]]
        carries = self.chains(specs, lags=[0, 0, 1])
        self.select(d, t, dd, carries[1])    # borrow of (a0 - a1) ? t + p : t
        self.pool.free(*t)
        self.pool.free(*dd)
        t0 = self.product(s, d)
        self.pool.free(*s)
        self.pool.free(*d)
        t1 = self.product(a0, a1)
        # t1 <- 2 t1 : two half-length chains cannot be split (one carry chain); interleave nothing -> use the
        # shift form instead: t1 = t1 << 1 via v_alignbit (no carries at all)
        for i in range(15, 0, -1):
            self.e.emit(f"v_alignbit_b32 v{t1[i]}, v{t1[i]}, v{t1[i - 1]}, 31", vw=[t1[i]])
        self.e.emit(f"v_lshlrev_b32_e32 v{t1[0]}, 1, v{t1[0]}", vw=[t1[0]])
        self.redc(t0, list(range(0, 8)))
        for r in t0:
            if r not in self.pool.free_regs:
                self.pool.free(r)
        self.redc(t1, list(range(8, 16)))
        for r in t1:
            if r not in self.pool.free_regs:
                self.pool.free(r)

    def r_mulfq(self):
        """A <- (A.c0 * B.c0, A.c1 * B.c0)"""
        k = list(range(16, 24))
        self.fips(list(range(0, 8)), k, list(range(0, 8)))
        self.fips(list(range(8, 16)), k, list(range(8, 16)))

    def r_fqmul(self):
        """A.c0 <- A.c0 * B.c0 (Fq)"""
        self.fips(list(range(0, 8)), list(range(16, 24)), list(range(0, 8)))

    def r_fqsqr(self):
        self.fips(list(range(0, 8)), list(range(0, 8)), list(range(0, 8)))

    def r_add(self):
        self.fq2_addsub("add")

    def r_sub(self):
        self.fq2_addsub("sub")

    def r_rsub(self):
        self.fq2_addsub("sub", swap=True)

    def r_dbl(self):
        """A <- 2A: shift left by one then conditional subtract (two components interleaved)."""
        t = [[self.pool.alloc() for _ in range(8)] for _ in range(2)]
        d = [[self.pool.alloc() for _ in range(8)] for _ in range(2)]
        for h in range(2):
            a = list(range(A0 + 8 * h, A0 + 8 * h + 8))
            for i in range(7, 0, -1):
                self.e.emit(f"v_alignbit_b32 v{t[h][i]}, v{a[i]}, v{a[i - 1]}, 31", vw=[t[h][i]])
            self.e.emit(f"v_lshlrev_b32_e32 v{t[h][0]}, 1, v{a[0]}", vw=[t[h][0]])
        specs = [[("sub", d[h][i], t[h][i], self.pv[i]) for i in range(8)] for h in range(2)]
        self.chains(specs)
        cs = ["vcc"] + S_CARRY
        for h in range(2):
            self.select(list(range(A0 + 8 * h, A0 + 8 * h + 8)), d[h], t[h], cs[h])
            self.pool.free(*t[h])
            self.pool.free(*d[h])

    def r_neg(self):
        """A <- -A = p - A, with 0 -> 0."""
        d = [[self.pool.alloc() for _ in range(8)] for _ in range(2)]
        specs = [[("sub", d[h][i], self.pv[i], A0 + 8 * h + i) for i in range(8)] for h in range(2)]
        self.chains(specs)
        for h in range(2):
            a = list(range(A0 + 8 * h, A0 + 8 * h + 8))
            z = self.pool.alloc()
            self.e.emit(f"v_or3_b32 v{z}, v{a[0]}, v{a[1]}, v{a[2]}", vw=[z])
            self.e.emit(f"v_or3_b32 v{z}, v{z}, v{a[3]}, v{a[4]}", vw=[z])
            self.e.emit(f"v_or3_b32 v{z}, v{z}, v{a[5]}, v{a[6]}", vw=[z])
            self.e.emit(f"v_or_b32_e32 v{z}, v{z}, v{a[7]}", vw=[z])
            self.e.emit(f"v_cmp_eq_u32_e32 vcc, 0, v{z}", w=["vcc"])
            self.select(a, d[h], a, "vcc")       # zero ? a (= 0) : p - a
            self.pool.free(z)
            self.pool.free(*d[h])

    def r_negc1(self):
        """A.c1 <- -A.c1 (conjugate_fp2)"""
        d = [self.pool.alloc() for _ in range(8)]
        a = list(range(A0 + 8, A0 + 16))
        self.chains([[("sub", d[i], self.pv[i], a[i]) for i in range(8)]])
        z = self.pool.alloc()
        self.e.emit(f"v_or3_b32 v{z}, v{a[0]}, v{a[1]}, v{a[2]}", vw=[z])
        self.e.emit(f"v_or3_b32 v{z}, v{z}, v{a[3]}, v{a[4]}", vw=[z])
        self.e.emit(f"v_or3_b32 v{z}, v{z}, v{a[5]}, v{a[6]}", vw=[z])
        self.e.emit(f"v_or_b32_e32 v{z}, v{z}, v{a[7]}", vw=[z])
        self.e.emit(f"v_cmp_eq_u32_e32 vcc, 0, v{z}", w=["vcc"])
        self.select(a, d, a, "vcc")
        self.pool.free(z, *d)

    def r_mulxi(self):
        """A <- (9 + u) A = (9 a0 - a1) + (a0 + 9 a1) u.
        n0 = 8 a0 + a0 - a1 + p in (0, 10p), n1 = 8 a1 + a1 + a0 in [0, 10p) as 9-limb integers (shift +
        five interleaved carry chains), then each is reduced by q*p with q estimated from the top bits
        (q_est in {q-1, q}, never above) and one conditional subtraction."""
        a0 = list(range(A0, A0 + 8))
        a1 = list(range(A0 + 8, A0 + 16))
        zero = self.pool.alloc()
        self.e.emit(f"v_mov_b32_e32 v{zero}, 0", vw=[zero])
        w = []
        for x in (a0, a1):
            s8 = [self.pool.alloc() for _ in range(9)]
            self.e.emit(f"v_lshlrev_b32_e32 v{s8[0]}, 3, v{x[0]}", vw=[s8[0]])
            for i in range(1, 8):
                self.e.emit(f"v_alignbit_b32 v{s8[i]}, v{x[i]}, v{x[i - 1]}, 29", vw=[s8[i]])
            self.e.emit(f"v_lshrrev_b32_e32 v{s8[8]}, 29, v{x[7]}", vw=[s8[8]])
            w.append(s8)
        w0, w1 = w

        def ext(v, i):
            return v[i] if i < 8 else zero
        ch = [
            [("add", w0[i], w0[i], ext(a0, i)) for i in range(9)],
            [("sub", w0[i], w0[i], ext(a1, i)) for i in range(9)],
            [("add", w0[i], w0[i], ext(self.pv, i)) for i in range(9)],
            [("add", w1[i], w1[i], ext(a1, i)) for i in range(9)],
            [("add", w1[i], w1[i], ext(a0, i)) for i in range(9)],
        ]
        self.chains(ch, lags=[0, 1, 2, 0, 1])
        for (wv, out) in ((w0, a0), (w1, a1)):
            top = self.pool.alloc()
            q = self.pool.alloc()
            self.e.emit(f"v_alignbit_b32 v{top}, v{wv[8]}, v{wv[7]}, 24", vw=[top])
            self.e.emit(f"v_mul_u32_u24_e32 v{q}, 1354, v{top}", vw=[q])
            self.e.emit(f"v_lshrrev_b32_e32 v{q}, 16, v{q}", vw=[q])
            prs = [self.pool.alloc_pair() for _ in range(8)]
            for i in range(8):
                self.e.emit(f"v_mad_u64_u32 v[{prs[i]}:{prs[i] + 1}], {S_CARRY[3]}, v{q}, {self.p[i]}, 0", w=[S_CARRY[3]], vw=[prs[i], prs[i] + 1])
            lo = [prs[i] for i in range(8)] + [zero]
            hi = [zero] + [prs[i] + 1 for i in range(8)]
            self.chains([[("sub", wv[i], wv[i], lo[i]) for i in range(9)], [("sub", wv[i], wv[i], hi[i]) for i in range(9)]], lags=[0, 1])
            for pr in prs:
                self.pool.free(pr, pr + 1)
            d = [self.pool.alloc() for _ in range(8)]
            self.chains([[("sub", d[i], wv[i], self.pv[i]) for i in range(8)]])
            self.select(out, d, wv[:8], "vcc")
            self.pool.free(top, q, *d)
        self.pool.free(zero, *w0)
        self.pool.free(*w1)


if __name__ == "__main__":
    e = Emitter()
    g = L1(e)
    for name in ("r_mul", "r_sqr", "r_mulfq", "r_add", "r_sub", "r_dbl", "r_neg", "r_mulxi"):
        e2 = Emitter()
        g2 = L1(e2)
        getattr(g2, name)()
        lines = e2.finalize()
        print(name, "instrs", len(lines), "nops", sum(1 for l in lines if l.startswith("s_nop")), "max tmp", max(g2.pool.used) if g2.pool.used else None)


# ------------------------------------------------------------------------------------------ code alignment post-pass
# Measured on gfx950 (tools/exp/l1_bench.py, DESIGN.md "issue model"): a lone wave issues one VALU instruction per 4 cycles, but an
# 8-byte instruction that is only 4-byte aligned costs ~1 extra cycle on average (it straddles a 32-byte fetch window every
# fourth time, +4 cycles).  align_code() keeps every 8-byte instruction 8-byte aligned: a 4-byte VOP1/VOP2 instruction in front
# of it is re-encoded as VOP3 (_e64, 8 bytes, same operation and speed) or, where that is impossible, an s_nop is inserted;
# labels are 8-byte aligned with s_nop padding.
_INLINE_INT = re.compile(r"^-?\d+$|^0x[0-9a-fA-F]+$")
_VOP3_ONLY = ("v_mad_", "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_hi_i32", "v_ashrrev_i64", "v_lshlrev_b64", "v_lshrrev_b64", "v_lshl_add_", "v_lshl_or_",
              "v_and_or_", "v_add3_", "v_alignbit_", "v_bfe_", "v_accvgpr_", "v_fma_", "v_perm_", "v_pk_", "v_mbcnt_", "v_readlane_", "v_writelane_",
              "v_add_lshl_", "v_xad_", "v_or3_", "v_dot", "v_mfma", "v_cvt_pk")
_MEM = ("ds_", "global_", "flat_", "buffer_", "scratch_", "s_load_", "s_store_", "s_buffer_", "s_memtime", "s_memrealtime", "s_dcache")
_NO_E64 = ("v_addc_co_u32", "v_subb_co_u32", "v_subbrev_co_u32", "v_add_co_u32", "v_sub_co_u32", "v_subrev_co_u32", "v_cndmask_b32", "v_cmp", "v_nop",
           "v_readfirstlane", "v_movrel", "v_swap", "v_fmac", "v_mac", "v_madmk", "v_madak", "v_fmamk", "v_fmaak")


def _operands(text):
    rest = text.split(None, 1)[1] if " " in text.strip() else ""
    return [t.strip() for t in re.split(r",(?![^\[]*\])", rest) if t.strip()]


def _has_literal(text):
    for t in _operands(text):
        t0 = t.split()[0]
        if _INLINE_INT.match(t0):
            v = int(t0, 0)
            if not -16 <= v <= 64:
                return True
    return False


def insn_size(text):
    """Encoded size in bytes (4 or 8) of one gfx950 instruction as the generators write it."""
    op = text.split()[0]
    if op.startswith(_MEM) or op.endswith("_e64") or op.startswith(_VOP3_ONLY):
        return 8
    if op.startswith("s_") or op.startswith("v_"):
        if op in ("s_waitcnt", "s_nop", "s_endpgm", "s_branch", "s_barrier", "s_sleep") or op.startswith(("s_cbranch", "s_call_b64", "s_setpc", "s_getpc")):
            return 4
        return 8 if _has_literal(text) else 4
    raise ValueError("unknown instruction class: " + text)


def _to_e64(text):
    """VOP3 re-encoding of a 4-byte VOP1/VOP2 instruction, or None when there is none with the same syntax."""
    op = text.split()[0]
    if not op.startswith("v_") or op.startswith(_NO_E64) or _has_literal(text):
        return None
    base = op[:-4] if op.endswith("_e32") else op
    if op.endswith("_e64"):
        return None
    return base + "_e64" + text[len(op):]


def align_code(lines):
    out, off, last = [], 0, None            # last: index in `out` of the previous instruction if it may be re-encoded
    for ln in lines:
        t = ln.strip()
        if not t or t.startswith((";", "//", ".")):
            out.append(ln)
            continue
        if t.endswith(":"):
            if off % 8:
                out.append("s_nop 0")
                off += 4
            out.append(ln)
            last = None
            continue
        size = insn_size(t)
        if size == 8 and off % 8:
            conv = _to_e64(out[last]) if last is not None else None
            if conv is not None:
                out[last] = conv
            else:
                out.append("s_nop 0")
            off += 4
        out.append(ln)
        off += size
        last = len(out) - 1 if size == 4 else None
    return out


def max_branch_distance(lines):
    """Largest |target - (pc + 4)| in bytes over all s_call_b64 / s_branch / s_cbranch_* with a label target."""
    off, lab, ins = 0, {}, []
    for l in lines:
        t = l.strip()
        if not t or t.startswith((";", "//", ".")):
            continue
        if t.endswith(":"):
            lab[t[:-1]] = off
            continue
        ins.append((off, t))
        off += insn_size(t)
    worst = 0
    for o, t in ins:
        op = t.split()[0]
        if op in ("s_call_b64", "s_branch") or op.startswith("s_cbranch"):
            tgt = t.split(",")[-1].strip() if op == "s_call_b64" else t.split()[-1]
            if tgt in lab:
                worst = max(worst, abs(lab[tgt] - (o + 4)))
    return worst
