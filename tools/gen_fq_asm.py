#!/usr/bin/env python3
"""Generator for the gfx950 inline-asm Montgomery kernels (8 x u32 limbs, R = 2^256).

Emits plonky2-bn254-pairing_amd/csrc/fq_asm_gen.h: leaf device functions whose bodies are ONE
inline-asm statement each, with operands bound to the physical VGPRs the AMDGPU calling
convention already uses for vector arguments (v0..v31) and temporaries taken from an
explicit clobber range.  The multiply-accumulate unit is

    v_mad_u64_u32  acc[lo:hi], vcc, a_i, b_j, acc[lo:hi]     ; 32x32 + 64 -> 64, carry -> vcc
    v_addc_co_u32  acc_top, vcc, 0, acc_top, vcc             ; carry into the third word

i.e. product scanning (column-wise) with a 96-bit column accumulator: 2 VALU instructions
per 32x32 limb product, no data-dependent control flow.  The modulus limbs and -p^-1 mod
2^32 are SGPR operands (VOP3 on gfx9 takes no 32-bit literal).

Routines:
  fq_mul      : fused product + Montgomery reduction (FIPS), result in [0,p)
  fq2_mul     : Karatsuba over Fq2 with lazy reduction: 3 double-width products, 2 REDCs
  fq2_sqr     : (a0+a1)(a0-a1), 2 a0 a1: 2 products, 2 REDCs
  fq2_mul_fq  : Fq2 x Fq (2 fused multiplies sharing one operand)
Run:  python tools/gen_fq_asm.py  (rewrites the header in place)
"""
import os
import sys

P_INT = 21888242871839275222246405745257275088696311157297823662689037894645226208583
P_LIMBS = [(P_INT >> (32 * i)) & 0xFFFFFFFF for i in range(8)]
N0 = (-pow(P_INT, -1, 1 << 32)) % (1 << 32)
assert N0 == 0xE4866389


def caller_saved_vgprs(first, last):
    """AMDGPU calling convention: v40-47, v56-63, v72-79, ... (every other block of 8 from v40) are
    callee-saved; everything else is free for a leaf routine to clobber without save/restore."""
    out = []
    for r in range(first, last + 1):
        if r >= 40 and ((r - 40) // 8) % 2 == 0:
            continue
        out.append(r)
    return out


class Asm:
    """Tiny macro-assembler with a physical VGPR pool and a gfx940/gfx950 hazard post-pass:
    a VALU that reads an SGPR / VCC written by a VALU needs 2 wait states in between
    (LLVM GCNHazardRecognizer `VALUWriteSGPRVALURead`, hasVDecCoExecHazard)."""

    def __init__(self, first_tmp, last_tmp, carry_regs):
        self.ins = []                 # (text, reads, writes)
        self.free_regs = caller_saved_vgprs(first_tmp, last_tmp)
        self.used = set()
        self.carry_regs = list(carry_regs)   # SGPR-pair operand strings for rotating carries
        self.carry_next = 0
        self.n_valu = 0

    def emit(self, s, r=(), w=()):
        self.ins.append((s, frozenset(r), frozenset(w)))

    def next_carry(self):
        c = self.carry_regs[self.carry_next % len(self.carry_regs)]
        self.carry_next += 1
        return c

    def finalize(self):
        """Insert s_nop so that VALU-written carry registers are read by a VALU >= 2 instructions later."""
        out = []
        last_write = {}
        for (s, r, w) in self.ins:
            need = 0
            for reg in r:
                if reg in last_write:
                    gap = len(out) - last_write[reg] - 1
                    need = max(need, 2 - gap)
            if need > 0:
                out.append("s_nop %d" % (need - 1))
            for reg in w:
                last_write[reg] = len(out)
            out.append(s)
        self.lines = out
        self.n_valu = sum(1 for l in out if l.startswith("v_"))
        self.n_nop = sum(1 for l in out if l.startswith("s_nop"))
        return out

    # ---- register pool ---------------------------------------------------
    def alloc(self):
        # prefer a register whose pair partner is busy, so aligned pairs stay available
        pick = None
        fs = set(self.free_regs)
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
        for i, r in enumerate(self.free_regs):
            if r % 2 == 0 and (r + 1) in self.free_regs:
                self.free_regs.remove(r)
                self.free_regs.remove(r + 1)
                self.used.update((r, r + 1))
                return r
        raise RuntimeError("out of VGPR pairs")

    def free(self, *regs):
        for r in regs:
            assert r not in self.free_regs
            self.free_regs.append(r)
        self.free_regs.sort()


# ---------------------------------------------------------------------------------------
# A more careful accumulator: keeps TWO aligned pairs (e0,e1),(f0,f1).  Column accumulates its
# low 64 bits in pair E and its third word in f1.  At the end of the column the low word is e0,
# and the carried value is (e1, f1); one move f0 <- e1 makes pair F the next column's low 64
# bits, and e1 (after the move) becomes the next column's third word (initialised by the first
# carry-producing addc in its e64 form, which writes 0+0+carry).  e0 is handed to the caller when
# the low word is a result limb, in which case a fresh even register replaces it.
# ---------------------------------------------------------------------------------------
class Col:
    def __init__(self, asm):
        self.a = asm
        self.E = asm.alloc_pair()
        self.F = asm.alloc_pair()
        self.cur = self.E         # pair holding lo64 of the running column
        self.oth = self.F
        self.top_init = False     # third word (oth+1) initialised?
        self.empty = True         # running column has value 0
        self.pending = []         # carry registers whose addc into `top` is still to be emitted

    def _top(self):
        return self.oth + 1

    def _addc(self, c):
        t = self._top()
        if self.top_init:
            self.a.emit(f"v_addc_co_u32_e64 v{t}, {c}, 0, v{t}, {c}", r=[c], w=[c])
        else:
            self.a.emit(f"v_addc_co_u32_e64 v{t}, {c}, 0, 0, {c}", r=[c], w=[c])
            self.top_init = True

    def _mad(self, A, B):
        P = f"v[{self.cur}:{self.cur + 1}]"
        if self.empty:
            c = self.a.next_carry()
            self.a.emit(f"v_mad_u64_u32 {P}, {c}, {A}, {B}, 0", w=[c])
            self.empty = False
            return
        c = self.a.next_carry()
        self.a.emit(f"v_mad_u64_u32 {P}, {c}, {A}, {B}, {P}", w=[c])
        self.pending.append(c)
        # keep two younger instructions between a mad and the addc that consumes its carry
        while len(self.pending) > 2:
            self._addc(self.pending.pop(0))

    def mac(self, a, b):
        A = f"v{a}" if isinstance(a, int) else a
        B = f"v{b}" if isinstance(b, int) else b
        self._mad(A, B)

    def add_word(self, w):
        if self.empty:
            self.a.emit(f"v_mov_b32_e32 v{self.cur}, v{w}")
            self.a.emit(f"v_mov_b32_e32 v{self.cur + 1}, 0")
            self.empty = False
            return
        self._mad(f"v{w}", "1")

    def flush(self):
        while self.pending:
            self._addc(self.pending.pop(0))

    def low(self):
        return self.cur

    def shift(self, keep_low):
        """End of column.  Returns the register with the low word if keep_low (caller owns it)."""
        assert not self.empty
        lo, mid = self.cur, self.cur + 1
        nlo, ntop_old = self.oth, self.oth + 1
        self.a.emit(f"v_mov_b32_e32 v{nlo}, v{mid}")
        self.flush()
        if not self.top_init:
            self.a.emit(f"v_mov_b32_e32 v{ntop_old}, 0")
        kept = None
        if keep_low:
            orphan = self.a.find_orphan()
            if orphan is not None:
                # compact: park the result limb in a register whose pair partner is busy anyway,
                # and keep our aligned pair for the next column
                self.a.emit(f"v_mov_b32_e32 v{orphan}, v{lo}")
                kept = orphan
                self.cur, self.oth = self.oth, self.cur
            else:
                kept = lo
                self.a.free(mid)
                newp = self.a.alloc_pair()
                self.cur, self.oth = self.oth, newp
        else:
            self.cur, self.oth = self.oth, self.cur
        self.top_init = False
        return kept

    def finish(self):
        """Returns (lo, mid, top_or_None) of the running column and releases nothing."""
        self.flush()
        return self.cur, self.cur + 1, (self._top() if self.top_init else None)

    def release(self, keep=()):
        for p in (self.cur, self.oth):
            for r in (p, p + 1):
                if r not in keep:
                    self.a.free(r)


def gen_product(asm, a, b, square=False):
    """16-limb product of 8-limb operands (register-number lists).  Returns 16 result registers."""
    col = Col(asm)
    out = []
    for k in range(15):
        lo_i = max(0, k - 7)
        hi_i = min(7, k)
        for i in range(lo_i, hi_i + 1):
            col.mac(a[i], b[k - i])
        out.append(col.shift(keep_low=True))
    lo, mid, top = col.finish()
    # after the 15th shift the running "column 15" holds the top limb in `lo`
    out.append(lo)
    col.release(keep=(lo,))
    return out


def gen_redc(asm, t, S, out_regs=None):
    """Montgomery reduction of a 16-limb value t (< p*2^256) -> 8 limbs in [0, 2p) ... then a
    conditional subtract of p -> [0,p).  S: dict with SGPR operand strings 'p'[8], 'n0'."""
    col = Col(asm)
    m = []
    for k in range(8):
        col.add_word(t[k])
        for i in range(k):
            col.mac(m[i], S["p"][k - i])
        mk = asm.alloc()
        asm.emit(f"v_mul_lo_u32 v{mk}, v{col.low()}, {S['n0']}")
        m.append(mk)
        col.mac(mk, S["p"][0])
        col.shift(keep_low=False)
    r = []
    for k in range(8, 16):
        col.add_word(t[k])
        for i in range(k - 7, 8):
            col.mac(m[i], S["p"][k - i])
        if k < 15:
            r.append(col.shift(keep_low=True))
    lo, mid, top = col.finish()
    r.append(lo)
    col.release(keep=(lo,))
    asm.free(*m)
    return cond_sub_p(asm, r, S, out_regs)


def gen_fips(asm, a, b, S, out_regs=None):
    """Fused product + reduction (FIPS).  Result in [0,p)."""
    col = Col(asm)
    m = []
    for k in range(8):
        for i in range(k + 1):
            col.mac(a[i], b[k - i])
        for i in range(k):
            col.mac(m[i], S["p"][k - i])
        mk = asm.alloc()
        asm.emit(f"v_mul_lo_u32 v{mk}, v{col.low()}, {S['n0']}")
        m.append(mk)
        col.mac(mk, S["p"][0])
        col.shift(keep_low=False)
    r = []
    for k in range(8, 15):
        for i in range(k - 7, 8):
            col.mac(a[i], b[k - i])
        for i in range(k - 7, 8):
            col.mac(m[i], S["p"][k - i])
        r.append(col.shift(keep_low=True))
    lo, mid, top = col.finish()
    r.append(lo)
    col.release(keep=(lo,))
    asm.free(*m)
    return cond_sub_p(asm, r, S, out_regs)


def load_pv(asm, S):
    """VGPR copies of the modulus limbs: a carry-in (VCC) plus an SGPR source would be two
    constant-bus reads, which gfx9 VOP2/VOP3 do not allow."""
    pv = [asm.alloc() for _ in range(8)]
    for i in range(8):
        asm.emit(f"v_mov_b32_e32 v{pv[i]}, {S['p'][i]}")
    S["pv"] = pv


def cond_sub_p(asm, r, S, out_regs=None):
    """r (8 regs, value < 2p) -> r mod p into out_regs (allocated if None).  Frees r."""
    d = [asm.alloc() for _ in range(8)]
    pv = S["pv"]
    for i in range(8):
        if i == 0:
            asm.emit(f"v_sub_co_u32_e32 v{d[i]}, vcc, v{r[i]}, v{pv[i]}", r=[], w=['vcc'])
        else:
            asm.emit(f"v_subb_co_u32_e32 v{d[i]}, vcc, v{r[i]}, v{pv[i]}, vcc", r=['vcc'], w=['vcc'])
    out = out_regs if out_regs is not None else [asm.alloc() for _ in range(8)]
    for i in range(8):
        # vcc = borrow -> keep r, else take d
        asm.emit(f"v_cndmask_b32_e32 v{out[i]}, v{d[i]}, v{r[i]}, vcc", r=['vcc'], w=[])
    asm.free(*d)
    for x in r:
        if x not in out:
            asm.free(x)
    return out


def add_nored(asm, a, b):
    """8-limb add without reduction (caller guarantees no overflow of 2^256)."""
    o = [asm.alloc() for _ in range(8)]
    for i in range(8):
        if i == 0:
            asm.emit(f"v_add_co_u32_e32 v{o[i]}, vcc, v{a[i]}, v{b[i]}", r=[], w=['vcc'])
        else:
            asm.emit(f"v_addc_co_u32_e32 v{o[i]}, vcc, v{a[i]}, v{b[i]}, vcc", r=['vcc'], w=['vcc'])
    return o


def sub_mod(asm, a, b, S):
    """(a - b) mod p for a,b in [0,p): 8 sub + 8 masked add."""
    o = [asm.alloc() for _ in range(8)]
    for i in range(8):
        if i == 0:
            asm.emit(f"v_sub_co_u32_e32 v{o[i]}, vcc, v{a[i]}, v{b[i]}", r=[], w=['vcc'])
        else:
            asm.emit(f"v_subb_co_u32_e32 v{o[i]}, vcc, v{a[i]}, v{b[i]}, vcc", r=['vcc'], w=['vcc'])
    msk = asm.alloc()
    asm.emit(f"v_cndmask_b32_e64 v{msk}, 0, -1, vcc", r=['vcc'], w=[])
    tmp = asm.alloc()
    for i in range(8):
        asm.emit(f"v_and_b32_e32 v{tmp}, {S['p'][i]}, v{msk}")
        if i == 0:
            asm.emit(f"v_add_co_u32_e32 v{o[i]}, vcc, v{o[i]}, v{tmp}", r=[], w=['vcc'])
        else:
            asm.emit(f"v_addc_co_u32_e32 v{o[i]}, vcc, v{o[i]}, v{tmp}, vcc", r=['vcc'], w=['vcc'])
    asm.free(msk, tmp)
    return o


def wide_sub(asm, x, y, n=16):
    """x -= y over n limbs in place; vcc = final borrow."""
    for i in range(n):
        if i == 0:
            asm.emit(f"v_sub_co_u32_e32 v{x[i]}, vcc, v{x[i]}, v{y[i]}", r=[], w=['vcc'])
        else:
            asm.emit(f"v_subb_co_u32_e32 v{x[i]}, vcc, v{x[i]}, v{y[i]}, vcc", r=['vcc'], w=['vcc'])


def wide_add(asm, x, y, n=16):
    for i in range(n):
        if i == 0:
            asm.emit(f"v_add_co_u32_e32 v{x[i]}, vcc, v{x[i]}, v{y[i]}", r=[], w=['vcc'])
        else:
            asm.emit(f"v_addc_co_u32_e32 v{x[i]}, vcc, v{x[i]}, v{y[i]}, vcc", r=['vcc'], w=['vcc'])


def cond_add_p_high(asm, x, S):
    """if vcc (borrow): x[8..15] += p   (x is a 16-limb value that went negative)."""
    msk = asm.alloc()
    tmp = asm.alloc()
    asm.emit(f"v_cndmask_b32_e64 v{msk}, 0, -1, vcc", r=['vcc'], w=[])
    for i in range(8):
        asm.emit(f"v_and_b32_e32 v{tmp}, {S['p'][i]}, v{msk}")
        if i == 0:
            asm.emit(f"v_add_co_u32_e32 v{x[8 + i]}, vcc, v{x[8 + i]}, v{tmp}", r=[], w=['vcc'])
        else:
            asm.emit(f"v_addc_co_u32_e32 v{x[8 + i]}, vcc, v{x[8 + i]}, v{tmp}, vcc", r=['vcc'], w=['vcc'])
    asm.free(msk, tmp)


# ---------------------------------------------------------------------------------------
N_CARRY = 4


def sregs(n_out):
    """Operand order: n_out in/out vectors, N_CARRY scratch SGPR pairs (early-clobber outputs),
    then the 9 SGPR constants."""
    base = n_out + N_CARRY
    return {"p": [f"%{base + i}" for i in range(8)], "n0": f"%{base + 8}"}


def carry_ops(n_out):
    return [f"%{n_out + i}" for i in range(N_CARRY)]


def routine_fq_mul():
    asm = Asm(16, 250, carry_ops(2))
    S = sregs(2)
    load_pv(asm, S)
    a = list(range(0, 8))
    b = list(range(8, 16))
    gen_fips(asm, a, b, S, out_regs=list(range(0, 8)))
    return asm


def routine_fq2_mul_fq():
    # a0 = v0-7, a1 = v8-15, k = v16-23 ; out (a0*k, a1*k) -> v0-15
    asm = Asm(24, 250, carry_ops(3))
    S = sregs(3)
    load_pv(asm, S)
    a0, a1, k = list(range(0, 8)), list(range(8, 16)), list(range(16, 24))
    gen_fips(asm, a0, k, S, out_regs=list(range(0, 8)))
    gen_fips(asm, a1, k, S, out_regs=list(range(8, 16)))
    return asm


def routine_fq2_mul():
    asm = Asm(32, 250, carry_ops(4))
    SREGS = sregs(4)
    load_pv(asm, SREGS)
    a0, a1, b0, b1 = (list(range(0, 8)), list(range(8, 16)), list(range(16, 24)), list(range(24, 32)))
    sa = add_nored(asm, a0, a1)
    sb = add_nored(asm, b0, b1)
    v0 = gen_product(asm, a0, b0)
    v1 = gen_product(asm, a1, b1)
    v2 = gen_product(asm, sa, sb)
    asm.free(*sa)
    asm.free(*sb)
    # c1 = v2 - v0 - v1 (>= 0)
    wide_sub(asm, v2, v0)
    wide_sub(asm, v2, v1)
    # c0 = v0 - v1 (+ p*2^256 if negative)
    wide_sub(asm, v0, v1)
    cond_add_p_high(asm, v0, SREGS)
    asm.free(*v1)
    # inputs dead now: v0..v31 reusable as outputs
    gen_redc(asm, v0, SREGS, out_regs=list(range(0, 8)))
    asm.free(*[r for r in v0 if r not in range(0, 8) and r not in asm.free_regs])
    gen_redc(asm, v2, SREGS, out_regs=list(range(8, 16)))
    return asm


def routine_fq2_sqr():
    # (a0 + a1 u)^2 = (a0+a1)(a0-a1) + 2 a0 a1 u
    asm = Asm(16, 250, carry_ops(2))
    SREGS = sregs(2)
    load_pv(asm, SREGS)
    a0, a1 = list(range(0, 8)), list(range(8, 16))
    s = add_nored(asm, a0, a1)           # < 2p
    d = sub_mod(asm, a0, a1, SREGS)      # [0,p)
    t0 = gen_product(asm, s, d)          # < 2p^2 < p*2^256
    asm.free(*s)
    asm.free(*d)
    t1 = gen_product(asm, a0, a1)
    # double t1 (16 limbs): t1 += t1  (< 2p^2)
    wide_add(asm, t1, t1)
    gen_redc(asm, t0, SREGS, out_regs=list(range(0, 8)))
    gen_redc(asm, t1, SREGS, out_regs=list(range(8, 16)))
    return asm


def c_escape(lines):
    return "\n".join('        "%s\\n\\t"' % l for l in lines)


def clobbers(asm, first_tmp):
    regs = [f'"v{r}"' for r in sorted(asm.used)]
    return ", ".join(regs + ['"vcc"'])


HEADER = '''// GENERATED by tools/gen_fq_asm.py -- do not edit by hand.
// gfx950 inline-asm Montgomery kernels: 8 x u32 limbs, R = 2^256, operands in the
// calling-convention VGPRs, modulus limbs / n0' as SGPR operands.
#pragma once
#include <stdint.h>

typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
struct Fq2V { u32x8 c0, c1; };

#define BN254_P_SGPR_OPERANDS \\
    "s"(0x%08xu), "s"(0x%08xu), "s"(0x%08xu), "s"(0x%08xu), "s"(0x%08xu), "s"(0x%08xu), "s"(0x%08xu), "s"(0x%08xu), "s"(0x%08xu)
''' % tuple(P_LIMBS + [N0])


def main():
    out_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "plonky2-bn254-pairing_amd", "csrc", "fq_asm_gen.h")
    parts = [HEADER]

    r = routine_fq_mul()
    parts.append(f'''
// a*b*R^-1 mod p, inputs in [0,p).  {(r.finalize(), r.n_valu)[1]} VALU instructions + {r.n_nop} s_nop.
__device__ __attribute__((noinline)) u32x8 fq_mul_asm(u32x8 a, u32x8 b) {{
    uint64_t cy0, cy1, cy2, cy3;
    asm volatile(
{c_escape(r.finalize())}
        : "+{{v[0:7]}}"(a), "+{{v[8:15]}}"(b), "=&s"(cy0), "=&s"(cy1), "=&s"(cy2), "=&s"(cy3)
        : BN254_P_SGPR_OPERANDS
        : {clobbers(r, 16)});
    return a;
}}
''')

    r = routine_fq2_mul_fq()
    parts.append(f'''
// (a0 + a1 u) * k, k in Fq.  {(r.finalize(), r.n_valu)[1]} VALU instructions + {r.n_nop} s_nop.
__device__ __attribute__((noinline)) Fq2V fq2_mul_fq_asm(u32x8 a0, u32x8 a1, u32x8 k) {{
    uint64_t cy0, cy1, cy2, cy3;
    asm volatile(
{c_escape(r.finalize())}
        : "+{{v[0:7]}}"(a0), "+{{v[8:15]}}"(a1), "+{{v[16:23]}}"(k), "=&s"(cy0), "=&s"(cy1), "=&s"(cy2), "=&s"(cy3)
        : BN254_P_SGPR_OPERANDS
        : {clobbers(r, 24)});
    Fq2V o; o.c0 = a0; o.c1 = a1; return o;
}}
''')

    r = routine_fq2_mul()
    parts.append(f'''
// (a0 + a1 u)(b0 + b1 u), u^2 = -1: Karatsuba, lazy reduction (3 products, 2 REDC).  {(r.finalize(), r.n_valu)[1]} VALU instructions + {r.n_nop} s_nop.
__device__ __attribute__((noinline)) Fq2V fq2_mul_asm(u32x8 a0, u32x8 a1, u32x8 b0, u32x8 b1) {{
    uint64_t cy0, cy1, cy2, cy3;
    asm volatile(
{c_escape(r.finalize())}
        : "+{{v[0:7]}}"(a0), "+{{v[8:15]}}"(a1), "+{{v[16:23]}}"(b0), "+{{v[24:31]}}"(b1), "=&s"(cy0), "=&s"(cy1), "=&s"(cy2), "=&s"(cy3)
        : BN254_P_SGPR_OPERANDS
        : {clobbers(r, 32)});
    Fq2V o; o.c0 = a0; o.c1 = a1; return o;
}}
''')

    r = routine_fq2_sqr()
    parts.append(f'''
// (a0 + a1 u)^2 = (a0+a1)(a0-a1) + 2 a0 a1 u.  {(r.finalize(), r.n_valu)[1]} VALU instructions + {r.n_nop} s_nop.
__device__ __attribute__((noinline)) Fq2V fq2_sqr_asm(u32x8 a0, u32x8 a1) {{
    uint64_t cy0, cy1, cy2, cy3;
    asm volatile(
{c_escape(r.finalize())}
        : "+{{v[0:7]}}"(a0), "+{{v[8:15]}}"(a1), "=&s"(cy0), "=&s"(cy1), "=&s"(cy2), "=&s"(cy3)
        : BN254_P_SGPR_OPERANDS
        : {clobbers(r, 16)});
    Fq2V o; o.c0 = a0; o.c1 = a1; return o;
}}
''')
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    with open(out_path, "w") as f:
        f.write("".join(parts))
    print("wrote", os.path.normpath(out_path))


if __name__ == "__main__":
    main()
