#!/usr/bin/env python3
"""ksim.py -- single-lane interpreter for the gfx950 instruction subset tools/kgen.py emits.

Runs the generated kernels (text after the hazard post-pass) on one lane with Python integers:
VGPR/AGPR/SGPR files, LDS and global memory as byte-addressed dicts of dwords, labels, branches,
s_call_b64 / s_setpc_b64.  Besides computing, it re-checks what the generator must guarantee:
  * VALU read of an SGPR/VCC written by a VALU < 2 instructions earlier (gfx940/950 hazard)
  * even alignment of 64-bit VGPR operands, 16-bit DS offsets, 13-bit global offsets
  * every register read was written before (catches allocation bugs)
It is test infrastructure (tests/test_kgen.py), not part of the product.
"""
import functools
import re

M32 = 0xFFFFFFFF
M64 = 0xFFFFFFFFFFFFFFFF


class SimError(Exception):
    pass


@functools.lru_cache(maxsize=None)
def _vpair(tok):
    """low register of a v[lo:hi] operand, or None (cached: the simulator parses every operand of every executed instruction)"""
    m = re.match(r"v\[(\d+):", tok)
    return int(m.group(1)) if m else None


@functools.lru_cache(maxsize=None)
def _vreg(tok):
    """n of a plain vN operand, or None"""
    return int(tok[1:]) if tok[0] == "v" and tok[1:].isdigit() else None


@functools.lru_cache(maxsize=None)
def _spair(tok):
    m = re.match(r"s\[(\d+):(\d+)\]", tok)
    return int(m.group(1)) if m else None


@functools.lru_cache(maxsize=None)
def _sreg(tok):
    m = re.match(r"s(\d+)$", tok)
    return int(m.group(1)) if m else None


def _split_args(rest):
    return [a.strip() for a in re.split(r",\s*(?![^\[]*\])", rest)] if rest else []


class Machine:
    def __init__(self, check_uninit=True):
        self.v = [None] * 256
        self.a = [None] * 256
        self.s = {}
        self.vcc = 0
        self.scc = 0
        self.exec = 1
        self.lds = {}
        self.gmem = {}
        self.check_uninit = check_uninit
        self.valu_w = {}      # sgpr name -> dyn. instruction counter of the last VALU write
        self.count = 0        # dynamic instruction count
        self.count_valu = 0
        self.count_nop = 0
        self.max_acc = 0      # largest |column accumulator| that fitted 64 signed bits
        self.exact = {}       # low register of a 64-bit accumulator pair -> its true integer value
        self.transient_wraps = 0
        self.l2_stack, self.l2_incl = [], {}     # profile mode: instructions inclusive of callees, per outermost L2 routine
        self.profile = None   # dict: (region label, opcode) -> dynamic count, when set to {} before run()
        self.hook = None      # profile mode: hook(opcode, args, region label, call stack) per executed instruction
        self.stack = []       # profile mode: [(return register, callee label)] of the s_call_b64 frames that are open
        self.call_log = None  # list: labels of the L2 routines called, in order (bound certification cross-check)
        self.max_stored = 0   # largest |signed dword| written to LDS (v3: limb magnitudes of stored values)

    # ---------------------------------------------------------------- scalar register helpers
    def sget(self, name):
        if name == "vcc":
            return self.vcc
        if name == "exec":
            return self.exec
        lo = _spair(name)
        if lo is not None:
            a, b = self.s.get(lo), self.s.get(lo + 1)
            if a is None or b is None:
                raise SimError(f"uninitialised {name}")
            return a | (b << 32)
        r_ = _sreg(name)
        if r_ is not None:
            x = self.s.get(r_)
            if x is None:
                raise SimError(f"uninitialised {name}")
            return x
        raise SimError("bad sgpr " + name)

    def sset(self, name, val):
        if name == "vcc":
            self.vcc = val & M64
            return
        if name == "exec":
            self.exec = val & M64
            return
        lo = _spair(name)
        if lo is not None:
            self.s[lo] = val & M32
            self.s[lo + 1] = (val >> 32) & M32
            return
        r_ = _sreg(name)
        if r_ is not None:
            self.s[r_] = val & M32
            return
        raise SimError("bad sgpr " + name)

    # 64-bit column accumulators may wrap around TRANSIENTLY (two's complement sums are exact mod 2^64: L1v4.kfips adds the
    # Karatsuba difference products before the terms that cancel them); what must never happen is that a wrapped value is
    # CONSUMED.  `exact` keeps the true integer of every pair written by v_mad_i64_i32 / v_lshl_add_u64; any other read of a
    # register of a pair whose true value does not fit 64 signed bits is an error.
    def exact_of(self, tok):
        """true value of a 64-bit accumulator operand (falls back to the signed register content)"""
        lo = _vpair(tok)
        c = self.vsrc64(tok, check=False)
        if lo is not None:
            ex = self.exact.get(lo)
            if ex is not None and (ex - c) % (1 << 64) == 0:
                return ex
        return c - (1 << 64) if c >> 63 else c

    def set_exact(self, lo, val):
        self.v[lo] = val & M32
        self.v[lo + 1] = (val >> 32) & M32
        self.exact[lo] = val
        if -(1 << 63) <= val < (1 << 63):
            self.max_acc = max(self.max_acc, abs(val))
        else:
            self.transient_wraps += 1

    def _consume(self, r):
        ex = self.exact.get(r & ~1)
        if ex is not None and not (-(1 << 63) <= ex < (1 << 63)):
            cur = (self.v[r & ~1] or 0) | ((self.v[(r & ~1) + 1] or 0) << 32)
            if (ex - cur) % (1 << 64) == 0:
                raise SimError(f"a wrapped 64-bit accumulator (v[{r & ~1}:{(r & ~1) + 1}], true value {ex}) is consumed")

    def vsrc(self, tok, valu=True):
        """32-bit source operand of a VALU instruction."""
        r = _vreg(tok)
        if r is not None:
            if self.exact:
                self._consume(r)
            x = self.v[r]
            if x is None:
                if self.check_uninit:
                    raise SimError(f"read of uninitialised {tok}")
                return 0
            return x
        if tok[0] == "s" or tok == "vcc":
            if valu:
                self._hazard(tok)
            return self.sget(tok) & M32
        if tok == "-1":
            return M32
        return int(tok, 0) & M32

    def vsrc64(self, tok, check=True):
        lo = _vpair(tok)
        if lo is not None:
            if lo % 2:
                raise SimError("odd-aligned 64-bit VGPR operand " + tok)
            if check:
                self._consume(lo)
            a, b = self.v[lo], self.v[lo + 1]
            if a is None or b is None:
                if self.check_uninit:
                    raise SimError(f"read of uninitialised {tok}")
                a, b = a or 0, b or 0
            return a | (b << 32)
        if tok.startswith("s["):                       # a 64-bit scalar operand (the digit extraction's rounding constant)
            self._hazard(tok)
            return self.sget(tok)
        return int(tok, 0) & M64

    def _hazard(self, reg):
        w = self.valu_w.get(reg)
        if w is not None and self.count - w - 1 < 2:
            raise SimError(f"hazard: VALU reads {reg} {self.count - w - 1} wait states after a VALU write (dyn #{self.count})")

    def carry_in(self, reg):
        self._hazard(reg)
        return self.sget(reg) & 1

    def carry_out(self, reg, bit):
        self.sset(reg, bit)
        self.valu_w[reg] = self.count

    def vset(self, tok, val):
        r = _vreg(tok)
        self.v[r] = val & M32
        if self.exact:
            self.exact.pop(r & ~1, None)


_PARSED = {}


def _parse(lines):
    key = id(lines)
    hit = _PARSED.get(key)
    if hit is not None and hit[0] is lines:
        return hit[1], hit[2]
    prog = []
    labels = {}
    for ln in lines:
        ln = ln.strip()
        if not ln or ln.startswith(";") or ln.startswith("//"):
            continue
        if ln.endswith(":"):
            labels[ln[:-1]] = len(prog)
            continue
        if ln.startswith("."):
            continue                                  # assembler directives (.p2align)
        op, _, rest = ln.partition(" ")
        if op.endswith("_e64"):
            op = op[:-4] + "_e32"                     # VOP3 re-encodings of VOP1/VOP2 instructions (kgen.align_code)
        prog.append((op, _split_args(rest.strip()), ln))
    _PARSED.clear()
    _PARSED[key] = (lines, prog, labels)
    return prog, labels


def run(lines, m, entry=None, max_steps=200_000_000, trace=None, start_pc=None, stop_label=None):
    """Execute `lines` on machine `m` until s_endpgm (returns None), or -- stop_label -- until control ARRIVES at that label
    (returns the program counter to resume from with start_pc): the lane-cooperative kernels are run one lane at a time, round
    by round (tests/test_cvm.py)."""
    prog, labels = _parse(lines)
    pc = start_pc if start_pc is not None else (labels[entry] if entry else 0)
    stop_pc = labels[stop_label] if stop_label is not None else None
    fresh = True
    v = m.v
    steps = 0
    region = None
    if m.profile is not None:          # region of an instruction = nearest preceding L1_/L2_/L3_/L_main label
        region, cur = [], "prologue"
        by_pc = {}
        for name, at in labels.items():
            if name.startswith(("L1_", "L2_", "L3_", "L_main", "LM_")):
                by_pc.setdefault(at, name)
        for i in range(len(prog)):
            cur = by_pc.get(i, cur)
            region.append(cur)
    while True:
        if pc >= len(prog):
            raise SimError("fell off the end of the program")
        if pc == stop_pc and not fresh:
            return pc
        fresh = False
        op, a, text = prog[pc]
        if region is not None:
            k_ = (region[pc], op)
            m.profile[k_] = m.profile.get(k_, 0) + 1
            if op == "s_call_b64" and (a[1].startswith("L2_") or (a[1].startswith("L_hop") and prog[labels[a[1]]][1][0].startswith("L2_"))):
                m.l2_stack.append(a[1])
            elif op == "s_setpc_b64" and a[0] == "s[56:57]" and m.l2_stack:
                m.l2_stack.pop()
            top = m.l2_stack[0] if m.l2_stack else "(main)"
            m.l2_incl[top] = m.l2_incl.get(top, 0) + 1
            if m.hook is not None:
                m.hook(op, a, region[pc], m.stack)
            if op == "s_call_b64":
                tgt = a[1]
                while prog[labels[tgt]][0] == "s_branch" and tgt.startswith("L_hop"):
                    tgt = prog[labels[tgt]][1][0]
                m.stack.append((a[0], tgt))
            elif op == "s_setpc_b64" and m.stack and m.stack[-1][0] == a[0]:
                m.stack.pop()
        pc += 1
        steps += 1
        if steps > max_steps:
            raise SimError("step limit")
        try:
            if op == "v_mad_u64_u32":
                r = m.vsrc(a[2]) * m.vsrc(a[3]) + m.vsrc64(a[4])
                lo = _vpair(a[0])
                if lo % 2:
                    raise SimError("odd-aligned 64-bit VGPR dest")
                v[lo] = r & M32
                v[lo + 1] = (r >> 32) & M32
                m.carry_out(a[1], (r >> 64) & 1)
                m.count_valu += 1
            elif op == "v_mad_i64_i32":
                x, y = m.vsrc(a[2]), m.vsrc(a[3])
                x = x - (1 << 32) if x >> 31 else x
                y = y - (1 << 32) if y >> 31 else y
                r = x * y + m.exact_of(a[4])
                lo = _vpair(a[0])
                if lo % 2:
                    raise SimError("odd-aligned 64-bit VGPR dest")
                m.set_exact(lo, r)
                m.carry_out(a[1], 0)
                m.count_valu += 1
            elif op == "v_lshl_add_u64":
                r = (m.exact_of(a[1]) << (m.vsrc(a[2]) & 7)) + m.exact_of(a[3])
                lo = _vpair(a[0])
                m.set_exact(lo, r)
                m.count_valu += 1
            elif op == "v_ashrrev_i64":
                sh = m.vsrc(a[1]) & 63
                c = m.vsrc64(a[2])
                c = c - (1 << 64) if c >> 63 else c
                r = c >> sh
                lo = _vpair(a[0])
                v[lo] = r & M32
                v[lo + 1] = (r >> 32) & M32
                m.count_valu += 1
            elif op == "v_ashrrev_i32_e32":
                x = m.vsrc(a[2])
                x = x - (1 << 32) if x >> 31 else x
                m.vset(a[0], x >> (m.vsrc(a[1]) & 31))
                m.count_valu += 1
            elif op == "v_lshl_add_u32":
                m.vset(a[0], (m.vsrc(a[1]) << (m.vsrc(a[2]) & 31)) + m.vsrc(a[3]))
                m.count_valu += 1
            elif op == "v_xad_u32":                 # (a ^ b) + c
                m.vset(a[0], (m.vsrc(a[1]) ^ m.vsrc(a[2])) + m.vsrc(a[3]))
                m.count_valu += 1
            elif op == "v_add_lshl_u32":            # (a + b) << c
                m.vset(a[0], ((m.vsrc(a[1]) + m.vsrc(a[2])) & M32) << (m.vsrc(a[3]) & 31))
                m.count_valu += 1
            elif op == "v_lshl_or_b32":
                m.vset(a[0], (m.vsrc(a[1]) << (m.vsrc(a[2]) & 31)) | m.vsrc(a[3]))
                m.count_valu += 1
            elif op == "v_sub_u32_e32":
                m.vset(a[0], m.vsrc(a[1]) - m.vsrc(a[2]))
                m.count_valu += 1
            elif op == "v_subrev_u32_e32":
                m.vset(a[0], m.vsrc(a[2]) - m.vsrc(a[1]))
                m.count_valu += 1
            elif op == "v_bfe_i32":
                w = m.vsrc(a[3]) & 31
                x = (m.vsrc(a[1]) >> (m.vsrc(a[2]) & 31)) & ((1 << w) - 1)
                m.vset(a[0], x - (1 << w) if w and (x >> (w - 1)) else x)
                m.count_valu += 1
            elif op == "v_bfe_u32":
                m.vset(a[0], (m.vsrc(a[1]) >> (m.vsrc(a[2]) & 31)) & ((1 << (m.vsrc(a[3]) & 31)) - 1))
                m.count_valu += 1
            elif op == "v_cmp_gt_i32_e32":
                x, y = m.vsrc(a[1]), m.vsrc(a[2])
                x = x - (1 << 32) if x >> 31 else x
                y = y - (1 << 32) if y >> 31 else y
                m.carry_out(a[0], 1 if x > y else 0)
                m.count_valu += 1
            elif op in ("v_addc_co_u32_e64", "v_addc_co_u32_e32"):
                r = m.vsrc(a[2]) + m.vsrc(a[3]) + m.carry_in(a[4])
                m.vset(a[0], r)
                m.carry_out(a[1], r >> 32)
                m.count_valu += 1
            elif op in ("v_add_co_u32_e32", "v_add_co_u32_e64"):
                r = m.vsrc(a[2]) + m.vsrc(a[3])
                m.vset(a[0], r)
                m.carry_out(a[1], r >> 32)
                m.count_valu += 1
            elif op in ("v_sub_co_u32_e32", "v_sub_co_u32_e64"):
                r = m.vsrc(a[2]) - m.vsrc(a[3])
                m.vset(a[0], r)
                m.carry_out(a[1], 1 if r < 0 else 0)
                m.count_valu += 1
            elif op in ("v_subb_co_u32_e32", "v_subb_co_u32_e64"):
                r = m.vsrc(a[2]) - m.vsrc(a[3]) - m.carry_in(a[4])
                m.vset(a[0], r)
                m.carry_out(a[1], 1 if r < 0 else 0)
                m.count_valu += 1
            elif op == "v_mov_b32_e32":
                m.vset(a[0], m.vsrc(a[1]))
                m.count_valu += 1
            elif op == "v_readfirstlane_b32":       # one simulated lane: the value must be wave-uniform (the lane-cooperative kernel's round kind)
                m.sset(a[0], m.vsrc(a[1]))
                m.valu_w[a[0]] = m.count
                m.count_valu += 1
            elif op == "v_mul_hi_i32":
                x, y = m.vsrc(a[1]), m.vsrc(a[2])
                x = x - (1 << 32) if x >> 31 else x
                y = y - (1 << 32) if y >> 31 else y
                m.vset(a[0], ((x * y) >> 32) & M32)
                m.count_valu += 1
            elif op == "v_mul_lo_u32":
                m.vset(a[0], m.vsrc(a[1]) * m.vsrc(a[2]))
                m.count_valu += 1
            elif op == "v_mul_u32_u24_e32":
                m.vset(a[0], (m.vsrc(a[1]) & 0xFFFFFF) * (m.vsrc(a[2]) & 0xFFFFFF))
                m.count_valu += 1
            elif op == "v_cndmask_b32_e32" or op == "v_cndmask_b32_e64":
                c = m.carry_in(a[3])
                m.vset(a[0], m.vsrc(a[2]) if c else m.vsrc(a[1]))
                m.count_valu += 1
            elif op == "v_xor_b32_e32":
                m.vset(a[0], m.vsrc(a[1]) ^ m.vsrc(a[2]))
                m.count_valu += 1
            elif op == "v_max_u32_e32":
                m.vset(a[0], max(m.vsrc(a[1]), m.vsrc(a[2])))
                m.count_valu += 1
            elif op == "v_add3_u32":
                m.vset(a[0], m.vsrc(a[1]) + m.vsrc(a[2]) + m.vsrc(a[3]))
                m.count_valu += 1
            elif op == "v_mad_u32_u24":
                m.vset(a[0], (m.vsrc(a[1]) & 0xFFFFFF) * (m.vsrc(a[2]) & 0xFFFFFF) + m.vsrc(a[3]))
                m.count_valu += 1
            elif op == "v_mad_u32_u16":            # D = S0.u16 * S1.u16 + S2.u32; op_sel:[h0,h1,0,0] takes the HIGH halves of S0 / S1
                last, _, mod = a[3].partition(" ")
                sel = [int(x) for x in re.search(r"op_sel:\[([0-9,]+)\]", mod).group(1).split(",")] if "op_sel" in mod else [0, 0, 0, 0]
                if sel[2] or sel[3]:
                    raise SimError("v_mad_u32_u16: op_sel on the 32-bit operands")
                x, y = m.vsrc(a[1]), m.vsrc(a[2])
                x = (x >> 16) & 0xFFFF if sel[0] else x & 0xFFFF
                y = (y >> 16) & 0xFFFF if sel[1] else y & 0xFFFF
                m.vset(a[0], x * y + m.vsrc(last))
                m.count_valu += 1
            elif op == "v_lshrrev_b64":
                r = m.vsrc64(a[2]) >> (m.vsrc(a[1]) & 63)
                lo = _vpair(a[0])
                if lo % 2:
                    raise SimError("odd-aligned 64-bit VGPR dest")
                v[lo] = r & M32
                v[lo + 1] = (r >> 32) & M32
                m.count_valu += 1
            elif op == "v_and_b32_e32":
                m.vset(a[0], m.vsrc(a[1]) & m.vsrc(a[2]))
                m.count_valu += 1
            elif op == "v_or_b32_e32":
                m.vset(a[0], m.vsrc(a[1]) | m.vsrc(a[2]))
                m.count_valu += 1
            elif op == "v_or3_b32":
                m.vset(a[0], m.vsrc(a[1]) | m.vsrc(a[2]) | m.vsrc(a[3]))
                m.count_valu += 1
            elif op == "v_alignbit_b32":
                x = ((m.vsrc(a[1]) << 32) | m.vsrc(a[2])) >> (m.vsrc(a[3]) & 31)
                m.vset(a[0], x)
                m.count_valu += 1
            elif op == "v_lshlrev_b32_e32":
                m.vset(a[0], m.vsrc(a[2]) << (m.vsrc(a[1]) & 31))
                m.count_valu += 1
            elif op == "v_lshrrev_b32_e32":
                m.vset(a[0], m.vsrc(a[2]) >> (m.vsrc(a[1]) & 31))
                m.count_valu += 1
            elif op == "v_add_u32_e32":
                m.vset(a[0], m.vsrc(a[1]) + m.vsrc(a[2]))
                m.count_valu += 1
            elif op == "v_min_u32_e32":
                m.vset(a[0], min(m.vsrc(a[1]), m.vsrc(a[2])))
                m.count_valu += 1
            elif op == "v_cmp_eq_u32_e32":
                m.carry_out(a[0], 1 if m.vsrc(a[1]) == m.vsrc(a[2]) else 0)
                m.count_valu += 1
            elif op == "v_cmp_ne_u32_e32":
                m.carry_out(a[0], 1 if m.vsrc(a[1]) != m.vsrc(a[2]) else 0)
                m.count_valu += 1
            elif op == "v_cmp_lt_u32_e32":
                m.carry_out(a[0], 1 if m.vsrc(a[1]) < m.vsrc(a[2]) else 0)
                m.count_valu += 1
            elif op == "v_cmp_gt_u32_e32":
                m.carry_out(a[0], 1 if m.vsrc(a[1]) > m.vsrc(a[2]) else 0)
                m.count_valu += 1
            elif op == "v_accvgpr_write_b32":
                m.a[int(a[0][1:])] = m.vsrc(a[1])
                m.count_valu += 1
            elif op == "v_accvgpr_read_b32":
                x = m.a[int(a[1][1:])]
                if x is None:
                    if m.check_uninit:
                        raise SimError("read of uninitialised " + a[1])
                    x = 0
                m.vset(a[0], x)
                m.count_valu += 1
            # ------------------------------------------------------------ SALU
            elif op == "s_nop":
                m.count += int(a[0], 0)
                m.count_nop += 1
            elif op == "s_mov_b32":
                m.sset(a[0], m.vsrc(a[1], valu=False))
            elif op == "s_mov_b64":
                m.sset(a[0], m.sget(a[1]) if (a[1][0] in "sve") else int(a[1], 0))
            elif op == "s_add_u32":
                r = m.vsrc(a[1], False) + m.vsrc(a[2], False)
                m.sset(a[0], r)
                m.scc = r >> 32
            elif op == "s_addc_u32":
                r = m.vsrc(a[1], False) + m.vsrc(a[2], False) + m.scc
                m.sset(a[0], r)
                m.scc = r >> 32
            elif op == "s_sub_u32":
                r = m.vsrc(a[1], False) - m.vsrc(a[2], False)
                m.sset(a[0], r)
                m.scc = 1 if r < 0 else 0
            elif op == "s_subb_u32":
                r = m.vsrc(a[1], False) - m.vsrc(a[2], False) - m.scc
                m.sset(a[0], r)
                m.scc = 1 if r < 0 else 0
            elif op in ("s_add_i32", "s_sub_i32"):                       # SCC = signed overflow
                x, y = m.vsrc(a[1], False) & M32, m.vsrc(a[2], False) & M32
                x = x - (1 << 32) if x >> 31 else x
                y = y - (1 << 32) if y >> 31 else y
                r = x + y if op == "s_add_i32" else x - y
                m.sset(a[0], r & M32)
                m.scc = 0 if -(1 << 31) <= r < (1 << 31) else 1
            elif op == "s_mul_hi_u32":
                m.sset(a[0], ((m.vsrc(a[1], False) & M32) * (m.vsrc(a[2], False) & M32)) >> 32)
            elif op == "s_mul_i32":
                m.sset(a[0], m.vsrc(a[1], False) * m.vsrc(a[2], False))
            elif op == "s_lshl_b32":
                r = (m.vsrc(a[1], False) << (m.vsrc(a[2], False) & 31)) & M32
                m.sset(a[0], r)
                m.scc = 1 if r else 0
            elif op == "s_lshr_b32":
                r = m.vsrc(a[1], False) >> (m.vsrc(a[2], False) & 31)
                m.sset(a[0], r)
                m.scc = 1 if r else 0
            elif op == "s_ashr_i32":
                x = m.vsrc(a[1], False) & M32
                x = x - (1 << 32) if x >> 31 else x
                r = (x >> (m.vsrc(a[2], False) & 31)) & M32
                m.sset(a[0], r)
                m.scc = 1 if r else 0
            elif op == "s_and_b32":
                r = m.vsrc(a[1], False) & m.vsrc(a[2], False)
                m.sset(a[0], r)
                m.scc = 1 if r else 0
            elif op == "s_or_b32":
                r = m.vsrc(a[1], False) | m.vsrc(a[2], False)
                m.sset(a[0], r)
                m.scc = 1 if r else 0
            elif op == "s_cselect_b32":
                m.sset(a[0], m.vsrc(a[1], False) if m.scc else m.vsrc(a[2], False))
            elif op == "s_load_dword":
                addr = m.sget(a[1]) + m.vsrc(a[2], False)
                if addr % 4 or addr not in m.gmem:
                    raise SimError(f"s_load_dword from unmapped / unaligned address {addr:#x}")
                m.sset(a[0], m.gmem[addr])
            elif op == "s_sext_i32_i8":
                x = m.vsrc(a[1], False) & 0xFF
                m.sset(a[0], (x - 256 if x & 0x80 else x) & M32)
            elif op in ("s_cmp_eq_i32", "s_cmp_gt_i32", "s_cmp_lt_i32"):
                x, y = m.vsrc(a[0], False) & M32, m.vsrc(a[1], False) & M32
                x = x - (1 << 32) if x >> 31 else x
                y = y - (1 << 32) if y >> 31 else y
                m.scc = 1 if (x == y if op == "s_cmp_eq_i32" else (x > y if op == "s_cmp_gt_i32" else x < y)) else 0
            elif op == "s_cmp_eq_u32":
                m.scc = 1 if m.vsrc(a[0], False) == m.vsrc(a[1], False) else 0
            elif op == "s_cmp_lg_u32":
                m.scc = 1 if m.vsrc(a[0], False) != m.vsrc(a[1], False) else 0
            elif op == "s_cmp_lt_u32":
                m.scc = 1 if m.vsrc(a[0], False) < m.vsrc(a[1], False) else 0
            elif op == "s_cmp_ge_u32":
                m.scc = 1 if m.vsrc(a[0], False) >= m.vsrc(a[1], False) else 0
            elif op == "s_cmp_gt_u32":
                m.scc = 1 if m.vsrc(a[0], False) > m.vsrc(a[1], False) else 0
            elif op == "s_cmp_le_u32":
                m.scc = 1 if m.vsrc(a[0], False) <= m.vsrc(a[1], False) else 0
            elif op == "s_bitcmp1_b64":
                m.scc = (m.sget(a[0]) >> (m.vsrc(a[1], False) & 63)) & 1
            elif op == "s_bitcmp1_b32":
                m.scc = (m.vsrc(a[0], False) >> (m.vsrc(a[1], False) & 31)) & 1
            elif op == "s_bitcmp0_b32":
                m.scc = 1 - ((m.vsrc(a[0], False) >> (m.vsrc(a[1], False) & 31)) & 1)
            elif op == "s_cbranch_scc1":
                if m.scc:
                    pc = labels[a[0]]
            elif op == "s_cbranch_scc0":
                if not m.scc:
                    pc = labels[a[0]]
            elif op == "s_branch":
                pc = labels[a[0]]
            elif op == "s_call_b64":
                m.sset(a[0], pc)          # "return address" = next instruction index
                tgt = a[1]
                while prog[labels[tgt]][0] == "s_branch" and tgt.startswith("L_hop"):      # trampolines of out-of-range calls
                    tgt = prog[labels[tgt]][1][0]
                if m.call_log is not None and tgt.startswith("L2_"):
                    m.call_log.append(tgt.replace("_%=", ""))
                pc = labels[a[1]]
            elif op == "s_setpc_b64":
                pc = m.sget(a[0])
            elif op == "s_waitcnt":
                pass
            elif op == "s_and_saveexec_b64":
                old = m.exec
                m.sset(a[0], old)
                m.exec = old & (m.sget(a[1]) & 1)
            elif op == "s_endpgm":
                m.count += 1
                return
            # ------------------------------------------------------------ memory
            elif op in ("ds_read_b32", "ds_write_b32"):
                off = 0
                for t in a:
                    for part in t.split():
                        if part.startswith("offset:"):
                            off = int(part[7:], 0)
                toks = [t.split()[0] for t in a]
                if op == "ds_read_b32":
                    x = m.lds.get(m.vsrc(toks[1]) + off)
                    if x is None:
                        raise SimError(f"LDS read of unwritten address {m.vsrc(toks[1]) + off}: {text}")
                    m.vset(toks[0], x)
                else:
                    base = m.vsrc(toks[0]) + off
                    if base % 4:
                        raise SimError("misaligned ds_write")
                    x = m.vsrc(toks[1])
                    if m.exec:
                        m.lds[base] = x
            elif op in ("ds_read_b128", "ds_write_b128", "ds_read_b64", "ds_write_b64"):
                ndw = 4 if op.endswith("b128") else 2
                off = 0
                regs = None
                addr = None
                for t in a:
                    for part in t.split():
                        if part.startswith("offset:"):
                            off = int(part[7:], 0)
                toks = [t.split()[0] for t in a]
                if off < 0 or off > 65535:
                    raise SimError("DS offset out of range: " + text)
                if op.startswith("ds_read"):
                    lo = int(re.match(r"v\[(\d+):", toks[0]).group(1))
                    if lo % 2:
                        raise SimError("odd-aligned DS destination tuple")
                    base = m.vsrc(toks[1])
                    for k in range(ndw):
                        x = m.lds.get(base + off + 4 * k)
                        if x is None:
                            raise SimError(f"LDS read of unwritten address {base + off + 4 * k}: {text}")
                        v[lo + k] = x
                else:
                    base = m.vsrc(toks[0])
                    lo = int(re.match(r"v\[(\d+):", toks[1]).group(1))
                    if (base + off) % (4 * ndw):
                        raise SimError("misaligned ds_write")
                    for k in range(ndw):
                        if v[lo + k] is None:
                            raise SimError("store of uninitialised register: " + text)
                        if m.exec:
                            m.lds[base + off + 4 * k] = v[lo + k]
                            sv = v[lo + k] - (1 << 32) if v[lo + k] >> 31 else v[lo + k]
                            m.max_stored = max(m.max_stored, abs(sv))
            elif op.startswith("global_load_dword") or op.startswith("global_store_dword"):
                n = {"": 1, "x2": 2, "x4": 4}[op.split("dword")[1]]
                off = 0
                toks = []
                for t in a:
                    parts = t.split()
                    toks.append(parts[0])
                    for part in parts[1:]:
                        if part.startswith("offset:"):
                            off = int(part[7:], 0)
                if off < -4096 or off > 4095:
                    raise SimError("global offset out of range: " + text)
                if op.startswith("global_load"):
                    dst, voff, sbase = toks[0], toks[1], toks[2]
                    addr = m.sget(sbase) + m.vsrc(voff) + off
                    mm = re.match(r"([va])\[?(\d+)", dst)               # gfx90a+: loads can target AGPRs directly
                    lo, bank = int(mm.group(2)), (v if mm.group(1) == "v" else m.a)
                    for k in range(n if m.exec else 0):               # a lane that EXEC masks off neither reads nor faults
                        x = m.gmem.get(addr + 4 * k)
                        if x is None:
                            raise SimError(f"global read of unwritten address {hex(addr + 4 * k)}: {text}")
                        bank[lo + k] = x
                else:
                    voff, src, sbase = toks[0], toks[1], toks[2]
                    addr = m.sget(sbase) + m.vsrc(voff) + off
                    lo = int(re.match(r"v\[?(\d+)", src).group(1))
                    for k in range(n):
                        if v[lo + k] is None:
                            raise SimError("store of uninitialised register: " + text)
                        if m.exec:
                            m.gmem[addr + 4 * k] = v[lo + k]
            else:
                raise SimError("unknown instruction: " + text)
        except SimError as ex:
            raise SimError(f"{ex} @ `{text}`") from None
        m.count += 1


def run_block(lines, m):
    """Run a straight-line routine body (no s_endpgm): appends one."""
    run(list(lines) + ["s_endpgm"], m)
