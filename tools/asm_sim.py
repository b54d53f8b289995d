#!/usr/bin/env python3
"""Single-lane interpreter for the instruction subset tools/gen_fq_asm.py emits.
Used by tests/test_asm_gen.py to check the generated Montgomery routines against
big-int arithmetic without a GPU (logic only -- hardware hazards are the generator's
post-pass' business, which this simulator also re-checks)."""
import re

M32 = 0xFFFFFFFF


class Sim:
    def __init__(self, operands):
        self.v = [0] * 256
        self.s = {}            # operand-name -> value (SGPR constants / carry pairs as 0/1)
        self.s.update(operands)
        self.vcc = 0
        self.last_valu_write = {}   # carry reg -> instruction index
        self.idx = 0

    def src(self, tok):
        tok = tok.strip()
        if tok.startswith("v["):
            m = re.match(r"v\[(\d+):(\d+)\]", tok)
            lo = int(m.group(1))
            return self.v[lo] | (self.v[lo + 1] << 32)
        if tok.startswith("v"):
            return self.v[int(tok[1:])]
        if tok == "vcc":
            return self.vcc
        if tok.startswith("%"):
            return self.s[tok]
        if tok == "-1":
            return M32
        return int(tok, 0)

    def check_read(self, reg):
        if reg in self.last_valu_write:
            gap = self.idx - self.last_valu_write[reg] - 1
            assert gap >= 2, f"hazard: {reg} read {gap} wait states after a VALU write (instr #{self.idx})"

    def setc(self, reg, val):
        if reg == "vcc":
            self.vcc = val
        else:
            self.s[reg] = val
        self.last_valu_write[reg] = self.idx

    def getc(self, reg):
        self.check_read(reg)
        return self.vcc if reg == "vcc" else self.s[reg]

    def run(self, lines):
        for line in lines:
            line = line.strip()
            if not line:
                continue
            op, rest = line.split(None, 1) if " " in line else (line, "")
            args = [a.strip() for a in re.split(r",\s*(?![^\[]*\])", rest)] if rest else []
            if op == "s_nop":
                self.idx += int(args[0], 0) + 1
                continue
            if op == "v_mad_u64_u32":
                d, c, a, b, acc = args
                r = self.src(a) * self.src(b) + self.src(acc)
                lo = int(re.match(r"v\[(\d+):", d).group(1))
                assert lo % 2 == 0, "64-bit VGPR operand must be even-aligned"
                self.v[lo] = r & M32
                self.v[lo + 1] = (r >> 32) & M32
                self.setc(c, r >> 64)
            elif op in ("v_addc_co_u32_e64", "v_addc_co_u32_e32"):
                d, co, a, b, ci = args
                r = self.src(a) + self.src(b) + self.getc(ci)
                self.v[int(d[1:])] = r & M32
                self.setc(co, r >> 32)
            elif op == "v_add_co_u32_e32":
                d, co, a, b = args
                r = self.src(a) + self.src(b)
                self.v[int(d[1:])] = r & M32
                self.setc(co, r >> 32)
            elif op == "v_sub_co_u32_e32":
                d, co, a, b = args
                r = self.src(a) - self.src(b)
                self.v[int(d[1:])] = r & M32
                self.setc(co, 1 if r < 0 else 0)
            elif op == "v_subb_co_u32_e32":
                d, co, a, b, ci = args
                r = self.src(a) - self.src(b) - self.getc(ci)
                self.v[int(d[1:])] = r & M32
                self.setc(co, 1 if r < 0 else 0)
            elif op == "v_mov_b32_e32":
                d, a = args
                self.v[int(d[1:])] = self.src(a) & M32
            elif op == "v_mul_lo_u32":
                d, a, b = args
                self.v[int(d[1:])] = (self.src(a) * self.src(b)) & M32
            elif op == "v_cndmask_b32_e32":
                d, a, b, c = args
                self.v[int(d[1:])] = self.src(b) if self.getc(c) else self.src(a)
            elif op == "v_cndmask_b32_e64":
                d, a, b, c = args
                self.v[int(d[1:])] = (self.src(b) if self.getc(c) else self.src(a)) & M32
            elif op == "v_and_b32_e32":
                d, a, b = args
                self.v[int(d[1:])] = self.src(a) & self.src(b)
            else:
                raise NotImplementedError(line)
            self.idx += 1
