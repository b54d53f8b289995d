#!/usr/bin/env python3
"""cvm_kernel.py -- the interpreter kernel of the lane-cooperative pairing (tools/cvm.py has the design and the program).

One wave per workgroup; NR = 16 lanes per item and four items per wave, or -- the builds for the smallest launches -- NR = 32 and two,
NR = 64 and one (the products of three and four pairings).  Every lane executes the same instruction stream;
what differs per lane is its table row (which LDS slots it reads and writes).  Per round:

    wait for the row / kind prefetched during the previous round, start the loads of the next ones
    branch on the round's kind (scalar):  m2 / m4 / m6: one Montgomery column pass over 2 / 4 / 6 limb-vector products (+ addend)
                                          l4 / l8:      one reducing 64-bit chain over 4 / 8 (coefficient, source) pairs
                                          inv:          the Fermat chain (sliding window, as KernelBuilder._fq_inv)
    write the result (and its negation, the `twin`) to the LDS slots the row names
(the kind of a round travels in its row; rows are fetched three rounds ahead)

LDS: per group n_slots x 48 bytes (nine limbs + a pad dword, 16-byte aligned for the 128-bit accesses).  A wave's LDS operations
execute in order, and the four groups never touch each other's slots, so no barrier is needed; the schedule guarantees that no slot
is written in the round that reads it last (tools/cvm.py: Program._allocate).

asm operands: %0 g1  %1 g2  %2 f_in  %3 out  %4 n  %5 k (pairs per item)  %6 program blob  %8 status  %9 threadIdx.x  %10 blockIdx.x
(the throughput kernels' list; %7, %11 unused).  Which inputs a program reads (G1 / G2 points of its k pairs, an Fq12) is in its blob.  Field arithmetic: tools/kgen4.py L1v4 on an explicit register plan (below).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from asmcore import Emitter, Pool, P_INT, align_code  # noqa: E402
import kgen4 as K4  # noqa: E402
from kgen4 import NL, L1v4, hx, bal_limbs, S_P, S_N0, S_REDN, S_M30, P_L, N0P, REDN_C, S_RET1  # noqa: E402
import cvm  # noqa: E402

NR = 16
GROUPS = 4
SLOT_BYTES = 48
ROW_DW = 8
HDR_DW = 24            # n_rounds, n_const, inputs_off, rows_off, consts_off, n_slots, trash, input chunks, 16 output slots (bytes offsets from the blob)
CONST_DW = 10


def OP(i):
    """operand block i: v[10 i : 10 i + 9] (nine limbs + the pad dword of the 8-byte tail access)"""
    return list(range(10 * i, 10 * i + NL))


OUT0, NEG0 = 130, 140
POOL_FIRST, POOL_LAST = 150, 203
# The row of the NEXT round is fetched into these registers as soon as the current one has been decoded.  (Rows fetched three rounds
# ahead through rotating buffers were measured SLOWER -- 1.336 ms against 1.277 ms for one pairing at that stage: the rows come from L2
# within a round, the copies cost more than the wait; what a round waits for is the LDS, the operand fetch.)
ROWN = 228
INV_SAFEGCD = bool(int(os.environ.get("CVM_INV_SAFEGCD", "1")))  # the inversion by divsteps instead of the Fermat chain
OVERLAP = bool(int(os.environ.get("CVM_OVERLAP", "1")))      # operand limbs that a pass needs late are fetched inside it
ADDR16 = bool(int(os.environ.get("CVM_ADDR16", "1")))        # slot addresses by v_mad_u32_u16 with op_sel straight from the packed row (one instruction instead of two / three)
V_LBASE, V_ROWOFF, V_ROLE, V_ITEM8, V_FLAG, V_T0, V_DST, V_TWIN, V_VALID, V_T1 = 236, 237, 238, 239, 240, 241, 242, 243, 244, 245
V_ELEM, V_DESC = 246, 247
S_BLOB, S_ROWS, S_NROUNDS, S_KIND, S_NCONST, S_NSLOTS, S_TMP = "s[48:49]", "s[50:51]", 52, 53, 65, 66, 67
S_CONSTS, S_G1, S_G2, S_OUT, S_N, S_NSTRIDE, S_STATUS = "s[68:69]", "s[70:71]", "s[72:73]", "s[74:75]", 76, 77, "s[78:79]"
S_FIN, S_INS, S_K, S_PITCH_IN, S_NCHUNK = "s[90:91]", "s[92:93]", 94, 95, 96
S_ITEM0, S_GSTEP, S_NCHUNK0 = 97, 98, 99      # first item of the wave's current pass, items per pass of the whole grid, input chunks per item
V_GRP = 208
S_H = 80               # s80..s87: header words / scratch
S_CNT = 88
S_SAVE = "s[60:61]"


# LDS layouts.  "aos48": a slot is 48 contiguous bytes (nine limbs + a pad dword; two 128-bit and one 64-bit access).  "split36": limbs
# 0..7 in a 32-byte slot, limb 8 in a separate array of dwords (two 128-bit and one 32-bit access, one more address computation
# per operand): 36 bytes per slot -- the pairing program's 242 slots are 35 KB per wave instead of 46, so FOUR waves fit a CU's
# 160 KB instead of three.  The host takes aos48 while a launch is at most three waves per CU and split36 beyond.
LAYOUTS = {"aos48": 48, "split36": 36}
V_LTOP, V_DST_T, V_TWIN_T, V_T1_T = 204, 205, 206, 207          # split36: base of the group's limb-8 array; companions of the fixed address registers


class VMKernel:
    def __init__(self, layout="aos48", nr=NR):
        assert layout in LAYOUTS and nr in (16, 32, 64)
        self.nr = nr
        self.lg = nr.bit_length() - 1             # log2(lanes per item)
        self.groups = 64 // nr
        self.e = Emitter()
        self.sizes = {}
        self.layout = layout
        self.split = layout == "split36"
        self.slot_bytes = LAYOUTS[layout]

    def l1(self):
        g = L1v4(self.e)
        g.pool = Pool(POOL_FIRST, POOL_LAST)
        if g.half is not None:
            g.half = "s[62:63]"          # (this kernel's scalar register map)
        return g

    # -------------------------------------------------------------- small helpers
    # An ADDRESS is (main register, limb-8 register | None).
    def addr_of_slot(self, dst, slot_reg):
        """dst <- address of the slot whose number is in slot_reg"""
        e = self.e
        if self.split:
            e.emit(f"v_lshl_add_u32 v{dst[0]}, v{slot_reg}, 5, v{V_LBASE}", vw=[dst[0]])
            e.emit(f"v_lshl_add_u32 v{dst[1]}, v{slot_reg}, 2, v{V_LTOP}", vw=[dst[1]])
        else:
            e.emit(f"v_mad_u32_u24 v{dst[0]}, v{slot_reg}, {SLOT_BYTES}, v{V_LBASE}", vw=[dst[0]])

    def slot_addr(self, dst, row_reg, hi):
        e = self.e
        if ADDR16:
            # one instruction per address: slot number (a 16-bit half of the row dword, picked by op_sel) x slot bytes + the group's base
            h = 1 if hi else 0
            if self.split:
                e.emit(f"v_mad_u32_u16 v{dst[0]}, v{row_reg}, 32, v{V_LBASE} op_sel:[{h},0,0,0]", vw=[dst[0]])
                e.emit(f"v_mad_u32_u16 v{dst[1]}, v{row_reg}, 4, v{V_LTOP} op_sel:[{h},0,0,0]", vw=[dst[1]])
            else:
                e.emit(f"v_mad_u32_u16 v{dst[0]}, v{row_reg}, {SLOT_BYTES}, v{V_LBASE} op_sel:[{h},0,0,0]", vw=[dst[0]])
            return
        if hi:
            e.emit(f"v_lshrrev_b32_e32 v{V_T0}, 16, v{row_reg}", vw=[V_T0])
        else:
            e.emit(f"v_and_b32_e32 v{V_T0}, 0xffff, v{row_reg}", vw=[V_T0])
        self.addr_of_slot(dst, V_T0)

    def lds_load_parts(self, blk0, addr):
        """the three accesses of one operand as (text, registers written): limbs 0..3, 4..7, 8 (+ the pad dword in aos48)"""
        main, top = addr
        tail = (f"ds_read_b32 v{blk0 + 8}, v{top}", [blk0 + 8]) if self.split else (f"ds_read_b64 v[{blk0 + 8}:{blk0 + 9}], v{main} offset:32", [blk0 + 8, blk0 + 9])
        return [(f"ds_read_b128 v[{blk0}:{blk0 + 3}], v{main}", list(range(blk0, blk0 + 4))),
                (f"ds_read_b128 v[{blk0 + 4}:{blk0 + 7}], v{main} offset:16", list(range(blk0 + 4, blk0 + 8))), tail]

    def lds_load(self, blk0, addr):
        for text, vw in self.lds_load_parts(blk0, addr):
            self.e.emit(text, kind="lds", vw=vw)

    def lds_store(self, addr, blk0):
        e = self.e
        main, top = addr
        e.emit(f"ds_write_b128 v{main}, v[{blk0}:{blk0 + 3}]", kind="lds")
        e.emit(f"ds_write_b128 v{main}, v[{blk0 + 4}:{blk0 + 7}] offset:16", kind="lds")
        if self.split:
            e.emit(f"ds_write_b32 v{top}, v{blk0 + 8}", kind="lds")
        else:
            e.emit(f"ds_write_b64 v{main}, v[{blk0 + 8}:{blk0 + 9}] offset:32", kind="lds")

    def fixed(self, reg):
        """the address held in one of the fixed registers V_DST / V_TWIN / V_T1"""
        return (reg, {V_DST: V_DST_T, V_TWIN: V_TWIN_T, V_T1: V_T1_T}[reg] if self.split else None)

    def decode(self, g, fields, first=None):
        """addresses (registers from the pool) of the slots named by `fields` = [(row dword, high half?)], read from the row registers
        BEFORE the next row is fetched into them.  first (optional): {field index: (operand block, part numbers)} -- those parts of the
        operand are requested as soon as its address exists, so that the LDS works while the remaining addresses are computed."""
        out = []
        for j, (dw, hi) in enumerate(fields):
            a = (g.pool.alloc(), g.pool.alloc() if self.split else None)
            self.slot_addr(a, ROWN + dw, hi)
            out.append(a)
            if first and j in first:
                blk0, which = first[j]
                pts = self.lds_load_parts(blk0, a)
                for k in which:
                    self.e.emit(pts[k][0], kind="lds", vw=pts[k][1])
        return out

    def next_row(self):
        """the next round's row into the row registers (everything that reads the current row has been issued)"""
        e = self.e
        b = ROWN
        e.emit(f"global_load_dwordx4 v[{b}:{b + 3}], v{V_ROWOFF}, {S_ROWS}", kind="vmem", vw=list(range(b, b + 4)))
        e.emit(f"global_load_dwordx4 v[{b + 4}:{b + 7}], v{V_ROWOFF}, {S_ROWS} offset:16", kind="vmem", vw=list(range(b + 4, b + 8)))
        e.emit(f"v_add_u32_e32 v{V_ROWOFF}, {self.nr * ROW_DW * 4}, v{V_ROWOFF}", vw=[V_ROWOFF])

    def finish(self, dst, twin):
        """result in OUT -> dst slot, its negation -> twin slot; next round"""
        e = self.e
        for i in range(NL):
            e.emit(f"v_sub_u32_e32 v{NEG0 + i}, 0, v{OUT0 + i}", vw=[NEG0 + i])
        self.lds_store(dst, OUT0)
        self.lds_store(twin, NEG0)
        e.salu("s_branch LC_round_%=")

    # -------------------------------------------------------------- prologue
    def prologue(self):
        e = self.e
        e.salu(f"s_mov_b64 {S_G1}, %0")
        e.salu(f"s_mov_b64 {S_G2}, %1")
        e.salu(f"s_mov_b64 {S_BLOB}, %6")
        e.salu(f"s_mov_b64 {S_FIN}, %2")
        e.salu(f"s_mov_b64 {S_OUT}, %3")
        e.salu(f"s_mov_b32 s{S_N}, %4")
        e.salu(f"s_mov_b32 s{S_K}, %5")
        e.salu(f"s_mov_b64 {S_STATUS}, %8")
        e.salu(f"s_mov_b32 s{S_TMP}, %10")
        for i, dst in enumerate((S_NROUNDS, S_NCONST, S_H, S_H + 1, S_H + 2, S_NSLOTS, S_H + 3, S_NCHUNK)):
            e.salu(f"s_load_dword s{dst}, {S_BLOB}, 0x{4 * i:x}")
        for i in range(NL):
            e.salu(f"s_mov_b32 s{S_P + i}, {hx(P_L[i])}")
        e.salu(f"s_mov_b32 s{S_N0}, 0x{N0P:x}")
        e.salu(f"s_mov_b32 s{S_REDN}, 0x{REDN_C:x}")
        e.salu(f"s_mov_b32 s{S_M30}, -30")
        e.salu(f"s_mov_b32 s62, 0x{1 << 28:x}")                      # s[62:63] = 2^28: the digit extraction's rounding constant (L1v4.half)
        e.salu("s_mov_b32 s63, 0")
        e.salu(f"s_lshl_b32 s{S_NSTRIDE}, s{S_N}, 3")               # bytes between the planes of the output (n items)
        e.salu(f"s_mul_i32 s{S_PITCH_IN}, s{S_NSTRIDE}, s{S_K}")      # ... of the inputs (n k elements: pair j of item g is element g k + j)
        e.emit(f"v_and_b32_e32 v{V_ROLE}, {self.nr - 1}, %9", vw=[V_ROLE])
        e.emit(f"v_lshrrev_b32_e32 v{V_GRP}, {self.lg}, %9", vw=[V_GRP])                   # group of the lane
        e.emit(f"v_mov_b32_e32 v{V_T0}, v{V_GRP}", vw=[V_T0])
        e.salu(f"s_lshl_b32 s{S_ITEM0}, s{S_TMP}, {6 - self.lg}")                          # the wave walks items block * G + group, + grid * G, ...
        e.salu(f"s_lshl_b32 s{S_GSTEP}, %11, {6 - self.lg}")
        e.raw("s_waitcnt lgkmcnt(0)")
        if self.split:
            e.salu(f"s_lshl_b32 s{S_TMP}, s{S_NSLOTS}, 5")                                  # a group's 32-byte slots
            e.emit(f"v_mul_lo_u32 v{V_LBASE}, v{V_T0}, s{S_TMP}", vw=[V_LBASE])
            e.salu(f"s_lshl_b32 s{S_TMP}, s{S_NSLOTS}, 2")                                  # its limb-8 dwords, behind all the groups' slots
            e.emit(f"v_mul_lo_u32 v{V_LTOP}, v{V_T0}, s{S_TMP}", vw=[V_LTOP])
            e.salu(f"s_lshl_b32 s{S_TMP}, s{S_NSLOTS}, {5 + 6 - self.lg}")
            e.emit(f"v_add_u32_e32 v{V_LTOP}, s{S_TMP}, v{V_LTOP}", vw=[V_LTOP])
        else:
            e.salu(f"s_mul_i32 s{S_TMP}, s{S_NSLOTS}, {SLOT_BYTES}")
            e.emit(f"v_mul_lo_u32 v{V_LBASE}, v{V_T0}, s{S_TMP}", vw=[V_LBASE])
        e.salu(f"s_mov_b32 s{S_NCHUNK0}, s{S_NCHUNK}")
        for r in (OUT0 + 9, NEG0 + 9):
            e.emit(f"v_mov_b32_e32 v{r}, 0", vw=[r])
        e.salu(f"s_add_u32 s50, s48, s{S_H + 1}")            # rows
        e.salu("s_addc_u32 s51, s49, 0")
        e.salu(f"s_add_u32 s68, s48, s{S_H + 2}")            # constants
        e.salu("s_addc_u32 s69, s49, 0")
        e.salu(f"s_add_u32 s92, s48, s{S_H}")                # input descriptors
        e.salu("s_addc_u32 s93, s49, 0")
        # ---- constant pool -> the group's slots [0, n_const): lane r copies constants r, r + 16, ... (clamped: the last one again)
        e.emit(f"v_mov_b32_e32 v{V_T1}, v{V_ROLE}", vw=[V_T1])
        e.salu(f"s_add_u32 s{S_CNT}, s{S_NCONST}, {self.nr - 1}")
        e.salu(f"s_lshr_b32 s{S_CNT}, s{S_CNT}, {self.lg}")
        e.salu(f"s_sub_u32 s{S_TMP}, s{S_NCONST}, 1")
        e.label("LC_const_%=")
        e.emit(f"v_min_u32_e32 v{V_T0}, s{S_TMP}, v{V_T1}", vw=[V_T0])
        e.emit(f"v_mul_u32_u24_e32 v{V_DST}, {4 * CONST_DW}, v{V_T0}", vw=[V_DST])
        e.emit(f"global_load_dwordx4 v[0:3], v{V_DST}, {S_CONSTS}", kind="vmem", vw=[0, 1, 2, 3])
        e.emit(f"global_load_dwordx4 v[4:7], v{V_DST}, {S_CONSTS} offset:16", kind="vmem", vw=[4, 5, 6, 7])
        e.emit(f"global_load_dwordx2 v[8:9], v{V_DST}, {S_CONSTS} offset:32", kind="vmem", vw=[8, 9])
        self.addr_of_slot(self.fixed(V_TWIN), V_T0)
        e.raw("s_waitcnt vmcnt(0)")
        self.lds_store(self.fixed(V_TWIN), 0)
        e.emit(f"v_add_u32_e32 v{V_T1}, {self.nr}, v{V_T1}", vw=[V_T1])
        e.salu(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1")
        e.salu(f"s_cmp_lg_u32 s{S_CNT}, 0")
        e.salu("s_cbranch_scc1 LC_const_%=")
        # ---- one item per group and pass (the grid is sized to what is resident: a wave walks its items)
        e.label("LC_item_%=")
        e.emit(f"v_add_u32_e32 v{V_ITEM8}, s{S_ITEM0}, v{V_GRP}", vw=[V_ITEM8])
        e.emit(f"v_cmp_gt_u32_e32 vcc, s{S_N}, v{V_ITEM8}", w=["vcc"])
        e.emit(f"v_cndmask_b32_e64 v{V_VALID}, 0, 1, vcc", r=["vcc"], vw=[V_VALID])
        e.salu(f"s_sub_u32 s{S_TMP}, s{S_N}, 1")
        e.emit(f"v_min_u32_e32 v{V_ITEM8}, s{S_TMP}, v{V_ITEM8}", vw=[V_ITEM8])             # lanes past the end redo the last item (they never store)
        e.emit(f"v_mul_lo_u32 v{V_ELEM}, v{V_ITEM8}, s{S_K}", vw=[V_ELEM])                  # first input element of the item
        e.emit(f"v_lshlrev_b32_e32 v{V_ITEM8}, 3, v{V_ITEM8}", vw=[V_ITEM8])
        e.emit(f"v_lshlrev_b32_e32 v{V_ROWOFF}, 5, v{V_ROLE}", vw=[V_ROWOFF])
        e.emit(f"v_mov_b32_e32 v{V_FLAG}, 0", vw=[V_FLAG])
        e.salu(f"s_mov_b32 s{S_NCHUNK}, s{S_NCHUNK0}")
        # ---- inputs: sixteen descriptors per chunk, one per lane: slot | Fq number << 16 | array << 20 | pair << 22.  The lane reads the
        # four words of that Fq from its array (the section of another array is skipped under EXEC: its pointer may be null), converts,
        # stores to the slot (padding descriptors: array 3, the trash slot).
        e.emit(f"v_lshlrev_b32_e32 v{V_T1}, 2, v{V_ROLE}", vw=[V_T1])
        e.label("LC_in_%=")
        e.emit(f"global_load_dword v{V_DESC}, v{V_T1}, {S_INS}", kind="vmem", vw=[V_DESC])
        for i in range(8):
            e.emit(f"v_mov_b32_e32 v{i}, 0", vw=[i])
        e.raw("s_waitcnt vmcnt(0)")
        e.emit(f"v_bfe_u32 v{V_T0}, v{V_DESC}, 16, 4", vw=[V_T0])                           # Fq number
        e.emit(f"v_lshlrev_b32_e32 v{V_T0}, 2, v{V_T0}", vw=[V_T0])
        e.emit(f"v_mul_lo_u32 v{V_T0}, v{V_T0}, s{S_PITCH_IN}", vw=[V_T0])
        e.emit(f"v_lshrrev_b32_e32 v{V_DST}, 22, v{V_DESC}", vw=[V_DST])                    # pair
        e.emit(f"v_add_u32_e32 v{V_DST}, v{V_DST}, v{V_ELEM}", vw=[V_DST])
        e.emit(f"v_lshl_add_u32 v{V_T0}, v{V_DST}, 3, v{V_T0}", vw=[V_T0])                  # byte offset of word 0
        e.emit(f"v_bfe_u32 v{V_TWIN}, v{V_DESC}, 20, 2", vw=[V_TWIN])                       # array
        for arr, base in ((cvm.ARR_G1, S_G1), (cvm.ARR_G2, S_G2), (cvm.ARR_F, S_FIN)):
            e.emit(f"v_cmp_eq_u32_e32 vcc, {arr}, v{V_TWIN}", w=["vcc"])
            e.raw("s_nop 1")
            e.salu(f"s_and_saveexec_b64 {S_SAVE}, vcc")
            e.emit(f"v_mov_b32_e32 v{V_DST}, v{V_T0}", vw=[V_DST])
            for l in range(4):
                e.emit(f"global_load_dwordx2 v[{2 * l}:{2 * l + 1}], v{V_DST}, {base}", kind="vmem", vw=[2 * l, 2 * l + 1])
                if l < 3:
                    e.emit(f"v_add_u32_e32 v{V_DST}, s{S_PITCH_IN}, v{V_DST}", vw=[V_DST])
            e.salu(f"s_mov_b64 exec, {S_SAVE}")
        e.raw("s_waitcnt vmcnt(0)")
        self.l1().r_cvtin()
        e.emit("v_mov_b32_e32 v9, 0", vw=[9])
        e.emit(f"v_and_b32_e32 v{V_T0}, 0xffff, v{V_DESC}", vw=[V_T0])
        self.addr_of_slot(self.fixed(V_DST), V_T0)
        self.lds_store(self.fixed(V_DST), 0)
        e.emit(f"v_add_u32_e32 v{V_T1}, {4 * self.nr}, v{V_T1}", vw=[V_T1])
        e.salu(f"s_sub_u32 s{S_NCHUNK}, s{S_NCHUNK}, 1")
        e.salu(f"s_cmp_lg_u32 s{S_NCHUNK}, 0")
        e.salu("s_cbranch_scc1 LC_in_%=")
        # ---- the first rows (the round's kind travels in the row: dword 7, high half)
        self.next_row()

    # -------------------------------------------------------------- the round loop
    def round_loop(self):
        e = self.e
        e.label("LC_round_%=")
        e.raw("s_waitcnt vmcnt(0)")
        e.emit(f"v_readfirstlane_b32 s{S_KIND}, v{ROWN + 7}", kind="valu")
        e.salu(f"s_lshr_b32 s{S_KIND}, s{S_KIND}, 16")
        # by frequency in the pairing program (the sixty-four-lane programs: two-product rounds and four-term combinations, tools/cvm.py Graph.full)
        for kind in ((cvm.K_L4, cvm.K_M2, cvm.K_M4, cvm.K_L8, cvm.K_M6, cvm.K_INV) if self.nr == 64 else (cvm.K_L4, cvm.K_M6, cvm.K_M2, cvm.K_M4, cvm.K_L8, cvm.K_INV)):
            e.salu(f"s_cmp_eq_u32 s{S_KIND}, {kind}")
            e.salu(f"s_cbranch_scc1 LC_k{kind}_%=")
        e.salu("s_branch LC_end_%=")

    def mul_handler(self, kind, nprod):
        e = self.e
        n0 = len(e.ins)
        e.label(f"LC_k{kind}_%=")
        g = self.l1()
        fields = [(i, h) for i in range(nprod) for h in (False, True)] + [(6, False), (6, True), (7, False)]
        out = list(range(OUT0, OUT0 + NL))
        prods = [(OP(2 * i), OP(2 * i + 1)) for i in range(nprod)]
        if OVERLAP:
            # column k of the pass needs limbs 0..k only: the low quarters of the product operands are fetched up front (each as soon as
            # its address is known), everything else -- the upper limbs, the addend (which enters behind column 8) -- rides in the
            # multiply runs of columns 0..3
            ad = self.decode(g, fields, first={i: (10 * i, (0,)) for i in range(2 * nprod)})
            self.next_row()
            parts = [self.lds_load_parts(10 * i, ad[i]) for i in range(2 * nprod)] + [self.lds_load_parts(10 * 12, ad[2 * nprod])]
            later = [pt[1] for pt in parts[:-1]] + [parts[-1][0], parts[-1][1]] + [pt[2] for pt in parts]
            fillers = [(t, vw, 0, 3) for t, vw in later] + [("s_waitcnt lgkmcnt(0)", [], 99, 4)]
            e.raw("s_waitcnt lgkmcnt(0)")
            g.fips(prods, out, inject=[(OP(12), 1)], fillers=fillers, gap=2)
        else:
            ad = self.decode(g, fields)
            self.next_row()
            for i in range(2 * nprod):
                self.lds_load(10 * i, ad[i])
            self.lds_load(10 * 12, ad[2 * nprod])
            e.raw("s_waitcnt lgkmcnt(0)")
            g.fips(prods, out, inject=[(OP(12), 1)])
        self.finish(ad[2 * nprod + 1], ad[2 * nprod + 2])
        self.sizes[cvm.KIND_NAME[kind]] = len(e.ins) - n0

    def lin_handler(self, kind, nsrc):
        e = self.e
        n0 = len(e.ins)
        e.label(f"LC_k{kind}_%=")
        g = self.l1()
        fields = [(i // 2, i % 2 == 1) for i in range(nsrc)] + [(6, True), (7, False)]
        # the chain starts from the TOP limbs (the quotient estimate), then walks up from limb 0: every source's tail and low quarter are
        # requested as soon as its address is known, the upper quarters last (waited for before limb 4)
        ad = self.decode(g, fields, first={i: (10 * i, (2, 0)) for i in range(nsrc)} if OVERLAP else None)
        co = [g.pool.alloc() for _ in range(nsrc)]
        for i in range(nsrc):
            e.emit(f"v_bfe_i32 v{co[i]}, v{ROWN + 4 + i // 4}, {8 * (i % 4)}, 8", vw=[co[i]])
        self.next_row()
        out = list(range(OUT0, OUT0 + NL))
        terms = [[(("v", co[i]), OP(i)) for i in range(nsrc)]]
        if OVERLAP:
            for i in range(nsrc):
                pt = self.lds_load_parts(10 * i, ad[i])[1]
                e.emit(pt[0], kind="lds", vw=pt[1])
            e.raw(f"s_waitcnt lgkmcnt({nsrc})")
            g.lincomb([out], terms, reduce=True, hooks={4: ["s_waitcnt lgkmcnt(0)"]})
        else:
            for i in range(nsrc):
                self.lds_load(10 * i, ad[i])
            e.raw("s_waitcnt lgkmcnt(0)")
            g.lincomb([out], terms, reduce=True)
        self.finish(ad[nsrc], ad[nsrc + 1])
        self.sizes[cvm.KIND_NAME[kind]] = len(e.ins) - n0

    def inv_handler(self):
        """OUT <- 1 / src; the zero-divisor flag is raised when src == 0 mod p.  INV_SAFEGCD: Bernstein-Yang divsteps
        (L1v4.fq_inv_safegcd, 17.5 k instructions); otherwise the Fermat chain src^(p - 2) (KernelBuilder._fq_inv's sliding window on
        plain register blocks, 61 k instructions; zero tested on the canonical form, like the throughput kernels' inversions)."""
        e = self.e
        n0 = len(e.ins)
        e.label(f"LC_k{cvm.K_INV}_%=")
        g0 = self.l1()
        ad = self.decode(g0, [(0, False), (6, True), (7, False)])
        for i, r in enumerate((V_T1, V_DST, V_TWIN)):                    # (the chain below takes the whole pool)
            for src, dst in zip(ad[i], self.fixed(r)):
                if src is not None:
                    e.emit(f"v_mov_b32_e32 v{dst}, v{src}", vw=[dst])
        self.next_row()
        if INV_SAFEGCD:
            self.lds_load(0, self.fixed(V_T1))
            e.raw("s_waitcnt lgkmcnt(0)")
            self.l1().fq_inv_safegcd(OP(0), OP(1), OP(2), OP(3), list(range(OUT0, OUT0 + NL)), f"s{S_CNT}", "LC_sg_%=", bad=V_T0)
            e.emit(f"v_or_b32_e32 v{V_FLAG}, v{V_FLAG}, v{V_T0}", vw=[V_FLAG])
            self.finish(self.fixed(V_DST), self.fixed(V_TWIN))
            self.sizes["inv"] = len(e.ins) - n0
            return
        self.lds_load(10 * 5, self.fixed(V_T1))                          # a -> block 5
        e.raw("s_waitcnt lgkmcnt(0)")
        RA, X2, T5, T7, A1, A3 = OP(0), OP(1), OP(3), OP(4), OP(5), OP(6)
        for i in range(NL):
            e.emit(f"v_mov_b32_e32 v{RA[i]}, v{A1[i]}", vw=[RA[i]])
        self.l1().r_cvtout()                                             # v0..7: canonical words
        e.emit(f"v_or3_b32 v{V_T0}, v0, v1, v2", vw=[V_T0])
        e.emit(f"v_or3_b32 v{V_T0}, v{V_T0}, v3, v4", vw=[V_T0])
        e.emit(f"v_or3_b32 v{V_T0}, v{V_T0}, v5, v6", vw=[V_T0])
        e.emit(f"v_or_b32_e32 v{V_T0}, v{V_T0}, v7", vw=[V_T0])
        e.emit(f"v_cmp_eq_u32_e32 vcc, 0, v{V_T0}", w=["vcc"])
        e.emit(f"v_cndmask_b32_e64 v{V_T0}, 0, 1, vcc", r=["vcc"], vw=[V_T0])
        e.emit(f"v_or_b32_e32 v{V_FLAG}, v{V_FLAG}, v{V_T0}", vw=[V_FLAG])
        for i in range(NL):
            e.emit(f"v_mov_b32_e32 v{RA[i]}, v{A1[i]}", vw=[RA[i]])
        from kgen4_prog import KernelBuilder
        first, sched = KernelBuilder.fqinv_schedule(3)
        self.l1().fips_sq(RA, X2)                                        # a^2
        self.l1().fips([(RA, X2)], A3)                                   # a^3
        self.l1().fips([(A3, X2)], T5)                                   # a^5
        self.l1().fips([(T5, X2)], T7)                                   # a^7
        src = {1: None, 3: A3, 5: T5, 7: T7}[first]
        if src is not None:
            for i in range(NL):
                e.emit(f"v_mov_b32_e32 v{RA[i]}, v{src[i]}", vw=[RA[i]])
        for nsq, val in sched:
            e.salu(f"s_mov_b32 s{S_CNT}, {nsq}")
            e.salu(f"s_call_b64 {S_RET1}, LC_inv_sq_%=")
            if val:
                e.salu(f"s_call_b64 {S_RET1}, LC_inv_m{val}_%=")
        for i in range(NL):
            e.emit(f"v_mov_b32_e32 v{OUT0 + i}, v{RA[i]}", vw=[OUT0 + i])
        self.finish(self.fixed(V_DST), self.fixed(V_TWIN))
        e.label("LC_inv_sq_%=")
        self.l1().fips_sq(RA, RA)
        e.salu(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1")
        e.salu(f"s_cmp_lg_u32 s{S_CNT}, 0")
        e.salu("s_cbranch_scc1 LC_inv_sq_%=")
        e.salu(f"s_setpc_b64 {S_RET1}")
        for v, blk in ((1, A1), (3, A3), (5, T5), (7, T7)):
            e.label(f"LC_inv_m{v}_%=")
            self.l1().fips([(RA, blk)], RA)
            e.salu(f"s_setpc_b64 {S_RET1}")
        self.sizes["inv"] = len(e.ins) - n0

    # -------------------------------------------------------------- epilogue
    def epilogue(self):
        """lane r < 12 of a valid item converts result number r (c0 / c1 of the six coefficients, interleaved) and stores it as
        Fq number (r & 1) * 6 + (r >> 1) of the MyFq12 output; the zero-divisor flag goes to the status word"""
        e = self.e
        e.label("LC_end_%=")
        e.emit(f"v_min_u32_e32 v{V_T1}, 15, v{V_ROLE}", vw=[V_T1])                          # (sixteen entries; lanes 12.. store nothing)
        e.emit(f"v_lshlrev_b32_e32 v{V_T1}, 2, v{V_T1}", vw=[V_T1])
        e.emit(f"global_load_dword v{V_DST}, v{V_T1}, {S_BLOB} offset:32", kind="vmem", vw=[V_DST])       # the lane's output slot
        e.raw("s_waitcnt vmcnt(0)")
        e.emit(f"v_mov_b32_e32 v{V_T0}, v{V_DST}", vw=[V_T0])
        self.addr_of_slot(self.fixed(V_DST), V_T0)
        self.lds_load(0, self.fixed(V_DST))
        e.raw("s_waitcnt lgkmcnt(0)")
        self.l1().r_cvtout()
        e.emit(f"v_and_b32_e32 v{V_T0}, 1, v{V_ROLE}", vw=[V_T0])
        e.emit(f"v_mul_u32_u24_e32 v{V_T0}, 6, v{V_T0}", vw=[V_T0])
        e.emit(f"v_lshrrev_b32_e32 v{V_T1}, 1, v{V_ROLE}", vw=[V_T1])
        e.emit(f"v_add_u32_e32 v{V_T0}, v{V_T0}, v{V_T1}", vw=[V_T0])
        e.emit(f"v_lshlrev_b32_e32 v{V_T0}, 2, v{V_T0}", vw=[V_T0])
        e.emit(f"v_mul_lo_u32 v{V_T0}, v{V_T0}, s{S_NSTRIDE}", vw=[V_T0])
        e.emit(f"v_add_u32_e32 v{V_T0}, v{V_T0}, v{V_ITEM8}", vw=[V_T0])
        e.emit(f"v_cmp_gt_u32_e32 vcc, 12, v{V_ROLE}", w=["vcc"])
        e.emit(f"v_cndmask_b32_e32 v{V_T1}, 0, v{V_VALID}, vcc", r=["vcc"], vw=[V_T1])
        e.emit(f"v_cmp_ne_u32_e32 vcc, 0, v{V_T1}", w=["vcc"])
        e.raw("s_nop 1")
        e.salu(f"s_and_saveexec_b64 {S_SAVE}, vcc")
        for l in range(4):
            e.emit(f"global_store_dwordx2 v{V_T0}, v[{2 * l}:{2 * l + 1}], {S_OUT}", kind="vmem")
            if l < 3:
                e.emit(f"v_add_u32_e32 v{V_T0}, s{S_NSTRIDE}, v{V_T0}", vw=[V_T0])
        e.emit(f"v_cmp_ne_u32_e32 vcc, 0, v{V_FLAG}", w=["vcc"])
        e.raw("s_nop 1")
        e.salu("s_and_saveexec_b64 s[58:59], vcc")
        e.emit(f"v_mov_b32_e32 v{V_T0}, 1", vw=[V_T0])
        e.emit(f"v_mov_b32_e32 v{V_T1}, 0", vw=[V_T1])
        e.emit(f"global_store_dword v{V_T1}, v{V_T0}, {S_STATUS}", kind="vmem")
        e.salu(f"s_mov_b64 exec, {S_SAVE}")
        e.raw("s_waitcnt vmcnt(0)")
        e.salu(f"s_add_u32 s{S_ITEM0}, s{S_ITEM0}, s{S_GSTEP}")
        e.salu(f"s_cmp_lt_u32 s{S_ITEM0}, s{S_N}")
        e.salu("s_cbranch_scc1 LC_item_%=")

    def build(self):
        self.prologue()
        self.round_loop()
        self.mul_handler(cvm.K_M6, 6)
        self.mul_handler(cvm.K_M4, 4)
        self.mul_handler(cvm.K_M2, 2)
        self.lin_handler(cvm.K_L4, 4)
        self.lin_handler(cvm.K_L8, 8)
        self.inv_handler()
        self.epilogue()
        # every 8-byte instruction 8-byte aligned (asmcore.align_code: a straddling one costs a cycle on average -- measured on this
        # kernel: 5.1 cycles per instruction of the column passes without it)
        if not int(os.environ.get("CVM_ALIGN", "1")):           # A/B switch (round 6: at two waves per SIMD a 32-bit VOP1 / VOP2 encoding issues in 2 cycles, its
            return [".p2align 3"] + self.e.finalize()            # VOP3 re-encoding in 4: profiles/r06_occupancy_calib.txt)
        if int(os.environ.get("CVM_ALIGN", "1")) == 2:            # A/B switch: alignment by s_nop only
            return [".p2align 3"] + align_code(self.e.finalize(), nop_only=True)
        return [".p2align 3"] + align_code(self.e.finalize())


def make_blob(enc):
    """the program blob as a list of dwords: header, input descriptors, rows (one END row more than rounds, + the look-ahead's reach),
    constants (internal form: nine balanced 29-bit limbs of c R' mod p, + a pad dword)"""
    NR = enc["nr"]
    assert NR in (16, 32, 64)
    n_rounds = len(enc["rows"])
    trash = enc.get("trash", enc["n_slots"] - 1)
    ins = [slot | fq << 16 | arr << 20 | pair << 22 for slot, arr, fq, pair in enc["inputs"]]
    assert all(fq < 16 and pair < 64 for _, _, fq, pair in enc["inputs"])
    while len(ins) % NR:
        ins.append(trash | cvm.ARR_NONE << 20)
    rows = []
    for row in enc["rows"]:
        for dw in row:
            rows += dw
    rows += [0] * (NR * ROW_DW * 2)                        # the END row, and the row the look-ahead reads behind it
    consts = []
    for c in enc["consts"]:
        consts += [w & 0xFFFFFFFF for w in bal_limbs(c * K4.RP % P_INT)] + [0]
    ins_off = 4 * HDR_DW
    rows_off = (ins_off + 4 * len(ins) + 15) // 16 * 16
    pad = (rows_off - ins_off) // 4 - len(ins)
    consts_off = rows_off + 4 * len(rows)
    assert len(enc["outputs"]) == 12
    hdr = [n_rounds, len(enc["consts"]), ins_off, rows_off, consts_off, enc["n_slots"], trash, len(ins) // NR]
    hdr += [enc["outputs"][i] if i < 12 else trash for i in range(16)]
    assert len(hdr) == HDR_DW
    return hdr + ins + [0] * pad + rows + consts


if __name__ == "__main__":
    k = VMKernel()
    lines = k.build()
    print(len(lines), "lines;", k.sizes)
