#!/usr/bin/env python3
"""kgen4_prog.py -- L2 (Fq12 arithmetic, curve steps) and L3 (kernels) of the generated gfx950 assembly.

Everything a lane computes is emitted from here as ONE inline-asm blob per kernel on top of the L1 field routines of
tools/kgen4.py (balanced signed radix-2^29 limbs, nine limbs, Montgomery R' = 2^261); hipcc only provides the kernel
descriptor and hands the kernel arguments over in SGPRs.

Model
  * "accumulator machine" on Fq2 values: block A = v[0:17], block B = v[18:35]; an L1 routine computes A <- op(A, B).
    Values live in SLOTS of 18 dwords (72 B): LDS (8 per lane: four ds_*_b128 chunks + one ds_*_b64 tail), VGPR homes
    (9), AGPRs (14) and, for cold Fq12 registers of the final exponentiation and per-pair state of the multi-pairing
    kernels, global scratch.  L2 code is a sequence of   p.A(x).mul(y).sub(v1).mulxi().add(v0).to(c0).
  * STATIC BOUND TRACKER (class Prog): every value carries (a) an interval, in units of 2^28, that contains all of its
    limbs, (b) a bound, in units of p, of the integer it represents.  Each multiplication asserts at generation time that
    its signed 64-bit column sums cannot overflow, `norm` (one balanced carry pass) / `redn` (normalise and subtract the
    right multiple of p) are inserted exactly where a bound would be exceeded, and KernelBuilder.certify_values() replays
    every kernel's data-independent call sequence on per-slot bounds (tests/test_bounds.py).
  * the same instruction stream is executed by tools/ksim.py (single-lane simulator) in the CPU tests.

Reference functions implemented here (file:line under /root/reference/src):
  miller_loop_native.rs:10-44 (line functions), :46-110 (sparse multiplications), :112-190 / :192-282 (Miller loops),
  :298-312 (twisted Frobenius); final_exp_native.rs:17-54 (frobenius_map_native), :56-84 (pow_native), :130-169
  (hard part), :171-181 (conjugate), :195-213 (easy part, final_exp_native).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from asmcore import Emitter, P_INT, BN_X, SIX_U_PLUS_2_NAF, SIX_U_PLUS_2_SHORT, max_branch_distance, place_with_islands  # noqa: E402
from kgen4 import DIGIT_ADD as K4_DIGIT_ADD, S_HALF as K4_S_HALF  # noqa: E402
from kgen4 import (A0, B0, HOME0, L1V4_NAMES, L1v4, LB, MUL3_KEEP_DY, N0P, N_AGPR_SLOTS, N_HOME, N_LDS_SLOTS, NL, P_L, REDN_C, S_M30, S_N0, S_P, S_REDN,  # noqa: E402
                   S_RET1, S_RET2, S_RET3, SLOT_DW, SLOT_BYTES, V_FLAG, V_GOFF, V_GOFF8, V_IDX, V_IDX8, V_LDS, V_LTAIL, V_TID, bal_limbs, hx, mont4)

# ---- scalar registers used by L2/L3 (all inside the clobbered range s36..s99) ----------------
S_TMP0, S_TMP1 = 60, 61
S_GADDR = "s[62:63]"      # address of the global slot being accessed
S_SCRATCH = "s[64:65]"    # scratch base of this workgroup
S_GSTRIDE = 66            # bytes between consecutive global slots
S_I = 67                  # Miller-loop digit index
S_NAF_NZ = "s[68:69]"     # 6u+2 NAF: non-zero mask, negative mask (digits 0..63)
S_NAF_NEG = "s[70:71]"
# (s50..s53, s72..s75 were freed when the x-power schedule became unrolled control code; since then: s50..s53 the split-loop experiment,
#  s[52:53] the digit extraction's rounding constant when that experiment is off, s72..s74 the alternating pair passes)
S_J = 76                  # pow_x digit index
S_GBASE = 77              # global Fq12 register operand of fq12_mul (slot number * stride, low 32 bits)
S_ITEM = 78
S_NITEMS = 79
S_G1 = "s[80:81]"
S_G2 = "s[82:83]"
S_OUT = "s[84:85]"
S_N = 86                  # batch size (u32)
S_GRID = 87
S_IOADDR = "s[88:89]"
S_NSTRIDE = 90            # n * 8 (bytes between limbs in the SoA batch)
S_PHASE = 91
S_STATUS = "s[92:93]"
S_FIN = "s[94:95]"
S_K = 96
S_JP = 97                 # pair counter of the multi-pairing kernels
S_SAVE_EXEC = "s[98:99]"
S_PB = 48                 # byte offset of the base's scratch register during the x-power routine
S_MODE_SINGLE, S_MODE_MULTI = 49, 51      # I/O layout of this launch (bits 28..30 of the kernel's k argument; KernelBuilder.s_mode): 1 inputs element-major,
                          # 2 output element-major, 4 ... in ark's Fq12 order.  (s49 is the k-pair kernels' S_GNEXT, s51 the split-loop experiment's cursor step)
S_IOSTRIDE = 75           # bytes between consecutive words of the array being walked: the plane stride (limb-major) or 8 (element-major)
S_TAB = "s[72:73]"        # fixed-G2 kernels: address of the current line triple of the current fixed pair in the line table (s74: the cursor in bytes)
S_TABCUR = 74
V_IOOFF = 247             # the lane's byte offset into the array being walked: index * 8 (limb-major) or index * 8 * words per element
MODE_IN_ELEMS, MODE_OUT_ELEMS, MODE_OUT_ARK = 0, 1, 2      # bit numbers in S_MODE
MODE_NO_OWN = 3            # fixed-G2 kernel: the groups have NO pair of their own -- every G2 point is one of the table's (a KZG-style check: e(P_1, Qfix_1) e(P_2, Qfix_2))
BLOCK = 256


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


class GlobLine:
    """Global slot k of the line triple at the phase-1 cursor S_LOFF (bytes inside the workgroup's scratch block)."""

    def __init__(self, k):
        self.kind, self.k = "globline", k


class Tab:
    """Slot k of the line triple at the table cursor S_TAB (fixed-G2 kernels): 72 contiguous bytes at S_TAB + 72 k + the lane's V_IOOFF."""

    def __init__(self, k):
        self.kind, self.k = "tab", k


class Const:
    """Fq2 constant (canonical integers), materialised with literal moves."""

    def __init__(self, c0, c1, name=""):
        self.kind, self.c0, self.c1, self.name = "const", c0, c1, name


def f2mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P_INT, (a[0] * b[1] + a[1] * b[0]) % P_INT)


def f2pow(a, e):
    r = (1, 0)
    for bit in bin(e)[2:]:
        r = f2mul(r, r)
        if bit == "1":
            r = f2mul(r, a)
    return r


def frob_const(k, i):
    """frob_coeffs(k)^i = xi^(i (p^k - 1)/6): the factor frobenius_map_native(a, k) puts on the coefficient of w^i
    (final_exp_native.rs:27,33-42)"""
    return f2pow(f2pow((9, 1), (P_INT ** k - 1) // 6), i)


def twist_consts():
    """c2 = xi^((p-1)/3), c3 = xi^((p-1)/2)   (miller_loop_native.rs:176-181)"""
    c = f2pow((9, 1), (P_INT - 1) // 6)
    c2 = f2mul(c, c)
    return c2, f2mul(c2, c)


def naf_masks(naf):
    nz = sum(1 << i for i, d in enumerate(naf) if d != 0)
    neg = sum(1 << i for i, d in enumerate(naf) if d < 0)
    return nz, neg


def _three_b():
    xi_inv_n = pow(82, -1, P_INT)          # 1/(9+u) = (9 - u)/82
    return Const(81 * xi_inv_n % P_INT, (-9 * xi_inv_n) % P_INT, "threeb")


THREE_B = _three_b()        # 3 b' = 9 / xi (twist curve y^2 = x^3 + 3/xi)

# ---- x-power schedule of the final exponentiation (F <- F^BN_X for cyclotomic F, three times per pairing).
# pow_native (final_exp_native.rs:56-84) walks the NAF of BN_X: 62 squarings + 23 multiplications.  The value F^x does
# not depend on the chain, so the kernels use a signed fixed-set recoding instead.  Rounds 2 - 4: digits in {0, +-1, +-5, +-9, +-13}, 61 S
# + 15 M.  Round 5 (tools/exp/xchain2.py: every set of up to three odd powers below 64, the optimal recoding of each by dynamic
# programming over (bit, carry), the table's own squarings and multiplications counted): digits in {0, +-1, +-15, +-19} -- 58 squarings +
# 11 multiplications in the loop, b^4, b^16 by four squarings, b^15 = b^16 conj(b), b^19 = b^15 b^4: **62 S + 13 M** (a squaring is 4.5 k
# instructions, a multiplication 12.5 k: -20.5 k per x-power).  Negative digits multiply by the conjugate (= inverse of a unitary element).
X_POWERS = (1, 15, 19)
X_HOT = 19                       # ten of the twelve non-zero digits: conj(b^19) stays in the on-chip register LREG
X_DIGITS = (-15, 0, 0, 0, 0, 0, 0, 0, 0, -19, 0, 0, 19, 0, 0, 0, 0, 0, 0, -19, 0, 0, 0, 0, -1, 0, 19, 0, 0, 0, 0, 0, 0, 0, -19, 0, 0, 0, 0, 0, 19,
            0, 0, 0, 0, 0, 0, 19, 0, 0, 0, 0, 0, -19, -19, 0, 0, 0, 19)            # least significant first
assert sum(d << i for i, d in enumerate(X_DIGITS)) == BN_X and X_DIGITS[-1] == X_HOT
assert all(d == 0 or abs(d) in X_POWERS for d in X_DIGITS)
G_POW = {15: 8}                  # scratch Fq12 register of b^15 (b itself: the caller's register; b^19: LREG)
G_B4 = 11                        # b^4 (only while the powers are built)
N_GREG = 12                      # Fq12 scratch registers G0..G11 = slots 0..71
GLOB_TMP0 = 6 * N_GREG           # eight overflow temporaries: slots 72..79
N_GSLOTS = GLOB_TMP0 + 8         # scratch slots of the single-pairing kernels; pair j of a multi kernel: N_GSLOTS + 7 j + ...
GCHUNK0 = 512                    # byte offset of the first 16-byte chunk plane inside a wave's part of a scratch slot (after the tails)

# timing experiments only (tools/exp/build_variant.sh; results are WRONG with any of these set): drop the streamed R stores /
# the pair-state prefetch loads of the multi-pairing loop, to see what the memory side of the stream costs
EXP_NO_RSTORE = bool(int(os.environ.get("KGEN_EXP_NO_RSTORE", "0")))
EXP_NO_PREFETCH = bool(int(os.environ.get("KGEN_EXP_NO_PREFETCH", "0")))
# Cache-policy bits of the scratch loads / stores.  Measured on the Groth16 shape (same-box A/B, profiles/r03_ab.txt): stores
# with system scope (sc0 sc1: written through, nothing lingers in L2 for data that is re-read 100 us later) +2.4 %, "nt" +1.7 %,
# sc0 / sc1 alone or nt + sc1 worse; any bit on the loads costs 0.3 .. 0.6 %.  Neutral on k_pairing (+0.1 %).
SCRATCH_LD_MOD = os.environ.get("KGEN_SCRATCH_LD_MOD", "")
SCRATCH_ST_MOD = os.environ.get("KGEN_SCRATCH_ST_MOD", "sc0 sc1")
EXP_NO_SCRATCH = bool(int(os.environ.get("KGEN_EXP_NO_SCRATCH", "0")))      # TIMING ONLY: no scratch load / store is emitted at all
# TIMING ONLY (results of groups of two and more pairs are WRONG): pair 1's X and Y (KGEN_EXP_R1=2) or X alone (=1) are neither prefetched
# nor stored in the streamed loop -- what a partly resident second pair could gain at most (round 5, profiles/r05_ab.txt)
EXP_R1 = int(os.environ.get("KGEN_EXP_R1", "0"))
# The line coefficients go from the fused point step to the sparse multiplication in registers (home blocks 7, 4, 5; the xi-multiplied
# copies are formed straight in the operand blocks) instead of through three AGPR slots and two temporaries: -180 moves per sparse
# multiplication, no LDS temporary left in it, and the freed slots take the result temporaries.
LINE_IN_REGS = bool(int(os.environ.get("KGEN_LINE_REGS", "1")))
# Round 4 -- LOOP FISSION of the Miller loop (untracked kernels; groups of up to FIS_MAX_K pairs).  The main loop's code -- f^2 (glue + the
# fused Fq6 multiplication: 40 KB), the fused doubling step (47 KB) or addition step (62 KB), the sparse multiplication (16 KB) -- is
# 103 .. 130 KB per iteration against a 64 KB instruction cache: the per-routine cycle counters (profiles/r04_l2_profile.json) show the
# Miller routines at 4.38 cycles per instruction where every leaf routine alone issues at 4.000 and the cyclotomic-squaring loop (38 KB)
# at 4.09.  The point steps do not depend on f, so the loop is split: PHASE 1 walks the whole chain of point steps of one pair with R
# resident in home registers (no LDS / scratch round trip per step) and leaves every step's line coefficients (three slots) in a per-lane
# line area of the scratch; PHASE 2 is the f loop -- f^2, then per pair one sparse multiplication by a line that was prefetched into three
# AGPR slots one step ahead.  Each loop's code fits the cache.  For k-pair groups this also REPLACES the R stream (load R / step / store R
# per pair and step, with its dependent round trip) by a line stream of the same volume that is written once and read once, sequentially.
# MEASURED, NOT ADOPTED (default off; profiles/r04_ab.txt "fission"): the split loop does what it was built for -- the wave's shader cycles
# fall (k_pairing 246.5 -> 245.3 M per wave, the Groth16 shape 130.2 -> 128.0 M; L2_sqr 4.28 -> 4.06 cycles per instruction) -- but the
# 40 GB of line traffic per launch are paid in CLOCK: the in-kernel clock fell from 2.37 to 2.19 GHz on the Groth16 shape and by 3 % on
# k_pairing (tools/clock_stamp.py, same box), so the launches got 2 .. 4 % slower.  HBM traffic costs package power, and the package is close
# enough to its limit that power is taken from the shader clock.
FISSION = bool(int(os.environ.get("KGEN_FISSION", "0")))
# Round 5 -- TWO DOUBLING ITERATIONS PER CHUNK, the lines kept on chip (the untracked one-pair kernel, k_pairing).  The cure for the
# instruction supply that costs no memory traffic: where digit i of the loop is zero, iterations i and i - 1 run as
#     dbl step (i), dbl step (i - 1)  |  f^2, f * line (i), f^2, f * line (i - 1)        (then iteration i - 1's addition step, if any)
# -- the second doubling step finds the 47 KB of the first one in the instruction cache, the second f^2 + sparse multiplication (56 KB)
# those of the first -- instead of [f^2, dbl step, sparse multiplication] twice (103 KB each through a 64 KB cache).  The point steps do
# not depend on f; the two lines wait in six slots that nothing in the loop touches (LDS 2 .. 5, which the Miller phase of this kernel
# never used, and the two AGPR slots of the addition point, free outside addition steps).  Iterations with a non-zero digit keep the
# round-4 sequence.  ONE copy of each code: the park set is picked by a scalar register (S_PARK).
# Round 5 -- the kernels whose Miller value goes straight into the final exponentiation (k_pairing, k_mpairing) walk the MINIMAL-WEIGHT
# 65-digit form of 6 x + 2 (22 non-zero digits: 64 doublings + 21 additions) instead of the reference's digit table (65 digits, 26 non-zero:
# 64 + 25): the chain changes the Miller value by factors from proper subfields only (vertical lines, line scales), which (p^6 - 1) kills --
# pairing(p, q) and the products of pairings come out limb for limb the same (tools/asmcore.py).  Four addition steps (16 k instructions
# each) less per pair.  k_miller / k_mmiller (the exact values) keep the reference's table.
SHORT_CHAIN = bool(int(os.environ.get("KGEN_SHORT_CHAIN", "1")))
CHUNK2 = bool(int(os.environ.get("KGEN_CHUNK2", "1")))
CHUNK_LAYOUT = bool(int(os.environ.get("KGEN_CHUNK_LAYOUT", "0")))     # measured: the contiguous f half LOSES the chunking's +0.24 % again (profiles/r05_ab.txt)
S_PARK = 50                                    # (s50 / s51: the split-loop experiment's cursors, free without it)
FIS_MAX_K = 4                                  # larger groups keep the streamed loop (the line area is 3 x FIS_STEPS slots per pair)
FIS_STEPS = 63 + sum(1 for d in SIX_U_PLUS_2_NAF[:64] if d)       # point steps of the main loop: 63 doublings + 26 additions
S_LOFF, S_LSTEP, S_LOFF2 = 50, 51, 52         # byte offsets (inside the workgroup's scratch block): phase-1 line cursor, its step, phase-2 cursor
S_LCNT = 53                                    # phase 2: lines not yet fetched
# Round 4: the Fq12 inversion of the easy part keeps its temporaries out of home blocks 0..7, so that its four Fq6 multiplications run on the
# fused L1 routine (mul6) instead of six generic Fq2 multiplications plus glue each
INV_FUSED = bool(int(os.environ.get("KGEN_INV_FUSED", "1")))
EXP_NO_SWAIT = bool(int(os.environ.get("KGEN_EXP_NO_SWAIT", "0")))          # no s_waitcnt at the start of a streamed step
# The next pair's prefetch is issued slot by slot behind the first four passes of the current pair's sparse multiplication instead
# of as one burst of 25 loads in front of the step: +1.9 % on the Groth16 shape (the four waves of a CU run in step: a burst is
# 100 KiB-lines at once into one L1)
SPREAD_PREFETCH = int(os.environ.get("KGEN_SPREAD_PF", "1"))
MARKERS = bool(int(os.environ.get("KGEN_MARKERS", "0")))      # tools/instr_histogram.py: LM_* labels at the phase changes inside routines
_marker_n = [0]
# Scratch layout.  "slot": [slot][workgroup][wave][...] -- a workgroup's 80+ slots lie a whole grid's worth apart (4.7 MB at a
# full grid: every slot access of a CU touches a different 2 MiB page).  "wg": [workgroup][slot][wave][...] -- all the slots
# of a workgroup are contiguous (kernel argument %7 is then the workgroup pitch, the slot pitch a constant).
SCRATCH_WG = os.environ.get("KGEN_SCRATCH", "wg") == "wg"
WG_SLOT_PITCH = BLOCK_LANES_SLOT = 256 * SLOT_BYTES            # bytes of one slot of one workgroup (4 waves x 4608 B)
# DIAGNOSTIC builds only (tools/exp/build_variant.sh stamp KGEN_CLOCK_STAMP=1; never the shipped header): every wave stamps s_memtime (shader
# cycles) and s_memrealtime (100 MHz) once in front of and once behind its item loop and leaves the two differences at the end of
# its workgroup's scratch block (the slack behind the last slot: the pitch is rounded up to 2 MiB), where no kernel code reads:
# the in-kernel clock = d(memtime) / d(memrealtime) x 100 MHz (MI355X guide, 'DVFS give-back' item 6).
CLOCK_STAMP = bool(int(os.environ.get("KGEN_CLOCK_STAMP", "0")))
# DIAGNOSTIC builds only (KGEN_PROFILE_L2=1, with KGEN_CLOCK_STAMP=1): every call of an L2 routine from the main program / the x-power
# control code is bracketed by s_memtime stamps; the wave accumulates, per routine, the shader cycles (inclusive) and the number of
# calls in lanes of three otherwise unused VGPRs (v248..v250, lane = routine id) and leaves them next to the clock stamps.  With the
# simulator's per-routine instruction counts (tools/l2_profile.py) this gives cycles per instruction for every routine.
PROFILE_L2 = bool(int(os.environ.get("KGEN_PROFILE_L2", "0")))
PROFILE_IDS = {}
PROFILE_OFFSET_FROM_END = 16384           # [wave][3][64] dwords in front of the clock stamps


def profile_id(name):
    name = name.replace("_%=", "")
    if name not in PROFILE_IDS:
        PROFILE_IDS[name] = len(PROFILE_IDS)
        assert len(PROFILE_IDS) <= 64
    return PROFILE_IDS[name]
STAMP_OFFSET_FROM_END = 4096
# multi-pairing kernels: the first doubling step and the two Frobenius addition steps of every pair run on the resident-slot routines
# L2_dblmul / L2_addmul -- 3 k - 1 step + sparse-multiplication pairs per group.  As COLD routines (generic point steps, L1 calls)
# they save 110 KB of code that is executed 11 times per k = 4 group; as hot ones (fused steps) those executions are cheaper.
MULTI_HOT_ENDS = bool(int(os.environ.get("KGEN_MULTI_HOT_ENDS", "0")))
ALIGN_CODE = bool(int(os.environ.get("KGEN_ALIGN", "1")))     # keep 8-byte instructions 8-byte aligned (asmcore.align_code)

# ---- static bound tracking -------------------------------------------------------------------------------------------
# LIMB intervals are in units of U = 2^28 (the magnitude bound of a normalised balanced limb) and cover all limbs of both
# Fq2 components; VALUE bounds are in units of p.
#   * a column pass over n_terms limb products of magnitudes (ma, mb) plus the reduction's 9 products of balanced m and
#     p limbs stays inside the signed 64-bit accumulator iff  n_terms ma mb + 9 + (carry) < 2^63 / 2^56 = 128;
#   * a reduction maps sum(a_i b_i) to sum(a_i b_i) / R' +- p/2 and R'/p = 169.6: values contract only while they are small;
#   * the top limb (weight 2^232) carries the representative: value v p <-> |top| = v / K_TOP units.
COL_BUDGET = 118.0
K_RP = float((1 << (NL * LB)) / P_INT)              # 169.6
K_TOP = float((1 << (LB * (NL - 1) + LB - 1)) / P_INT)   # 84.8: value (in p) per unit of the top limb
V_CAP = 84.0                # nothing larger is ever stored or multiplied: the top limb then stays within one unit
V_STORE = 4.0               # contract: bound (in p) of a value another routine left in a slot (routine boundaries)
V_REDN_AT = 4.0             # a store that has to normalise anyway reduces as well when the value bound exceeds this
STORE_MAG = 2.05            # stored values may keep limbs of up to two units (sums / differences of two normalised values)
LIMB_MAG = 6.9              # int32 limbs that `norm` may meet: |limb| + 2^28 < 2^31
R_NORM = (-1.0, 1.0)


def _ldm():
    return (" " + SCRATCH_LD_MOD) if SCRATCH_LD_MOD else ""


def _stm():
    return (" " + SCRATCH_ST_MOD) if SCRATCH_ST_MOD else ""


def mag(r):
    return max(abs(r[0]), abs(r[1]))


def r_add(a, b):
    return (a[0] + b[0], a[1] + b[1])


def r_sub(a, b):
    return (a[0] - b[1], a[1] - b[0])


def r_neg(a):
    return (-a[1], -a[0])


def r_hull(a, b):
    return (min(a[0], b[0]), max(a[1], b[1]))


class Prog:
    """Accumulator-machine program over slots with static limb / value bounds."""

    UNKNOWN = (-STORE_MAG, STORE_MAG)      # contract for values stored by other routines
    STORED_NORM = R_NORM                   # slots in `norm_keys` hold NORMALISED values on every routine boundary

    def __init__(self, e, l1_labels):
        self.e = e
        self.l1 = l1_labels
        self.tagA = None
        self.tagB = None
        self.free_tmp = []
        self.lds_pending = False
        self.vm_pending = False
        self.stats = {}
        self.rA = None
        self.slot_r = {}
        self.vA = V_STORE
        self.slot_v = {}
        self.max_v = 0.0
        self.norm_keys = frozenset()
        self.entry_v = {}           # certification: bounds of the values other routines left in the slots
        self.default_v = V_STORE
        self.read_keys = {}         # slot keys whose entry bound was used -> the bound
        self.temp_keys = frozenset()
        self.homes_free = False
        self.cold = False
        self._blocks_reserved = False

    # ---------------------------------------------------------------- bounds
    @staticmethod
    def key(slot):
        if slot.kind in ("globdyn", "globline", "tab"):
            return (slot.kind, slot.k)
        if slot.kind == "const":
            return ("const", slot.name)
        return (slot.kind, slot.idx)

    def v_of(self, slot):
        if slot.kind == "const":
            return 1.0
        k = self.key(slot)
        if k in self.slot_v:
            return self.slot_v[k]
        v = self.entry_v.get(k, self.default_v)
        self.read_keys[k] = v
        return v

    def r_of(self, slot):
        if slot.kind == "const":
            return R_NORM
        k = self.key(slot)
        return self.slot_r.get(k, self.STORED_NORM if k in self.norm_keys else self.UNKNOWN)

    def r_norm(self, v=None):
        """Limb interval of a normalised value: limbs 0..NL-2 in [-1, 1) units; the signed top limb carries the value."""
        t = (self.vA if v is None else v) / K_TOP
        return (-max(1.0, t), max(1.0, t))

    def _need(self, ok, what):
        if not ok:
            raise AssertionError("bound violated: " + what)

    def _count(self, k):
        self.stats[k] = self.stats.get(k, 0) + 1

    def marker(self, name):
        """profiling builds only (KGEN_MARKERS=1): a label the simulator's region map picks up; the shipped code has none"""
        if MARKERS:
            self.wait()
            _marker_n[0] += 1
            self.e.label(f"LM_{name}_{_marker_n[0]}_%=")

    # ---------------------------------------------------------------- data movement (18-dword slots)
    N_B128 = SLOT_DW // 4          # four 16-byte chunks ...
    TAIL_DW = SLOT_DW % 4          # ... and one 8-byte tail

    def _lds_addr(self, slot, c):
        """(address VGPR, offset) of 16-byte chunk c of an LDS slot: [slot][chunk][lane] uint4"""
        ci = slot.idx * self.N_B128 + c
        return V_LDS + ci // 16, (ci % 16) * 4096

    def _lds_tail(self, slot):
        return V_LTAIL, slot.idx * 2048

    def _glob_base(self, slot):
        if slot.kind == "glob":
            self.e.salu(f"s_mul_i32 s{S_TMP0}, s{S_GSTRIDE}, {slot.idx}")
        elif slot.kind == "globline":
            self.e.salu(f"s_mul_i32 s{S_TMP0}, s{S_GSTRIDE}, {slot.k}")
            self.e.salu(f"s_add_u32 s{S_TMP0}, s{S_TMP0}, s{S_LOFF}")
        else:
            self.e.salu(f"s_mul_i32 s{S_TMP0}, s{S_GSTRIDE}, {slot.k}")
            self.e.salu(f"s_add_u32 s{S_TMP0}, s{S_TMP0}, s{S_GBASE}")
        self.e.salu(f"s_add_u32 s62, s64, s{S_TMP0}")
        self.e.salu("s_addc_u32 s63, s65, 0")

    def load(self, blk, slot):
        e = self.e
        if slot.kind == "lds":
            for c in range(self.N_B128):
                base, off = self._lds_addr(slot, c)
                e.emit(f"ds_read_b128 v[{blk + 4 * c}:{blk + 4 * c + 3}], v{base} offset:{off}", kind="lds", vw=range(blk + 4 * c, blk + 4 * c + 4))
            base, off = self._lds_tail(slot)
            e.emit(f"ds_read_b64 v[{blk + 16}:{blk + 17}], v{base} offset:{off}", kind="lds", vw=[blk + 16, blk + 17])
            self.lds_pending = True
        elif slot.kind == "home":
            r0 = HOME0 + SLOT_DW * slot.idx
            for i in range(SLOT_DW):
                e.emit(f"v_mov_b32_e32 v{blk + i}, v{r0 + i}", vw=[blk + i])
        elif slot.kind == "agpr":
            for i in range(SLOT_DW):
                e.emit(f"v_accvgpr_read_b32 v{blk + i}, a{SLOT_DW * slot.idx + i}", vw=[blk + i])
        elif slot.kind == "const":
            w = bal_limbs(mont4(slot.c0)) + bal_limbs(mont4(slot.c1))
            for i in range(SLOT_DW):
                e.emit(f"v_mov_b32_e32 v{blk + i}, {hx(w[i])}", vw=[blk + i])
        elif slot.kind == "tab":
            for c in range(self.N_B128):
                e.emit(f"global_load_dwordx4 v[{blk + 4 * c}:{blk + 4 * c + 3}], v{V_IOOFF}, {S_TAB} offset:{SLOT_BYTES * slot.k + 16 * c}", kind="vmem",
                       vw=range(blk + 4 * c, blk + 4 * c + 4))
            e.emit(f"global_load_dwordx2 v[{blk + 16}:{blk + 17}], v{V_IOOFF}, {S_TAB} offset:{SLOT_BYTES * slot.k + 64}", kind="vmem", vw=[blk + 16, blk + 17])
            self.vm_pending = True
        elif slot.kind in ("glob", "globdyn", "globline") and EXP_NO_SCRATCH:
            pass
        elif slot.kind in ("glob", "globdyn", "globline"):
            self._glob_base(slot)
            for c in range(self.N_B128):
                e.emit(f"global_load_dwordx4 v[{blk + 4 * c}:{blk + 4 * c + 3}], v{V_GOFF}, {S_GADDR} offset:{GCHUNK0 + 1024 * c}" + _ldm(), kind="vmem",
                       vw=range(blk + 4 * c, blk + 4 * c + 4))
            e.emit(f"global_load_dwordx2 v[{blk + 16}:{blk + 17}], v{V_GOFF8}, {S_GADDR} offset:0" + _ldm(), kind="vmem", vw=[blk + 16, blk + 17])
            self.vm_pending = True
        else:
            raise ValueError(slot.kind)
        self._count("ld_" + slot.kind)

    def store(self, blk, slot):
        e = self.e
        if slot.kind == "lds":
            for c in range(self.N_B128):
                base, off = self._lds_addr(slot, c)
                e.emit(f"ds_write_b128 v{base}, v[{blk + 4 * c}:{blk + 4 * c + 3}] offset:{off}", kind="lds")
            base, off = self._lds_tail(slot)
            e.emit(f"ds_write_b64 v{base}, v[{blk + 16}:{blk + 17}] offset:{off}", kind="lds")
        elif slot.kind == "home":
            r0 = HOME0 + SLOT_DW * slot.idx
            for i in range(SLOT_DW):
                e.emit(f"v_mov_b32_e32 v{r0 + i}, v{blk + i}", vw=[r0 + i])
        elif slot.kind == "agpr":
            for i in range(SLOT_DW):
                e.emit(f"v_accvgpr_write_b32 a{SLOT_DW * slot.idx + i}, v{blk + i}")
        elif slot.kind == "tab":
            for c in range(self.N_B128):
                e.emit(f"global_store_dwordx4 v{V_IOOFF}, v[{blk + 4 * c}:{blk + 4 * c + 3}], {S_TAB} offset:{SLOT_BYTES * slot.k + 16 * c}", kind="vmem",
                       store=range(blk + 4 * c, blk + 4 * c + 4))
            e.emit(f"global_store_dwordx2 v{V_IOOFF}, v[{blk + 16}:{blk + 17}], {S_TAB} offset:{SLOT_BYTES * slot.k + 64}", kind="vmem", store=[blk + 16, blk + 17])
            e.raw("s_nop 1")
        elif slot.kind in ("glob", "globdyn", "globline") and EXP_NO_SCRATCH:
            pass
        elif slot.kind in ("glob", "globdyn", "globline"):
            self._glob_base(slot)
            for c in range(self.N_B128):
                e.emit(f"global_store_dwordx4 v{V_GOFF}, v[{blk + 4 * c}:{blk + 4 * c + 3}], {S_GADDR} offset:{GCHUNK0 + 1024 * c}" + _stm(), kind="vmem",
                       store=range(blk + 4 * c, blk + 4 * c + 4))
            e.emit(f"global_store_dwordx2 v{V_GOFF8}, v[{blk + 16}:{blk + 17}], {S_GADDR} offset:0" + _stm(), kind="vmem", store=[blk + 16, blk + 17])
            e.raw("s_nop 1")        # wide-store data hazard: the next VALU write of the block may sit behind a call
        else:
            raise ValueError(slot.kind)
        self._count("st_" + slot.kind)

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
        self.rA = None

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

    def set_A_fresh(self, v=0.51):
        """A was filled by hand-written code with a normalised value of at most v p."""
        self.tagA = None
        self.rA = R_NORM
        self.vA = v

    # a call/return pair costs a lone wave ~70 cycles (two taken branches, each refilling the instruction buffer):
    # the short routines are inlined (cold routines keep calling the longer ones: code size)
    INLINE_SET = ("add", "sub", "rsub", "dbl", "neg", "negc1", "norm", "mulxi", "mulxir", "redn", "dblstep", "addstep")   # (the fused steps have one call site each)
    INLINE_COLD = ("add", "sub", "rsub", "dbl", "neg", "negc1")

    def _raw_call(self, name):
        self.wait()
        base = name.split("_h")[0]
        if base in (self.INLINE_COLD if self.cold else self.INLINE_SET):
            g = L1v4(self.e)
            if "_h" in name:
                g.home_variant(base, int(name.split("_h")[1]))
            else:
                getattr(g, "r_" + name)()
        else:
            self.e.salu(f"s_call_b64 {S_RET1}, {self.l1[name]}")
        self.tagA = None
        self._count(name)

    def norm(self):
        """A <- normalised A.  Limbs beyond what the 32-bit carry pass takes go through the 64-bit chain (redn)."""
        rA = self.rA if self.rA is not None else self.UNKNOWN
        if mag(rA) > LIMB_MAG or self.vA > V_REDN_AT:
            return self.redn()
        self._raw_call("norm")
        self.rA = self.r_norm()
        return self

    def redn(self):
        """Normalise and bring the representative back to (-0.51 p, 0.51 p) (L1 redn: quotient from the top limb)."""
        self._need(self.vA <= 8 * V_CAP, f"redn of a value of {self.vA} p")       # |top limb| must fit an int32 with room
        self._raw_call("redn")
        self.vA = 0.51
        self.rA = R_NORM
        return self

    @staticmethod
    def cols(n_terms, ra, rb):
        return n_terms * mag(ra) * mag(rb)

    def call(self, name, rB=None, direct=None, vB=1.0, k=None):
        vA = self.vA
        rA = self.rA if self.rA is not None else self.UNKNOWN
        if name in ("mul", "mulfq", "fqmul"):
            n = 2 * NL if name == "mul" else NL
            if self.cols(n, rA, rB) > COL_BUDGET or vA > V_CAP:
                self.norm()
                rA, vA = self.rA, self.vA
            self._need(self.cols(n, rA, rB) <= COL_BUDGET, f"{name} {rA} {rB}")
            v_out = (2 if name == "mul" else 1) * vA * vB / K_RP + 0.5
            out = self.r_norm(v_out)
        elif name in ("sqr", "fqsqr"):
            t = r_add(rA, rA) if name == "sqr" else rA
            if self.cols(NL, t, t) > COL_BUDGET or vA > V_CAP:
                self.norm()
                rA, vA = self.rA, self.vA
                t = r_add(rA, rA) if name == "sqr" else rA
            self._need(self.cols(NL, t, t) <= COL_BUDGET, f"{name} {rA}")
            v_out = (4 if name == "sqr" else 1) * vA * vA / K_RP + 0.5
            out = self.r_norm(v_out)
        elif name in ("add", "sub", "rsub"):
            f = {"add": r_add, "sub": r_sub, "rsub": lambda x, y: r_sub(y, x)}[name]
            if mag(f(rA, rB)) > LIMB_MAG:
                self.norm()
                rA, vA = self.rA, self.vA
            out = f(rA, rB)
            self._need(mag(out) <= LIMB_MAG, f"{name} {rA} {rB}")
            v_out = vA + vB
        elif name == "dbl":
            if 2 * mag(rA) > LIMB_MAG:
                self.norm()
                rA, vA = self.rA, self.vA
            out = (2 * rA[0], 2 * rA[1])
            v_out = 2 * vA
        elif name == "neg":
            out, v_out = r_neg(rA), vA
        elif name == "negc1":
            out, v_out = r_hull(rA, r_neg(rA)), vA
        elif name == "scale":
            # A <- k A on 64-bit chains (k: inline constant), normalised
            self._need(mag(rA) <= 7.9 and -16 <= k <= 64, f"scale {k} {rA}")
            v_out = abs(k) * vA
            out = self.r_norm(v_out)
            self.wait()
            g = L1v4(self.e)
            g.lincomb(list(g.fq2(A0)), [[(k, g.blk(A0, 0))], [(k, g.blk(A0, 1))]])
            self.tagA = None
            self._count("scale")
            self.vA, self.rA = v_out, out
            self._need(v_out <= V_CAP, f"scale: value {v_out}")
            return self
        elif name == "mulxi":
            # 64-bit chains: any int32 limbs in, normalised out; values grow tenfold
            self._need(mag(rA) <= 7.9, f"mulxi {rA}")
            v_out = 10 * vA
            if v_out > V_CAP / 2 or k == "reduce":
                name = direct = "mulxir"
                v_out = 0.51
            out = self.r_norm(v_out)
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

    def mul(self, y): return self._bin("mul", y)
    def add(self, y): return self._bin("add", y)
    def sub(self, y): return self._bin("sub", y)
    def rsub(self, y): return self._bin("rsub", y)
    def mulfq(self, y): return self._bin("mulfq", y)      # A * (Fq in y.c0)

    def mulfq_c1(self, y):
        """A * (the Fq in y.c1): y is a slot that packs two Fq values (the evaluation point (Px, Py) of a fixed pair)"""
        self._B(y)
        self.wait()
        for i in range(NL):
            self.e.emit(f"v_mov_b32_e32 v{B0 + i}, v{B0 + NL + i}", vw=[B0 + i])
        self.tagB = None
        return self.call("mulfq", self.r_of(y), vB=self.v_of(y))

    def sqr(self): return self.call("sqr")
    def dbl(self): return self.call("dbl")
    def neg(self): return self.call("neg")
    def conj(self): return self.call("negc1")
    def mulxi(self, reduce=False): return self.call("mulxi", k="reduce" if reduce else None)
    def scale(self, k): return self.call("scale", k=k)

    def v_limit(self, dst):
        """Largest value bound (in p) that may be left in `dst`: routine temporaries up to the cap, everything that
        crosses a routine boundary at most V_STORE (the bound every routine assumes for its inputs)."""
        return V_CAP if self.key(dst) in self.temp_keys else V_STORE

    def to(self, dst):
        rA = self.rA if self.rA is not None else self.UNKNOWN
        lim = 1.0 if self.key(dst) in self.norm_keys else STORE_MAG
        if self.vA > self.v_limit(dst):
            self.redn()
        elif mag(rA) > max(lim, self.vA / K_TOP):
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

    def mov(self, dst, src):
        self.A(src).to(dst)

    # ---------------------------------------------------------------- temp slots
    def set_temps(self, slots):
        self.free_tmp = list(slots)
        self.temp_keys = frozenset(self.key(t) for t in slots)

    def tmp(self):
        return self.free_tmp.pop(0)

    def rel(self, *slots):
        for s in slots:
            assert s not in self.free_tmp
            self.free_tmp.insert(0, s)

    # ---- operand blocks H0..H3 of the three-term multiply (home registers 0..3 used as raw blocks)
    MUL3_SCRATCH = (6, 7, 8)          # home blocks the three-term multiply (L1 mul3: a Karatsuba pass) uses as scratch

    def reserve_blocks(self, count=4, scratch=()):
        """home blocks 0..count-1 become raw operand blocks, `scratch` blocks are handed to the L1 routine: neither may hold
        a temporary of this program until release_blocks()"""
        self._saved_tmp = self.free_tmp
        taken = lambda t: t.kind == "home" and (t.idx < count or t.idx in scratch)
        self._held = [t for t in self.free_tmp if taken(t)]
        self.free_tmp = [t for t in self.free_tmp if not taken(t)]
        assert len(self._held) == count + len(scratch), f"home blocks 0..{count - 1} and {scratch} must be free"
        self._scratch_reserved = tuple(scratch)
        self.tagH = [None] * 4
        self.eH = [None] * 4
        self.vH = [V_STORE] * 4
        self._dy_for = {"B": None, 1: None, 3: None}       # which operand the kept y-side differences of mul3 were formed from
        self._blocks_reserved = True

    def release_blocks(self):
        self.free_tmp = self._held + self.free_tmp
        self._blocks_reserved = False
        self._scratch_reserved = ()

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
        self._need(2 * NL * worst <= COL_BUDGET, f"mul3 {worst}")
        vs = (self.vA, self.v_of(y), *self.vH)
        self._need(max(vs) <= V_CAP, f"mul3 operand value {vs}")
        # L1v4.kfips forms c1 - c0 / c0 - c1 of every operand limb by limb: below 2^31 as long as the limbs stay under 3.9 units
        self._need(max(mag(rA), mag(self.r_of(y)), *(mag(h) for h in self.eH)) <= 3.9, "mul3 operand limbs")
        assert getattr(self, "_scratch_reserved", ()) == self.MUL3_SCRATCH, "mul3 needs home blocks 6, 7, 8 as scratch"
        self.vA = 2 * (self.vA * self.v_of(y) + self.vH[0] * self.vH[1] + self.vH[2] * self.vH[3]) / K_RP + 0.5
        if MUL3_KEEP_DY:               # the y-side difference vectors live in the routine's scratch across passes: (re)formed when an operand changed
            for w, tag in (("B", self.tagB), (1, self.tagH[1]), (3, self.tagH[3])):
                assert tag is not None
                if self._dy_for[w] is not tag:
                    self.wait()
                    L1v4(self.e).mul3_dy(w)
                    self._dy_for[w] = tag
                    self._count("mul3_dy")
        self._raw_call("mul3")
        self.rA = self.r_norm()
        return self                                  # (all five other operands survive)

    def mul2a(self):
        """A <- A + H0*H1 + H2*H3 (the value in A enters the pass's upper half: no product for it, no separate addition pass)"""
        rA = self.rA if self.rA is not None else self.UNKNOWN
        worst = mag(self.eH[0]) * mag(self.eH[1]) + mag(self.eH[2]) * mag(self.eH[3])
        self._need(2 * NL * worst + mag(rA) / 256.0 <= COL_BUDGET, f"mul2a {worst}")
        vs = (self.vA, *self.vH[:4])
        self._need(max(vs) <= V_CAP, f"mul2a operand value {vs}")
        self._need(max(mag(rA), *(mag(h) for h in self.eH[:4])) <= 3.9, "mul2a operand limbs")
        assert getattr(self, "_scratch_reserved", ()) == self.MUL3_SCRATCH, "mul2a needs home blocks 6, 7, 8 as scratch"
        self.vA = self.vA + 2 * (self.vH[0] * self.vH[1] + self.vH[2] * self.vH[3]) / K_RP + 0.5
        if MUL3_KEEP_DY:
            for w, tag in ((1, self.tagH[1]), (3, self.tagH[3])):
                assert tag is not None
                if self._dy_for[w] is not tag:
                    self.wait()
                    L1v4(self.e).mul3_dy(w)
                    self._dy_for[w] = tag
                    self._count("mul3_dy")
        self._raw_call("mul2a")
        self.rA = self.r_norm()
        return self

    # ================================================================ L2 algorithms: Fq6 / Fq12
    def _load_norm_sum(self, blk, s1, s2):
        """register block blk <- s1 (+ s2), NORMALISED limbs (the fused routines take one-unit operands)."""
        m_ = mag(self.r_of(s1)) + (mag(self.r_of(s2)) if s2 is not None else 0.0)
        v_ = self.v_of(s1) + (self.v_of(s2) if s2 is not None else 0.0)
        if m_ > 1.0 or v_ > K_TOP:
            self.A(s1)
            if s2 is not None:
                self.add(s2)
            self.norm()
            self.wait()
            if blk != A0:
                for i in range(SLOT_DW):
                    self.e.emit(f"v_mov_b32_e32 v{blk + i}, v{A0 + i}", vw=[blk + i])
            self.tagA = None
            return self.vA
        assert s2 is None
        if blk == A0:
            self.A(s1)
        elif blk == B0:
            self._B(s1)
        else:
            self.load(blk, s1)
        return v_

    def _load_sum(self, blk, s1, s2, limit):
        """register block blk <- s1 (+ s2); limbs of at most `limit` units: the raw limb-wise sum when it fits, normalised
        otherwise.  Returns (limb magnitude, value bound)."""
        m_ = mag(self.r_of(s1)) + (mag(self.r_of(s2)) if s2 is not None else 0.0)
        v_ = self.v_of(s1) + (self.v_of(s2) if s2 is not None else 0.0)
        if m_ > limit or v_ > K_TOP:
            v_ = self._load_norm_sum(blk, s1, s2)
            return 1.0, v_
        if s2 is None:
            if blk == A0:
                self.A(s1)
            elif blk == B0:
                self._B(s1)
            else:
                self.load(blk, s1)
            return m_, v_
        self.load(blk, s1)
        self.load(A0, s2)
        self.tagA = None
        self.wait()
        for i in range(SLOT_DW):
            self.e.emit(f"v_add_u32_e32 v{blk + i}, v{blk + i}, v{A0 + i}", vw=[blk + i])
        return m_, v_

    def _mul6_regs(self, a, b, a_plus=None, b_plus=None):
        """Fused Fq6 multiplication (L1 mul6: schoolbook over Fq2, one dual column pass per output coefficient with the a1 / a2
        products as Karatsuba products) of (a + a_plus) by (b + b_plus), coefficient-wise sums formed while the operands are
        loaded into the home blocks: the a side may stay an unnormalised sum (two units), the b side is normalised.
        Returns the three result 'slots' [c0, c1, c2]: c0 = HOME(6), c1 = HOME(2), c2 = block A (None), all normalised; home
        blocks 1, 2, 6, 7, block A and block B are clobbered."""
        ma = mb = va = vb = 0.0
        for k, s_ in enumerate(b):
            m_, v_ = self._load_sum(HOME0 + SLOT_DW * (3 + k), s_, b_plus[k] if b_plus else None, 1.0)
            mb, vb = max(mb, m_), max(vb, v_)
        for k, s_ in enumerate(a):
            m_, v_ = self._load_sum(HOME0 + SLOT_DW * k, s_, a_plus[k] if a_plus else None, 2.0)
            ma, va = max(ma, m_), max(va, v_)
        # worst column: three Fq2 products of ma x mb limbs (54 limb products) + nine reduction products.  (Inside a pass the
        # Karatsuba difference products may carry the imaginary accumulator beyond 64 bits before U and W cancel them: sums are
        # exact mod 2^64 and the value that is finally shifted out is the true one -- tools/ksim.py checks exactly that.)
        self._need(2 * NL * 3 * ma * mb <= COL_BUDGET, f"mul6 operand limbs {ma} {mb}")
        self._need(10 * va <= V_CAP and vb <= V_CAP, f"mul6 operand values {va} {vb}")
        self._raw_call("mul6")
        res = [HOME(6, "mul6.c0"), HOME(2, "mul6.c1"), None]
        v0 = 42 * va * vb / K_RP + 0.5            # c0 = a0 b0 + (xi a1) b2 + (xi a2) b1: 2 (1 + 10 + 10) va vb / K + 1/2
        v1 = 24 * va * vb / K_RP + 0.5            # c1 = a0 b1 + a1 b0 + (xi a2) b2
        v2 = 6 * va * vb / K_RP + 0.5             # c2 = a0 b2 + a1 b1 + a2 b0
        for slot, v in ((res[0], v0), (res[1], v1)):
            self._need(v <= V_CAP, f"mul6 result value {v}")
            self.slot_r[self.key(slot)] = self.r_norm(v)
            self.slot_v[self.key(slot)] = v
            self.max_v = max(self.max_v, v)
        self.vA = v2
        self.rA = self.r_norm()
        self.tagA = self.tagB = None
        return res

    def fq6_mul(self, a, b, out, a_plus=None, b_plus=None):
        """(a0, a1, a2)(b0, b1, b2) in Fq2[v]/(v^3 - xi): ONE fused L1 routine (mul6: operands in the home blocks) when the
        routine keeps no temporary in the home registers; six calls plus glue otherwise."""
        if not self.homes_free:
            assert a_plus is None and b_plus is None
            return self._fq6_mul_generic(a, b, out)
        res = self._mul6_regs(a, b, a_plus, b_plus)
        self.to(out[2])                                       # c2 sits in block A
        self.A(res[0]).to(out[0])
        self.A(res[1]).to(out[1])

    def _fq6_mul_generic(self, a, b, out):
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

    def fq12_mul(self, F, Bs, conj_b=False):
        """F <- F * B (Karatsuba over Fq6: T0 = A0 B0, T1 = A1 B1, M = (A0 + A1)(B0 + B1)) on the fused Fq6 multiplication; B is
        not modified.  THREE temporaries: M is formed first (its sums while the operands are loaded); T0 then goes straight to
        the places of A0 -- dead from there on -- and leaves M; T1 completes both halves from the registers:
            F0 = T0.0 + xi T1.2   F2 = T0.1 + T1.0   F4 = T0.2 + T1.1        F1, F3, F5 = M - T0 - T1
        (Six temporaries -- T0 and T1 held until the end -- cost the final exponentiation its last free on-chip slots: with three,
        the eight LDS slots hold a whole Fq12 register next to f and the multiplication operand.)"""
        if conj_b:
            for k in (1, 3, 5):
                self.A(Bs[k]).neg().to(Bs[k])
        assert self.homes_free, "fq12_mul runs on the fused Fq6 multiplication (home blocks 0..7 must be free)"
        if self.FUSED_GLUE and all(s_.kind != "home" for s_ in list(F) + list(Bs)):
            return self._fq12_mul_fused(F, Bs)
        A_0, A_1 = [F[0], F[2], F[4]], [F[1], F[3], F[5]]
        B_0, B_1 = [Bs[0], Bs[2], Bs[4]], [Bs[1], Bs[3], Bs[5]]
        M = [self.tmp() for _ in range(3)]
        res = self._mul6_regs(A_0, B_0, A_1, B_1)                 # (A0 + A1)(B0 + B1): c2 in block A
        self.to(M[2])
        self.A(res[0]).to(M[0])
        self.A(res[1]).to(M[1])
        res = self._mul6_regs(A_0, B_0)                           # T0; A0 = F0, F2, F4 is dead now
        self.to(F[4]).rsub(M[2]).to(M[2])                         # F4 <- T0.2 ; M.2 -= T0.2
        self.A(res[0]).to(F[0])
        self.A(M[0]).sub(res[0]).to(M[0])
        self.A(res[1]).to(F[2])
        self.A(M[1]).sub(res[1]).to(M[1])
        res = self._mul6_regs(A_1, B_1)                           # T1; A1 = F1, F3, F5 is dead now
        self.to(F[5])                                             # T1.2 parked in the place it finally leaves through
        self.mulxi().add(F[0]).to(F[0])
        self.A(M[2]).sub(F[5]).to(F[5])
        self.A(res[0]).add(F[2]).to(F[2])
        self.A(M[0]).sub(res[0]).to(F[1])
        self.A(res[1]).add(F[4]).to(F[4])
        self.A(M[1]).sub(res[1]).to(F[3])
        self.rel(*M)

    # ---- register-level glue around the fused Fq6 multiplication (FUSED_GLUE): the operands of the second product are built
    # from what the first one left in the home blocks (a0 and b survive mul6), sums that must be normalised come out of ONE 64-bit
    # chain each, the recombination runs on registers and every result is stored once.
    def _mul6_call(self, ma, mb, va, vb):
        """the fused Fq6 multiplication on operands that are already in home blocks 0..2 (limbs of ma units, value va) and 3..5
        (mb, vb); returns the value bounds of c0 (home 6), c1 (home 2), c2 (block A), all normalised"""
        self._need(2 * NL * 3 * ma * mb <= COL_BUDGET, f"mul6 operand limbs {ma} {mb}")
        self._need(10 * va <= V_CAP and vb <= V_CAP, f"mul6 operand values {va} {vb}")
        self.wait()
        self._raw_call("mul6")
        vs = (42 * va * vb / K_RP + 0.5, 24 * va * vb / K_RP + 0.5, 6 * va * vb / K_RP + 0.5)
        for v in vs:
            self._need(v <= V_CAP, f"mul6 result value {v}")
            self.max_v = max(self.max_v, v)
        self.tagA = self.tagB = None
        return vs

    def _blk_sum(self, dst, terms, v):
        """register block dst <- sum of coef * (register block) over terms, NORMALISED on one 64-bit chain per component; reduced as
        well when the value bound v exceeds what a normalising store would keep (cf. norm()).  Returns the value bound left."""
        g = L1v4(self.e)
        d = g.fq2(dst)
        red = v > V_REDN_AT
        self._need(v <= 8 * V_CAP, f"chain on a value of {v} p")
        g.lincomb([d[0], d[1]], [[(c, g.fq2(b)[0]) for c, b in terms], [(c, g.fq2(b)[1]) for c, b in terms]], reduce=red)
        return 0.51 if red else v

    def _blk_store(self, blk, dst, limbs, v):
        """dst <- register block blk (limbs of `limbs` units, value bound v): normalised / reduced first exactly where to() would"""
        g = L1v4(self.e)
        b = g.fq2(blk)
        lim = 1.0 if self.key(dst) in self.norm_keys else STORE_MAG
        need_norm = limbs > max(lim, v / K_TOP)
        if v > self.v_limit(dst) or (need_norm and (v > V_REDN_AT or limbs > LIMB_MAG)):
            self._need(v <= 8 * V_CAP and limbs <= 7.9, f"redn of {v} p, limbs {limbs}")
            g.lincomb([b[0], b[1]], [[(1, b[0])], [(1, b[1])]], reduce=True)
            limbs, v = 1.0, 0.51
            self._count("redn")
        elif need_norm:
            self._need(limbs <= LIMB_MAG, f"norm of limbs {limbs}")
            g.norm_limbs(b[0])
            g.norm_limbs(b[1])
            limbs = 1.0
            self._count("norm")
        self.wait()
        self.store(blk, dst)
        self._need(v <= V_CAP, f"value bound {v} p at store")
        self.slot_r[self.key(dst)] = self.r_norm(v) if limbs <= 1.0 else (-limbs, limbs)
        self.slot_v[self.key(dst)] = v
        self.max_v = max(self.max_v, v)
        if self.tagA is dst:
            self.tagA = None
        if self.tagB is dst:
            self.tagB = None

    def _lw_blocks(self, op, dst, a, b):
        for i in range(SLOT_DW):
            self.e.emit(f"{op} v{dst + i}, v{a + i}, v{b + i}", vw=[dst + i])

    def _fq12_sqr_fused(self, F):
        Hb = lambda k: HOME0 + SLOT_DW * k
        for s_ in F:
            self._need(mag(self.r_of(s_)) <= 1.0, f"fq12_sqr: {s_} is not normalised")
        v = max(self.v_of(s_) for s_ in F)
        T = [self.tmp() for _ in range(3)]
        self.tagA = self.tagB = None
        # t = A0 A1
        for k, i in enumerate((0, 2, 4)):
            self.load(Hb(k), F[i])
        for k, i in enumerate((1, 3, 5)):
            self.load(Hb(3 + k), F[i])
        t0, t1, t2 = self._mul6_call(1.0, 1.0, v, v)              # t0 home 6, t1 home 2, t2 block A; home 0 = F0, homes 3..5 = F1, F3, F5 survive
        for blk, dst, tv in ((Hb(6), T[0], t0), (Hb(2), T[1], t1), (A0, T[2], t2)):
            self._blk_store(blk, dst, 1.0, tv)
        # u = (A0 + A1)(A0 + v A1): a' = (F0 + F1, F2 + F3, F4 + F5) raw sums, b' = (F0 + xi F5, F2 + F1, F4 + F3) off the chains
        self.load(Hb(1), F[2])
        self.load(Hb(2), F[4])
        self.wait()
        g = L1v4(self.e)
        h0, h5 = g.fq2(Hb(0)), g.fq2(Hb(5))
        a_ = g.fq2(A0)
        v_s0 = 11 * v                                             # F0 + xi F5 = (F0.0 + 9 F5.0 - F5.1, F0.1 + 9 F5.1 + F5.0)
        red0 = v_s0 > V_REDN_AT
        g.lincomb([a_[0], a_[1]], [[(1, h0[0]), (9, h5[0]), (-1, h5[1])], [(1, h0[1]), (9, h5[1]), (1, h5[0])]], reduce=red0)
        vb0 = 0.51 if red0 else v_s0
        vb1 = self._blk_sum(Hb(6), [(1, Hb(1)), (1, Hb(3))], 2 * v)       # F2 + F1
        vb2 = self._blk_sum(B0, [(1, Hb(2)), (1, Hb(4))], 2 * v)          # F4 + F3
        for k in range(3):
            self._lw_blocks("v_add_u32_e32", Hb(k), Hb(k), Hb(3 + k))     # a' (two units)
        for dst, src in ((Hb(3), A0), (Hb(4), Hb(6)), (Hb(5), B0)):
            for i in range(SLOT_DW):
                self.e.emit(f"v_mov_b32_e32 v{dst + i}, v{src + i}", vw=[dst + i])
        u0, u1, u2 = self._mul6_call(2.0, 1.0, 2 * v, max(vb0, vb1, vb2))
        # recombination on registers: t0, t1, t2 -> home blocks 3, 4, 5
        for k in range(3):
            self.load(Hb(3 + k), T[k])
        self.wait()
        self._lw_blocks("v_sub_u32_e32", A0, A0, Hb(5))
        self._lw_blocks("v_sub_u32_e32", A0, A0, Hb(4))
        self._blk_store(A0, F[4], 3.0, u2 + t2 + t1)                      # F4 = u2 - t2 - t1
        h7 = g.fq2(Hb(7))
        vx = 10 * t2                                                      # X = xi t2 -> home 7 (normalised; reduced when large)
        redx = vx > V_REDN_AT
        g2 = L1v4(self.e)
        g2.lincomb([h7[0], h7[1]], [[(9, h5[0]), (-1, h5[1])], [(9, h5[1]), (1, h5[0])]], reduce=redx)
        vx = 0.51 if redx else vx
        self._lw_blocks("v_sub_u32_e32", Hb(6), Hb(6), Hb(3))
        self._lw_blocks("v_sub_u32_e32", Hb(6), Hb(6), Hb(7))
        self._blk_store(Hb(6), F[0], 3.0, u0 + t0 + vx)                   # F0 = u0 - t0 - xi t2
        self._lw_blocks("v_sub_u32_e32", Hb(2), Hb(2), Hb(4))
        self._lw_blocks("v_sub_u32_e32", Hb(2), Hb(2), Hb(3))
        self._blk_store(Hb(2), F[2], 3.0, u1 + t1 + t0)                   # F2 = u1 - t1 - t0
        for k, (dst, tv) in enumerate(((F[1], t0), (F[3], t1), (F[5], t2))):
            for i in range(SLOT_DW):
                r = Hb(3 + k) + i
                self.e.emit(f"v_lshlrev_b32_e32 v{r}, 1, v{r}", vw=[r])
            self._blk_store(Hb(3 + k), dst, 2.0, 2 * tv)                  # F1, F3, F5 = 2 t
        self.rel(*T)
        self.tagA = self.tagB = None

    def _fq12_mul_fused(self, F, Bs):
        """fq12_mul with register-level glue: operands go straight into the home blocks of the fused Fq6 multiplication (sums formed
        there), results leave straight from the blocks it fills, the last recombination runs on registers."""
        Hb = lambda k: HOME0 + SLOT_DW * k
        for s_ in list(F) + list(Bs):
            self._need(mag(self.r_of(s_)) <= 1.0, f"fq12_mul: {s_} is not normalised")
        va = max(self.v_of(s_) for s_ in F)
        vb = max(self.v_of(s_) for s_ in Bs)
        M = [self.tmp() for _ in range(3)]
        self.tagA = self.tagB = None
        g = L1v4(self.e)
        # M = (A0 + A1)(B0 + B1): a' raw sums (two units), b' normalised (reduced when the bound asks for it)
        for k in range(3):
            self.load(Hb(k), F[2 * k])
            self.load(Hb(3 + k), Bs[2 * k])
        vbs = []
        for k in range(3):
            self.load(A0, F[2 * k + 1])
            self.load(B0, Bs[2 * k + 1])
            self.wait()
            self._lw_blocks("v_add_u32_e32", Hb(k), Hb(k), A0)
            self._lw_blocks("v_add_u32_e32", Hb(3 + k), Hb(3 + k), B0)
            b = g.fq2(Hb(3 + k))
            if 2 * vb > V_REDN_AT:
                g.lincomb([b[0], b[1]], [[(1, b[0])], [(1, b[1])]], reduce=True)
                vbs.append(0.51)
            else:
                g.norm_limbs(b[0])
                g.norm_limbs(b[1])
                vbs.append(2 * vb)
        m0, m1, m2 = self._mul6_call(2.0, 1.0, 2 * va, max(vbs))
        for blk, dst, mv in ((Hb(6), M[0], m0), (Hb(2), M[1], m1), (A0, M[2], m2)):
            self._blk_store(blk, dst, 1.0, mv)
        # T0 = A0 B0 -> straight into the places of A0 (dead from here on)
        for k in range(3):
            self.load(Hb(k), F[2 * k])
            self.load(Hb(3 + k), Bs[2 * k])
        t0 = self._mul6_call(1.0, 1.0, va, vb)
        for blk, dst, tv in ((Hb(6), F[0], t0[0]), (Hb(2), F[2], t0[1]), (A0, F[4], t0[2])):
            self._blk_store(blk, dst, 1.0, tv)
        t0 = [self.slot_v[self.key(F[i])] for i in (0, 2, 4)]         # (a store may have reduced its value)
        # T1 = A1 B1, then everything else on registers: T1 in home 6, home 2, block A
        for k in range(3):
            self.load(Hb(k), F[2 * k + 1])
            self.load(Hb(3 + k), Bs[2 * k + 1])
        t1 = self._mul6_call(1.0, 1.0, va, vb)
        for k in range(3):
            self.load(Hb(3 + k), F[2 * k])                            # T0.0, T0.1, T0.2
        for blk, src in ((Hb(0), M[0]), (Hb(1), M[1]), (Hb(7), M[2])):
            self.load(blk, src)
        self.wait()
        mv = [self.slot_v[self.key(m_)] for m_ in M]
        # F0 = T0.0 + xi T1.2 (one chain, into block B)
        h3, a_, bb = g.fq2(Hb(3)), g.fq2(A0), g.fq2(B0)
        v0 = t0[0] + 10 * t1[2]
        red = v0 > V_REDN_AT
        g.lincomb([bb[0], bb[1]], [[(1, h3[0]), (9, a_[0]), (-1, a_[1])], [(1, h3[1]), (9, a_[1]), (1, a_[0])]], reduce=red)
        self._blk_store(B0, F[0], 1.0, 0.51 if red else v0)
        self._lw_blocks("v_sub_u32_e32", Hb(0), Hb(0), Hb(3))
        self._lw_blocks("v_sub_u32_e32", Hb(0), Hb(0), Hb(6))
        self._blk_store(Hb(0), F[1], 3.0, mv[0] + t0[0] + t1[0])      # F1 = M0 - T0.0 - T1.0
        self._lw_blocks("v_add_u32_e32", Hb(6), Hb(6), Hb(4))
        self._blk_store(Hb(6), F[2], 2.0, t0[1] + t1[0])              # F2 = T0.1 + T1.0
        self._lw_blocks("v_sub_u32_e32", Hb(1), Hb(1), Hb(4))
        self._lw_blocks("v_sub_u32_e32", Hb(1), Hb(1), Hb(2))
        self._blk_store(Hb(1), F[3], 3.0, mv[1] + t0[1] + t1[1])      # F3 = M1 - T0.1 - T1.1
        self._lw_blocks("v_add_u32_e32", Hb(2), Hb(2), Hb(5))
        self._blk_store(Hb(2), F[4], 2.0, t0[2] + t1[1])              # F4 = T0.2 + T1.1
        self._lw_blocks("v_sub_u32_e32", Hb(7), Hb(7), Hb(5))
        self._lw_blocks("v_sub_u32_e32", Hb(7), Hb(7), A0)
        self._blk_store(Hb(7), F[5], 3.0, mv[2] + t0[2] + t1[2])      # F5 = M2 - T0.2 - T1.2
        self.rel(*M)
        self.tagA = self.tagB = None

    FUSED_GLUE = bool(int(os.environ.get("KGEN_FUSED_GLUE", "1")))

    def fq12_sqr(self, F):
        """F <- F^2 (complex squaring over Fq6: t = A0 A1, u = (A0 + A1)(A0 + v A1))."""
        if self.FUSED_GLUE and self.homes_free and all(s_.kind != "home" for s_ in F):
            return self._fq12_sqr_fused(F)
        A_0, A_1 = [F[0], F[2], F[4]], [F[1], F[3], F[5]]
        T = [self.tmp() for _ in range(3)]
        S0 = self.tmp()
        assert self.homes_free, "fq12_sqr runs on the fused Fq6 multiplication (home blocks 0..7 must be free)"
        if True:
            self.fq6_mul(A_0, A_1, T)
            self.A(F[5]).mulxi().add(F[0]).to(S0)                 # first coefficient of A0 + v A1
            U = self._mul6_regs(A_0, [S0, F[2], F[4]], A_1, [None, F[1], F[3]])
            self.sub(T[2]).sub(T[1]).to(F[4])                     # u2 - t2 - t1   (u2 is in block A)
            X = S0
            self.A(T[2]).mulxi().to(X)
            self.A(U[0]).sub(T[0]).sub(X).to(F[0])
            self.A(U[1]).sub(T[1]).sub(T[0]).to(F[2])
        self.A(T[0]).dbl().to(F[1])
        self.A(T[1]).dbl().to(F[3])
        self.A(T[2]).dbl().to(F[5])
        self.rel(S0, *T)

    # ================================================================ sparse multiplications (miller_loop_native.rs:46-110)
    def mul_by_034(self, F, L0, L3, L4, between=None):
        """f *= L0 + L3 w^3 + L4 w^4 with one reduction per output coefficient (xi folded into the line).
        between(i): the caller's code behind pass i (the multi-pairing kernels spread their prefetch loads there)."""
        between = between or (lambda i: None)
        self.marker("mul034")
        if L3.kind == "home":
            return self._mul_by_034_regs(F, L0, L3, L4, between)
        self.reserve_blocks(scratch=self.MUL3_SCRATCH)
        L3x, L4x = self.tmp(), self.tmp()
        self.A(L3).mulxi().to(L3x)
        self.A(L4).mulxi().to(L4x)
        c = [self.tmp() for _ in range(3)]
        # c0 = a0 L0 + a3 xiL3 + a2 xiL4 ; c1 = a1 L0 + a4 xiL3 + a3 xiL4 ; c2 = a2 L0 + a5 xiL3 + a4 xiL4
        for k, (i0, i3, i4) in enumerate(((0, 3, 2), (1, 4, 3), (2, 5, 4))):
            self.ldH(1, L3x).ldH(3, L4x).ldH(0, F[i3]).ldH(2, F[i4])
            self.A(F[i0]).mul3(L0).to(c[k])
            between(k)
        # c3 = a3 L0 + a0 L3 + a5 xiL4   (a3 is not read again: the result goes straight to its place)
        self.ldH(1, L3).ldH(3, L4x).ldH(0, F[0]).ldH(2, F[5])
        self.A(F[3]).mul3(L0).to(F[3])
        between(3)
        # c4 = a4 L0 + a1 L3 + a0 L4 ; c5 = a5 L0 + a2 L3 + a1 L4
        self.ldH(1, L3).ldH(3, L4).ldH(0, F[1]).ldH(2, F[0])
        self.A(F[4]).mul3(L0).to(F[4])
        between(4)
        self.ldH(0, F[2]).ldH(2, F[1])
        self.A(F[5]).mul3(L0).to(F[5])
        for k in range(3):
            self.mov(F[k], c[k])
        self.rel(L3x, L4x, *c)
        self.release_blocks()

    def mul_by_034_one(self, F, L3, L4):
        """f *= 1 + L3 w^3 + L4 w^4: the line's constant coefficient is ONE (a table line of a fixed G2 point, divided by its own constant coefficient
        when the table was made), so every output is its own input plus TWO products -- six mul2a passes, 12 Fq2 products instead of 18.
            c0 = a0 + a3 xiL3 + a2 xiL4   c1 = a1 + a4 xiL3 + a3 xiL4   c2 = a2 + a5 xiL3 + a4 xiL4
            c3 = a3 + a0 L3 + a5 xiL4     c4 = a4 + a1 L3 + a0 L4       c5 = a5 + a2 L3 + a1 L4"""
        self.marker("mul034one")
        self.reserve_blocks(scratch=self.MUL3_SCRATCH)
        L3x, L4x = self.tmp(), self.tmp()
        self.A(L3).mulxi().to(L3x)
        self.A(L4).mulxi().to(L4x)
        c = [self.tmp() for _ in range(3)]
        for k, (i0, i3, i4) in enumerate(((0, 3, 2), (1, 4, 3), (2, 5, 4))):
            self.ldH(1, L3x).ldH(3, L4x).ldH(0, F[i3]).ldH(2, F[i4])
            self.A(F[i0]).mul2a().to(c[k])
        self.ldH(1, L3).ldH(3, L4x).ldH(0, F[0]).ldH(2, F[5])
        self.A(F[3]).mul2a().to(F[3])
        self.ldH(1, L3).ldH(3, L4).ldH(0, F[1]).ldH(2, F[0])
        self.A(F[4]).mul2a().to(F[4])
        self.ldH(0, F[2]).ldH(2, F[1])
        self.A(F[5]).mul2a().to(F[5])
        for k in range(3):
            self.mov(F[k], c[k])
        self.rel(L3x, L4x, *c)
        self.release_blocks()

    # ---- the same two multiplications with the line where the fused point step left it: La in home block 7, Lb in home block 4,
    # Lc in home block 5 (LINE_REGS).  La goes to block B once (all six passes use it), Lb / Lc stay parked in home blocks 4 / 5 --
    # the three-term multiply does not touch them -- and are copied, or multiplied by xi on the way, straight into the operand
    # blocks 1 / 3 as the passes need them.  No slot is written for the line; the three result temporaries are the only ones.
    LINE_REGS = (HOME(7, "La"), HOME(4, "Lb"), HOME(5, "Lc"))

    def _hold_line(self, La, Lb, Lc):
        assert (La.idx, Lb.idx, Lc.idx) == (7, 4, 5)
        self._B(La)                                               # block B <- La (home block 7 becomes scratch of the multiply)
        self.wait()
        held = [t for t in self.free_tmp if t.kind == "home" and t.idx in (4, 5)]
        assert len(held) == 2, "home blocks 4, 5 must be free temporaries of the routine"
        self.free_tmp = [t for t in self.free_tmp if t not in held]
        self.reserve_blocks(scratch=self.MUL3_SCRATCH)
        return held

    def _xi_into(self, k, src, tag):
        """home block k <- xi * src (src: a parked home slot), normalised, on the 64-bit chains; no slot is touched"""
        self._need(mag(self.r_of(src)) <= 7.9, f"xi of {src}")
        v = 10 * self.v_of(src)
        self._need(v <= V_CAP, f"xi {src}: value {v}")
        self.wait()
        g = L1v4(self.e)
        d, s_ = g.fq2(HOME0 + SLOT_DW * k), g.fq2(HOME0 + SLOT_DW * src.idx)
        g.lincomb([d[0], d[1]], [[(9, s_[0]), (-1, s_[1])], [(9, s_[1]), (1, s_[0])]])
        self.tagH[k], self.eH[k], self.vH[k] = tag, self.r_norm(v), v
        self._count("xi_into")

    def _mul_by_034_regs(self, F, L0, L3, L4, between):
        held = self._hold_line(L0, L3, L4)
        L3x, L4x = Slot("regs", 1, "xiLb"), Slot("regs", 3, "xiLc")
        self._xi_into(1, L3, L3x)
        self._xi_into(3, L4, L4x)
        c = [self.tmp() for _ in range(3)]
        for k, (i0, i3, i4) in enumerate(((0, 3, 2), (1, 4, 3), (2, 5, 4))):
            self.ldH(0, F[i3]).ldH(2, F[i4])
            self.A(F[i0]).mul3(L0).to(c[k])
            between(k)
        self.ldH(1, L3).ldH(0, F[0]).ldH(2, F[5])                   # c3 = a3 L0 + a0 L3 + a5 xiL4
        self.A(F[3]).mul3(L0).to(F[3])
        between(3)
        self.ldH(3, L4).ldH(0, F[1]).ldH(2, F[0])                   # c4 = a4 L0 + a1 L3 + a0 L4
        self.A(F[4]).mul3(L0).to(F[4])
        between(4)
        self.ldH(0, F[2]).ldH(2, F[1])                              # c5 = a5 L0 + a2 L3 + a1 L4
        self.A(F[5]).mul3(L0).to(F[5])
        for k in range(3):
            self.mov(F[k], c[k])
        self.rel(*c)
        self.release_blocks()
        self.free_tmp = held + self.free_tmp

    def _mul_by_235_regs(self, F, L2, L3, L5, between):
        held = self._hold_line(L2, L3, L5)
        L3x, L5x = Slot("regs", 1, "xiLb"), Slot("regs", 3, "xiLc")
        t0, t1, t2 = self.tmp(), self.tmp(), self.tmp()
        for dst, (i2, i3, i5) in ((t0, (4, 3, 1)), (t1, (5, 4, 2))):                 # c0, c1 = xi (a b2 + a' b3 + a'' b5)
            self.ldH(1, L3).ldH(3, L5).ldH(0, F[i3]).ldH(2, F[i5])
            self.A(F[i2]).mul3(L2).mulxi(reduce=True).to(dst)
            between(0 if dst is t0 else 1)
        self._xi_into(3, L5, L5x)
        self.ldH(0, F[0]).ldH(2, F[4])                                               # c3 = a1 b2 + a0 b3 + xi a4 b5
        self.A(F[1]).mul3(L2).to(t2)
        between(2)
        self.ldH(0, F[1]).ldH(2, F[5])                                               # c4 -> its place
        self.A(F[2]).mul3(L2).to(F[4])
        between(3)
        self.mov(F[1], t1)
        self._xi_into(1, L3, L3x)
        self.ldH(0, F[5]).ldH(2, F[3])                                               # c2 = a0 b2 + xi (a5 b3 + a3 b5), into the freed temporary
        self.A(F[0]).mul3(L2).to(t1)
        between(4)
        self.ldH(1, L3).ldH(3, L5).ldH(0, F[2]).ldH(2, F[0])                         # c5 -> its place
        self.A(F[3]).mul3(L2).to(F[5])
        self.mov(F[0], t0)
        self.mov(F[2], t1)
        self.mov(F[3], t2)
        self.rel(t0, t1, t2)
        self.release_blocks()
        self.free_tmp = held + self.free_tmp

    def mul_by_235(self, F, L2, L3, L5, between=None):
        """f *= L2 w^2 + L3 w^3 + L5 w^5, same scheme.  Five temporaries (two xi-multiplied line coefficients, three results
        that wait for their place): the order below frees the places as early as possible, and the two coefficients that are
        xi times a plain sum take the xi afterwards.
            c0 = xi (a4 b2 + a3 b3 + a1 b5)   c1 = xi (a5 b2 + a4 b3 + a2 b5)   c2 = a0 b2 + xi (a5 b3 + a3 b5)
            c3 = a1 b2 + a0 b3 + xi a4 b5     c4 = a2 b2 + a1 b3 + xi a5 b5     c5 = a3 b2 + a2 b3 + a0 b5"""
        between = between or (lambda i: None)
        self.marker("mul235")
        if L3.kind == "home":
            return self._mul_by_235_regs(F, L2, L3, L5, between)
        self.reserve_blocks(scratch=self.MUL3_SCRATCH)
        L3x, L5x = self.tmp(), self.tmp()
        self.A(L3).mulxi().to(L3x)
        self.A(L5).mulxi().to(L5x)
        t0, t1, t2 = self.tmp(), self.tmp(), self.tmp()
        for dst, (i2, i3, i5) in ((t0, (4, 3, 1)), (t1, (5, 4, 2))):                 # c0, c1 (reduced after the xi: they are stored)
            self.ldH(1, L3).ldH(3, L5).ldH(0, F[i3]).ldH(2, F[i5])
            self.A(F[i2]).mul3(L2).mulxi(reduce=True).to(dst)
            between(0 if dst is t0 else 1)
        self.ldH(1, L3).ldH(3, L5x).ldH(0, F[0]).ldH(2, F[4])                        # c3: a4 has now been read by c0, c1, c3
        self.A(F[1]).mul3(L2).to(t2)
        between(2)
        self.ldH(1, L3).ldH(3, L5x).ldH(0, F[1]).ldH(2, F[5])                        # c4 -> its place; a1 read by c0, c3, c4
        self.A(F[2]).mul3(L2).to(F[4])
        between(3)
        self.mov(F[1], t1)
        self.ldH(1, L3x).ldH(3, L5x).ldH(0, F[5]).ldH(2, F[3])                       # c2 (into the freed temporary); a5 read by c1, c2, c4
        self.A(F[0]).mul3(L2).to(t1)
        between(4)
        self.ldH(1, L3).ldH(3, L5).ldH(0, F[2]).ldH(2, F[0])                         # c5 -> its place
        self.A(F[3]).mul3(L2).to(F[5])
        self.mov(F[0], t0)
        self.mov(F[2], t1)
        self.mov(F[3], t2)
        self.rel(L3x, L5x, t0, t1, t2)
        self.release_blocks()

    # ================================================================ G2 steps (homogeneous projective, no inversions)
    # The reference steps in affine coordinates with one Fq2 inversion each (miller_loop_native.rs:157,167,186); the
    # projective line values are the reference's un-normalised affine line values (:10-44) times a known Fq2 factor (Z^2 for
    # tangents, Z for chords), tracked in `scale` by the kernels that must return the exact miller_loop_native value.
    # ---- fused point steps (L1 dblstep / addstep: everything in named register blocks, no marshalling between the ~20 field
    # operations of a step); used when all nine home blocks are this routine's temporaries
    FUSED_STEPS = bool(int(os.environ.get("KGEN_FUSED_STEPS", "1")))

    def _fused_ok(self):
        homes = {t.idx for t in self.free_tmp if t.kind == "home"}
        return self.FUSED_STEPS and homes == set(range(N_HOME)) and not self.cold

    def _load_fq(self, reg0, slot):
        """v[reg0 : reg0 + NL] <- the c0 component (an Fq value) of a register-resident slot"""
        self._need(mag(self.r_of(slot)) <= 1.0, f"{slot} is not normalised")
        for i in range(NL):
            if slot.kind == "agpr":
                self.e.emit(f"v_accvgpr_read_b32 v{reg0 + i}, a{SLOT_DW * slot.idx + i}", vw=[reg0 + i])
            elif slot.kind == "home":
                self.e.emit(f"v_mov_b32_e32 v{reg0 + i}, v{HOME0 + SLOT_DW * slot.idx + i}", vw=[reg0 + i])
            else:
                raise ValueError(slot.kind)
        return self.v_of(slot)

    def _step_in(self, slots, alt=None):
        """home block k <- slots[k].  alt = (condition emitter, other slots for the first three, label maker): the R operands
        come from `alt` slots when the run-time condition holds (SCC set by the emitter) -- pair 0 of a multi-pairing group keeps
        its R in LDS."""
        vs = []
        if alt is not None:
            cond, other, lab = alt[:3]
            part = alt[4] if len(alt) > 4 else None          # (condition emitter, slots): pair 1's first coordinates live on chip as well
            l_alt, l_done = lab("L_si_alt"), lab("L_si_done")
            for s_ in list(slots[:3]) + list(other) + (list(part[1]) if part else []):
                self._need(mag(self.r_of(s_)) <= 1.0, f"fused step operand {s_} is not normalised")
                vs.append(self.v_of(s_))
            self.wait()
            cond(self.e)
            self.e.salu(f"s_cbranch_scc1 {l_alt}")
            if part:
                l_part = lab("L_si_part")
                part[0](self.e)
                self.e.salu(f"s_cbranch_scc1 {l_part}")
            for k in range(3):
                self.load(HOME0 + SLOT_DW * k, slots[k])
            self.wait()
            self.e.salu(f"s_branch {l_done}")
            if part:
                self.e.label(l_part)
                for k in range(3):
                    self.load(HOME0 + SLOT_DW * k, part[1][k] if k < len(part[1]) else slots[k])
                self.wait()
                self.e.salu(f"s_branch {l_done}")
            self.e.label(l_alt)
            for k in range(3):
                self.load(HOME0 + SLOT_DW * k, other[k])
            self.wait()
            self.e.label(l_done)
            slots = slots[3:]
            base = 3
        else:
            base = 0
        for k, s_ in enumerate(slots):
            self._need(mag(self.r_of(s_)) <= 1.0, f"fused step operand {s_} is not normalised")
            if not (s_.kind == "home" and s_.idx == base + k):          # (phase 1 of the split loop: R stays in home blocks 0..2 from step to step)
                self.load(HOME0 + SLOT_DW * (base + k), s_)
            vs.append(self.v_of(s_))
        return max(vs)

    def _step_out_r(self, blocks, R, vs, alt):
        """the three coordinates of the new R: home blocks -> R slots, or -> the `alt` slots when the run-time condition holds"""
        if alt is None:
            for k, dst, v in zip(blocks, R, vs):
                self._step_out(k, dst, v)
            return
        cond, other, lab = alt[:3]
        turn = alt[3] if len(alt) > 3 else None          # (condition emitter, slots): the pair that opens the next pass as well keeps its R on chip
        part = alt[4] if len(alt) > 4 else None          # (condition emitter, slots): pair 1's first coordinates stay on chip, the rest streams
        l_alt, l_done = lab("L_so_alt"), lab("L_so_done")
        self.wait()
        cond(self.e)
        self.e.salu(f"s_cbranch_scc1 {l_alt}")
        if part is not None:
            l_part = lab("L_so_part")
            part[0](self.e)
            self.e.salu(f"s_cbranch_scc1 {l_part}")
        if turn is not None:
            l_turn = lab("L_so_turn")
            turn[0](self.e)
            self.e.salu(f"s_cbranch_scc1 {l_turn}")
        if EXP_R1:
            l_r1, l_r1d = lab("L_so_r1"), lab("L_so_r1d")
            self.e.salu(f"s_cmp_eq_u32 s{S_JP}, 1")
            self.e.salu(f"s_cbranch_scc1 {l_r1}")
        for k, dst, v in zip(blocks, R, vs):
            self._step_out(k, dst, v)
        if EXP_R1:
            self.wait()
            self.e.salu(f"s_branch {l_r1d}")
            self.e.label(l_r1)
            for k, dst, v in list(zip(blocks, R, vs))[EXP_R1:]:
                self._step_out(k, dst, v)
            self.e.label(l_r1d)
        self.wait()
        self.e.salu(f"s_branch {l_done}")
        if turn is not None:
            self.e.label(l_turn)
            for k, dst, v in zip(blocks, turn[1], vs):
                self._step_out(k, dst, v)
            self.wait()
            self.e.salu(f"s_branch {l_done}")
        if part is not None:
            n1 = len(part[1])
            self.e.label(l_part)
            for k, dst, v in list(zip(blocks, part[1], vs))[:n1]:
                self._step_out(k, dst, v)
            if turn is not None:                          # (k = 2: pair 1 is also the pair that turns)
                l_pt = lab("L_so_part_turn")
                turn[0](self.e)
                self.e.salu(f"s_cbranch_scc1 {l_pt}")
            for k, dst, v in list(zip(blocks, R, vs))[n1:]:
                self._step_out(k, dst, v)
            self.wait()
            self.e.salu(f"s_branch {l_done}")
            if turn is not None:
                self.e.label(l_pt)
                for k, dst, v in list(zip(blocks, turn[1], vs))[n1:]:
                    self._step_out(k, dst, v)
                self.wait()
                self.e.salu(f"s_branch {l_done}")
        self.e.label(l_alt)
        for k, dst, v in zip(blocks, other, vs):
            self._step_out(k, dst, v)
        self.wait()
        self.e.label(l_done)

    def _step_out(self, k, dst, v, limbs=1.0):
        """dst <- home block k (a result of value bound v; normalised, or with limbs of up to `limbs` units)"""
        self._need(v <= self.v_limit(dst) and v <= V_CAP, f"fused step result {dst}: {v} p")
        self._need(limbs <= (1.0 if self.key(dst) in self.norm_keys else STORE_MAG), f"fused step result {dst}: limbs of {limbs} units")
        if dst.kind == "home" and dst.idx == k:
            pass                                       # the result stays where the routine left it
        elif not (EXP_NO_RSTORE and dst.kind == "globdyn"):
            self.store(HOME0 + SLOT_DW * k, dst)
        self.slot_r[self.key(dst)] = self.r_norm(v) if limbs <= 1.0 else (-max(limbs, v / K_TOP), max(limbs, v / K_TOP))
        self.slot_v[self.key(dst)] = v
        self.max_v = max(self.max_v, v)

    def _load_point_p(self, Pt, load_p):
        """block B <- (Px limbs, Py limbs) of the evaluation point; load_p: the caller's own code for it (multi-pairing kernels)"""
        if load_p is None:
            return max(self._load_fq(B0, Pt[0]), self._load_fq(B0 + NL, Pt[1]))
        for s_ in Pt:
            self._need(mag(self.r_of(s_)) <= 1.0, f"{s_} is not normalised")
        load_p(self)
        return max(self.v_of(Pt[0]), self.v_of(Pt[1]))

    def _scale_from_z(self, scale, square):
        """scale *= Z (or Z^2) with Z where the step has just put it (home block 2): the R slots differ from pair to pair"""
        z = HOME(2, "Z")
        self.slot_r[self.key(z)], self.slot_v[self.key(z)] = R_NORM, V_STORE
        self.tagA = None
        self.A(z)
        if square:
            self.sqr()
        self.mul(scale).to(scale)
        self.wait()
        self.tagA = self.tagB = None

    def _dbl_step_fused(self, R, Pt, line, out=None, after_load=None, load_p=None, alt_r=None, scale_in=None):
        v = self._step_in(R, alt_r)
        if scale_in is not None:
            self._scale_from_z(scale_in, square=True)
        vp = self._load_point_p(Pt, load_p)
        self.tagA = self.tagB = None
        if after_load:
            self.wait()
            after_load()
        R = out or R
        self.marker("dblstep")
        self._raw_call("dblstep")
        self.marker("stepout")
        # transfer function of L1v4.r_dblstep (the xi^2-scaled doubling): B = Y^2, N = 9 Z^2, H = 2 Y Z, T / S = xi B -+ 3 N
        sq = lambda x: 4 * x * x / K_RP + 0.5
        ml = lambda x, y: 2 * x * y / K_RP + 0.5
        bq = c = sq(v)
        hh = sq(2 * v) + bq + c
        xb, xh, n = 10 * bq, 10 * hh, 9 * c
        t_ = xb + 3 * n
        self._need(max(xb, xh, t_) <= V_CAP, f"dblstep operand values {xb} {xh} {t_}")
        # Y3 = S S + N (-12 N): two two-product passes on normalised S, N, M = -12 N (sums / differences of two components: two
        # units each): 9 (2 2 + 2 2) = 72 units per column; the value in front of its reducing chain
        m_s, m_n, m_m = max(1.0, t_ / K_TOP), max(1.0, n / K_TOP), max(1.0, 12 * n / K_TOP)       # limb magnitudes (the top limb carries the value)
        self._need(12 * n <= 2 * V_CAP and NL * (2 * m_s * 2 * m_s + 2 * m_n * 2 * m_m) <= COL_BUDGET, f"dblstep: Y3 passes, S {t_} p, M = -12 N of {12 * n} p")
        y3 = (4 * t_ * t_ + 48 * n * n) / K_RP + 0.5
        self._need(max(4 * ml(xb, xh), 20 * ml(ml(v, v), t_), y3, sq(t_) + 12 * sq(n)) <= 8 * V_CAP, "dblstep: values in front of the reducing chains")
        self._step_out_r((0, 1, 2), R, (0.51, 0.51, 0.51), alt_r)
        self._step_out(7, line[0], xb + n, limbs=2.0)
        self._step_out(4, line[1], hh * vp / K_RP + 0.5)
        self._step_out(5, line[2], 3 * sq(v) * vp / K_RP + 0.5)
        self.wait()

    def _add_step_fused(self, R, Q, Pt, line, out=None, after_load=None, load_p=None, load_q=None, alt_r=None, scale_in=None):
        if load_q is None:
            v = self._step_in(list(R) + list(Q), alt_r)
        else:                       # home blocks 3, 4 <- (x2, y2) by the caller's own code
            v = self._step_in(list(R), alt_r)
        if scale_in is not None:
            self._scale_from_z(scale_in, square=False)
        if load_q is not None:
            for s_ in Q:
                self._need(mag(self.r_of(s_)) <= 1.0, f"fused step operand {s_} is not normalised")
                v = max(v, self.v_of(s_))
            load_q(self)
        vp = self._load_point_p(Pt, load_p)
        self.tagA = self.tagB = None
        if after_load:
            self.wait()
            after_load()
        R = out or R
        self.marker("addstep")
        self._raw_call("addstep")
        self.marker("stepout")
        sq = lambda x: 4 * x * x / K_RP + 0.5
        ml = lambda x, y: 2 * x * y / K_RP + 0.5
        th = mu = v + ml(v, v)
        cc = d = sq(th)
        e_, fz, g = ml(mu, d), ml(v, cc), ml(v, d)
        hh = e_ + fz + 2 * g
        v5 = 2 * ml(v, v) - 0.5                                                # L5 = X y2 - x2 Y (one reduction): block A
        in_regs = line[1].kind == "home" and line[1].idx == 4
        if not in_regs:
            self.vA, self.rA, self.tagA = v5, None, None
            self.rA = self.r_norm()
            self.to(line[2])
        self._step_out_r((6, 4, 2), R, (ml(mu, hh), 2 * (th * (g + hh) + e_ * v) / K_RP + 0.5, ml(v, e_)), alt_r)
        self._step_out(7, line[0], mu * vp / K_RP + 0.5)
        if in_regs:                     # home 4 (Y3) has left: L3 parks there, L5 in home 5 (blocks 8 and A belong to the sparse multiplication)
            assert (line[0].kind, line[0].idx, line[2].kind, line[2].idx) == ("home", 7, "home", 5)
            self.wait()
            for i in range(SLOT_DW):
                self.e.emit(f"v_mov_b32_e32 v{HOME0 + SLOT_DW * 4 + i}, v{HOME0 + SLOT_DW * 8 + i}", vw=[HOME0 + SLOT_DW * 4 + i])
            for i in range(SLOT_DW):
                self.e.emit(f"v_mov_b32_e32 v{HOME0 + SLOT_DW * 5 + i}, v{A0 + i}", vw=[HOME0 + SLOT_DW * 5 + i])
            for dst, vv in ((line[1], th * vp / K_RP + 0.5), (line[2], v5)):
                self._need(vv <= self.v_limit(dst) and vv <= V_CAP, f"fused step result {dst}: {vv} p")
                self.slot_r[self.key(dst)] = self.r_norm(vv)
                self.slot_v[self.key(dst)] = vv
                self.max_v = max(self.max_v, vv)
            self.tagA = None
        else:
            self._step_out(8, line[1], th * vp / K_RP + 0.5)
        self.wait()

    def _store_line(self, dsts):
        """the line the fused step left in home blocks 7, 4, 5 (LINE_REGS) -> three slots (phase 1 of the split loop: the line area)"""
        for src, dst in zip(self.LINE_REGS, dsts):
            self.wait()
            self.store(HOME0 + SLOT_DW * src.idx, dst)
            k_ = self.key(dst)
            self.slot_r[k_], self.slot_v[k_] = self.r_of(src), self.v_of(src)
            self.max_v = max(self.max_v, self.v_of(src))

    def dbl_step(self, R, Pt, line, scale=None, out=None, after_load=None, load_p=None, alt_r=None, line_store=None):
        """R=(X,Y,Z) <- 2R ; line = (L0, L3, L4) of the tangent at the old R evaluated at P (Pt = (PX, PY) slots, scalar in c0).
        scale: slot of the running line scale s <- s * Z^2 (the caller squares it with f)."""
        X, Y, Z = R
        L0, L3, L4 = line
        if self._fused_ok():
            if scale is not None and alt_r is None:
                self.A(Z).sqr().mul(scale).to(scale)
            if LINE_IN_REGS:
                line = self.LINE_REGS
            self._dbl_step_fused(R, Pt, line, out, after_load, load_p, alt_r, scale_in=(scale if alt_r is not None else None))
            if line_store is not None:
                assert LINE_IN_REGS
                self._store_line(line_store)
            return line
        assert out is None and after_load is None and load_p is None and alt_r is None
        Bq, C, E, Fv, H, T = [self.tmp() for _ in range(6)]
        self.A(Y).sqr().to(Bq)
        self.A(Z).sqr().to(C)
        if scale is not None:
            self.A(scale).mul(C).to(scale)
        self.A(C).mul(THREE_B).to(E)
        self.A(E).dbl().add(E).to(Fv)
        self.A(Y).add(Z).sqr().sub(Bq).sub(C).to(H)            # H = 2 Y Z = (Y + Z)^2 - Y^2 - Z^2
        # line
        self.A(C).scale(9).to(T)                               # 9 C
        self.A(Bq).mulxi().sub(T).to(L0)
        self.A(H).mulfq(Pt[1]).to(L3)                          # H * Py
        self.A(X).sqr().to(T)
        self.A(T).dbl().add(T).mulfq(Pt[0]).neg().to(L4)       # -3 X^2 * Px
        # point
        self.A(Bq).sub(Fv).to(T)
        self.A(X).mul(Y).dbl().mul(T).to(X)
        self.A(E).sqr().to(T)
        self.A(T).scale(12).to(T)                              # 12 E^2
        self.A(Bq).add(Fv).sqr().sub(T).to(Y)
        self.A(Bq).mul(H).scale(4).to(Z)
        self.rel(Bq, C, E, Fv, H, T)
        return line

    def add_step(self, R, Q, Pt, line, scale=None, update=True, out=None, after_load=None, load_p=None, load_q=None, alt_r=None, line_store=None):
        """R <- R + Q (Q = (x2, y2) affine slots); line = (L2, L3, L5) of the chord through old R and Q at P."""
        X, Y, Z = R
        x2, y2 = Q
        L2, L3, L5 = line
        if update and self._fused_ok():
            if scale is not None and alt_r is None:
                self.A(scale).mul(Z).to(scale)
            if LINE_IN_REGS:
                line = self.LINE_REGS
            self._add_step_fused(R, Q, Pt, line, out, after_load, load_p, load_q, alt_r, scale_in=(scale if alt_r is not None else None))
            if line_store is not None:
                assert LINE_IN_REGS
                self._store_line(line_store)
            return line
        assert out is None and after_load is None and load_p is None and load_q is None and alt_r is None
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
        return line


    def pt_add(self, R, Q):
        """R <- R + Q (Q affine), the point part of add_step alone (fixed-base scalar multiplication of the input generator)."""
        X, Y, Z = R
        x2, y2 = Q
        th, mu, T, U, Cc, D, E = [self.tmp() for _ in range(7)]
        self.A(y2).mul(Z).rsub(Y).to(th)
        self.A(x2).mul(Z).rsub(X).to(mu)
        self.A(th).sqr().to(Cc)
        self.A(mu).sqr().to(D)
        self.A(mu).mul(D).to(E)
        self.A(Z).mul(Cc).to(Cc)
        self.A(X).mul(D).to(D)
        self.A(D).dbl().to(T)
        self.A(E).add(Cc).sub(T).to(T)
        self.A(mu).mul(T).to(X)
        self.A(E).mul(Y).to(U)
        self.A(D).sub(T).mul(th).sub(U).to(Y)
        self.A(Z).mul(E).to(Z)
        self.rel(th, mu, T, U, Cc, D, E)


    # ---- Jacobian arithmetic on the twist (x = X / Z^2, y = Y / Z^3), no lines: the G2 subgroup check (KernelBuilder subcheck).
    # The formulas of csrc/bn254_point_checks.h (dbl-2009-l, madd-2007-bl, add-2007-bl) WITHOUT their exceptional branches: an
    # exceptional case makes H = 0, hence Z3 = 0, and Z stays 0 through every later step -- the caller treats a final Z = 0 as "not in the
    # subgroup" (a point of the r-torsion never meets one: the partial scalars of [x]Q are below 2^63, tests/test_point_checks.py).
    def jac_dbl(self, R):
        X, Y, Z = R
        tA, tB, tC, tD, tE = [self.tmp() for _ in range(5)]
        self.A(X).sqr().to(tA)
        self.A(Y).sqr().to(tB)
        self.A(tB).sqr().to(tC)
        self.A(X).add(tB).sqr().sub(tA).sub(tC).dbl().to(tD)          # D = 2 ((X + B)^2 - A - C)
        self.A(Y).mul(Z).dbl().to(Z)                                  # Z3 = 2 Y Z
        self.A(tA).scale(3).to(tE)                                    # E = 3 A
        self.A(tE).sqr().sub(tD).sub(tD).to(X)                        # X3 = E^2 - 2 D
        self.A(tC).scale(8).to(tC)
        self.A(tD).sub(X).mul(tE).sub(tC).to(Y)                       # Y3 = E (D - X3) - 8 C
        self.rel(tA, tB, tC, tD, tE)

    def jac_madd(self, R, Q):
        """R <- R + (qx, qy, 1)"""
        X, Y, Z = R
        qx, qy = Q
        zz, h, hh, i, j, r, v = [self.tmp() for _ in range(7)]
        self.A(Z).sqr().to(zz)
        self.A(qx).mul(zz).sub(X).to(h)                               # H = U2 - X1
        self.A(qy).mul(Z).mul(zz).sub(Y).dbl().to(r)                  # r = 2 (S2 - Y1)
        self.A(h).sqr().to(hh)
        self.A(Z).add(h).sqr().sub(zz).sub(hh).to(Z)                  # Z3 = (Z1 + H)^2 - Z1Z1 - HH
        self.A(hh).scale(4).to(i)
        self.A(h).mul(i).to(j)
        self.A(X).mul(i).to(v)
        self.A(r).sqr().sub(j).sub(v).sub(v).to(X)                    # X3 = r^2 - J - 2 V
        self.A(Y).mul(j).dbl().to(hh)
        self.A(v).sub(X).mul(r).sub(hh).to(Y)                         # Y3 = r (V - X3) - 2 Y1 J
        self.rel(zz, h, hh, i, j, r, v)

    def jac_add(self, P1, P2):
        """P1 <- P1 + P2 (both projective)"""
        X1, Y1, Z1 = P1
        X2, Y2, Z2 = P2
        z1z1, z2z2, u1, s1, h, r, i, j, v = [self.tmp() for _ in range(9)]
        self.A(Z1).sqr().to(z1z1)
        self.A(Z2).sqr().to(z2z2)
        self.A(X1).mul(z2z2).to(u1)
        self.A(X2).mul(z1z1).sub(u1).to(h)                            # H = U2 - U1
        self.A(Y1).mul(Z2).mul(z2z2).to(s1)
        self.A(Y2).mul(Z1).mul(z1z1).sub(s1).dbl().to(r)              # r = 2 (S2 - S1)
        self.A(Z1).add(Z2).sqr().sub(z1z1).sub(z2z2).mul(h).to(Z1)    # Z3 = ((Z1 + Z2)^2 - Z1Z1 - Z2Z2) H
        self.A(h).dbl().sqr().to(i)                                   # I = (2 H)^2
        self.A(h).mul(i).to(j)
        self.A(u1).mul(i).to(v)
        self.A(r).sqr().sub(j).sub(v).sub(v).to(X1)
        self.A(s1).mul(j).dbl().to(i)
        self.A(v).sub(X1).mul(r).sub(i).to(Y1)
        self.rel(z1z1, z2z2, u1, s1, h, r, i, j, v)


# ======================================================================================================================
class _PhaseList(list):
    """The builder's section list: remembers in which phase (Miller loop / final exponentiation) a section was added."""

    def __init__(self, kb):
        super().__init__()
        self.kb = kb

    def append(self, e):
        self.kb.section_phase[id(e)] = self.kb._phase
        super().append(e)


class KernelBuilder:
    """Assembles one kernel blob.  Operand order of the asm statement (all inputs):
       %0 g1 (s64)  %1 g2 (s64)  %2 f_in (s64)  %3 out (s64)  %4 n (s32)  %5 k (s32)  %6 scratch (s64)
       %7 scratch pitch in bytes (s32): between the workgroups' blocks (layout "wg") / between slots (layout "slot")  %8 status (s64)  %9 tid (v32)  %10 block id (s32)  %11 grid size (s32)"""

    # slot map -------------------------------------------------------------------------------------
    # The Miller loop touches the Fq12 accumulator f in every routine (18 + 6 slot accesses per sparse multiplication): there
    # it lives in AGPR slots (72 issue cycles per access, no wait) and the cold point coordinates in LDS; the final
    # exponentiation needs the AGPR slots for the multiplication operand, so f moves to LDS at the phase boundary.
    F_IN_AGPR = bool(int(os.environ.get("KGEN_F_AGPR", "1")))
    F_LDS = [LDS(i, f"F{i}") for i in range(6)]
    F_AGPR = [AGPR(i, f"F{i}") for i in range(6)]
    F_AGPR_FEXP = bool(int(os.environ.get("KGEN_F_AGPR_FEXP", "1")))
    BOP = [AGPR(i, f"B{i}") for i in ((6, 7, 8, 10, 11, 12) if (F_IN_AGPR and F_AGPR_FEXP) else (0, 1, 2, 3, 4, 5))]      # fq12_mul operand copy (final exponentiation only)
    FQINV_BASE = AGPR(9, "fqinv_base")
    LINE = [AGPR(6, "La"), AGPR(7, "Lb"), AGPR(8, "Lc")]
    if F_IN_AGPR:
        R = [LDS(6, "RX"), LDS(7, "RY"), AGPR(9, "RZ")]          # RZ shares AGPR 9 with the Fq-inversion base (R is dead by then)
        QX, QY = LDS(0, "QX"), LDS(1, "QY")
        PX, PY = AGPR(10, "PX"), AGPR(11, "PY")
        SX, SY = AGPR(12, "SX"), AGPR(13, "SY")                  # the affine point of the current addition step
        SCALE = LDS(2, "scale")
        MILLER_FREE = ([], [LDS(3), LDS(4), LDS(5)])             # free (AGPR, LDS) slots of the Miller phase (+ the scale slot when untracked)
    else:
        R = [LDS(6, "RX"), LDS(7, "RY"), AGPR(9, "RZ")]
        SCALE = AGPR(11, "scale")
        QX, QY, PX, PY = AGPR(0, "QX"), AGPR(1, "QY"), AGPR(2, "PX"), AGPR(3, "PY")
        SX, SY = AGPR(4, "SX"), AGPR(5, "SY")
        MILLER_FREE = ([AGPR(10), AGPR(12), AGPR(13)], [])

    @property
    def F(self):
        """the running Fq12 accumulator: AGPR slots in the Miller phase, LDS in the final exponentiation"""
        if self.F_IN_AGPR and self.F_AGPR_FEXP:
            return self.F_AGPR
        return self.F_AGPR if (self.F_IN_AGPR and self.do_miller and self._phase == "miller") else self.F_LDS

    PAIR_SLOT0 = N_GSLOTS               # scratch slots of pair j: PAIR_SLOT0 + 7 j + {PX, PY, QX, QY, RX, RY, RZ}

    COLD = ("L2_inv", "L2_frob1", "L2_frob2", "L2_frob3", "L2_dblfirst", "L2_addmul_last", "L2_descale", "L2_fqinv")

    MAX_FIXED = 4             # fixed pairs per group of the fixed-G2 kernel: their evaluation points fill LDS slots 2..5
    FIX_P = [LDS(2 + j, f"Pfix{j}") for j in range(4)]

    def __init__(self, do_miller=True, do_fexp=True, track=False, multi=False, helper=False, generate=False, subcheck=False, fixed=False, lines=False):
        """track: keep the running line scale and divide it out (the exact miller_loop_native value).
        multi: k pairs per lane with a shared f (multi_miller_loop_native, miller_loop_native.rs:192-282).
        helper: the batched public helpers of the reference on Fq12 batches -- MyFq12 `Mul`, frobenius_map_native
        (final_exp_native.rs:17-54), pow_native (:56-84) -- selected at run time by the kernel's `k` argument."""
        if helper:
            do_miller, do_fexp, track, multi = False, True, False, False
        if generate or subcheck or lines:
            do_miller, do_fexp, track, multi = False, False, False, False
        if fixed:
            do_miller, do_fexp, track, multi = True, True, False, False
        # fixed: pairing(P0, Q0) x prod_j pairing(P_j, Qfix_j) with the Qfix_j THE SAME for every group of the batch (a Groth16 verifier's
        # beta / gamma / delta): their line coefficients come from a table that the `lines` kernel makes once (_fixed_routines).
        self.fixed, self.lines = fixed, lines
        self.generate = generate
        # subcheck: ark's `G2Affine::new` contract (miller_loop_native.rs:303,311) -- is the lane's G2 point in the r-torsion?  One verdict word per
        # point into `out`; the point state lives in AGPR slots only (_subcheck_routines)
        self.subcheck = subcheck
        if subcheck:
            self.R = [AGPR(0, "RX"), AGPR(1, "RY"), AGPR(2, "RZ")]
            self.QX, self.QY = AGPR(3, "QX"), AGPR(4, "QY")
            self.SX, self.SY = self.QX, AGPR(5, "SY")
        self.do_miller, self.do_fexp, self.track = do_miller, do_fexp, track
        self.multi = multi
        self.helper = helper
        self.labels = {n: f"L1_{n}_%=" for n in L1V4_NAMES}
        for op in ("add", "sub", "rsub"):
            for i in range(N_HOME):
                self.labels[f"{op}_h{i}"] = f"L1_{op}_h{i}_%="
        self.sections = []
        self._phase = "miller"
        self._cold = False
        self._uid = 0
        if multi and not MULTI_HOT_ENDS:       # the resident-slot step routines only run for the first and the last steps of a multi kernel
            self.COLD = self.COLD + ("L2_dblmul", "L2_addmul")

    def lab(self, name):
        return f"{name}_%="

    @property
    def s_mode(self):
        """scalar register of the launch's I/O layout bits, or None where the kernel is limb-major only (k_op: its k argument is full; k_generate;
        the split-loop experiment of the k-pair kernels, which owns every spare scalar register)"""
        if self.helper or self.generate or self.lines or (self.multi and FISSION):
            return None
        return S_MODE_MULTI if self.multi else S_MODE_SINGLE

    @property
    def naf(self):
        """the digits of 6 x + 2 this kernel's Miller loop walks (least significant first; the top one is R = Q, f = 1)"""
        if self.subcheck:            # the scalar multiplication [x]Q walks the non-adjacent form of x (63 digits, 24 non-zero), padded to the masks' 64
            d, n = [], BN_X
            while n:
                z = (2 - n % 4) if n & 1 else 0
                d.append(z)
                n = (n - z) // 2
            return d + [0] * (66 - len(d))
        # (an untracked Miller value is only ever the input of a final exponentiation -- in the same kernel or, for the Miller-only k-pair kernel of the
        # spread route, in a later one -- and that does not see the chain)
        if SHORT_CHAIN and ((self.do_miller and not self.track and not FISSION) or self.lines):
            return SIX_U_PLUS_2_SHORT
        return SIX_U_PLUS_2_NAF

    @property
    def naf_first(self):
        """digit index of the first doubling (L2_dblfirst): 63 for a 65-digit form (the reference's table, the minimal-weight form); a
        66-digit form (the canonical NAF: first = 64) must have digit 64 zero -- the loop's 64-bit digit masks hold digits 0..63"""
        assert len(self.naf) - 2 == 63 or self.naf[64] == 0
        return len(self.naf) - 2

    @property
    def chunk2(self):
        """two doubling iterations per chunk with both lines parked on chip: the untracked one-pair kernels (k_pairing)"""
        return (CHUNK2 and self.do_miller and not self.track and not self.multi and not self.helper and not self.generate and LINE_IN_REGS
                and Prog.FUSED_STEPS and not FISSION and self.F_IN_AGPR and not self.fixed)

    def chunk_park(self):
        return [[LDS(3, "parkA0"), LDS(4, "parkA1"), LDS(5, "parkA2")], [LDS(2, "parkB0"), self.SX, self.SY]]

    def _chunk_routines(self):
        """L2_dbl_p: the fused doubling step on the resident R, its line -> park set S_PARK.  L2_sp034_c: park set S_PARK -> the line
        registers (home blocks 7, 4, 5), then the sparse multiplication."""
        park = self.chunk_park()
        bounds = {}
        temps = [HOME(i) for i in range(N_HOME)] + list(self.LINE) + [GLOB(GLOB_TMP0 + i) for i in range(8)]
        pkeys = frozenset(Prog.key(s_) for set_ in park for s_ in set_)
        L = self.lab

        def dbl_p(p):
            e = p.e
            p.temp_keys = p.temp_keys | pkeys            # the parked lines have their own contract (recorded below), not V_STORE
            p.dbl_step(self.R, (self.PX, self.PY), self.LINE)
            p.wait()
            u = self.uid()
            e.salu(f"s_cmp_eq_u32 s{S_PARK}, 0")
            e.salu(f"s_cbranch_scc0 {L(f'L_pk1_{u}')}")
            p._store_line(park[0])
            p.wait()
            e.salu(f"s_branch {L(f'L_pkd_{u}')}")
            e.label(L(f"L_pk1_{u}"))
            p._store_line(park[1])
            p.wait()
            e.label(L(f"L_pkd_{u}"))
            for i, src in enumerate(Prog.LINE_REGS):
                bounds[i] = (mag(p.r_of(src)), p.v_of(src))
        flat = [s_ for set_ in park for s_ in set_]
        self.l2_routine("L2_dbl_p", dbl_p, temps, local=list(self.LINE) + flat)      # (the parked lines travel to L2_sp034_c with the bounds recorded above)

        def sp_c(p):
            e = p.e
            p.reset_tags()
            p.temp_keys = p.temp_keys | pkeys
            u = self.uid()
            e.salu(f"s_cmp_eq_u32 s{S_PARK}, 0")
            e.salu(f"s_cbranch_scc0 {L(f'L_up1_{u}')}")
            for dst, src in zip(Prog.LINE_REGS, park[0]):
                p.load(HOME0 + SLOT_DW * dst.idx, src)
            p.wait()
            e.salu(f"s_branch {L(f'L_upd_{u}')}")
            e.label(L(f"L_up1_{u}"))
            for dst, src in zip(Prog.LINE_REGS, park[1]):
                p.load(HOME0 + SLOT_DW * dst.idx, src)
            p.wait()
            e.label(L(f"L_upd_{u}"))
            for i, dst in enumerate(Prog.LINE_REGS):
                m_, v_ = bounds[i]
                p._need(v_ <= V_CAP, f"line coefficient of {v_} p")
                p.slot_r[p.key(dst)] = (-m_, m_) if m_ > 1.0 else p.r_norm(v_)
                p.slot_v[p.key(dst)] = v_
            p.mul_by_034(self.F, *Prog.LINE_REGS)
        self.l2_routine("L2_sp034_c", sp_c, temps, local=list(self.LINE) + flat)

    @property
    def fission(self):
        """split Miller loop (see FISSION): the untracked kernels -- k_pairing, k_mpairing (groups of up to FIS_MAX_K pairs)"""
        return FISSION and self.do_miller and not self.track and LINE_IN_REGS and Prog.FUSED_STEPS

    # phase 2 of the split loop: the next line waits in three AGPR slots (free there: in the one-pair kernel RZ, PX, PY are parked in LDS meanwhile)
    LINE_BUF = [AGPR(9, "bLa"), AGPR(10, "bLb"), AGPR(11, "bLc")]
    FIS_PARK = [LDS(3, "parkRZ"), LDS(4, "parkPX"), LDS(5, "parkPY")]

    def _emit_line_prefetch(self, e):
        """LINE_BUF <- the line triple at the phase-2 cursor S_LOFF2 (global loads straight into the AGPRs, nobody waits here); cursor += 3 slots.
        Nothing is fetched behind the last line (S_LCNT counts the lines left)."""
        skip = self.lab(f"L_lpf_none_{self.uid()}")
        e.salu(f"s_cmp_eq_u32 s{S_LCNT}, 0")
        e.salu(f"s_cbranch_scc1 {skip}")
        e.salu(f"s_sub_u32 s{S_LCNT}, s{S_LCNT}, 1")
        for k, dst in enumerate(self.LINE_BUF):
            if EXP_NO_SCRATCH:
                break
            a0 = SLOT_DW * dst.idx
            if k == 0:
                e.salu(f"s_add_u32 s62, s64, s{S_LOFF2}")
            else:
                e.salu(f"s_mul_i32 s{S_TMP0}, s{S_GSTRIDE}, {k}")
                e.salu(f"s_add_u32 s{S_TMP0}, s{S_TMP0}, s{S_LOFF2}")
                e.salu(f"s_add_u32 s62, s64, s{S_TMP0}")
            e.salu("s_addc_u32 s63, s65, 0")
            for c in range(Prog.N_B128):
                e.emit(f"global_load_dwordx4 a[{a0 + 4 * c}:{a0 + 4 * c + 3}], v{V_GOFF}, {S_GADDR} offset:{GCHUNK0 + 1024 * c}" + _ldm(), kind="vmem")
            e.emit(f"global_load_dwordx2 a[{a0 + 16}:{a0 + 17}], v{V_GOFF8}, {S_GADDR} offset:0" + _ldm(), kind="vmem")
        e.salu(f"s_mul_i32 s{S_TMP0}, s{S_GSTRIDE}, 3")
        e.salu(f"s_add_u32 s{S_LOFF2}, s{S_LOFF2}, s{S_TMP0}")
        e.label(skip)

    def _fission_routines(self):
        """L2 routines of the split Miller loop.  Phase 1: L2_dbl_f / L2_add_f -- one fused point step with R in home blocks 0..2 on entry
        and exit, its line -> the three slots at the cursor S_LOFF, cursor += S_LSTEP; L2_r2h / L2_h2r move R between its resident
        slots and the home blocks.  Phase 2: L2_sp034_f / L2_sp235_f -- wait for the prefetched line, take it into home blocks 7, 4, 5,
        prefetch the next one, sparse multiplication; L2_ln_pf issues the first prefetch."""
        Hs = [HOME(0, "X"), HOME(1, "Y"), HOME(2, "Z")]
        LG = [GlobLine(0), GlobLine(1), GlobLine(2)]
        hkeys = frozenset(Prog.key(h) for h in Hs)
        lkeys = [Prog.key(g) for g in LG]
        bounds = {}

        def phase1(kind):
            def body(p):
                p.norm_keys = p.norm_keys | hkeys                       # R arrives normalised in the home blocks (the contract of its slots)
                p.temp_keys = p.temp_keys | frozenset(lkeys)            # the line area has its own contract (checked below), not V_STORE
                if kind == "dbl":
                    p.dbl_step(Hs, (self.PX, self.PY), self.LINE, line_store=LG)
                else:
                    p.add_step(Hs, (self.SX, self.SY), (self.PX, self.PY), self.LINE, line_store=LG)
                p.wait()
                p.e.salu(f"s_add_u32 s{S_LOFF}, s{S_LOFF}, s{S_LSTEP}")
                for i, k_ in enumerate(lkeys):
                    r_, v_ = p.slot_r[k_], p.slot_v[k_]
                    o = bounds.get((kind, i), (0.0, 0.0))
                    bounds[(kind, i)] = (max(o[0], mag(r_)), max(o[1], v_))
            return body
        tm = self.miller_temps(extra=tuple(self.LINE))
        self.l2_routine("L2_dbl_f", phase1("dbl"), tm, local=self.LINE)
        self.l2_routine("L2_add_f", phase1("add"), tm, local=self.LINE)
        self.fis_line_bounds = bounds

        def r2h(p):
            for k, src in enumerate(self.R):
                p._need(mag(p.r_of(src)) <= 1.0, f"{src} is not normalised")
                p.load(HOME0 + SLOT_DW * k, src)
            p.wait()
        def h2r(p):
            p.norm_keys = p.norm_keys | hkeys
            for k, dst in enumerate(self.R):
                p.wait()
                p.store(HOME0 + SLOT_DW * k, dst)
                p.slot_r[p.key(dst)], p.slot_v[p.key(dst)] = R_NORM, V_STORE
            p.max_v = max(p.max_v, V_STORE)
            p.wait()
        self.l2_routine("L2_r2h", r2h, tm)
        self.l2_routine("L2_h2r", h2r, tm)

        def phase2(kind):
            def body(p):
                e = p.e
                e.raw("s_waitcnt vmcnt(0)")                               # the prefetched line has landed
                p.reset_tags()
                for i, (dst, src) in enumerate(zip(Prog.LINE_REGS, self.LINE_BUF)):
                    p.load(HOME0 + SLOT_DW * dst.idx, src)
                    m_, v_ = bounds[(kind, i)]
                    p._need(v_ <= V_CAP, f"line coefficient of {v_} p")
                    p.slot_r[p.key(dst)] = (-m_, m_) if m_ > 1.0 else p.r_norm(v_)
                    p.slot_v[p.key(dst)] = v_
                p.wait()
                self._emit_line_prefetch(e)                              # the buffer is free again: the next line travels under this multiplication
                if kind == "dbl":
                    p.mul_by_034(self.F, *Prog.LINE_REGS)
                else:
                    p.mul_by_235(self.F, *Prog.LINE_REGS)
            return body
        tm2 = self.miller_temps(extra=(*self.LINE, self.SX, self.SY))
        self.l2_routine("L2_sp034_f", phase2("dbl"), tm2, local=self.LINE)
        self.l2_routine("L2_sp235_f", phase2("add"), tm2, local=self.LINE)
        self.l2_routine("L2_ln_pf", lambda p: self._emit_line_prefetch(p.e), tm2)
        if not self.multi:               # one-pair kernel: RZ, PX, PY wait in LDS while their AGPR slots serve as the line buffer
            def park(p, back):
                for a_, l_ in zip((self.R[2], self.PX, self.PY), self.FIS_PARK):
                    if back:
                        p.A(l_).to(a_)
                    else:
                        p.A(a_).to(l_)
                p.wait()
            self.l2_routine("L2_fpark", lambda p: park(p, False), tm2)
            self.l2_routine("L2_funpark", lambda p: park(p, True), tm2)

    def uid(self):
        self._uid += 1
        return self._uid

    # ---------------------------------------------------------------------------------------------
    def norm_keys(self, phase):
        """Slots that hold normalised values on every routine boundary of `phase` (Prog.norm_keys)."""
        keys = [Prog.key(s_) for s_ in self.F]
        if phase == "fexp":
            keys += [Prog.key(s_) for s_ in self.BOP] + [Prog.key(s_) for s_ in self.LREG] + [("globdyn", i) for i in range(6)]
        else:       # the point state of the Miller loop (and its per-pair copies in scratch): operands of the fused steps
            keys += [Prog.key(s_) for s_ in (*self.R, self.QX, self.QY, self.PX, self.PY, self.SX, self.SY)]
            keys += [("globdyn", i) for i in range(7)]
            if self.fixed:
                keys += [Prog.key(s_) for s_ in self.FIX_P]
            if self.multi and self.r0_resident():
                keys += [Prog.key(s_) for s_ in self.R0_LDS] + [Prog.key(s_) for s_ in self.r1_slots()]
            if self.fission:
                keys += [Prog.key(s_) for s_ in self.LINE_BUF] + ([] if self.multi else [Prog.key(s_) for s_ in self.FIS_PARK])
        return frozenset(keys)

    def new_prog(self, temps, phase=None):
        if not hasattr(self, "l2_bodies"):
            self.l2_bodies, self.l2_exit, self.l2_maxv, self.l2_phase = {}, {}, {}, {}
        e = Emitter()
        p = Prog(e, self.labels)
        p.set_temps(temps)
        p.norm_keys = self.norm_keys(phase or self._phase)
        p.homes_free = not any(t.kind == "home" and t.idx < 8 for t in temps)      # mul6's workspace: home blocks 0..7
        p.cold = self._cold
        return e, p

    def l2_routine(self, name, body, temps, local=(), entry=None):
        """Also records the value bounds (units of p) the routine leaves in every non-temporary slot, given that all
        its inputs were below V_STORE p: the basis of the inductive certification in certify_values().
        entry: {slot key: bound} -- TIGHTER bounds this routine may assume for some of its inputs (its call sites guarantee them:
        certify_values() checks every call against them)."""
        self._cold = name in self.COLD
        e, p = self.new_prog(temps)
        if entry:
            assert all(v <= V_STORE for v in entry.values())
            p.entry_v = dict(entry)
            if not hasattr(self, "l2_entry"):
                self.l2_entry = {}
            self.l2_entry[name] = dict(entry)
        # `local`: named slots that are written and consumed inside this routine (the line coefficients): no contract at its exit
        p.temp_keys = p.temp_keys | {Prog.key(s_) for s_ in local}
        e.label(self.lab(name))
        body(p)
        p.wait()
        e.salu(f"s_setpc_b64 {S_RET2}")
        self.sections.append(e)
        if not hasattr(self, "l2_section"):
            self.l2_section = {}
        self.l2_section[name] = e
        self._cold = False
        tk = {Prog.key(t) for t in temps} | {Prog.key(s_) for s_ in local}
        self.l2_bodies[name] = (body, temps)
        self.l2_phase[name] = self._phase
        # (home registers never carry a value across a routine boundary: they are every routine's workspace)
        # (... nor does the line area of the split loop: its bounds travel from the phase-1 routines to the phase-2 ones, _fission_routines)
        self.l2_exit[name] = {k: v for k, v in p.slot_v.items() if k not in tk and k[0] not in ("home", "globline")}      # ("tab": to() keeps the contract)
        self.l2_maxv[name] = p.max_v
        return p

    def miller_temps(self, extra=(), no_homes=False, in_loop=False):
        """Fast temporaries of the Miller-loop routines: the home registers (only block 8 in routines built on the fused
        Fq6 multiplication, whose workspace is blocks 0..7), the free AGPR slots, routine-specific dead slots; global
        scratch slots only as overflow."""
        homes = [HOME(8)] if no_homes else [HOME(i) for i in range(N_HOME)]
        fa, fl = self.MILLER_FREE
        if self.fission and not self.multi:      # LDS 3..5 hold RZ, PX, PY during phase 2 of the split loop
            fl = []
        if self.fixed:                           # LDS 2..5 hold the fixed pairs' evaluation points for the whole Miller loop
            return homes + fa + list(extra) + [GLOB(GLOB_TMP0 + i) for i in range(8)]
        if in_loop and self.multi and self.r0_resident():    # (f^2 of the streamed loop) LDS 3..5 hold pair 0's R, LDS 2 pair 1's X there: never temporaries
            return (homes + fa + list(extra) + ([] if self.track or self.SCALE.kind == "agpr" or self.r1_slots() else [self.SCALE])
                    + [GLOB(GLOB_TMP0 + i) for i in range(8)])
        return (homes + fa + ([] if self.track else ([self.SCALE] if self.SCALE.kind == "agpr" else [])) + list(extra) + fl
                + ([] if self.track or self.SCALE.kind == "agpr" else [self.SCALE]) + [GLOB(GLOB_TMP0 + i) for i in range(8)])

    # An on-chip Fq12 register of the final exponentiation: six of the eight LDS slots (free there since fq12_mul works with three
    # temporaries -- home block 8, AGPR 13 and, outside the inversion, AGPR 9).  It holds conj(b^19) during an x-power (most of the
    # twelve digit multiplications use it: no operand fetch) and T0 during the y-chain.
    LREG = [LDS(i, f"L{i - 2}") for i in range(2, 8)]

    def fexp_temps(self, no_homes=False, lds=False):
        """lds: the LDS slots as temporaries -- only for the routines of the easy part (inversion), which run before the on-chip
        register LREG is live."""
        # fastest first: home registers, then AGPR slots (72 cycles either way), then LDS (a slot store costs 130-270 cycles)
        homes = [HOME(8)] if no_homes else [HOME(i) for i in range(N_HOME)]
        if self.F_IN_AGPR and self.F_AGPR_FEXP:      # f: AGPR 0..5, the multiplication operand: AGPR 6..8, 10..12 (9: the Fq-inversion base)
            return homes + [AGPR(13)] + ([LDS(i) for i in range(N_LDS_SLOTS)] if lds else []) + [GLOB(GLOB_TMP0 + i) for i in range(8)]
        return homes + [AGPR(i) for i in (6, 7, 8, 10, 11, 12, 13)] + [LDS(6), LDS(7)] + [GLOB(GLOB_TMP0 + i) for i in range(8)]

    # ---------------------------------------------------------------------------------------------
    def build(self):
        main = Emitter()
        self.prologue(main)
        l1 = {}                                  # one section per leaf routine: their order is chosen below
        for n in L1V4_NAMES:
            l1[n] = Emitter()
            l1[n].label(self.labels[n])
            getattr(L1v4(l1[n]), "r_" + n)()
            l1[n].salu(f"s_setpc_b64 {S_RET1}")
        for op in ("add", "sub", "rsub"):
            for i in range(N_HOME):
                n = f"{op}_h{i}"
                l1[n] = Emitter()
                l1[n].label(self.labels[n])
                L1v4(l1[n]).home_variant(op, i)
                l1[n].salu(f"s_setpc_b64 {S_RET1}")
        self.sections = _PhaseList(self)
        self.section_phase = {}
        self.control_sections = []
        self._phase = "miller"
        if self.do_miller:
            sc = self.SCALE if self.track else None
            # during f^2 the line (AGPR 6..8) and the addition point (AGPR 4, 5) are dead
            # during f^2 the line and the addition point are dead (in the multi kernels the S slots hold the prefetched pair)
            if self.chunk2:      # (the addition point's slots and LDS 2..5 hold parked lines while f^2 runs)
                sqr_temps = [HOME(8)] + list(self.LINE) + [GLOB(GLOB_TMP0 + i) for i in range(8)]
            else:
                sqr_temps = self.miller_temps(extra=(tuple(self.LINE) if self.multi else (self.SX, self.SY, *self.LINE)), no_homes=True, in_loop=True)
            self.l2_routine("L2_sqr", lambda p: p.fq12_sqr(self.F), sqr_temps)
            # (routines that run the fused steps hand the line over in registers: the LINE slots are then ordinary temporaries)
            hot = lambda name: LINE_IN_REGS and Prog.FUSED_STEPS and name not in self.COLD
            line_tmp = lambda name: tuple(self.LINE) if hot(name) else ()
            if not (self.fission and not self.multi):           # (the one-pair split loop has no use for it)
                self.l2_routine("L2_dblmul", lambda p: p.mul_by_034(self.F, *p.dbl_step(self.R, (self.PX, self.PY), self.LINE, scale=sc)),
                                self.miller_temps(extra=(*line_tmp("L2_dblmul"), self.SX, self.SY)), local=self.LINE)
            self.l2_routine("L2_dblfirst", lambda p: self._dbl_first(p), self.miller_temps(), local=self.LINE)

            def addmul(p, update):
                line = p.add_step(self.R, (self.SX, self.SY), (self.PX, self.PY), self.LINE, scale=sc, update=update)
                glob = [t for t in p.free_tmp if t.kind == "glob"]
                p.free_tmp = [t for t in p.free_tmp if t.kind != "glob"] + [self.SX, self.SY] + glob    # S is dead now
                p.temp_keys = p.temp_keys | {Prog.key(self.SX), Prog.key(self.SY)}
                p.mul_by_235(self.F, *line)

            self.l2_routine("L2_addmul", lambda p: addmul(p, True), self.miller_temps(extra=line_tmp("L2_addmul")), local=self.LINE)
            self.l2_routine("L2_addmul_last", lambda p: addmul(p, False), self.miller_temps(), local=self.LINE)
            if self.fission:
                self._fission_routines()
            if self.chunk2:
                self._chunk_routines()
            if self.multi:
                self._stream_routines(sc)
            if self.track:
                self.l2_routine("L2_fqinv", self._fq_inv, self.miller_temps())
                self.l2_routine("L2_descale", self._descale, self.miller_temps())
                self.l2_routine("L2_sqscale", lambda p: p.A(self.SCALE).sqr().to(self.SCALE), self.miller_temps())
        if self.generate:
            gt = self.gen_temps()
            self.l2_routine("L2_fqinv", self._fq_inv, gt)
            self.l2_routine("L2_ptadd", lambda p: p.pt_add(self.R, (self.SX, self.SY)), gt)
            self.l2_routine("L2_affine", self._to_affine, gt)
        if self.subcheck:
            self._subcheck_routines()
        if self.fixed:
            self._fixed_routines()
        if self.lines:
            self._lines_routines()
        self._phase = "fexp"
        if self.do_fexp:
            if not (self.do_miller and self.track):
                self.l2_routine("L2_fqinv", self._fq_inv, self.fexp_temps(lds=True))
            self.l2_routine("L2_cyc", self._cyc_run, self.fexp_temps())
            self.l2_routine("L2_redF", self._reduce_f, self.fexp_temps())
            self._mulG_routines()
            for k in (1, 2, 3):
                self.l2_routine(f"L2_frob{k}", lambda p, k=k: self._frobenius(p, k), self.fexp_temps())
            self.l2_routine("L2_inv", self._fq12_inv, self.fexp_temps(no_homes=INV_FUSED, lds=True))
            self.l2_routine("L2_cpB", lambda p: [p.A(self.F[i]).to(self.BOP[i]) for i in range(6)], self.fexp_temps())
            self.l2_routine("L2_stL", lambda p: [p.A(self.F[i]).to(self.LREG[i]) for i in range(6)], self.fexp_temps())
            self.l2_routine("L2_ldL", lambda p: [p.A(self.LREG[i]).to(self.F[i]) for i in range(6)], self.fexp_temps())
            self.l2_routine("L2_stG", lambda p: [p.A(self.F[i]).to(GlobDyn(i)) for i in range(6)], self.fexp_temps())
            self.l2_routine("L2_ldG", lambda p: self.batch_load_globdyn(p.e, p, range(6), self.F), self.fexp_temps())
            self.l2_routine("L2_ldGc", lambda p: (self.batch_load_globdyn(p.e, p, range(6), self.F),
                                                  [p.A(self.F[i]).neg().to(self.F[i]) for i in (1, 3, 5)]), self.fexp_temps())
            self.l2_routine("L2_conjF", lambda p: [p.A(self.F[i]).neg().to(self.F[i]) for i in (1, 3, 5)], self.fexp_temps())
            self._powx_routine()
            if self.helper:
                for k in range(4, 12):
                    self.l2_routine(f"L2_frob{k}", lambda p, k=k: self._frobenius(p, k), self.fexp_temps())
                self.l2_routine("L2_sqrF", lambda p: p.fq12_sqr(self.F), self.fexp_temps(no_homes=True, lds=True))
        self._phase = "miller"              # the main program only touches F
        self.main_body(main)
        # Layout: s_call_b64 / s_branch reach +-128 KB.  The leaf routines (called from everywhere) and the main control
        # code sit in the middle; Miller-loop routines in front of that block, final-exponentiation routines behind it, the
        # small ones (called from the main program) nearest to the middle, the big cold ones (Fq12 inversion) at the far ends.
        size = {id(e): len(e.finalize()) for e in self.sections}
        first = sorted([e for e in self.sections if self.section_phase[id(e)] == "miller"], key=lambda e: -size[id(e)])
        second = sorted([e for e in self.sections if self.section_phase[id(e)] == "fexp"], key=lambda e: size[id(e)])
        tot = lambda lst: sum(size[id(e)] for e in lst)
        while second and tot(second) - tot(first) > size[id(second[-1])]:      # one-phase kernels: balance the two sides
            first.insert(0, second.pop())
        while first and tot(first) - tot(second) > size[id(first[0])]:
            second.append(first.pop(0))
        hot_l1 = ()
        if self.chunk2 and CHUNK_LAYOUT:
            # the f half of a chunk -- f^2 glue, sparse-multiplication glue, mul3, mul6 and the small routines they call: 56 KB -- as ONE
            # contiguous stretch of the image (it is what has to survive in the 64 KB instruction cache from the first f^2 to the second)
            for name in ("L2_sqr", "L2_sp034_c"):
                sec = self.l2_section[name]
                for lst in (first, second):
                    if any(x is sec for x in lst):
                        lst[:] = [x for x in lst if x is not sec]
                first.append(sec)
            hot_l1 = ("mul3", "mul6", "norm", "redn", "mulxi", "mulxir")
        main.salu(f"s_branch {self.lab('L_exit')}")
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
        if hot_l1:
            order = [o for n_ in hot_l1 for o in order if o[2] == n_] + [o for o in order if o[2] not in hot_l1]
            secs = [self._pro] + first + [l1[n] for _, _, n in order if n in hot_l1] + [main] + self.control_sections + [l1[n] for _, _, n in order if n not in hot_l1] + second + [tail]
        else:
            secs = [self._pro] + first + [main] + self.control_sections + [l1[n] for _, _, n in order] + second + [tail]
        # transfers that cannot reach their target (+-128 KB) go through one-instruction trampolines between the sections
        out, self.n_trampolines = place_with_islands([e.finalize() for e in secs], 131072 - 1024, self.lab)
        assert max_branch_distance(out) < 131072 - 512
        return out

    # ------------------------------------------------------------------ value-bound certification
    # Limb bounds are closed per routine (every store enforces its limit, every multiplication its column sums).
    # VALUE bounds cross routine boundaries through one contract: every routine is generated assuming that whatever it
    # finds in a slot is below V_STORE p, and Prog.to() makes it leave at most V_STORE p in every slot that outlives it
    # (its temporaries may hold up to V_CAP p).  Bounds are monotone in the inputs, so the shipped code is safe whenever
    # the contract holds at every call: certify_values() walks the kernel's data-independent call sequence (NAF digits
    # only), checks the recorded exit bounds of every routine on it against the contract and returns the sequence, which
    # tests/test_kgen4.py compares with the simulator's call log.
    def _check_routine(self, name):
        for k_, v_ in self.l2_exit[name].items():
            assert v_ <= V_STORE, f"{name} leaves {v_} p in {k_}"
        assert self.l2_maxv[name] <= V_CAP, f"{name} stores {self.l2_maxv[name]} p"
        return self.l2_maxv[name]

    def certify_values(self, k_pairs=1, own_pair=True):
        """Returns a report dict (call sequence, largest stored bound); raises on any contract violation.
        own_pair=False: the fixed-G2 kernel on groups without a pair of their own (MODE_NO_OWN): f = 1, then only squarings and table lines."""
        worst, calls, seq = 0.0, 0, []

        bound = {}                               # what the walked routines left in the slots (only read where a routine assumes less than the contract)
        fixed_after = ("L2_dblfirst", "L2_dblmul", "L2_addmul", "L2_addmul_last") if self.fixed else ()

        def run(name, label=None):
            nonlocal worst, calls
            worst = max(worst, self._check_routine(name))
            for k_, v_ in getattr(self, "l2_entry", {}).get(name, {}).items():
                assert bound.get(k_, V_STORE) <= v_, f"{name} assumes {v_} p in {k_}, its caller leaves {bound.get(k_, V_STORE)}"
            bound.update(self.l2_exit[name])
            calls += 1
            seq.append(label or name)
            if name in ("L2_descale", "L2_inv"):
                seq.append("L2_fqinv")           # nested: the Fq inversion (fixed exponent)
            if name in fixed_after:              # the fixed-G2 kernel: the current table line of every fixed pair (the most there can be)
                lines()

        def lines():
            for j in range(self.MAX_FIXED):
                run(f"L2_fix_{j}")
                run(self.fsp_variant[j])

        assert self.main_prog.max_v <= V_STORE, "the main program stores across routine boundaries only"
        fis = self.fission and k_pairs <= FIS_MAX_K
        if self.do_miller and fis:
            run("L2_dblfirst")
            for _ in range(k_pairs - 1):
                run("L2_dblmul")
            for _ in range(k_pairs):                                    # phase 1: every pair's chain of point steps
                run("L2_r2h")
                for i in range(63, -1, -1):
                    if i != 63:
                        run("L2_dbl_f")
                    if SIX_U_PLUS_2_NAF[i] != 0:
                        run("L2_add_f")
                run("L2_h2r")
            if not self.multi:
                run("L2_fpark")
            run("L2_ln_pf")
            for i in range(63, -1, -1):                                 # phase 2: the f loop
                if i != 63:
                    run("L2_sqr")
                    for _ in range(k_pairs):
                        run("L2_sp034_f")
                if SIX_U_PLUS_2_NAF[i] != 0:
                    for _ in range(k_pairs):
                        run("L2_sp235_f")
            if not self.multi:
                run("L2_funpark")
            for _ in range(k_pairs):
                run("L2_addmul")
                run("L2_addmul_last")
        if self.do_miller and not fis and not own_pair:
            assert self.fixed
            bound.update({Prog.key(s_): 1.01 for s_ in self.F})         # f = 1 (the main program's stores)
            lines()
            for i in range(self.naf_first, -1, -1):
                if i != self.naf_first:
                    run("L2_sqr")
                    lines()
                if self.naf[i] != 0:
                    run("L2_fxred")
                    lines()
            for _ in range(2):
                run("L2_fxred")
                lines()
        if self.do_miller and not fis and own_pair:
            run("L2_dblfirst")
            for _ in range(k_pairs - 1):
                run("L2_dblmul")
            dn, an = ("L2_dblmul_s", "L2_addmul_s") if self.multi else ("L2_dblmul", "L2_addmul")
            if self.multi:
                run("L2_prefetch")
            naf, first = self.naf, self.naf_first
            i = first
            while i >= 0:
                if i != first:
                    if self.chunk2 and naf[i] == 0 and i > 0:
                        for name in ("L2_dbl_p", "L2_dbl_p", "L2_sqr", "L2_sp034_c", "L2_sqr", "L2_sp034_c"):
                            run(name)
                        i -= 1
                    else:
                        run("L2_sqr")
                        if self.track:
                            run("L2_sqscale")
                        for _ in range(k_pairs):
                            run(dn)
                if naf[i] != 0:
                    for _ in range(k_pairs):
                        run(an)
                i -= 1
            for _ in range(k_pairs):                                    # per pair: + Q1, then - Q2
                run("L2_addmul")
                run("L2_addmul_last")
            if self.track:
                run("L2_descale")
        if self.do_fexp and not self.helper:
            def mul_by(conj=False):
                run("L2_mul_body", label="L2_mulGc" if conj else "L2_mulG")

            for op in self.fexp_trace:
                if op[0] == "st":
                    seq.append("L2_stG")
                elif op[0] == "ld":
                    seq.append("L2_ldGc" if op[2] else "L2_ldG")
                elif op[0] == "mul":
                    mul_by(op[2])
                elif op[0] == "pf":
                    seq.append("L2_pfB")
                elif op[0] == "mulw":
                    run("L2_mul_body", label="L2_mulGc_w" if op[1] else "L2_mulG_w")
                elif op[0] == "powx":
                    for o, a in self.powx_ops(store_base=op[2]):
                        if o == "cyc":
                            run("L2_cyc", label="L2_cycN" if a > 1 else "L2_cyc")
                        elif o == "call":
                            if a in ("L2_mulL", "L2_mulLc", "L2_mul_body"):
                                run("L2_mul_body", label=a)
                            else:
                                run(a)
                        elif o in ("st", "ld", "pf"):
                            seq.append({"st": "L2_stG", "ld": "L2_ldG", "pf": "L2_pfB"}[o])
                        else:
                            run("L2_mul_body", label={"mul": "L2_mulG", "mulc": "L2_mulGc", "mul_w": "L2_mulG_w", "mulc_w": "L2_mulGc_w"}[o])
                elif op[1] in ("L2_mulL", "L2_mulLc", "L2_mul_body"):
                    run("L2_mul_body", label=op[1])
                else:
                    run(op[1])
        return {"max_stored": worst, "calls": calls, "sequence": seq}

    def certify_helper(self):
        """The helper kernel's loops are data dependent (NAF of the caller's exponent): the contract must hold for ANY order
        of squarings and multiplications, which is exactly what the per-routine check gives."""
        worst = 0.0
        for name in ["L2_inv", "L2_sqrF", "L2_mul_body", "L2_stG", "L2_ldG"] + [f"L2_frob{k}" for k in range(1, 12)]:
            worst = max(worst, self._check_routine(name))
        return {"fixed_point": V_STORE, "max_stored": worst}

    def _cyc_run(self, p):
        """F <- F^(2^n), n cyclotomic squarings with F resident in home blocks 3..8 for the whole run (L1v4.cyc3): entry L2_cyc does
        one, entry L2_cycN S_J + 1 (the x-powers have runs of up to eight squarings between two multiplications)."""
        e = p.e
        e.salu(f"s_mov_b32 s{S_J}, 0")
        e.label(self.lab("L2_cycN"))
        p.reset_tags()
        for i in range(6):
            p._need(mag(p.r_of(self.F[i])) <= 1.0, f"cyclotomic squaring: {self.F[i]} is not normalised")
            p._need(p.v_of(self.F[i]) <= V_CAP / 12, "cyclotomic squaring operand value")       # S = xi b + a enters a product
            p.load(HOME0 + SLOT_DW * (3 + i), self.F[i])
        p.wait()
        loop = self.lab(f"L_cyc_loop_{self.uid()}")
        e.label(loop)
        L1v4(e).cyc3()
        p._count("cyc3")
        e.salu(f"s_sub_u32 s{S_J}, s{S_J}, 1")
        e.salu(f"s_cbranch_scc0 {loop}")
        for i in range(6):
            p.store(HOME0 + SLOT_DW * (3 + i), self.F[i])
            k_ = p.key(self.F[i])
            p.slot_r[k_], p.slot_v[k_] = R_NORM, 0.51
        p.max_v = max(p.max_v, 0.51)
        p.reset_tags()

    def _reduce_f(self, p):
        """F <- the same residues with representatives back in (-0.51 p, 0.51 p) (L1 redn)."""
        for k in range(6):
            p.A(self.F[k]).redn().to(self.F[k])

    def powx_ops(self, store_base):
        """The x-power F <- F^BN_X as a straight list of steps (emitted by _powx_routine, replayed by certify_values):
        ("st" | "ld" | "mul" | "mulc" | "pf", register) with register = an Fq12 scratch register number or "base" (the caller's),
        ("call", L2 routine).  b^19 -- ten of the twelve digit multiplications -- never goes to scratch: conj(b^19) sits in the on-chip
        register LREG (negative digits multiply by it, positive ones conjugate f around the multiplication instead)."""
        ops = [("st", "base")] if store_base else []
        ops += [("pf", "base"), ("cyc", 2), ("st", G_B4), ("cyc", 2),                  # b^4 (kept), b^16 (the operand b is fetched under the squarings)
                ("mulc_w", "base"), ("pf", G_B4), ("st", G_POW[15]),                   # b^15 = b^16 conj(b)
                ("mul_w", G_B4),                                                       # b^19 = b^15 b^4
                ("call", "L2_conjF"), ("call", "L2_stL"), ("call", "L2_conjF")]        # conj(b^19) -> LREG; F = b^19: the top digit
        assert X_DIGITS[-1] == X_HOT
        run = 0                                                                        # squarings since the last multiplication: ONE resident run
        for d in reversed(X_DIGITS[:-1]):
            run += 1
            if d == 0:
                continue
            if abs(d) == X_HOT:
                ops.append(("cyc", run))
                ops += [("call", "L2_mulL")] if d < 0 else [("call", "L2_conjF"), ("call", "L2_mulL"), ("call", "L2_conjF")]
            else:
                reg = "base" if abs(d) == 1 else G_POW[abs(d)]
                ops += [("pf", reg), ("cyc", run), ("mulc_w" if d < 0 else "mul_w", reg)]      # the fetch runs under the squarings
            run = 0
        if run:
            ops.append(("cyc", run))
        return ops

    def _powx_routine(self):
        """F <- F^BN_X for cyclotomic F (base b = F on entry, S_GBASE = its scratch register): the X_DIGITS schedule, unrolled
        (control code only: every step is a call).  Same value as pow_native(a, [BN_X]) (final_exp_native.rs:56-84) for unitary a.
        Entry L3_powx stores b into its register first, L3_powx_ns finds it there already."""
        assert X_DIGITS[-1] == X_HOT
        e = Emitter()
        L = self.lab
        name = {"st": "L2_stG", "ld": "L2_ldG", "mul": "L2_mulG", "mulc": "L2_mulGc", "pf": "L2_pfB", "mul_w": "L2_mulG_w", "mulc_w": "L2_mulGc_w"}
        e.label(L("L3_powx"))
        e.salu(f"s_mov_b32 s{S_PB}, s{S_GBASE}")
        self.emit_call2(e, L('L2_stG'))
        e.salu(f"s_branch {L('L3_powx_go')}")
        e.label(L("L3_powx_ns"))
        e.salu(f"s_mov_b32 s{S_PB}, s{S_GBASE}")
        e.label(L("L3_powx_go"))
        for op, arg in self.powx_ops(store_base=False):
            if op == "call":
                self.emit_call2(e, L(arg))
                continue
            if op == "cyc":                                         # a run of `arg` squarings
                if arg > 1:
                    e.salu(f"s_mov_b32 s{S_J}, {arg - 1}")
                self.emit_call2(e, L('L2_cycN' if arg > 1 else 'L2_cyc'))
                continue
            if op not in ("mul_w", "mulc_w"):                       # (the waiting entries find the operand fetched by "pf")
                if arg == "base":
                    e.salu(f"s_mov_b32 s{S_GBASE}, s{S_PB}")
                else:
                    e.salu(f"s_mul_i32 s{S_GBASE}, s{S_GSTRIDE}, {6 * arg}")
            self.emit_call2(e, L(name[op]))
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
            if EXP_NO_SCRATCH:
                break
            e.salu(f"s_mul_i32 s{S_TMP0}, s{S_GSTRIDE}, {k}")
            e.salu(f"s_add_u32 s{S_TMP0}, s{S_TMP0}, s{S_GBASE}")
            e.salu(f"s_add_u32 s62, s64, s{S_TMP0}")
            e.salu("s_addc_u32 s63, s65, 0")
            for c in range(Prog.N_B128):
                r = land[n] + 4 * c
                e.emit(f"global_load_dwordx4 v[{r}:{r + 3}], v{V_GOFF}, {S_GADDR} offset:{GCHUNK0 + 1024 * c}" + _ldm(), kind="vmem", vw=range(r, r + 4))
            r = land[n] + 16
            e.emit(f"global_load_dwordx2 v[{r}:{r + 1}], v{V_GOFF8}, {S_GADDR} offset:0" + _ldm(), kind="vmem", vw=[r, r + 1])
        e.raw("s_waitcnt vmcnt(0)")
        for n, d in enumerate(dests):
            p.store(land[n], d) if d.kind != "home" or HOME0 + SLOT_DW * d.idx != land[n] else None
            p.slot_r.pop(p.key(d), None)
            p.slot_v.pop(p.key(d), None)
        p.reset_tags()

    def _emit_load_bop(self, e, p):
        """multiplication operand (AGPR slots BOP) <- the Fq12 scratch register at S_GBASE: global loads straight into the
        AGPRs (no landing VGPRs, no moves); nobody waits here."""
        p.reset_tags()
        p.wait()
        for k, dst in enumerate(self.BOP):
            a0 = SLOT_DW * dst.idx
            p.slot_r.pop(p.key(dst), None)
            p.slot_v.pop(p.key(dst), None)
            if EXP_NO_SCRATCH:
                continue
            e.salu(f"s_mul_i32 s{S_TMP0}, s{S_GSTRIDE}, {k}")
            e.salu(f"s_add_u32 s{S_TMP0}, s{S_TMP0}, s{S_GBASE}")
            e.salu(f"s_add_u32 s62, s64, s{S_TMP0}")
            e.salu("s_addc_u32 s63, s65, 0")
            for c in range(Prog.N_B128):
                e.emit(f"global_load_dwordx4 a[{a0 + 4 * c}:{a0 + 4 * c + 3}], v{V_GOFF}, {S_GADDR} offset:{GCHUNK0 + 1024 * c}" + _ldm(), kind="vmem")
            e.emit(f"global_load_dwordx2 a[{a0 + 16}:{a0 + 17}], v{V_GOFF8}, {S_GADDR} offset:0" + _ldm(), kind="vmem")
            p.slot_r.pop(p.key(dst), None)
            p.slot_v.pop(p.key(dst), None)

    def _mulG_routines(self):
        """F <- F * G (G = Fq12 in scratch at S_GBASE); L2_mulGc multiplies by conjugate_fp12(G).  Entry points:
        L2_pfB issues the operand loads and returns (the x-power loop calls it BEFORE the cyclotomic squaring that precedes
        the multiplication: the ~3 us of a 27 KB-per-wave fetch that every wave of the chip issues at about the same time
        run under the squaring); L2_mulG_w / L2_mulGc_w wait for them and multiply; L2_mulG / L2_mulGc do both."""
        # three temporaries: home block 8 and the two AGPR slots that are free here (9 is the Fq inversion's base: not running)
        e, p = self.new_prog([HOME(8), AGPR(13), self.FQINV_BASE] + [GLOB(GLOB_TMP0 + i) for i in range(8)])
        e.label(self.lab("L2_pfB"))
        self._emit_load_bop(e, p)
        e.salu(f"s_setpc_b64 {S_RET2}")
        # operand = the on-chip register: copied into the operand slots (138 instructions, no memory wait to speak of);
        # L2_mulLc multiplies by its conjugate
        for name, conj in (("L2_mulL", False), ("L2_mulLc", True)):
            e.label(self.lab(name))
            p.reset_tags()
            for i in range(6):
                p.A(self.LREG[i])
                if conj and i % 2:
                    p.neg()
                p.to(self.BOP[i])
            p.wait()
            e.salu(f"s_branch {self.lab('L2_mul_body')}")
        e.label(self.lab("L2_mulGc"))
        self._emit_load_bop(e, p)
        e.label(self.lab("L2_mulGc_w"))
        e.raw("s_waitcnt vmcnt(0)")
        p.reset_tags()
        for i in (1, 3, 5):
            p.A(self.BOP[i]).neg().to(self.BOP[i])
        p.wait()
        e.salu(f"s_branch {self.lab('L2_mul_body')}")
        e.label(self.lab("L2_mulG"))
        p.reset_tags()
        self._emit_load_bop(e, p)
        e.label(self.lab("L2_mulG_w"))
        e.raw("s_waitcnt vmcnt(0)")
        e.label(self.lab("L2_mul_body"))
        p.reset_tags()
        p.slot_r.clear()
        p.slot_v.clear()
        p.fq12_mul(self.F, self.BOP)
        p.wait()
        e.salu(f"s_setpc_b64 {S_RET2}")
        self.sections.append(e)
        self.l2_exit["L2_mul_body"] = {k: v for k, v in p.slot_v.items() if k not in p.temp_keys and k[0] != "home"}
        self.l2_maxv["L2_mul_body"] = p.max_v

    # ---------------------------------------------------------------------------------------------
    def prologue(self, main):
        e = Emitter()
        self._pro = e
        e.salu(f"s_mov_b64 {S_G1}, %0")
        e.salu(f"s_mov_b64 {S_G2}, %1")
        e.salu(f"s_mov_b64 {S_FIN}, %2")
        e.salu(f"s_mov_b64 {S_OUT}, %3")
        e.salu(f"s_mov_b32 s{S_N}, %4")
        if self.s_mode is None:
            e.salu(f"s_mov_b32 s{S_K}, %5")                        # (k_op: k = op | power << 8 | naf_len << 16; limb-major I/O only)
        else:
            e.salu(f"s_and_b32 s{S_K}, %5, 0x0fffffff")
            e.salu(f"s_lshr_b32 s{self.s_mode}, %5, 28")
        e.salu(f"s_mov_b32 s{S_GSTRIDE}, " + (f"{WG_SLOT_PITCH}" if SCRATCH_WG else "%7"))
        e.salu(f"s_mov_b64 {S_STATUS}, %8")
        e.salu(f"s_mov_b32 s{S_ITEM}, %10")
        e.salu(f"s_mov_b32 s{S_GRID}, %11")
        # scratch base of this workgroup: scratch + block * 256 * 72 ; lane offset = tid * 72
        # (64-bit product: with the line area of the split loop a full grid of k = 4 scratch blocks passes 4 GiB)
        e.salu(f"s_mul_i32 s{S_TMP0}, %10, " + ("%7" if SCRATCH_WG else f"{BLOCK * SLOT_BYTES}"))
        e.salu(f"s_mul_hi_u32 s{S_TMP1}, %10, " + ("%7" if SCRATCH_WG else f"{BLOCK * SLOT_BYTES}"))
        e.salu(f"s_mov_b64 {S_SCRATCH}, %6")
        e.salu(f"s_add_u32 s64, s64, s{S_TMP0}")
        e.salu(f"s_addc_u32 s65, s65, s{S_TMP1}")
        # scratch slot of a workgroup = 4 waves x 4608 B; inside a wave's part the 64 lanes' 8-byte tails come first (512 B),
        # then the four 16-byte chunks as [chunk][lane] (1 KiB each): every slot access instruction touches ONE contiguous
        # 1 KiB (16 cache lines) instead of 64 lines at a 72-byte lane stride
        e.emit(f"v_lshrrev_b32_e32 v{V_GOFF}, 6, %9", vw=[V_GOFF])
        e.emit(f"v_mul_u32_u24_e32 v{V_GOFF}, {64 * SLOT_BYTES}, v{V_GOFF}", vw=[V_GOFF])
        e.emit(f"v_and_b32_e32 v{V_GOFF8}, 63, %9", vw=[V_GOFF8])
        e.emit(f"v_lshl_add_u32 v{V_GOFF8}, v{V_GOFF8}, 3, v{V_GOFF}", vw=[V_GOFF8])
        e.emit(f"v_and_b32_e32 v{V_IDX}, 63, %9", vw=[V_IDX])
        e.emit(f"v_lshl_add_u32 v{V_GOFF}, v{V_IDX}, 4, v{V_GOFF}", vw=[V_GOFF])
        e.emit(f"v_lshlrev_b32_e32 v{V_LDS}, 4, %9", vw=[V_LDS])
        e.emit(f"v_add_u32_e32 v{V_LDS + 1}, 0x10000, v{V_LDS}", vw=[V_LDS + 1])
        e.emit(f"v_lshlrev_b32_e32 v{V_LTAIL}, 3, %9", vw=[V_LTAIL])
        e.emit(f"v_add_u32_e32 v{V_LTAIL}, 0x{N_LDS_SLOTS * Prog.N_B128 * 4096:x}, v{V_LTAIL}", vw=[V_LTAIL])
        e.emit(f"v_mov_b32_e32 v{V_TID}, %9", vw=[V_TID])
        for i in range(NL):
            e.salu(f"s_mov_b32 s{S_P + i}, {hx(P_L[i])}")
        e.salu(f"s_mov_b32 s{S_N0}, 0x{N0P:x}")
        e.salu(f"s_mov_b32 s{S_REDN}, 0x{REDN_C:x}")
        e.salu(f"s_mov_b32 s{S_M30}, -30")
        if K4_DIGIT_ADD:
            e.salu(f"s_mov_b32 s{K4_S_HALF}, 0x{1 << (LB - 1):x}")
            e.salu(f"s_mov_b32 s{K4_S_HALF + 1}, 0")
        nz, neg = naf_masks(self.naf[:64])
        e.salu(f"s_mov_b32 s68, 0x{nz & 0xFFFFFFFF:x}")
        e.salu(f"s_mov_b32 s69, 0x{nz >> 32:x}")
        e.salu(f"s_mov_b32 s70, 0x{neg & 0xFFFFFFFF:x}")
        e.salu(f"s_mov_b32 s71, 0x{neg >> 32:x}")
        e.salu(f"s_lshl_b32 s{S_NSTRIDE}, s{S_N}, 3")
        e.salu(f"s_add_u32 s{S_NITEMS}, s{S_N}, 255")
        e.salu(f"s_lshr_b32 s{S_NITEMS}, s{S_NITEMS}, 8")
        e.emit(f"v_mov_b32_e32 v{V_FLAG}, 0", vw=[V_FLAG])
        e.salu(f"s_branch {self.lab('L_main')}")

    # ---------------------------------------------------------------------------------------------
    def io_walk_begin(self, e, base, words=None, out=False):
        """The walk over the words of the lane's element of the batch at `base` starts.  Limb-major batches (the engine's own layout): word w
        of element i at (w n + i) * 8 -- V_IDX8 = i * 8 is the lane's offset, consecutive words lie S_NSTRIDE = 8 n apart.  ELEMENT-major
        batches (`words` per element: what the reference's callers hold, src/pairing.rs:20, miller_loop_native.rs:324; S_MODE says which side
        of the launch is): word w of element i at (i words + w) * 8.  Same loads and stores either way: only the lane offset and the step
        differ (576 bytes per pairing against 2.3 M multiply-adds: coalescing is not what matters here)."""
        e.salu(f"s_mov_b64 {S_IOADDR}, {base}")
        if words is None or self.s_mode is None:
            e.salu(f"s_mov_b32 s{S_IOSTRIDE}, s{S_NSTRIDE}")
            e.emit(f"v_mov_b32_e32 v{V_IOOFF}, v{V_IDX8}", vw=[V_IOOFF])
            return
        e.salu(f"s_bitcmp1_b32 s{self.s_mode}, {MODE_OUT_ELEMS if out else MODE_IN_ELEMS}")
        e.salu(f"s_cselect_b32 s{S_IOSTRIDE}, 8, s{S_NSTRIDE}")
        e.salu(f"s_cselect_b32 s{S_TMP0}, {words}, 1")
        e.emit(f"v_mul_lo_u32 v{V_IOOFF}, v{V_IDX8}, s{S_TMP0}", vw=[V_IOOFF])

    def io_walk_next(self, e):
        e.salu(f"s_add_u32 s88, s88, s{S_IOSTRIDE}")
        e.salu("s_addc_u32 s89, s89, 0")

    def io_load_fq(self, e, reg0):
        """Loads one Fq (4 u64 limbs of the SoA batch at the walking address) into v[reg0:reg0+7]."""
        for l in range(4):
            e.emit(f"global_load_dwordx2 v[{reg0 + 2 * l}:{reg0 + 2 * l + 1}], v{V_IOOFF}, {S_IOADDR}", kind="vmem", vw=[reg0 + 2 * l, reg0 + 2 * l + 1])
            self.io_walk_next(e)

    def io_store_fq(self, e, reg0):
        for l in range(4):
            e.emit(f"global_store_dwordx2 v{V_IOOFF}, v[{reg0 + 2 * l}:{reg0 + 2 * l + 1}], {S_IOADDR}", kind="vmem")
            self.io_walk_next(e)

    def zero_block(self, e, blk, n=SLOT_DW):
        for i in range(n):
            e.emit(f"v_mov_b32_e32 v{blk + i}, 0", vw=[blk + i])

    def one_into_A(self, e):
        w = bal_limbs(mont4(1))
        for i in range(NL):
            e.emit(f"v_mov_b32_e32 v{A0 + i}, {hx(w[i])}", vw=[A0 + i])
        self.zero_block(e, A0 + NL, NL)

    def cvt_call(self, e, name):
        e.salu(f"s_call_b64 {S_RET1}, {self.labels[name]}")

    def gsel(self, e, j):
        """S_GBASE <- byte offset of Fq12 scratch register j."""
        e.salu(f"s_mul_i32 s{S_GBASE}, s{S_GSTRIDE}, {6 * j}")

    def call2(self, e, name):
        self.emit_call2(e, self.lab(name))

    def emit_call2(self, e, label):
        """s_call_b64 S_RET2, label -- in profiling builds bracketed by cycle stamps that are accumulated per routine"""
        if not PROFILE_L2 or self.multi:
            e.salu(f"s_call_b64 {S_RET2}, {label}")
            return
        rid = profile_id(label)
        e.salu("s_memtime s[72:73]")
        e.raw("s_waitcnt lgkmcnt(0)")
        e.salu(f"s_call_b64 {S_RET2}, {label}")
        e.salu("s_memtime s[60:61]")
        e.raw("s_waitcnt lgkmcnt(0)")
        e.salu("s_sub_u32 s60, s60, s72")
        e.salu("s_subb_u32 s61, s61, s73")
        e.raw(f"v_readlane_b32 s74, v248, {rid}")
        e.raw(f"v_readlane_b32 s75, v249, {rid}")
        e.raw("s_nop 3")
        e.salu("s_add_u32 s74, s74, s60")
        e.salu("s_addc_u32 s75, s75, s61")
        e.raw("s_nop 3")
        e.raw(f"v_writelane_b32 v248, s74, {rid}")
        e.raw(f"v_writelane_b32 v249, s75, {rid}")
        e.raw(f"v_readlane_b32 s74, v250, {rid}")
        e.raw("s_nop 3")
        e.salu("s_add_u32 s74, s74, 1")
        e.raw("s_nop 3")
        e.raw(f"v_writelane_b32 v250, s74, {rid}")

    def io_load_fq2_into_A(self, e, p, c1_present=True):
        """Loads c0 (and c1) of the SoA batch at the walking address, converts to internal form -> block A."""
        if c1_present:
            self.io_load_fq(e, B0)                 # c0 packed -> v[18:25] (block B as staging)
            self.io_load_fq(e, A0)                 # c1 packed -> v[0:7]
            e.raw("s_waitcnt vmcnt(0)")
            self.cvt_call(e, "cvtin")              # A.c0 <- internal(c1)
            for i in range(NL):
                e.emit(f"v_mov_b32_e32 v{A0 + NL + i}, v{A0 + i}", vw=[A0 + NL + i])
            for i in range(8):
                e.emit(f"v_mov_b32_e32 v{A0 + i}, v{B0 + i}", vw=[A0 + i])
            self.cvt_call(e, "cvtin")
        else:
            self.io_load_fq(e, A0)
            e.raw("s_waitcnt vmcnt(0)")
            self.cvt_call(e, "cvtin")
            self.zero_block(e, A0 + NL, NL)
        p.set_A_fresh()
        p.tagB = None

    def _dbl_first(self, p):
        """i = 63: R = Q -> 2Q, f = dense(tangent line) (miller_loop_native.rs:127-149); scale stays 1."""
        p.dbl_step(self.R, (self.PX, self.PY), self.LINE, scale=None)
        p.mov(self.F[0], self.LINE[0])
        p.mov(self.F[3], self.LINE[1])
        p.mov(self.F[4], self.LINE[2])
        p.wait()
        self.zero_block(p.e, A0)
        p.set_A_fresh(0.0)
        for k in (1, 2, 5):
            p.to(self.F[k])

    FQINV_WINDOW = 3
    FQINV_WIDE_M = bool(int(os.environ.get("KGEN_FQINV_WIDE_M", "1")))
    INV_SAFEGCD = bool(int(os.environ.get("KGEN_INV_SAFEGCD", "1")))   # A/B switch: the Fq inversion by divsteps (L1v4.fq_inv_safegcd) instead of the Fermat chain
    FQINV_OUT_V = 0.56 if INV_SAFEGCD else (4.2 if FQINV_WIDE_M else 0.51)       # value bound (units of p) of the inverse left in block A

    @staticmethod
    def fqinv_schedule(w):
        """Sliding-window schedule of a^(p-2): (first odd value, [(squarings, odd multiplier)] from the top bit down)."""
        bits = bin(P_INT - 2)[2:]
        n, i, out, first = len(bits), 0, [], None
        while i < n:
            if bits[i] == "0":
                j = i
                while j < n and bits[j] == "0":
                    j += 1
                zeros, i = j - i, j
            else:
                zeros = 0
            if i >= n:
                out.append((zeros, 0))                    # trailing zeros (none for p - 2, which is odd)
                break
            j = min(i + w, n)
            while bits[j - 1] == "0":
                j -= 1
            val = int(bits[i:j], 2)
            if first is None:
                first = val
            else:
                out.append((zeros + (j - i), val))
            i = j
        return first, out

    def _fq_inv(self, p):
        """A.c0 <- A.c0^(p-2) (Fermat; fixed exponent, uniform control flow).  Input / output normalised.
        Sliding window of three bits over the odd powers a, a^3, a^5, a^7 (a, a^3 in the AGPR slot FQINV_BASE, a^5, a^7 in the top
        eighteen pool registers): 253 squarings -- each 45 limb products, L1v4.fips_sq -- and 56 + 4 multiplications instead of
        253 and 109 full products.  Called with S_RET2; uses S_RET1 for its own two subroutines and s[60:61] as scratch."""
        e = p.e
        L = self.lab
        uid = self.uid()
        if self.INV_SAFEGCD:
            # round 4: Bernstein-Yang divsteps, 17.5 k instructions (most of them 32-bit logic) against the chain's 87 k slots with
            # their 30 k multiply-adds; f, d, e of the iteration in A.c1 and block B, the rest in the pool
            p.wait()
            p.tagA = p.tagB = None
            L1v4(e).fq_inv_safegcd(L1v4.blk(A0, 0), L1v4.blk(A0, 1), L1v4.blk(B0, 0), L1v4.blk(B0, 1), L1v4.blk(A0, 0), f"s{S_TMP0}",
                                   L(f"L_fqinv_sg_{uid}"))
            p.set_A_fresh(self.FQINV_OUT_V)
            return
        base = self.FQINV_BASE
        first, sched = self.fqinv_schedule(self.FQINV_WINDOW)
        assert self.FQINV_WINDOW == 3 and all(v in (1, 3, 5, 7) for _, v in sched) and first in (1, 3, 5, 7)
        RA, X2, RB = L1v4.blk(A0, 0), L1v4.blk(A0, 1), L1v4.blk(B0, 0)
        T5, T7 = list(range(58, 67)), list(range(67, 76))          # the top of the pool: kept out of the leaf routines' hands below
        a9 = SLOT_DW * base.idx

        def l1():
            g = L1v4(e)
            g.pool.free_regs = [r for r in g.pool.free_regs if r < 58]
            return g

        # (round 4: every product of the chain with 32-bit Montgomery digits, L1v4.fips wide_m: one normalised value times another
        # is 9 units per column, 18 for the doubled operand of a squaring, and a result of v^2 / 169.6 + 4 p stays at 4.1 p from
        # step to step -- nine instructions less per product, 313 products)
        W = self.FQINV_WIDE_M
        p.wait()
        p.tagA = p.tagB = None
        for i in range(NL):
            e.emit(f"v_accvgpr_write_b32 a{a9 + i}, v{RA[i]}")                       # a
        l1().fips_sq(RA, X2, wide_m=W)                                              # a^2
        l1().fips([(RA, X2)], RB, wide_m=W)                                         # a^3
        for i in range(NL):
            e.emit(f"v_accvgpr_write_b32 a{a9 + NL + i}, v{RB[i]}")
        l1().fips([(RB, X2)], T5, wide_m=W)                                         # a^5
        l1().fips([(T5, X2)], T7, wide_m=W)                                         # a^7
        src = {1: None, 3: RB, 5: T5, 7: T7}[first]                                # the top window
        if src is not None:
            for i in range(NL):
                e.emit(f"v_mov_b32_e32 v{RA[i]}, v{src[i]}", vw=[RA[i]])
        sq_l = L(f"L_fqinv_sq_{uid}")
        mul_l = {v: L(f"L_fqinv_m{v}_{uid}") for v in (1, 3, 5, 7)}
        for nsq, val in sched:
            e.salu(f"s_mov_b32 s{S_TMP0}, {nsq - 1}")
            e.salu(f"s_call_b64 {S_RET1}, {sq_l}")
            if val:
                e.salu(f"s_call_b64 {S_RET1}, {mul_l[val]}")
        e.salu(f"s_branch {L(f'L_fqinv_done_{uid}')}")
        # subroutine: S_TMP0 + 1 squarings of RA
        e.label(sq_l)
        l1().fips_sq(RA, RA, wide_m=W)
        e.salu(f"s_sub_u32 s{S_TMP0}, s{S_TMP0}, 1")
        e.salu(f"s_cbranch_scc0 {sq_l}")
        e.salu(f"s_setpc_b64 {S_RET1}")
        # subroutines: RA *= a^v
        for v in (1, 3, 5, 7):
            e.label(mul_l[v])
            if v in (1, 3):
                off = a9 + (NL if v == 3 else 0)
                for i in range(NL):
                    e.emit(f"v_accvgpr_read_b32 v{RB[i]}, a{off + i}", vw=[RB[i]])
                l1().fips([(RA, RB)], RA, wide_m=W)
            else:
                l1().fips([(RA, T5 if v == 5 else T7)], RA, wide_m=W)
            e.salu(f"s_setpc_b64 {S_RET1}")
        e.label(L(f"L_fqinv_done_{uid}"))
        p.set_A_fresh(self.FQINV_OUT_V)

    def _fq2_inv_inline(self, p, src, dst):
        """dst <- 1/src (Fq2): conj(src) / (c0^2 + c1^2); sets the zero-divisor flag when the norm is 0."""
        e = p.e
        n0, tmp = p.tmp(), p.tmp()
        p.A(src)
        p.call("fqsqr")
        p.to(n0)                                                  # n0.c0 = c0^2
        p.A(src)
        p.wait()
        for i in range(NL):
            e.emit(f"v_mov_b32_e32 v{A0 + i}, v{A0 + NL + i}", vw=[A0 + i])
        rS, vS = p.rA, p.vA
        p.tagA = None
        p.rA, p.vA = rS, vS
        p.call("fqsqr")
        p.add(n0).redn()                                         # A.c0 = c0^2 + c1^2, reduced (the Fermat loop squares it unchecked)
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
        p.set_A_fresh(self.FQINV_OUT_V)
        p.tagB = None
        p.to(tmp)
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
        for i in range(6):
            g = frob_const(k, i)
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

    # ---------------------------------------------------------------------------------------------
    def main_body(self, e):
        L = self.lab
        e.label(L("L_main"))
        if PROFILE_L2:
            assert CLOCK_STAMP
            for r in (248, 249, 250):
                e.emit(f"v_mov_b32_e32 v{r}, 0")
        if CLOCK_STAMP:
            assert SCRATCH_WG
            # (no scalar register is free in every kernel any more: the start stamps wait in four lanes of v251)
            e.salu("s_memtime s[60:61]")
            e.salu("s_memrealtime s[88:89]")
            e.raw("s_waitcnt lgkmcnt(0)")
            for i, sr in enumerate((60, 61, 88, 89)):
                e.raw(f"v_writelane_b32 v251, s{sr}, {i}")
        e.label(L("L_item"))
        e.salu(f"s_cmp_ge_u32 s{S_ITEM}, s{S_NITEMS}")
        e.salu(f"s_cbranch_scc1 {L('L_done')}")
        e.salu(f"s_lshl_b32 s{S_TMP0}, s{S_ITEM}, 8")
        e.emit(f"v_add_u32_e32 v{V_IDX}, s{S_TMP0}, v{V_TID}", vw=[V_IDX])
        e.salu(f"s_sub_u32 s{S_TMP1}, s{S_N}, 1")
        e.emit(f"v_min_u32_e32 v{V_IDX8}, s{S_TMP1}, v{V_IDX}", vw=[V_IDX8])
        e.emit(f"v_lshlrev_b32_e32 v{V_IDX8}, 3, v{V_IDX8}", vw=[V_IDX8])
        e.emit(f"v_mov_b32_e32 v{V_FLAG}, 0", vw=[V_FLAG])
        p = Prog(e, self.labels)
        p.set_temps(self.miller_temps())
        p.temp_keys = frozenset()            # the main program's stores all cross routine boundaries
        p.norm_keys = self.norm_keys("miller")
        self.main_prog = p
        if self.generate or self.subcheck or self.lines:
            (self.generate_main if self.generate else (self.subcheck_main if self.subcheck else self.lines_main))(e, p)
            e.salu(f"s_add_u32 s{S_ITEM}, s{S_ITEM}, s{S_GRID}")
            e.salu(f"s_branch {L('L_item')}")
            e.label(L("L_done"))
            return
        if self.do_miller and self.multi:
            self.miller_main_multi(e, p)
        elif self.fixed:
            self.miller_main_fixed(e, p)
        elif self.do_miller:
            self.miller_main(e, p)
        else:
            self.load_fq12_into_F(e, p, S_FIN)
        if self.do_miller and self.do_fexp and self.F_IN_AGPR and not self.F_AGPR_FEXP:      # phase boundary: f moves from the AGPR slots to LDS
            p.reset_tags()
            src = self.F
            self._phase = "fexp"
            p.norm_keys = self.norm_keys("fexp")
            for a_, l_ in zip(src, self.F):
                p.mov(l_, a_)
            p.reset_tags()
        if self.helper:
            self.helper_main(e, p)
        elif self.do_fexp:
            self.fexp_main(e, p)
        self.store_out(e, p)
        self._phase = "miller"
        e.salu(f"s_add_u32 s{S_ITEM}, s{S_ITEM}, s{S_GRID}")
        e.salu(f"s_branch {L('L_item')}")
        e.label(L("L_done"))
        if CLOCK_STAMP:
            e.salu("s_memtime s[60:61]")
            e.salu("s_memrealtime s[88:89]")
            e.raw("s_waitcnt lgkmcnt(0)")
            for i in range(4):
                e.raw(f"v_readlane_b32 s{72 + i}, v251, {i}")
            e.raw("s_nop 3")
            e.salu("s_sub_u32 s60, s60, s72")
            e.salu("s_subb_u32 s61, s61, s73")
            e.salu("s_sub_u32 s88, s88, s74")
            e.salu("s_subb_u32 s89, s89, s75")
            for i, sr in enumerate((60, 61, 88, 89)):
                e.emit(f"v_mov_b32_e32 v{36 + i}, s{sr}", vw=[36 + i])
            e.salu(f"s_sub_u32 s72, %7, {STAMP_OFFSET_FROM_END}")
            e.emit(f"v_lshrrev_b32_e32 v40, 10, v{V_LDS}", vw=[40])          # wave number (V_LDS = tid * 16)
            e.emit("v_lshlrev_b32_e32 v40, 4, v40", vw=[40])
            e.emit("v_add_u32_e32 v40, s72, v40", vw=[40])
            e.emit(f"global_store_dwordx4 v40, v[36:39], {S_SCRATCH}", kind="vmem")
            if PROFILE_L2:      # [wave][counter][lane]: wave * 768 + counter * 256 + lane * 4
                e.salu(f"s_sub_u32 s72, %7, {PROFILE_OFFSET_FROM_END}")
                e.emit(f"v_lshrrev_b32_e32 v40, 10, v{V_LDS}", vw=[40])
                e.emit("v_mul_u32_u24_e32 v40, 768, v40", vw=[40])
                e.emit(f"v_lshrrev_b32_e32 v41, 4, v{V_LDS}", vw=[41])
                e.emit("v_and_b32_e32 v41, 63, v41", vw=[41])
                e.emit("v_lshl_add_u32 v40, v41, 2, v40", vw=[40])
                e.emit("v_add_u32_e32 v40, s72, v40", vw=[40])
                for i, r in enumerate((248, 249, 250)):
                    e.emit(f"global_store_dword v40, v{r}, {S_SCRATCH} offset:{256 * i}", kind="vmem")
            e.raw("s_waitcnt vmcnt(0)")

    def load_fq12_into_F(self, e, p, ptr):
        """F <- the lane's MyFq12 of the SoA batch at `ptr` (components 0..5 are the c0 parts of w^0..w^5, 6..11 the c1
        parts): two passes over the planes, c0 parts first into AGPR staging (the operand slots, free at that point)."""
        stage = SLOT_DW * self.BOP[0].idx
        self.io_walk_begin(e, ptr, 48)
        for k in range(6):
            self.io_load_fq(e, A0)
            e.raw("s_waitcnt vmcnt(0)")
            self.cvt_call(e, "cvtin")
            for i in range(NL):
                e.emit(f"v_accvgpr_write_b32 a{stage + NL * k + i}, v{A0 + i}")       # c0 of coefficient k
        for k in range(6):
            self.io_load_fq(e, A0)
            e.raw("s_waitcnt vmcnt(0)")
            self.cvt_call(e, "cvtin")
            for i in range(NL):
                e.emit(f"v_mov_b32_e32 v{A0 + NL + i}, v{A0 + i}", vw=[A0 + NL + i])
            for i in range(NL):
                e.emit(f"v_accvgpr_read_b32 v{A0 + i}, a{stage + NL * k + i}", vw=[A0 + i])
            p.set_A_fresh()
            p.to(self.F[k])
        p.reset_tags()

    # ---------------------------------------------------------------------------------------------
    # batched helpers: k argument = op | power << 8 | naf_len << 16
    OP_MUL, OP_FROB, OP_POW = 0, 1, 2

    def helper_main(self, e, p):
        L = self.lab

        def gsel(j):
            self.gsel(e, j)

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
        c2("L2_inv")
        gsel(1); c2("L2_stG")                                     # G1 = 1/a   (`res / a` on a -1 digit, final_exp_native.rs:72-75)
        gsel(0); c2("L2_ldG")                                     # res = a (the top digit)
        e.label(L("L_h_noinv"))
        e.salu(f"s_lshr_b32 s{S_J}, s{S_K}, 16")
        e.salu(f"s_sub_u32 s{S_J}, s{S_J}, 2")
        e.salu(f"s_cbranch_scc1 {L('L_h_done')}")                 # a single digit: a^1
        e.label(L("L_h_ploop"))
        c2("L2_sqrF")
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
        c2("L2_mulG")
        e.label(L("L_h_pnext"))
        e.salu(f"s_sub_u32 s{S_J}, s{S_J}, 1")
        e.salu(f"s_cbranch_scc0 {L('L_h_ploop')}")
        e.label(L("L_h_done"))
        p.reset_tags()

    def _twist_consts(self):
        c2, c3 = twist_consts()
        return Const(c2[0], c2[1], "c2"), Const(c3[0], c3[1], "c3")

    def _frobenius_points(self, p):
        """Q1 = pi(Q) = (c2 conj(x), c3 conj(y)) -> S ; -Q2 = (c2 conj(Q1.x), c3 neg_conj(Q1.y)) -> the Q slots, which are dead
        from here on (miller_loop_native.rs:298-312).  (The sparse multiplication inside L2_addmul uses the S slots as
        temporaries, so -Q2 must exist before the first call.)"""
        C2, C3 = self._twist_consts()
        p.reset_tags()
        p.A(self.QX).conj().mul(C2).to(self.SX)
        p.A(self.QY).conj().mul(C3).to(self.SY)
        p.A(self.SX).conj().mul(C2).to(self.QX)
        p.A(self.SY).conj().neg().mul(C3).to(self.QY)

    def _select_pm_q(self, e, p):
        """S <- +-Q by the sign of the current digit"""
        L = self.lab
        p.reset_tags()
        p.mov(self.SX, self.QX)
        p.A(self.QY)
        p.wait()
        u = self.uid()
        e.salu(f"s_bitcmp1_b64 {S_NAF_NEG}, s{S_I}")
        e.salu(f"s_cbranch_scc0 {L(f'L_mpos_{u}')}")
        e.salu(f"s_call_b64 {S_RET1}, {self.labels['neg']}")
        e.label(L(f"L_mpos_{u}"))
        p.tagA = None
        p.to(self.SY)

    def _fission_loops(self, e, p, single):
        """The main loop of the Miller loop, split (FISSION).  PHASE 1: for every pair the whole chain of point steps, R in home
        registers, one line triple per step into the line area (layout [step][pair][3 slots]: phase 2 reads it sequentially).
        PHASE 2: the f loop.  single: the one-pair kernel (resident point state; RZ, PX, PY parked in LDS during phase 2);
        otherwise the pairs' state is in their scratch blocks (S_K pairs), as the first steps left it and the end steps expect it."""
        L = self.lab
        u = self.uid()
        if single:
            e.salu(f"s_mul_i32 s{S_LSTEP}, s{S_GSTRIDE}, 3")
            e.salu(f"s_mul_i32 s{S_LOFF}, s{S_GSTRIDE}, {N_GSLOTS}")
        else:
            e.salu(f"s_mul_i32 s{S_TMP0}, s{S_K}, 3")
            e.salu(f"s_mul_i32 s{S_LSTEP}, s{S_GSTRIDE}, s{S_TMP0}")                     # a step's triples of all pairs lie together

        def chain():
            if not single:
                self.pair_in(e, p, with_q=True)
                # line area: behind the pairs' blocks; pair j starts at triple j
                e.salu(f"s_mul_i32 s{S_TMP0}, s{S_K}, 7")
                e.salu(f"s_add_u32 s{S_TMP0}, s{S_TMP0}, {self.PAIR_SLOT0}")
                e.salu(f"s_mul_i32 s{S_TMP1}, s{S_JP}, 3")
                e.salu(f"s_add_u32 s{S_TMP0}, s{S_TMP0}, s{S_TMP1}")
                e.salu(f"s_mul_i32 s{S_LOFF}, s{S_TMP0}, s{S_GSTRIDE}")
            self.call2(e, "L2_r2h")
            e.salu(f"s_mov_b32 s{S_I}, 63")
            e.label(L(f"L_f1_loop_{u}"))
            e.salu(f"s_cmp_eq_u32 s{S_I}, 63")
            e.salu(f"s_cbranch_scc1 {L(f'L_f1_skip_{u}')}")
            self.call2(e, "L2_dbl_f")
            e.label(L(f"L_f1_skip_{u}"))
            e.salu(f"s_bitcmp1_b64 {S_NAF_NZ}, s{S_I}")
            e.salu(f"s_cbranch_scc0 {L(f'L_f1_noadd_{u}')}")
            self._select_pm_q(e, p)
            self.call2(e, "L2_add_f")
            e.label(L(f"L_f1_noadd_{u}"))
            e.salu(f"s_sub_u32 s{S_I}, s{S_I}, 1")
            e.salu(f"s_cbranch_scc0 {L(f'L_f1_loop_{u}')}")
            self.call2(e, "L2_h2r")
            if not single:
                self.pair_out(e, p)
        if single:
            chain()
            self.call2(e, "L2_fpark")
            e.salu(f"s_mul_i32 s{S_LOFF2}, s{S_GSTRIDE}, {N_GSLOTS}")
        else:
            self.pair_loop(e, f"fis1_{u}", chain)
            e.salu(f"s_mul_i32 s{S_TMP0}, s{S_K}, 7")
            e.salu(f"s_add_u32 s{S_TMP0}, s{S_TMP0}, {self.PAIR_SLOT0}")
            e.salu(f"s_mul_i32 s{S_LOFF2}, s{S_TMP0}, s{S_GSTRIDE}")
            e.raw("s_waitcnt vmcnt(0)")                                                    # every line has been written
        p.reset_tags()
        if single:
            e.salu(f"s_mov_b32 s{S_LCNT}, {FIS_STEPS}")
        else:
            e.salu(f"s_mul_i32 s{S_LCNT}, s{S_K}, {FIS_STEPS}")
        self.call2(e, "L2_ln_pf")
        per_pair = (lambda name: self.call2(e, name)) if single else (lambda name: self.pair_loop(e, f"{name}_{u}", lambda: self.call2(e, name)))
        e.salu(f"s_mov_b32 s{S_I}, 63")
        e.label(L(f"L_f2_loop_{u}"))
        e.salu(f"s_cmp_eq_u32 s{S_I}, 63")
        e.salu(f"s_cbranch_scc1 {L(f'L_f2_skip_{u}')}")
        self.call2(e, "L2_sqr")
        per_pair("L2_sp034_f")
        e.label(L(f"L_f2_skip_{u}"))
        e.salu(f"s_bitcmp1_b64 {S_NAF_NZ}, s{S_I}")
        e.salu(f"s_cbranch_scc0 {L(f'L_f2_noadd_{u}')}")
        per_pair("L2_sp235_f")
        e.label(L(f"L_f2_noadd_{u}"))
        e.salu(f"s_sub_u32 s{S_I}, s{S_I}, 1")
        e.salu(f"s_cbranch_scc0 {L(f'L_f2_loop_{u}')}")
        e.raw("s_waitcnt vmcnt(0)")                                                        # the last (unused) prefetch
        if single:
            self.call2(e, "L2_funpark")
        p.reset_tags()

    def miller_main(self, e, p):
        L = self.lab
        self.io_walk_begin(e, S_G1, 8)
        self.io_load_fq2_into_A(e, p, c1_present=False)          # Px
        p.to(self.PX)
        self.io_load_fq2_into_A(e, p, c1_present=False)          # Py
        p.to(self.PY)
        self.io_walk_begin(e, S_G2, 16)
        self.io_load_fq2_into_A(e, p)                            # Q.x
        p.to(self.QX)
        p.to(self.R[0])
        self.io_load_fq2_into_A(e, p)                            # Q.y
        p.to(self.QY)
        p.to(self.R[1])
        self.one_into_A(e)
        p.set_A_fresh()
        p.to(self.R[2])
        if self.track:
            p.to(self.SCALE)
        p.reset_tags()
        self.call2(e, "L2_dblfirst")
        if self.fission:
            self._fission_loops(e, p, single=True)
            return self._miller_end_steps(e, p)
        first = self.naf_first
        e.salu(f"s_mov_b32 s{S_I}, {first}")
        e.label(L("L_mloop"))
        e.salu(f"s_cmp_eq_u32 s{S_I}, {first}")
        e.salu(f"s_cbranch_scc1 {L('L_mskip') if first == 63 else L('L_mnoadd')}")       # (digit 64 of the canonical NAF is zero and outside the masks)
        if self.chunk2:
            # digit i zero and an iteration i - 1 exists: the two doubling steps first (lines parked), then the two f^2 + sparse
            # multiplications; iteration i - 1's addition step (if its digit is not zero) follows below
            e.salu(f"s_bitcmp1_b64 {S_NAF_NZ}, s{S_I}")
            e.salu(f"s_cbranch_scc1 {L('L_msingle')}")
            e.salu(f"s_cmp_eq_u32 s{S_I}, 0")
            e.salu(f"s_cbranch_scc1 {L('L_msingle')}")
            for j in (0, 1):
                e.salu(f"s_mov_b32 s{S_PARK}, {j}")
                self.call2(e, "L2_dbl_p")
            for j in (0, 1):
                self.call2(e, "L2_sqr")
                e.salu(f"s_mov_b32 s{S_PARK}, {j}")
                self.call2(e, "L2_sp034_c")
            e.salu(f"s_sub_u32 s{S_I}, s{S_I}, 1")
            e.salu(f"s_branch {L('L_mskip')}")
            e.label(L("L_msingle"))
        self.call2(e, "L2_sqr")
        if self.track:
            self.call2(e, "L2_sqscale")
        self.call2(e, "L2_dblmul")
        e.label(L("L_mskip"))
        e.salu(f"s_bitcmp1_b64 {S_NAF_NZ}, s{S_I}")
        e.salu(f"s_cbranch_scc0 {L('L_mnoadd')}")
        self._select_pm_q(e, p)
        self.call2(e, "L2_addmul")
        e.label(L("L_mnoadd"))
        e.salu(f"s_sub_u32 s{S_I}, s{S_I}, 1")
        e.salu(f"s_cbranch_scc0 {L('L_mloop')}")
        self._miller_end_steps(e, p)

    def _miller_end_steps(self, e, p):
        """+ pi(Q), then the line through the result and -pi^2(Q) (miller_loop_native.rs:176-187)"""
        self._frobenius_points(p)
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
    def pair_select(self, e):
        """S_GBASE <- byte offset of pair S_JP's scratch block."""
        e.salu(f"s_mul_i32 s{S_TMP0}, s{S_JP}, 7")
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

    def pair_loop(self, e, name, body, alternating=False):
        """for S_JP in 0..k-1: body().  alternating (the streamed passes of the main loop, BOUSTROPHEDON): the pass runs in the
        direction S_DIR (0 -> k-1 or k-1 -> 0) and flips it, so that the pair that closes one pass opens the next."""
        L = self.lab
        if alternating and self.boustrophedon():
            e.salu(f"s_sub_u32 s{S_TMP0}, s{S_K}, 1")
            e.salu(f"s_cmp_gt_i32 s{self.S_DIR}, 0")
            e.salu(f"s_cselect_b32 s{S_JP}, 0, s{S_TMP0}")
            e.salu(f"s_mov_b32 s{self.S_CNT}, s{S_K}")
            e.label(L(f"L_pl_{name}"))
            self.pair_select(e)
            body()
            e.salu(f"s_add_i32 s{S_JP}, s{S_JP}, s{self.S_DIR}")
            e.salu(f"s_sub_u32 s{self.S_CNT}, s{self.S_CNT}, 1")
            e.salu(f"s_cmp_lg_u32 s{self.S_CNT}, 0")
            e.salu(f"s_cbranch_scc1 {L(f'L_pl_{name}')}")
            e.salu(f"s_sub_i32 s{self.S_DIR}, 0, s{self.S_DIR}")
            return
        e.salu(f"s_mov_b32 s{S_JP}, 0")
        e.label(L(f"L_pl_{name}"))
        self.pair_select(e)
        body()
        e.salu(f"s_add_u32 s{S_JP}, s{S_JP}, 1")
        e.salu(f"s_cmp_lt_u32 s{S_JP}, s{S_K}")
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
            e.salu(f"s_lshl_b32 s{S_TMP1}, s{S_JP}, 3")
            e.emit(f"v_add_u32_e32 v{V_IDX8}, s{S_TMP1}, v{V_IDX8}", vw=[V_IDX8])
            self.io_walk_begin(e, S_G1, 8)
            self.io_load_fq2_into_A(e, p, c1_present=False)
            p.to(GlobDyn(0))
            self.io_load_fq2_into_A(e, p, c1_present=False)
            p.to(GlobDyn(1))
            self.io_walk_begin(e, S_G2, 16)
            self.io_load_fq2_into_A(e, p)
            p.to(GlobDyn(2))
            p.to(GlobDyn(4))
            self.io_load_fq2_into_A(e, p)
            p.to(GlobDyn(3))
            p.to(GlobDyn(5))
            self.one_into_A(e)
            p.set_A_fresh()
            p.to(GlobDyn(6))
            p.wait()
            e.salu(f"s_lshl_b32 s{S_TMP1}, s{S_JP}, 3")
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
            e.salu(f"s_cmp_eq_u32 s{S_JP}, 0")
            e.salu(f"s_cbranch_scc0 {L('L_mf_rest')}")
            self.call2(e, "L2_dblfirst")
            e.salu(f"s_branch {L('L_mf_done')}")
            e.label(L("L_mf_rest"))
            self.call2(e, "L2_dblmul")
            e.label(L("L_mf_done"))
            self.pair_out(e, p)

        self.pair_loop(e, "first", first_step)
        if self.fission:                 # groups of up to FIS_MAX_K pairs: the split loop (no pair state on chip, no R stream)
            e.raw("s_waitcnt vmcnt(0)")
            e.salu(f"s_cmp_le_u32 s{S_K}, {FIS_MAX_K}")
            e.salu(f"s_cbranch_scc0 {L('L_mf_streamed')}")
            self._fission_loops(e, p, single=False)
            e.salu(f"s_branch {L('L_mf_ends')}")
            e.label(L("L_mf_streamed"))
        # resident-P mode: the evaluation points move into the LDS slots the one-pair routines above no longer need
        e.salu(f"s_cmp_le_u32 s{S_K}, {self.RES_K}")
        e.salu(f"s_cbranch_scc0 {L('L_mf_nopack')}")
        self.pair_loop(e, "pack", lambda: self._emit_pack_p(e, p))
        e.label(L("L_mf_nopack"))
        if self.r0_resident():                                        # pair 0's R moves on chip for the whole loop
            e.salu(f"s_mov_b32 s{S_JP}, 0")
            self.pair_select(e)
            p.reset_tags()
            e.raw("s_waitcnt vmcnt(0)")                               # (the stores of the first steps have been acknowledged)
            for k_, dst in zip((4, 5, 6), self.R0_LDS):
                p.A(GlobDyn(k_)).to(dst)
            p.wait()
            p.reset_tags()
        if self.r1_slots():                                           # ... and pair 1's resident coordinates (every group has a pair 1)
            e.salu(f"s_mov_b32 s{S_JP}, 1")
            self.pair_select(e)
            p.reset_tags()
            for k_, dst in zip((4, 5, 6), self.r1_slots()):
                p.A(GlobDyn(k_)).to(dst)
            p.wait()
            p.reset_tags()
        e.salu(f"s_mul_i32 s{self.S_GNEXT}, s{S_GSTRIDE}, {self.PAIR_SLOT0}")      # prime the stream: pair 0
        if self.boustrophedon():
            e.salu(f"s_mov_b32 s{self.S_DIR}, 1")
            e.salu(f"s_mov_b32 s{self.S_NEXTP}, 0")
            e.salu(f"s_mov_b32 s{S_JP}, -1")                              # (no pair is "the same as the next one" yet)
        self.call2(e, "L2_prefetch")
        first = self.naf_first
        e.salu(f"s_mov_b32 s{S_I}, {first}")
        e.label(L("L_mloop"))
        e.salu(f"s_cmp_eq_u32 s{S_I}, {first}")
        e.salu(f"s_cbranch_scc1 {L('L_mskip') if first == 63 else L('L_mnoadd')}")
        self.call2(e, "L2_sqr")
        if self.track:
            self.call2(e, "L2_sqscale")

        def dbl_pair():
            self.pair_select_next(e)
            self.call2(e, "L2_dblmul_s")

        self.pair_loop(e, "dbl", dbl_pair, alternating=True)
        e.label(L("L_mskip"))
        e.salu(f"s_bitcmp1_b64 {S_NAF_NZ}, s{S_I}")
        e.salu(f"s_cbranch_scc0 {L('L_mnoadd')}")

        def add_pair():
            self.pair_select_next(e)
            self.call2(e, "L2_addmul_s")

        self.pair_loop(e, "add", add_pair, alternating=True)
        e.label(L("L_mnoadd"))
        e.salu(f"s_sub_u32 s{S_I}, s{S_I}, 1")
        e.salu(f"s_cbranch_scc0 {L('L_mloop')}")

        if self.boustrophedon():
            # the last pass ran upwards (S_DIR has been flipped to -1 behind it) <=> pair k - 1 closed it and its R sits in the
            # prefetch buffer: back to its scratch block for the end steps
            e.salu(f"s_cmp_lt_i32 s{self.S_DIR}, 0")
            e.salu(f"s_cbranch_scc0 {L('L_mb_noflush')}")
            e.salu(f"s_sub_u32 s{S_JP}, s{S_K}, 1")
            self.pair_select(e)
            p.reset_tags()
            buf = self.BUF
            for k_, src in zip((4, 5, 6), (buf["RX"], buf["RY"], buf["RZ"])):
                p.A(src).to(GlobDyn(k_))
            p.wait()
            e.raw("s_waitcnt vmcnt(0)")
            p.reset_tags()
            e.label(L("L_mb_noflush"))
        if self.r0_resident():                                        # ... and back to its scratch block for the end steps
            e.salu(f"s_mov_b32 s{S_JP}, 0")
            self.pair_select(e)
            p.reset_tags()
            for k_, src in zip((4, 5, 6), self.R0_LDS):
                p.A(src).to(GlobDyn(k_))
            p.wait()
            e.raw("s_waitcnt vmcnt(0)")                               # acknowledged before the end steps read the block back
            p.reset_tags()
        if self.r1_slots():
            e.salu(f"s_mov_b32 s{S_JP}, 1")
            self.pair_select(e)
            p.reset_tags()
            for k_, src in zip((4, 5, 6), self.r1_slots()):
                p.A(src).to(GlobDyn(k_))
            p.wait()
            e.raw("s_waitcnt vmcnt(0)")
            p.reset_tags()

        if self.fission:
            e.label(L("L_mf_ends"))

        def end_pair():
            self.pair_in(e, p, with_q=True)
            self._frobenius_points(p)
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


    # ---------------------------------------------------------------------------------------------
    # synthetic inputs: P = [s] G1, Q = [t] G2 by the fixed-base radix-16 method over the table of tools/gen_tables.py
    # (stands in for G1Affine::rand / G2Affine::rand, /root/reference/src/pairing.rs:65-66).
    # kernel arguments: %0 g1_out  %1 g2_out  %2 table  %3 seed (64-bit value)  %4 n
    GEN_WORDS = HOME0 + SLOT_DW * 8            # v[228:235]: the scalars s (4 dwords) and t (4 dwords) ; v[236:243]: SplitMix64 scratch
    GEN_ENTRY_BYTES = 4 * 4 * NL               # x.c0, x.c1, y.c0, y.c1

    def gen_temps(self):
        return [HOME(i) for i in range(8)] + [AGPR(i) for i in (6, 7, 8, 10, 11, 12, 13)] + [GLOB(GLOB_TMP0 + i) for i in range(8)]

    def _to_affine(self, p):
        """S <- (X / Z, Y / Z)"""
        inv, z = p.tmp(), p.tmp()
        p.mov(z, self.R[2])                       # RZ shares its AGPR slot with the Fq-inversion base
        self._fq2_inv_inline(p, z, inv)
        p.rel(z)
        p.A(self.R[0]).mul(inv).to(self.SX)
        p.A(self.R[1]).mul(inv).to(self.SY)
        p.rel(inv)

    def _splitmix(self, e, st, z, t, dst):
        """dst (one VGPR pair: 64 bits) <- next SplitMix64 output; st: the state pair, z / t: scratch pairs (all even-aligned)"""
        P2 = lambda r: f"v[{r}:{r + 1}]"

        def mul64(c):
            e.salu(f"s_mov_b32 s{S_TMP0}, 0x{c & 0xFFFFFFFF:x}")
            e.salu(f"s_mov_b32 s{S_TMP1}, 0x{c >> 32:x}")
            e.emit(f"v_mul_lo_u32 v{t}, v{z + 1}, s{S_TMP0}", vw=[t])
            e.emit(f"v_mul_lo_u32 v{t + 1}, v{z}, s{S_TMP1}", vw=[t + 1])
            e.emit(f"v_mad_u64_u32 {P2(z)}, vcc, v{z}, s{S_TMP0}, 0", w=["vcc"], vw=[z, z + 1])
            e.emit(f"v_add3_u32 v{z + 1}, v{z + 1}, v{t}, v{t + 1}", vw=[z + 1])

        def xorshift(k):
            e.emit(f"v_lshrrev_b64 {P2(t)}, {k}, {P2(z)}", vw=[t, t + 1])
            e.emit(f"v_xor_b32_e32 v{z}, v{z}, v{t}", vw=[z])
            e.emit(f"v_xor_b32_e32 v{z + 1}, v{z + 1}, v{t + 1}", vw=[z + 1])

        e.emit(f"v_add_co_u32_e32 v{st}, vcc, 0x7f4a7c15, v{st}", w=["vcc"], vw=[st])
        e.emit(f"v_mov_b32_e32 v{t}, 0x9e3779b9", vw=[t])                 # (a literal next to the VCC carry-in would need two constant-bus reads)
        e.emit(f"v_addc_co_u32_e32 v{st + 1}, vcc, v{t}, v{st + 1}, vcc", r=["vcc"], w=["vcc"], vw=[st + 1])
        e.emit(f"v_mov_b32_e32 v{z}, v{st}", vw=[z])
        e.emit(f"v_mov_b32_e32 v{z + 1}, v{st + 1}", vw=[z + 1])
        xorshift(30)
        mul64(0xBF58476D1CE4E5B9)
        xorshift(27)
        mul64(0x94D049BB133111EB)
        xorshift(31)
        e.emit(f"v_mov_b32_e32 v{dst}, v{z}", vw=[dst])
        e.emit(f"v_mov_b32_e32 v{dst + 1}, v{z + 1}", vw=[dst + 1])

    def _gen_load_entry(self, e, p, word, first):
        """S <- table entry of the digit at (scalar word VGPR `word`, nibble S_J) in window S_I of the current curve
        (S_PHASE = byte offset of the curve's sub-table); first: into R = (x, y, 1) instead."""
        off, d = V_IDX8, V_FLAG            # free here: the output index is recomputed before the stores
        e.salu(f"s_lshl_b32 s{S_TMP0}, s{S_J}, 2")
        e.emit(f"v_lshrrev_b32_e32 v{d}, s{S_TMP0}, v{word}", vw=[d])
        e.emit(f"v_and_b32_e32 v{d}, 15, v{d}", vw=[d])
        e.emit(f"v_max_u32_e32 v{d}, 1, v{d}", vw=[d])                       # digits are 1..15 (a zero nibble counts as 1)
        # entry = (window * 15 + digit - 1) ; byte offset = entry * 144 + curve offset
        e.salu(f"s_mul_i32 s{S_TMP1}, s{S_I}, {15 * self.GEN_ENTRY_BYTES}")
        e.salu(f"s_add_u32 s{S_TMP1}, s{S_TMP1}, s{S_PHASE}")
        e.salu(f"s_sub_u32 s{S_TMP1}, s{S_TMP1}, {self.GEN_ENTRY_BYTES}")
        e.emit(f"v_mul_u32_u24_e32 v{off}, {self.GEN_ENTRY_BYTES}, v{d}", vw=[off])
        e.emit(f"v_add_u32_e32 v{off}, s{S_TMP1}, v{off}", vw=[off])
        for blk, base in ((A0, 0), (B0, 4 * SLOT_DW)):
            for c in range(Prog.N_B128):
                e.emit(f"global_load_dwordx4 v[{blk + 4 * c}:{blk + 4 * c + 3}], v{off}, {S_FIN} offset:{base + 16 * c}", kind="vmem",
                       vw=range(blk + 4 * c, blk + 4 * c + 4))
            e.emit(f"global_load_dwordx2 v[{blk + 16}:{blk + 17}], v{off}, {S_FIN} offset:{base + 64}", kind="vmem", vw=[blk + 16, blk + 17])
        e.raw("s_waitcnt vmcnt(0)")
        p.set_A_fresh()
        p.tagB = None
        p.to(self.R[0] if first else self.SX)
        p.wait()
        for i in range(SLOT_DW):
            e.emit(f"v_mov_b32_e32 v{A0 + i}, v{B0 + i}", vw=[A0 + i])
        p.set_A_fresh()
        p.to(self.R[1] if first else self.SY)
        p.reset_tags()

    def generate_main(self, e, p):
        L = self.lab
        W = self.GEN_WORDS
        st, z, t = W + 8, W + 10, W + 12
        # SplitMix64 state = seed ^ (0xD1B54A32D192ED03 * (index + 1))
        e.emit(f"v_lshrrev_b32_e32 v{z}, 3, v{V_IDX8}", vw=[z])           # V_IDX8 = clamped index * 8
        e.emit(f"v_add_u32_e32 v{z}, 1, v{z}", vw=[z])
        e.salu(f"s_mov_b32 s{S_TMP0}, 0xd192ed03")
        e.salu(f"s_mov_b32 s{S_TMP1}, 0xd1b54a32")
        e.emit(f"v_mul_lo_u32 v{t}, v{z}, s{S_TMP1}", vw=[t])
        e.emit(f"v_mad_u64_u32 v[{st}:{st + 1}], vcc, v{z}, s{S_TMP0}, 0", w=["vcc"], vw=[st, st + 1])
        e.emit(f"v_add_u32_e32 v{st + 1}, v{st + 1}, v{t}", vw=[st + 1])
        e.emit(f"v_xor_b32_e32 v{st}, s84, v{st}", vw=[st])                # seed: the `out` argument (s[84:85])
        e.emit(f"v_xor_b32_e32 v{st + 1}, s85, v{st + 1}", vw=[st + 1])
        for k in range(4):                                                 # s = draws 0, 1 ; t = draws 2, 3 (low word first)
            self._splitmix(e, st, z, t, W + 2 * k)
        for curve in range(2):
            e.salu(f"s_mov_b32 s{S_PHASE}, {curve * 32 * 15 * self.GEN_ENTRY_BYTES}")
            e.salu(f"s_mov_b32 s{S_I}, 0")
            for w in range(4):
                word = W + 4 * curve + w
                e.salu(f"s_mov_b32 s{S_J}, 0")
                if w == 0:                                                 # window 0: R = (entry, 1)
                    self._gen_load_entry(e, p, word, first=True)
                    self.one_into_A(e)
                    p.set_A_fresh()
                    p.to(self.R[2])
                    p.reset_tags()
                    e.salu(f"s_mov_b32 s{S_J}, 1")
                    e.salu(f"s_mov_b32 s{S_I}, 1")
                e.label(L(f"L_gen_{curve}_{w}"))
                self._gen_load_entry(e, p, word, first=False)
                self.call2(e, "L2_ptadd")
                e.salu(f"s_add_u32 s{S_I}, s{S_I}, 1")
                e.salu(f"s_add_u32 s{S_J}, s{S_J}, 1")
                e.salu(f"s_cmp_lt_u32 s{S_J}, 8")
                e.salu(f"s_cbranch_scc1 {L(f'L_gen_{curve}_{w}')}")
            self.call2(e, "L2_affine")
            # store: canonical external form, SoA planes of this curve's output batch
            e.salu(f"s_sub_u32 s{S_TMP1}, s{S_N}, 1")
            e.emit(f"v_min_u32_e32 v{V_IDX8}, s{S_TMP1}, v{V_IDX}", vw=[V_IDX8])
            e.emit(f"v_lshlrev_b32_e32 v{V_IDX8}, 3, v{V_IDX8}", vw=[V_IDX8])
            e.emit(f"v_cmp_gt_u32_e32 vcc, s{S_N}, v{V_IDX}", w=["vcc"])
            e.raw("s_nop 1")
            e.salu(f"s_and_saveexec_b64 {S_SAVE_EXEC}, vcc")
            self.io_walk_begin(e, S_G1 if curve == 0 else S_G2)
            p.reset_tags()
            for src in (self.SX, self.SY):
                for half in range(1 if curve == 0 else 2):
                    p.load(A0, src)
                    p.wait()
                    if half == 1:
                        for i in range(NL):
                            e.emit(f"v_mov_b32_e32 v{A0 + i}, v{A0 + NL + i}", vw=[A0 + i])
                    self.cvt_call(e, "cvtout")
                    self.io_store_fq(e, A0)
                    e.raw("s_nop 1")
            e.salu(f"s_mov_b64 exec, {S_SAVE_EXEC}")
            e.raw("s_waitcnt vmcnt(0)")
            e.emit(f"v_mov_b32_e32 v{V_FLAG}, 0", vw=[V_FLAG])
        e.emit(f"v_lshrrev_b32_e32 v{V_TID}, 4, v{V_LDS}", vw=[V_TID])
        p.reset_tags()


    # ---------------------------------------------------------------------------------------------
    # FIXED G2 POINTS.  multi_miller_loop_native (miller_loop_native.rs:192-282) takes any pairs; a Groth16 verifier calls it with three of its
    # four G2 points -- beta, gamma, delta of the verifying key -- THE SAME for every proof.  The point steps of such a pair do not depend on
    # the lane at all: the `lines` kernel walks them once per fixed point (one lane each, the generic step routines, evaluation point (1, 1)) and
    # leaves every step's line coefficients in a table; the `fixed` kernel is k_pairing for the group's own pair plus, per step and fixed
    # pair, one table line scaled by that pair's (Px, Py) and one sparse multiplication -- no point step, no R anywhere but the variable
    # pair's resident one.  Same chain as the fused kernels (the table is made for it); any Fq2 factor of a line dies in the easy part.
    #   table: [fixed pair][line][FIX_LINE_SLOTS slots of 72 bytes], lines in the order the loop consumes them: the first doubling, then per digit
    #   the doubling and (digit != 0) the addition, then the two Frobenius steps.
    @property
    def n_fixed_lines(self):
        return 1 + self.naf_first + sum(1 for d in self.naf[:self.naf_first + 1] if d) + 2

    FIX_LINE_SLOTS = 4        # table slots per line: denominator (only while the table is made), B, C, prefix product (ditto)

    def _tab_cursor(self, e, base, pair):
        """S_TAB <- base + pair * (bytes per pair) + S_TABCUR"""
        lo, hi = (int(x) for x in base.strip("s[]").split(":"))
        off = pair * self.n_fixed_lines * self.FIX_LINE_SLOTS * SLOT_BYTES
        e.salu(f"s_add_u32 s72, s{lo}, s{S_TABCUR}")
        e.salu(f"s_addc_u32 s73, s{hi}, 0")
        if off:
            e.salu(f"s_add_u32 s72, s72, 0x{off:x}")
            e.salu("s_addc_u32 s73, s73, 0")

    # Every table line is left in ONE shape, 1 + B Py w^3 + C Px w^4:
    #   a doubling's line  L0 + H Py w^3 - 3 X^2 Px w^4            is divided by L0;
    #   an addition's line -mu Py w^2 + theta Px w^3 + L5 w^5      is multiplied by w / (xi L5)   (w^6 = xi).
    # The factors 1 / L0, 1 / (xi L5) lie in Fq2; the w's of the lines that are followed by a squaring end up as even powers, and the two Frobenius lines
    # bring w^2: everything lies in Fq6 and dies in the easy part like the projective lines' own scales.  With the constant coefficient ONE a sparse
    # multiplication is six two-product passes (Prog.mul_by_034_one) instead of six three-product ones, and one routine serves every line.
    def _fixed_routines(self):
        # (the third waiting result of the multiplication goes through the global scratch: handing it the line's unused constant slot, AGPR 6,
        # instead makes no measurable difference -- 36 more AGPR moves per line against ten memory instructions whose latency the passes hide)
        tm = self.miller_temps(extra=(self.SX, self.SY))
        n_last = sum(1 for d in self.naf[:1] if d) + 2            # lines behind the last squaring: the digit-0 addition (none: 6x + 2 is even) + the Frobenius pair
        assert n_last % 2 == 0, "an odd number of w factors would survive the easy part"

        def fixline(j):
            def body(p):
                e = p.e
                self._tab_cursor(e, S_FIN, j)
                e.emit(f"v_mov_b32_e32 v{V_IOOFF}, 0", vw=[V_IOOFF])
                Pj = self.FIX_P[j]
                # the two coefficients travel together (block A and a home temporary): ONE exposed memory latency per line
                t1 = p.tmp()
                assert t1.kind == "home"
                p.A(Tab(1))
                p.load(HOME0 + SLOT_DW * t1.idx, Tab(2))
                p.slot_r[p.key(t1)], p.slot_v[p.key(t1)] = p.UNKNOWN, V_STORE
                p.mulfq_c1(Pj).to(self.LINE[1])                     # B Py
                p.A(t1).mulfq(Pj).to(self.LINE[2])                  # C Px
                p.rel(t1)
            return body
        for j in range(self.MAX_FIXED):
            self.l2_routine(f"L2_fix_{j}", fixline(j), tm)
        # A pass of the unit-coefficient multiplication ADDS to its input (out = a + two products / R' +- p/2): f grows by about 0.6 p per line instead
        # of contracting, and a routine that had to assume the general 4 p at its entry would reduce every output (six reducing chains per line).
        # So the multiplication is built per position j in the run of fixed lines, each variant assuming what its predecessor leaves (the first: the
        # largest bounds the own pair's routines leave); a variant whose output would pass the contract reduces there, and a later position
        # whose entry bounds an earlier variant covers reuses it.  certify_values() checks every call against the assumed bounds.
        # (groups without a pair of their own: an addition step's lines follow the doubling step's directly -- f is brought back in between)
        self.l2_routine("L2_fxred", self._reduce_f, tm)
        Fk = [Prog.key(s_) for s_ in self.F]
        before = ("L2_dblfirst", "L2_dblmul", "L2_addmul", "L2_addmul_last", "L2_sqr", "L2_fxred")   # (L2_sqr, L2_fxred: groups without a pair of their own, MODE_NO_OWN)
        ent = {k: max(max(self.l2_exit[n].get(k, 0.0) for n in before), 1.01) for k in Fk}               # (1.01: f = 1 at the start of such a group)
        line = {Prog.key(s_): max(self.l2_exit[f"L2_fix_{j}"][Prog.key(s_)] for j in range(self.MAX_FIXED)) for s_ in self.LINE[1:]}     # (the scaled coefficients: 0.6 p)
        variants, self.fsp_variant = [], []
        for j in range(self.MAX_FIXED):
            cover = next((v for v in variants if all(ent[k] <= v[1][k] for k in Fk)), None)
            if cover is None:
                name = f"L2_fsp1_{len(variants)}"
                self.l2_routine(name, lambda p: p.mul_by_034_one(self.F, self.LINE[1], self.LINE[2]), tm, entry={**ent, **line})
                cover = (name, dict(ent), {k: self.l2_exit[name][k] for k in Fk})
                variants.append(cover)
            self.fsp_variant.append(cover[0])
            ent = dict(cover[2])

    def _fixed_lines(self, e):
        """f *= the current line of every fixed pair (S_K of them); the table cursor moves on by one line"""
        u = self.uid()
        done = self.lab(f"L_fx_done_{u}")
        for j in range(self.MAX_FIXED):
            e.salu(f"s_cmp_gt_u32 s{S_K}, {j}")
            e.salu(f"s_cbranch_scc0 {done}")
            self.call2(e, f"L2_fix_{j}")
            self.call2(e, self.fsp_variant[j])
        e.label(done)
        e.salu(f"s_add_u32 s{S_TABCUR}, s{S_TABCUR}, {self.FIX_LINE_SLOTS * SLOT_BYTES}")

    def miller_main_fixed(self, e, p):
        """g1: 1 + S_K points per group, group-major (the group's own P first, then the P_j of the fixed pairs); g2: the group's own Q;
        f_in: the line table of the S_K fixed points.  Mode bit MODE_NO_OWN: the groups have no pair of their own -- g1: S_K points per group,
        g2 unused, f starts at one and only the table lines are multiplied in (the loop's squarings, S_K lines per step, the final exponentiation)."""
        L = self.lab
        own = lambda label: (e.salu(f"s_bitcmp1_b32 s{self.s_mode}, {MODE_NO_OWN}"), e.salu(f"s_cbranch_scc1 {label}"))      # skip the own pair's work
        e.salu(f"s_bitcmp0_b32 s{self.s_mode}, {MODE_NO_OWN}")      # SCC = 1 iff the group has its own pair
        e.salu(f"s_addc_u32 s{S_TMP1}, s{S_K}, 0")                   # points per group: S_K (+ 1)
        e.salu(f"s_mul_i32 s{S_NSTRIDE}, s{S_N}, s{S_TMP1}")
        e.salu(f"s_lshl_b32 s{S_NSTRIDE}, s{S_NSTRIDE}, 3")          # bytes between limb planes of the G1 batch
        e.emit(f"v_mul_lo_u32 v{V_IDX8}, v{V_IDX8}, s{S_TMP1}", vw=[V_IDX8])   # byte offset of the group's first point
        own(L("L_fx_noP"))
        self.io_walk_begin(e, S_G1, 8)
        self.io_load_fq2_into_A(e, p, c1_present=False)          # Px
        p.to(self.PX)
        self.io_load_fq2_into_A(e, p, c1_present=False)          # Py
        p.to(self.PY)
        e.emit(f"v_add_u32_e32 v{V_IDX8}, 8, v{V_IDX8}", vw=[V_IDX8])
        e.label(L("L_fx_noP"))
        p.reset_tags()
        skip = L("L_fx_pts")
        for j in range(self.MAX_FIXED):                          # (Px_j, Py_j) packed into one slot: c0 = Px, c1 = Py
            e.salu(f"s_cmp_gt_u32 s{S_K}, {j}")
            e.salu(f"s_cbranch_scc0 {skip}")
            self.io_walk_begin(e, S_G1, 8)
            self.io_load_fq2_into_A(e, p)
            p.to(self.FIX_P[j])
            p.reset_tags()
            e.emit(f"v_add_u32_e32 v{V_IDX8}, 8, v{V_IDX8}", vw=[V_IDX8])
        e.label(skip)
        e.salu(f"s_lshl_b32 s{S_NSTRIDE}, s{S_N}, 3")                # G2 and the result: one element per group
        e.salu(f"s_sub_u32 s{S_TMP1}, s{S_N}, 1")
        e.emit(f"v_min_u32_e32 v{V_IDX8}, s{S_TMP1}, v{V_IDX}", vw=[V_IDX8])
        e.emit(f"v_lshlrev_b32_e32 v{V_IDX8}, 3, v{V_IDX8}", vw=[V_IDX8])
        e.salu(f"s_mov_b32 s{S_TABCUR}, 0")
        own(L("L_fx_noQ"))
        self.io_walk_begin(e, S_G2, 16)
        self.io_load_fq2_into_A(e, p)                            # Q.x
        p.to(self.QX)
        p.to(self.R[0])
        self.io_load_fq2_into_A(e, p)                            # Q.y
        p.to(self.QY)
        p.to(self.R[1])
        self.one_into_A(e)
        p.set_A_fresh()
        p.to(self.R[2])
        p.reset_tags()
        self.call2(e, "L2_dblfirst")
        e.salu(f"s_branch {L('L_fx_f0')}")
        e.label(L("L_fx_noQ"))
        p.reset_tags()
        self.one_into_A(e)                                       # f = 1
        p.set_A_fresh()
        p.to(self.F[0])
        p.wait()
        self.zero_block(e, A0)
        p.set_A_fresh(0.0)
        for k_ in range(1, 6):
            p.to(self.F[k_])
        p.reset_tags()
        e.label(L("L_fx_f0"))
        self._fixed_lines(e)
        first = self.naf_first
        assert first == 63
        e.salu(f"s_mov_b32 s{S_I}, {first}")
        e.label(L("L_mloop"))
        e.salu(f"s_cmp_eq_u32 s{S_I}, {first}")
        e.salu(f"s_cbranch_scc1 {L('L_mskip')}")
        self.call2(e, "L2_sqr")
        own(L("L_fx_nodbl"))
        self.call2(e, "L2_dblmul")
        e.label(L("L_fx_nodbl"))
        self._fixed_lines(e)
        e.label(L("L_mskip"))
        e.salu(f"s_bitcmp1_b64 {S_NAF_NZ}, s{S_I}")
        e.salu(f"s_cbranch_scc0 {L('L_mnoadd')}")
        self._own_or_reduce(e, p, L, "noadd", lambda: (self._select_pm_q(e, p), self.call2(e, "L2_addmul")))
        self._fixed_lines(e)
        e.label(L("L_mnoadd"))
        e.salu(f"s_sub_u32 s{S_I}, s{S_I}, 1")
        e.salu(f"s_cbranch_scc0 {L('L_mloop')}")
        # + pi(Q), then the line through the result and -pi^2(Q)
        self._own_or_reduce(e, p, L, "nofr1", lambda: (self._frobenius_points(p), self.call2(e, "L2_addmul")))
        self._fixed_lines(e)
        self._own_or_reduce(e, p, L, "nofr2", lambda: (p.reset_tags(), p.mov(self.SX, self.QX), p.mov(self.SY, self.QY), self.call2(e, "L2_addmul_last")))
        self._fixed_lines(e)
        p.reset_tags()

    def _own_or_reduce(self, e, p, L, tag, own_work):
        """the own pair's step in front of a run of table lines -- or, for groups without one (MODE_NO_OWN), f brought back to (-0.51 p, 0.51 p): the
        previous run's lines have grown it, and the first line's multiplication assumes what the own pair's routines leave"""
        e.salu(f"s_bitcmp1_b32 s{self.s_mode}, {MODE_NO_OWN}")
        e.salu(f"s_cbranch_scc1 {L('L_fx_' + tag)}")
        own_work()
        e.salu(f"s_branch {L('L_fx_' + tag + '_done')}")
        e.label(L("L_fx_" + tag))
        self.call2(e, "L2_fxred")
        e.label(L("L_fx_" + tag + "_done"))
        p.reset_tags()

    # the table's maker: one fixed G2 point per lane (g2: S_N points), out = the table.  Pass 1 walks the point steps and leaves, per line, the
    # denominator D (L0, or xi L5), the two numerators and the product of all EARLIER denominators; one Fq2 inversion of the whole product; pass 2
    # walks the lines backwards (Montgomery's trick) and leaves B = N1 / D, C = N2 / D.
    LN_ACC, LN_INV = AGPR(0, "lnacc"), AGPR(1, "lninv")          # (no f in this kernel: its AGPR slots are free)

    def _lines_routines(self):
        self.COLD = self.COLD + ("L2_ldbl", "L2_ladd", "L2_ladd_last", "L2_linv", "L2_lnorm")
        tm = self.miller_temps()
        step_b = self.FIX_LINE_SLOTS * SLOT_BYTES

        def step(kind):
            def body(p):
                if kind == "dbl":
                    p.dbl_step(self.R, (self.PX, self.PY), self.LINE)
                    D, N1, N2 = self.LINE
                else:
                    p.add_step(self.R, (self.SX, self.SY), (self.PX, self.PY), self.LINE, update=(kind == "add"))
                    N1, N2, D = self.LINE
                    p.A(D).mulxi().to(D)
                p.A(N1).to(Tab(1))
                p.A(N2).to(Tab(2))
                p.A(self.LN_ACC).to(Tab(3))                         # the product of the denominators BEFORE this line
                p.A(D).to(Tab(0))
                p.mul(self.LN_ACC).to(self.LN_ACC)
                p.wait()
                p.e.salu(f"s_add_u32 s72, s72, {step_b}")
                p.e.salu("s_addc_u32 s73, s73, 0")
            return body
        self.l2_routine("L2_ldbl", step("dbl"), tm, local=self.LINE)
        self.l2_routine("L2_ladd", step("add"), tm, local=self.LINE)
        self.l2_routine("L2_ladd_last", step("last"), tm, local=self.LINE)
        self.l2_routine("L2_fqinv", self._fq_inv, tm)
        self.l2_routine("L2_linv", lambda p: self._fq2_inv_inline(p, self.LN_ACC, self.LN_INV), tm)

        def norm_line(p):
            t = p.tmp()
            p.A(Tab(3)).mul(self.LN_INV).to(t)                      # 1 / D of this line
            p.A(Tab(0)).mul(self.LN_INV).to(self.LN_INV)            # 1 / (product of the earlier denominators)
            p.A(Tab(1)).mul(t).to(Tab(1))
            p.A(Tab(2)).mul(t).to(Tab(2))
            p.rel(t)
            p.wait()
            p.e.salu(f"s_sub_u32 s72, s72, {step_b}")
            p.e.salu("s_subb_u32 s73, s73, 0")
        self.l2_routine("L2_lnorm", norm_line, tm)

    def lines_main(self, e, p):
        L = self.lab
        # lanes past the last point would redo it INTO THE SAME PART OF THE TABLE -- harmless for pass 1's plain stores, not for pass 2, which reads
        # what it overwrites (another wave's lane may have been there already): they sit the whole item out
        e.emit(f"v_cmp_gt_u32_e32 vcc, s{S_N}, v{V_IDX}", w=["vcc"])
        e.raw("s_nop 1")
        e.salu(f"s_and_saveexec_b64 {S_SAVE_EXEC}, vcc")
        self.io_walk_begin(e, S_G2)
        self.io_load_fq2_into_A(e, p)
        p.to(self.QX)
        p.to(self.R[0])
        self.io_load_fq2_into_A(e, p)
        p.to(self.QY)
        p.to(self.R[1])
        self.one_into_A(e)
        p.set_A_fresh()
        p.to(self.R[2])
        p.to(self.PX)                                            # the lines are left WITHOUT an evaluation point: (Px, Py) = (1, 1)
        p.to(self.PY)
        p.to(self.LN_ACC)
        p.reset_tags()
        e.salu(f"s_mov_b64 {S_TAB}, {S_OUT}")
        e.salu(f"s_mov_b32 s{S_TMP0}, 0x{self.n_fixed_lines * self.FIX_LINE_SLOTS * SLOT_BYTES:x}")
        e.emit(f"v_lshrrev_b32_e32 v{V_IOOFF}, 3, v{V_IDX8}", vw=[V_IOOFF])       # (clamped) index of this lane's point
        e.emit(f"v_mul_lo_u32 v{V_IOOFF}, v{V_IOOFF}, s{S_TMP0}", vw=[V_IOOFF])   # its part of the table
        self.call2(e, "L2_ldbl")
        first = self.naf_first
        e.salu(f"s_mov_b32 s{S_I}, {first}")
        e.label(L("L_mloop"))
        e.salu(f"s_cmp_eq_u32 s{S_I}, {first}")
        e.salu(f"s_cbranch_scc1 {L('L_mskip')}")
        self.call2(e, "L2_ldbl")
        e.label(L("L_mskip"))
        e.salu(f"s_bitcmp1_b64 {S_NAF_NZ}, s{S_I}")
        e.salu(f"s_cbranch_scc0 {L('L_mnoadd')}")
        self._select_pm_q(e, p)
        self.call2(e, "L2_ladd")
        e.label(L("L_mnoadd"))
        e.salu(f"s_sub_u32 s{S_I}, s{S_I}, 1")
        e.salu(f"s_cbranch_scc0 {L('L_mloop')}")
        self._frobenius_points(p)
        self.call2(e, "L2_ladd")
        p.reset_tags()
        p.mov(self.SX, self.QX)
        p.mov(self.SY, self.QY)
        self.call2(e, "L2_ladd_last")
        e.raw("s_waitcnt vmcnt(0)")
        p.reset_tags()
        # pass 2: one inversion, then the lines backwards (the cursor stands behind the last line)
        self.call2(e, "L2_linv")
        e.salu(f"s_sub_u32 s72, s72, {self.FIX_LINE_SLOTS * SLOT_BYTES}")
        e.salu("s_subb_u32 s73, s73, 0")
        e.salu(f"s_mov_b32 s{S_I}, {self.n_fixed_lines - 1}")
        e.label(L("L_lnorm"))
        self.call2(e, "L2_lnorm")
        e.salu(f"s_sub_u32 s{S_I}, s{S_I}, 1")
        e.salu(f"s_cbranch_scc0 {L('L_lnorm')}")
        e.raw("s_waitcnt vmcnt(0)")
        e.salu(f"s_mov_b64 exec, {S_SAVE_EXEC}")
        p.reset_tags()

    # ---------------------------------------------------------------------------------------------
    # G2 subgroup check (bn254_check_points_ex with BN254_CHECK_SUBGROUP): ark's `G2Affine::new` asserts it, the reference calls that on the
    # Frobenius images of Q (miller_loop_native.rs:303,311).  Criterion (El Housni - Guillevic - Piellard, ePrint 2022/348; the same as
    # csrc/bn254_point_checks.h, which stays as this kernel's cross-check):
    #     [x + 1] Q + psi([x] Q) + psi^2([x] Q) == psi^3([2 x] Q),      psi = the reference's twisted_frobenius (:298-304)
    # [x]Q by the non-adjacent form of x, top digit first: 62 doublings + 23 mixed additions of +-Q in Jacobian coordinates; then three psi
    # (two Fq2 products each), one mixed and two general additions, one doubling, and the comparison of the two projective points by
    # cross-multiplication, on canonical limbs (cvtout).  Everything lives in AGPR slots and home registers: no LDS, no scratch.
    SUB_B = [AGPR(6, "BX"), AGPR(7, "BY"), AGPR(8, "BZ")]
    SUB_C = [AGPR(9, "CX"), AGPR(10, "CY"), AGPR(11, "CZ")]
    V_SUB_DIFF, V_SUB_Z = 252, 253          # OR of the canonical limbs of the two cross-differences / of Z_lhs Z_rhs (free VGPRs: V_P3T of the k-pair kernels)

    def sub_temps(self):
        return [HOME(i) for i in range(8)] + [AGPR(12), AGPR(13)] + [LDS(i) for i in range(N_LDS_SLOTS)] + [GLOB(GLOB_TMP0 + i) for i in range(8)]

    def _sub_psi(self, p, src, dst):
        """dst <- psi(src) on Jacobian coordinates: (c2 conj(X), c3 conj(Y), conj(Z))"""
        C2, C3 = self._twist_consts()
        p.A(src[0]).conj().mul(C2).to(dst[0])
        p.A(src[1]).conj().mul(C3).to(dst[1])
        p.A(src[2]).conj().to(dst[2])

    def _sub_or_limbs(self, p, slot, acc, first):
        """v[acc] (|)= OR of the sixteen canonical dwords of the Fq2 in `slot` (zero iff the value is 0 mod p in both components)"""
        e = p.e
        for half in range(2):
            p.tagA = None
            p.A(slot)
            p.redn()
            p.wait()
            if half == 1:
                for i in range(NL):
                    e.emit(f"v_mov_b32_e32 v{A0 + i}, v{A0 + NL + i}", vw=[A0 + i])
            self.cvt_call(e, "cvtout")
            for i in range(8):
                if first and half == 0 and i == 0:
                    e.emit(f"v_mov_b32_e32 v{acc}, v{A0}", vw=[acc])
                else:
                    e.emit(f"v_or_b32_e32 v{acc}, v{acc}, v{A0 + i}", vw=[acc])
        p.reset_tags()

    def _sub_final(self, p):
        a, B, C = self.R, self.SUB_B, self.SUB_C
        self._sub_psi(p, a, B)                         # psi([x]Q)
        self._sub_psi(p, B, C)                         # psi^2([x]Q)
        p.jac_madd(a, (self.QX, self.QY))              # [x + 1]Q
        p.jac_add(a, B)
        p.jac_add(a, C)                                # the left-hand side
        self._sub_psi(p, C, B)                         # psi^3([x]Q)
        p.jac_dbl(B)                                   # the right-hand side
        # equal as points: X1 Z2^2 == X2 Z1^2 and Y1 Z2^3 == Y2 Z1^3, and neither is the point at infinity
        z1z1, z2z2, t, d = [p.tmp() for _ in range(4)]
        p.A(a[2]).sqr().to(z1z1)
        p.A(B[2]).sqr().to(z2z2)
        p.A(B[0]).mul(z1z1).to(t)
        p.A(a[0]).mul(z2z2).sub(t).to(d)
        self._sub_or_limbs(p, d, self.V_SUB_DIFF, True)
        p.A(B[1]).mul(a[2]).mul(z1z1).to(t)
        p.A(a[1]).mul(B[2]).mul(z2z2).sub(t).to(d)
        self._sub_or_limbs(p, d, self.V_SUB_DIFF, False)
        p.A(a[2]).mul(B[2]).to(d)
        self._sub_or_limbs(p, d, self.V_SUB_Z, True)
        p.rel(z1z1, z2z2, t, d)

    def _subcheck_routines(self):
        st = self.sub_temps()
        self.l2_routine("L2_sdbl", lambda p: p.jac_dbl(self.R), st)
        self.l2_routine("L2_smadd", lambda p: p.jac_madd(self.R, (self.QX, self.SY)), st)
        self.l2_routine("L2_sfin", self._sub_final, st)

    def subcheck_main(self, e, p):
        L = self.lab
        top = max(i for i, d in enumerate(self.naf) if d)
        assert self.naf[top] == 1 and top == 62
        self.io_walk_begin(e, S_G2)
        self.io_load_fq2_into_A(e, p)                            # Q.x
        p.to(self.QX)
        p.to(self.R[0])
        self.io_load_fq2_into_A(e, p)                            # Q.y
        p.to(self.QY)
        p.to(self.R[1])
        self.one_into_A(e)
        p.set_A_fresh()
        p.to(self.R[2])
        p.reset_tags()
        e.salu(f"s_mov_b32 s{S_I}, {top - 1}")
        e.label(L("L_sloop"))
        self.call2(e, "L2_sdbl")
        e.salu(f"s_bitcmp1_b64 {S_NAF_NZ}, s{S_I}")
        e.salu(f"s_cbranch_scc0 {L('L_snoadd')}")
        p.reset_tags()                                           # SY <- +-Q.y by the digit's sign
        p.A(self.QY)
        p.wait()
        e.salu(f"s_bitcmp1_b64 {S_NAF_NEG}, s{S_I}")
        e.salu(f"s_cbranch_scc0 {L('L_spos')}")
        e.salu(f"s_call_b64 {S_RET1}, {self.labels['neg']}")
        e.label(L("L_spos"))
        p.tagA = None
        p.to(self.SY)
        self.call2(e, "L2_smadd")
        e.label(L("L_snoadd"))
        e.salu(f"s_sub_u32 s{S_I}, s{S_I}, 1")
        e.salu(f"s_cbranch_scc0 {L('L_sloop')}")
        p.reset_tags()
        self.call2(e, "L2_sfin")
        # verdict word: 1 = the criterion does not hold (a cross-difference is not zero, or one side is the point at infinity)
        D, Z = self.V_SUB_DIFF, self.V_SUB_Z
        e.emit(f"v_cmp_ne_u32_e32 vcc, 0, v{D}", w=["vcc"])
        e.raw("s_nop 1")
        e.emit(f"v_cndmask_b32_e64 v{D}, 0, 1, vcc", r=["vcc"], vw=[D])
        e.emit(f"v_cmp_eq_u32_e32 vcc, 0, v{Z}", w=["vcc"])
        e.raw("s_nop 1")
        e.emit(f"v_cndmask_b32_e64 v{Z}, 0, 1, vcc", r=["vcc"], vw=[Z])
        e.emit(f"v_or_b32_e32 v{D}, v{D}, v{Z}", vw=[D])
        e.emit(f"v_lshlrev_b32_e32 v{Z}, 2, v{V_IDX}", vw=[Z])
        e.emit(f"v_cmp_gt_u32_e32 vcc, s{S_N}, v{V_IDX}", w=["vcc"])      # lanes past the end of the batch do not store
        e.raw("s_nop 1")
        e.salu(f"s_and_saveexec_b64 {S_SAVE_EXEC}, vcc")
        e.emit(f"global_store_dword v{Z}, v{D}, {S_OUT}", kind="vmem")
        e.salu(f"s_mov_b64 exec, {S_SAVE_EXEC}")
        e.raw("s_waitcnt vmcnt(0)")
        e.emit(f"v_lshrrev_b32_e32 v{V_TID}, 4, v{V_LDS}", vw=[V_TID])
        p.reset_tags()

    # ---------------------------------------------------------------------------------------------
    # multi-pairing main loop: STREAMED pair state.  The k pairs of a group share f, so their points take turns; their
    # state (P, Q, R) lives in scratch.  Swapping it through resident slots (load, wait, compute, store) leaves the global
    # latency exposed twice per pair and step -- 14 % of the Groth16-shape kernel's time.  Instead the NEXT pair's state is
    # fetched straight into five AGPR slots (global loads can target AGPRs, no VGPR is needed) while the current pair's step
    # and its sparse multiplication run; the fused step reads its operands from that buffer and writes the new R from its
    # register blocks straight back to scratch.
    #   * k > RES_K: the buffer takes P and R of the next pair; Q (needed by the 27 addition steps only) is fetched
    #     synchronously there.
    #   * k <= RES_K (the Groth16 shape, k = 4): P never changes, so every pair's (Px, Py) sits packed in ONE slot of the four
    #     LDS slots that are idle during the loop (the resident Q and R slots of the one-pair routines): the stream carries R only
    #     (3 slots in, 3 out per step instead of 5 + 3), and the two freed buffer slots take the next pair's Q whenever the next
    #     step is an addition -- nothing is fetched synchronously any more.
    R0_LDS = [LDS(3, "R0X"), LDS(4, "R0Y"), LDS(5, "R0Z")]    # pair 0's R during the streamed loop

    def r0_resident(self):
        return bool(int(os.environ.get("KGEN_R0_LDS", "1"))) and LINE_IN_REGS and Prog.FUSED_STEPS and SPREAD_PREFETCH

    # Round 5 -- pair 1 PARTLY resident.  The untracked kernels (k_mpairing) have one LDS slot left in the streamed loop (slot 2, the line
    # scale of the exact-value kernels): pair 1's X lives there for the whole loop and only Y, Z stream -- 4 of the 30 slot moves of two
    # passes over four pairs (-13 % of the R stream).  A timing-only build that drops those moves altogether measured +0.52 % (X alone)
    # and +1.14 % (X and Y) on the Groth16 shape (profiles/r05_ab.txt): what residency can gain at most.  A second coordinate needs a
    # second slot: every other slot is taken.
    R1_LDS = [LDS(2, "R1X"), LDS(7, "R1Y")]
    # The second slot (7) is the home of pair 3's packed evaluation point in resident-P mode: that point moves into the 16 KB of LDS the
    # eight slots leave (sixteen dwords per lane: four 16-byte chunk planes behind the tails, at byte P3_LDS_BASE) and its last two limbs
    # into two VGPRs that nothing else uses (V_P3T) -- no repacking, one scalar branch where a pair's P is written or read.
    P3_LDS_BASE = N_LDS_SLOTS * (Prog.N_B128 * 4096 + 2048)
    V_P3T = 252

    def r1_slots(self):
        """the LDS slots of pair 1's resident coordinates (X first), or [] (exact-value kernels: slot 2 holds the line scale)"""
        n = int(os.environ.get("KGEN_R1_LDS", "2"))
        return self.R1_LDS[:n] if (self.multi and not self.track and self.boustrophedon()) else []

    def p3_moved(self):
        return len(self.r1_slots()) >= 2

    S_GNEXT = 49               # byte offset of the NEXT pair's scratch block
    S_DIR, S_CNT, S_NEXTP = 72, 73, 74       # boustrophedon passes: direction (+1 / -1), pairs left in the pass, index of the pair of the NEXT step

    def boustrophedon(self):
        """Round 4: the passes over the k pairs of a group alternate their direction, so that the pair that closes a pass opens the
        next one: its R never leaves the chip in between -- pair 0 is resident anyway, pair k - 1 goes from the step's register
        blocks straight into the prefetch buffer (AGPR) instead of through scratch: one store and one load of R less per two
        passes (k = 4: -17 % of the R stream, k = 2: -50 %)."""
        return bool(int(os.environ.get("KGEN_BOUSTRO", "1"))) and self.multi and self.r0_resident()
    RES_K = 4                  # largest k whose evaluation points stay on chip
    RES_P_LDS = (0, 1, 6, 7)   # LDS slot of pair j's packed (Px, Py)

    @property
    def BUF(self):
        return {"PX": AGPR(10, "bPX"), "PY": AGPR(11, "bPY"), "RX": AGPR(12, "bRX"), "RY": AGPR(13, "bRY"), "RZ": AGPR(9, "bRZ")}

    def _emit_buf_loads(self, e, pairs):
        """buffer slot <- scratch slot k of the pair whose block starts at S_GNEXT, for (buffer name, k) in pairs; nobody waits"""
        for name, k in pairs:
            if EXP_NO_PREFETCH or EXP_NO_SCRATCH:
                break
            a0 = SLOT_DW * self.BUF[name].idx
            e.salu(f"s_mul_i32 s{S_TMP0}, s{S_GSTRIDE}, {k}")
            e.salu(f"s_add_u32 s{S_TMP0}, s{S_TMP0}, s{self.S_GNEXT}")
            e.salu(f"s_add_u32 s62, s64, s{S_TMP0}")
            e.salu("s_addc_u32 s63, s65, 0")
            for c in range(Prog.N_B128):
                e.emit(f"global_load_dwordx4 a[{a0 + 4 * c}:{a0 + 4 * c + 3}], v{V_GOFF}, {S_GADDR} offset:{GCHUNK0 + 1024 * c}" + _ldm(), kind="vmem")
            e.emit(f"global_load_dwordx2 a[{a0 + 16}:{a0 + 17}], v{V_GOFF8}, {S_GADDR} offset:0" + _ldm(), kind="vmem")

    def _emit_prefetch_part(self, e, part, q):
        """the prefetch, slot group by slot group (KGEN_SPREAD_PF): part 0..2 = RX, RY, RZ; part 3 = P or Q (as _emit_prefetch)"""
        L = self.lab
        if part < 3:
            if self.boustrophedon():                      # the next step's pair is pair 0 (resident) or this very pair (it turns: its R stays in the buffer)
                u = self.uid()
                e.salu(f"s_cmp_eq_u32 s{self.S_NEXTP}, 0")
                e.salu(f"s_cbranch_scc1 {L(f'L_pf_r0_{u}')}")
                e.salu(f"s_cmp_eq_u32 s{self.S_NEXTP}, s{S_JP}")
                e.salu(f"s_cbranch_scc1 {L(f'L_pf_r0_{u}')}")
                if (EXP_R1 and part < EXP_R1) or part < len(self.r1_slots()):     # pair 1's resident coordinates are on chip
                    e.salu(f"s_cmp_eq_u32 s{self.S_NEXTP}, 1")
                    e.salu(f"s_cbranch_scc1 {L(f'L_pf_r0_{u}')}")
            elif self.r0_resident():                      # the next pair is pair 0 (the index wraps): its R is on chip
                u = self.uid()
                e.salu(f"s_add_u32 s{S_TMP0}, s{S_JP}, 1")
                e.salu(f"s_cmp_lt_u32 s{S_TMP0}, s{S_K}")
                e.salu(f"s_cbranch_scc0 {L(f'L_pf_r0_{u}')}")
            self._emit_buf_loads(e, ((("RX", 4), ("RY", 5), ("RZ", 6))[part],))
            if self.r0_resident():
                e.label(L(f"L_pf_r0_{u}"))
            return
        u = self.uid()
        e.salu(f"s_cmp_le_u32 s{S_K}, {self.RES_K}")
        e.salu(f"s_cbranch_scc1 {L(f'L_pf_res_{u}')}")
        self._emit_buf_loads(e, (("PX", 0), ("PY", 1)))
        e.salu(f"s_branch {L(f'L_pf_done_{u}')}")
        e.label(L(f"L_pf_res_{u}"))
        if q == "last" and self.boustrophedon():         # only behind the pair that closes the doubling pass (it opens the addition pass)
            e.salu(f"s_cmp_lg_u32 s{self.S_NEXTP}, s{S_JP}")
            e.salu(f"s_cbranch_scc1 {L(f'L_pf_done_{u}')}")
            e.salu(f"s_bitcmp1_b64 {S_NAF_NZ}, s{S_I}")
            e.salu(f"s_cbranch_scc0 {L(f'L_pf_done_{u}')}")
        elif q == "last":
            e.salu(f"s_add_u32 s{S_TMP0}, s{S_JP}, 1")
            e.salu(f"s_cmp_lt_u32 s{S_TMP0}, s{S_K}")
            e.salu(f"s_cbranch_scc1 {L(f'L_pf_done_{u}')}")
            e.salu(f"s_bitcmp1_b64 {S_NAF_NZ}, s{S_I}")
            e.salu(f"s_cbranch_scc0 {L(f'L_pf_done_{u}')}")
        self._emit_buf_loads(e, (("PX", 2), ("PY", 3)))
        e.label(L(f"L_pf_done_{u}"))

    def _emit_prefetch(self, e, q="always"):
        """buffer <- state of the pair whose scratch block starts at S_GNEXT (R; P or -- resident-P mode -- Q); nobody waits here.
        q: when the resident-P mode also fetches Q: "always", or "last" = only behind the LAST pair's doubling step of an iteration
        whose digit is non-zero (the next step is then pair 0's addition step)."""
        L = self.lab
        u = self.uid()
        self._emit_buf_loads(e, (("RX", 4), ("RY", 5), ("RZ", 6)))
        e.salu(f"s_cmp_le_u32 s{S_K}, {self.RES_K}")
        e.salu(f"s_cbranch_scc1 {L(f'L_pf_res_{u}')}")
        self._emit_buf_loads(e, (("PX", 0), ("PY", 1)))
        e.salu(f"s_branch {L(f'L_pf_done_{u}')}")
        e.label(L(f"L_pf_res_{u}"))
        if q == "last":
            e.salu(f"s_add_u32 s{S_TMP0}, s{S_JP}, 1")
            e.salu(f"s_cmp_lt_u32 s{S_TMP0}, s{S_K}")
            e.salu(f"s_cbranch_scc1 {L(f'L_pf_done_{u}')}")           # not the last pair: a doubling step follows
            e.salu(f"s_bitcmp1_b64 {S_NAF_NZ}, s{S_I}")
            e.salu(f"s_cbranch_scc0 {L(f'L_pf_done_{u}')}")           # zero digit: a doubling step follows
        self._emit_buf_loads(e, (("PX", 2), ("PY", 3)))               # the freed P slots take (x2, y2)
        e.label(L(f"L_pf_done_{u}"))

    def _res_p_addr(self, e, v_chunks, v_tail):
        """v_chunks / v_tail <- LDS byte addresses (16-byte chunk plane 0, tail plane) of the packed P of pair S_JP"""
        a, b, c, d = self.RES_P_LDS
        assert (a, b) == (0, 1) and d == c + 1, "slot(j) = j for j < 2, j + c - 2 above"
        e.salu(f"s_cmp_ge_u32 s{S_JP}, 2")
        e.salu(f"s_cselect_b32 s{S_TMP0}, {c - 2}, 0")
        e.salu(f"s_add_u32 s{S_TMP0}, s{S_TMP0}, s{S_JP}")
        e.salu(f"s_lshl_b32 s{S_TMP1}, s{S_TMP0}, {(Prog.N_B128 * 4096).bit_length() - 1}")       # [slot][chunk][lane] uint4: 16 KiB per slot
        e.emit(f"v_add_u32_e32 v{v_chunks}, s{S_TMP1}, v{V_LDS}", vw=[v_chunks])
        e.salu(f"s_lshl_b32 s{S_TMP1}, s{S_TMP0}, 11")                                            # tails: 2 KiB per slot
        e.emit(f"v_add_u32_e32 v{v_tail}, s{S_TMP1}, v{V_LTAIL}", vw=[v_tail])

    def _emit_pack_p(self, e, p):
        """resident-P mode: LDS slot of pair S_JP <- (Px, Py) packed into one slot (S_GBASE = the pair's scratch block)"""
        p.reset_tags()
        p.load(A0, GlobDyn(0))
        p.load(B0, GlobDyn(1))
        p.wait()
        for i in range(NL):
            e.emit(f"v_mov_b32_e32 v{A0 + NL + i}, v{B0 + i}", vw=[A0 + NL + i])
        va, vt = B0, B0 + 1                                   # (block B is dead now)
        L = self.lab
        u = self.uid()
        if self.p3_moved():                                   # pair 3: the chunk planes behind the slots, the tail in two VGPRs (slot 7 holds pair 1's Y)
            e.salu(f"s_cmp_eq_u32 s{S_JP}, 3")
            e.salu(f"s_cbranch_scc1 {L(f'L_pk_p3_{u}')}")
        self._res_p_addr(e, va, vt)
        for c in range(Prog.N_B128):
            e.emit(f"ds_write_b128 v{va}, v[{A0 + 4 * c}:{A0 + 4 * c + 3}] offset:{4096 * c}", kind="lds")
        e.emit(f"ds_write_b64 v{vt}, v[{A0 + 16}:{A0 + 17}]", kind="lds")
        if self.p3_moved():
            e.salu(f"s_branch {L(f'L_pk_done_{u}')}")
            e.label(L(f"L_pk_p3_{u}"))
            e.emit(f"v_add_u32_e32 v{va}, 0x{self.P3_LDS_BASE:x}, v{V_LDS}", vw=[va])
            for c in range(Prog.N_B128):
                e.emit(f"ds_write_b128 v{va}, v[{A0 + 4 * c}:{A0 + 4 * c + 3}] offset:{4096 * c}", kind="lds")
            e.emit(f"v_mov_b32_e32 v{self.V_P3T}, v{A0 + 16}", vw=[self.V_P3T])
            e.emit(f"v_mov_b32_e32 v{self.V_P3T + 1}, v{A0 + 17}", vw=[self.V_P3T + 1])
            e.label(L(f"L_pk_done_{u}"))
        p.reset_tags()

    def _stream_routines(self, sc):
        buf = self.BUF
        Rb, Pb = [buf["RX"], buf["RY"], buf["RZ"]], (buf["PX"], buf["PY"])
        Rout = [GlobDyn(4), GlobDyn(5), GlobDyn(6)]
        res0 = self.r0_resident()
        temps = ([HOME(i) for i in range(N_HOME)] + (list(self.LINE) if LINE_IN_REGS and Prog.FUSED_STEPS else [])
                 + ([] if res0 else self.MILLER_FREE[1]) + [GLOB(GLOB_TMP0 + i) for i in range(8)])
        L = self.lab
        # pair 0's R never leaves the chip: it lives in the three LDS slots the sparse multiplications no longer need (R0_LDS); the
        # steps pick their R source / destination by the pair index
        alt = (lambda e_: e_.salu(f"s_cmp_eq_u32 s{S_JP}, 0"), self.R0_LDS, lambda n: L(f"{n}_{self.uid()}")) if res0 else None
        if self.boustrophedon():                   # the pair that also opens the next pass: R straight into the prefetch buffer
            alt = alt + ((lambda e_: e_.salu(f"s_cmp_eq_u32 s{self.S_NEXTP}, s{S_JP}"), Rb),)
        if self.r1_slots():                        # pair 1: its first coordinates from / to their LDS slots, the others stream
            alt = alt + ((lambda e_: e_.salu(f"s_cmp_eq_u32 s{S_JP}, 1"), self.r1_slots()),)

        def load_p(p):
            """block B <- (Px, Py): from the buffer, or -- resident-P mode -- from the pair's packed LDS slot"""
            e = p.e
            u = self.uid()
            e.salu(f"s_cmp_le_u32 s{S_K}, {self.RES_K}")
            e.salu(f"s_cbranch_scc1 {L(f'L_lp_res_{u}')}")
            p._load_fq(B0, Pb[0])
            p._load_fq(B0 + NL, Pb[1])
            e.salu(f"s_branch {L(f'L_lp_done_{u}')}")
            e.label(L(f"L_lp_res_{u}"))
            va, vt = A0, A0 + 1                               # block A is free until the step itself
            if self.p3_moved():
                e.salu(f"s_cmp_eq_u32 s{S_JP}, 3")
                e.salu(f"s_cbranch_scc1 {L(f'L_lp_p3_{u}')}")
            self._res_p_addr(e, va, vt)
            for c in range(Prog.N_B128):
                e.emit(f"ds_read_b128 v[{B0 + 4 * c}:{B0 + 4 * c + 3}], v{va} offset:{4096 * c}", kind="lds", vw=range(B0 + 4 * c, B0 + 4 * c + 4))
            e.emit(f"ds_read_b64 v[{B0 + 16}:{B0 + 17}], v{vt}", kind="lds", vw=[B0 + 16, B0 + 17])
            e.raw("s_waitcnt lgkmcnt(0)")
            if self.p3_moved():
                e.salu(f"s_branch {L(f'L_lp_done_{u}')}")
                e.label(L(f"L_lp_p3_{u}"))
                e.emit(f"v_add_u32_e32 v{va}, 0x{self.P3_LDS_BASE:x}, v{V_LDS}", vw=[va])
                for c in range(Prog.N_B128):
                    e.emit(f"ds_read_b128 v[{B0 + 4 * c}:{B0 + 4 * c + 3}], v{va} offset:{4096 * c}", kind="lds", vw=range(B0 + 4 * c, B0 + 4 * c + 4))
                e.emit(f"v_mov_b32_e32 v{B0 + 16}, v{self.V_P3T}", vw=[B0 + 16])
                e.emit(f"v_mov_b32_e32 v{B0 + 17}, v{self.V_P3T + 1}", vw=[B0 + 17])
                e.raw("s_waitcnt lgkmcnt(0)")
            e.label(L(f"L_lp_done_{u}"))

        def load_q(p):
            """home blocks 3, 4 <- (x2, y2): the prefetched copy (resident-P mode) or synchronously from the pair's scratch block"""
            e = p.e
            u = self.uid()
            e.salu(f"s_cmp_le_u32 s{S_K}, {self.RES_K}")
            e.salu(f"s_cbranch_scc1 {L(f'L_lq_res_{u}')}")
            p.load(HOME0 + SLOT_DW * 3, GlobDyn(2))
            p.load(HOME0 + SLOT_DW * 4, GlobDyn(3))
            p.wait()
            e.salu(f"s_branch {L(f'L_lq_done_{u}')}")
            e.label(L(f"L_lq_res_{u}"))
            p.load(HOME0 + SLOT_DW * 3, Pb[0])
            p.load(HOME0 + SLOT_DW * 4, Pb[1])
            e.label(L(f"L_lq_done_{u}"))

        def spread(p, q):
            def between(i):
                if i < 4:
                    p.wait()
                    self._emit_prefetch_part(p.e, i, q)
            return between if SPREAD_PREFETCH else None

        def dbl_s(p):
            if not EXP_NO_SWAIT:
                p.e.raw("s_waitcnt vmcnt(0)")                   # the prefetch of this pair has landed
            line = p.dbl_step(Rb, Pb, self.LINE, scale=sc, out=Rout, after_load=(None if SPREAD_PREFETCH else lambda: self._emit_prefetch(p.e, q="last")), load_p=load_p,
                              alt_r=alt)
            p.mul_by_034(self.F, *line, between=spread(p, "last"))

        def add_s(p):
            e = p.e
            if not EXP_NO_SWAIT:
                e.raw("s_waitcnt vmcnt(0)")

            def after():
                # S = +-Q by the sign of the current digit: y2 sits in home block 4
                e.salu(f"s_bitcmp1_b64 {S_NAF_NEG}, s{S_I}")
                e.salu(f"s_cbranch_scc0 {L('L_as_pos')}")
                for i in range(SLOT_DW):
                    r = HOME0 + SLOT_DW * 4 + i
                    e.emit(f"v_sub_u32_e32 v{r}, 0, v{r}", vw=[r])
                e.label(L("L_as_pos"))
                if not SPREAD_PREFETCH:
                    self._emit_prefetch(e, q="always")
            line = p.add_step(Rb, (GlobDyn(2), GlobDyn(3)), Pb, self.LINE, scale=sc, out=Rout, after_load=after, load_p=load_p, load_q=load_q, alt_r=alt)
            p.mul_by_235(self.F, *line, between=spread(p, "always"))

        self.l2_routine("L2_dblmul_s", dbl_s, temps, local=self.LINE)
        self.l2_routine("L2_addmul_s", add_s, temps, local=self.LINE)
        self.l2_routine("L2_prefetch", lambda p: (self._emit_prefetch_part(p.e, 3, "always") if res0 else self._emit_prefetch(p.e, q="always")), temps)

    def pair_select_next(self, e):
        """S_GNEXT <- byte offset of the scratch block of the pair of the NEXT step: pair (S_JP + 1) mod k, or -- alternating passes --
        pair S_JP + S_DIR, and this very pair again when it closes the pass (S_NEXTP <- that pair's index)"""
        if self.boustrophedon():
            e.salu(f"s_add_i32 s{S_TMP0}, s{S_JP}, s{self.S_DIR}")
            e.salu(f"s_cmp_lt_u32 s{S_TMP0}, s{S_K}")                    # unsigned: -1 is out of range too
            e.salu(f"s_cselect_b32 s{S_TMP0}, s{S_TMP0}, s{S_JP}")
            e.salu(f"s_mov_b32 s{self.S_NEXTP}, s{S_TMP0}")
        else:
            e.salu(f"s_add_u32 s{S_TMP0}, s{S_JP}, 1")
            e.salu(f"s_cmp_lt_u32 s{S_TMP0}, s{S_K}")
            e.salu(f"s_cselect_b32 s{S_TMP0}, s{S_TMP0}, 0")
        e.salu(f"s_mul_i32 s{S_TMP0}, s{S_TMP0}, 7")
        e.salu(f"s_add_u32 s{S_TMP0}, s{S_TMP0}, {self.PAIR_SLOT0}")
        e.salu(f"s_mul_i32 s{self.S_GNEXT}, s{S_TMP0}, s{S_GSTRIDE}")

    # ---------------------------------------------------------------------------------------------
    def fexp_main(self, e, p):
        """final_exp_native on F (LDS), F-centric schedule (tests/sched_model.py: final_exp_gpu)."""
        G0, GM, G2, G3, G4, G5, G6, G7 = range(8)
        tr = self.fexp_trace = []         # the straight-line call sequence, replayed by the bound certification

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

        def pf(j):
            """the operand of the NEXT multiplication is fetched under whatever runs in between (nothing there touches the operand slots)"""
            tr.append(("pf", j))
            self.gsel(e, j)
            self.call2(e, "L2_pfB")

        def mulw(conj=False):
            tr.append(("mulw", conj))
            self.call2(e, "L2_mulGc_w" if conj else "L2_mulG_w")

        def powx(j, stored):
            """F <- F^x with register j as the base's scratch register; stored: it already holds F"""
            tr.append(("powx", j, not stored))
            self.gsel(e, j)
            e.salu(f"s_call_b64 {S_RET3}, {self.lab('L3_powx_ns' if stored else 'L3_powx')}")

        def c2(n):
            tr.append(("call", n))
            self.call2(e, n)
        # Scratch is touched only where a value has to outlive the three on-chip Fq12 places (f, the multiplication operand, the
        # LDS register): an operand that IS the current f is copied on chip (cpB), one that has work in front of it is fetched under
        # that work (pf ... mulw); only the second of two back-to-back multiplications waits for its operand.
        cpB = lambda: c2("L2_cpB")                    # operand slots <- f
        mulB = lambda: c2("L2_mul_body")              # f *= the operand slots as they are
        # easy part (:195-206): f2 = conj(a) / a ; f = frob(f2, 2) * f2 -- no scratch at all
        cpB(); c2("L2_inv"); mulw(conj=True); cpB(); c2("L2_frob2"); mulB()
        # hard part (:130-169); m stays in the on-chip register while its three Frobenius images are multiplied together
        st(GM); c2("L2_stL")
        c2("L2_frob1"); cpB()
        c2("L2_ldL"); c2("L2_frob2"); mulB(); cpB()
        c2("L2_ldL"); c2("L2_frob3"); mulB(); st(G2)                  # y0
        c2("L2_ldL"); powx(GM, True)                                  # mx   (the x-power stores its base itself: the next one's
        powx(G3, False)                                               # mx2   base register IS the result register)
        powx(G4, False); st(G5)                                       # mx3
        ld(G3); c2("L2_frob1"); st(G6)                                # mxp
        ld(G4); pf(G3); c2("L2_frob1"); mulw(); st(G7)                # mx * mx2p
        ld(G4); c2("L2_frob2"); st(G3)                                # y2
        ld(G5); cpB(); c2("L2_frob1"); mulB(); c2("L2_conjF")         # y6
        pf(G7); c2("L2_cyc")                                          # T0 = y6^2
        mulw(conj=True)                                               # * y4
        mul(G4, conj=True)                                            # * y5
        c2("L2_stL")                                                  # T0 lives in the on-chip register from here on
        pf(G4); ld(G6, conj=True); mulw(conj=True)                    # T1 = y3 * y5
        c2("L2_mulL"); st(G5)                                         # T1 *= T0
        pf(G3); c2("L2_ldL"); mulw(); c2("L2_stL")                    # T0 = y2 * T0
        ld(G5); c2("L2_cyc"); c2("L2_mulL"); pf(GM); c2("L2_cyc"); st(G5)     # T1 = (T1^2 * T0)^2
        mulw(conj=True); c2("L2_stL")                                 # T0 = T1 * y1
        pf(G2); ld(G5); mulw(); cpB()                                 # T1 = T1 * y0 (-> the operand slots)
        c2("L2_ldL"); c2("L2_cyc"); mulB()                            # T0 = T0^2 * T1

    def store_out(self, e, p):
        p.reset_tags()
        e.emit(f"v_cmp_gt_u32_e32 vcc, s{S_N}, v{V_IDX}", w=["vcc"])      # lanes past the end of the batch do not store
        e.raw("s_nop 1")
        e.salu(f"s_and_saveexec_b64 {S_SAVE_EXEC}, vcc")
        self.io_walk_begin(e, S_OUT, 48, out=True)
        # MyFq12 coefficient c = half * 6 + k goes to position c of the element -- or, element-major in ark's Fq12 order (the `.into()` of
        # src/pairing.rs:21; bn254_myfq12_to_ark_index), to position j(c): the walk then jumps by (j(c) - j(c - 1) - 1) * 32 bytes in between
        ark_pos = {(2 * kk + h) + 6 * ee: (h * 3 + kk) * 2 + ee for h in range(2) for kk in range(3) for ee in range(2)}
        prev = -1
        for half in range(2):
            for k in range(6):
                jump = 32 * (ark_pos[half * 6 + k] - prev - 1)
                prev = ark_pos[half * 6 + k]
                if jump and self.s_mode is not None:
                    e.salu(f"s_bitcmp1_b32 s{self.s_mode}, {MODE_OUT_ARK}")
                    e.salu(f"s_cselect_b32 s{S_TMP0}, 0x{jump & 0xFFFFFFFF:x}, 0")
                    e.salu(f"s_ashr_i32 s{S_TMP1}, s{S_TMP0}, 31")
                    e.salu(f"s_add_u32 s88, s88, s{S_TMP0}")
                    e.salu(f"s_addc_u32 s89, s89, s{S_TMP1}")
                p.load(A0, self.F[k])
                p.wait()
                if half == 1:
                    for i in range(NL):
                        e.emit(f"v_mov_b32_e32 v{A0 + i}, v{A0 + NL + i}", vw=[A0 + i])
                self.cvt_call(e, "cvtout")
                self.io_store_fq(e, A0)
                e.raw("s_nop 1")
        e.emit(f"v_cmp_ne_u32_e32 vcc, 0, v{V_FLAG}", w=["vcc"])          # zero-divisor flag -> status word
        e.raw("s_nop 1")
        e.salu("s_and_saveexec_b64 s[60:61], vcc")
        e.emit(f"v_mov_b32_e32 v{V_IDX8}, 1", vw=[V_IDX8])
        e.emit("v_mov_b32_e32 v36, 0", vw=[36])
        e.emit(f"global_store_dword v36, v{V_IDX8}, {S_STATUS}", kind="vmem")
        e.salu(f"s_mov_b64 exec, {S_SAVE_EXEC}")
        e.raw("s_waitcnt vmcnt(0)")
        e.emit(f"v_lshrrev_b32_e32 v{V_TID}, 4, v{V_LDS}", vw=[V_TID])
