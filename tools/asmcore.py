#!/usr/bin/env python3
"""asmcore.py -- shared plumbing of the gfx950 kernel generator (tools/kgen4.py, tools/kgen4_prog.py):

  * Emitter: instruction list with SGPR/VCC read-write annotations and the hazard post-pass
    (gfx940/gfx950: a VALU may not read an SGPR/VCC a VALU wrote < 2 instructions earlier; a wide global store's data
    registers may not be overwritten by a VALU < 2 instructions later),
  * Pool: temporary VGPR allocator (even-aligned pairs for 64-bit operands),
  * align_code / insn_size / max_branch_distance: the code-alignment post-pass (every 8-byte instruction 8-byte aligned --
    a lone wave loses ~1 cycle per misaligned 8-byte instruction) and the +-128 KB branch-range check.
"""
import re

P_INT = 21888242871839275222246405745257275088696311157297823662689037894645226208583
BN_X = 4965661367192848881
SIX_U_PLUS_2_NAF = [
    0, 0, 0, 1, 0, 1, 0, -1, 0, 0, 1, -1, 0, 0, 1, 0, 0, 1, 1, 0, -1, 0, 0, 1, 0, -1, 0, 0, 0, 0,
    1, 1, 1, 0, 0, -1, 0, 0, 1, 0, 0, 0, 0, 0, -1, 0, 0, 1, 1, 0, 0, -1, 0, 0, 0, 1, 1, 0, -1, 0,
    0, 1, 0, 1, 1,
]


def canonical_naf(n):
    """the non-adjacent form of n, least significant digit first (minimal weight among the signed binary representations)"""
    out = []
    while n:
        z = 2 - (n % 4) if n & 1 else 0
        n -= z
        out.append(z)
        n >>= 1
    return out


# The reference's table above has 65 digits, 26 of them non-zero: 64 doublings and 25 additions behind the top digit.  The canonical NAF of
# 6 x + 2 has 66 digits and 22 non-zero ones (one doubling more, four additions less); the NAF of 6 x + 2 - 2^64 under the top digit 2^64 has
# 65 digits and 22 non-zero ones -- the same 64 doublings as the reference, FOUR additions less (no signed binary form of 65 digits has fewer
# non-zero digits: checked exhaustively by tests/test_consts.py).  The Miller VALUE depends on the chain (by factors from proper subfields:
# vertical lines, the projective lines' scales), pairing(p, q) = final_exp_native(miller_loop_native(q, p)) does not -- (p^6 - 1) kills
# them -- so every path that ends in the final exponentiation walks this form (round 5); miller_loop_native / multi_miller_loop_native
# themselves keep the reference's table, digit for digit.
SIX_U_PLUS_2_SHORT = (lambda t: t + [0] * (64 - len(t)) + [1])(canonical_naf(6 * BN_X + 2 - (1 << 64)))
assert len(SIX_U_PLUS_2_SHORT) == 65 and sum(1 for d in SIX_U_PLUS_2_SHORT if d) == 22
assert sum(d << k for k, d in enumerate(SIX_U_PLUS_2_SHORT)) == sum(d << k for k, d in enumerate(SIX_U_PLUS_2_NAF)) == 6 * BN_X + 2


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


def align_code(lines, src_map=None, nop_only=False):
    """src_map (optional, an empty list): receives, for every output line, the index of the input line it is (None for an
    inserted s_nop); re-encoded lines keep their index.  nop_only: never re-encode a 32-bit VOP1 / VOP2 instruction as VOP3 to gain the four
    bytes -- pad with `s_nop 0` instead (kernels that run two waves per SIMD: the VOP3 form costs 4 issue cycles there, the 32-bit form 2, and
    the other wave issues under the s_nop; profiles/r06_occupancy_calib.txt)."""
    out, off, last = [], 0, None            # last: index in `out` of the previous instruction if it may be re-encoded
    src = [] if src_map is None else src_map

    def put(line, i):
        out.append(line)
        src.append(i)

    for i, ln in enumerate(lines):
        t = ln.strip()
        if not t or t.startswith((";", "//", ".")):
            put(ln, i)
            continue
        if t.endswith(":"):
            if off % 8:
                put("s_nop 0", None)
                off += 4
            put(ln, i)
            last = None
            continue
        size = insn_size(t)
        if size == 8 and off % 8:
            conv = _to_e64(out[last]) if (last is not None and not nop_only) else None
            if conv is not None:
                out[last] = conv
            else:
                put("s_nop 0", None)
            off += 4
        put(ln, i)
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


def branch_table(lines):
    """(label -> byte offset, [(byte offset, line index, opcode, target label)]) of all label-targeted branches / calls"""
    off, lab, ins = 0, {}, []
    for idx, l in enumerate(lines):
        t = l.strip()
        if not t or t.startswith((";", "//", ".")):
            continue
        if t.endswith(":"):
            lab[t[:-1]] = off
            continue
        op = t.split()[0]
        if op in ("s_call_b64", "s_branch") or op.startswith("s_cbranch"):
            tgt = t.split(",")[-1].strip() if op == "s_call_b64" else t.split()[-1]
            ins.append((off, idx, op, tgt))
        off += insn_size(t)
    return lab, ins




def place_with_islands(blocks, reach, mk_label):
    """Concatenates code blocks (lists of lines; no block falls through into the next) and makes every s_call_b64 / s_branch /
    s_cbranch reach its target within `reach` bytes: a transfer that cannot is retargeted to a one-instruction trampoline
    (`label: s_branch target`) in an island between two blocks near the midpoint; trampolines hop again if needed.  A
    trampoline costs one extra taken branch and leaves the return address untouched.  Returns (lines, trampolines)."""
    islands = [[] for _ in range(len(blocks) + 1)]
    blocks = [list(b) for b in blocks]
    n_hops = 0
    for _ in range(400):
        flat, where = [], []                  # the concatenation and, per line, (collection, index) it came from
        for i in range(len(blocks) + 1):
            flat.append(mk_label(f"L_isl{i}") + ":")
            where.append(None)
            for j, l in enumerate(islands[i]):
                flat.append(l)
                where.append((islands[i], j))
            if i < len(blocks):
                for j, l in enumerate(blocks[i]):
                    flat.append(l)
                    where.append((blocks[i], j))
        src = []
        out = align_code(flat, src)
        lab, ins = branch_table(out)
        bad = [(abs(lab[t] - (o + 4)), o, idx, t) for o, idx, op, t in ins if t in lab and abs(lab[t] - (o + 4)) >= reach]
        if not bad:
            return [".p2align 3"] + out, n_hops
        _, o, idx, t = max(bad)
        mid = (o + lab[t]) // 2
        k = min(range(len(blocks) + 1), key=lambda i: abs(lab[mk_label(f"L_isl{i}")] - mid))
        hop = mk_label(f"L_hop{n_hops}")
        n_hops += 1
        coll, j = where[src[idx]]
        line = coll[j]
        assert line.rstrip().endswith(t), (line, t)
        coll[j] = line[: line.rfind(t)] + hop
        islands[k] += [hop + ":", f"s_branch {t}"]
    raise RuntimeError("island placement did not converge")
