#!/usr/bin/env python3
"""cvm_sim.py -- the lane-cooperative kernel (tools/cvm_kernel.py) on the single-lane interpreter (tools/ksim.py): the lanes of one
group are run ONE AT A TIME, round by round (each up to its next arrival at the round label), sharing the LDS and global-memory
dictionaries.  That is equivalent to lockstep execution because a round never reads a slot written in the same round
(tools/cvm.py: Program._allocate).  Test infrastructure (tests/test_cvm.py)."""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ksim as S  # noqa: E402
import cvm_kernel as CK  # noqa: E402

G1B, G2B, FINB, BLOBB, OUTB, STAT = 0x10000, 0x20000, 0x28000, 0x1000000, 0x30000, 0x40000


def concretize(lines):
    ops = {"%0": "s[2:3]", "%1": "s[4:5]", "%2": "s[6:7]", "%3": "s[8:9]", "%4": "s10", "%5": "s11", "%6": "s[12:13]", "%7": "s14", "%8": "s[16:17]",
           "%9": "v255", "%10": "s18", "%11": "s19"}
    out = []
    for l in lines:
        l = re.sub(r"%(1[01]|\d)(?!\d)", lambda mo: ops["%" + mo.group(1)], l)
        out.append(l.replace("_%=", "_0"))
    return out


def simulate(lines, blob, g1_words=(), g2_words=(), n=1, block=0, tids=None, k=1, fin_words=None):
    """One group of lanes on the program `blob`; g1 / g2 / f_in: the SoA u64 words of the batch (n items of k pairs; a missing
    array is a null pointer, as at the C boundary) -> (global memory, machines, rounds run)"""
    lines = concretize(lines) + ["s_endpgm"]
    if tids is None:
        tids = range(CK.NR)
    lds, gmem = {}, {}
    for i, w in enumerate(blob):
        gmem[BLOBB + 4 * i] = w & 0xFFFFFFFF
    for base, words in ((G1B, g1_words), (G2B, g2_words), (FINB, fin_words or ())):
        for i, w in enumerate(words):
            gmem[base + 8 * i] = w & 0xFFFFFFFF
            gmem[base + 8 * i + 4] = (w >> 32) & 0xFFFFFFFF
    ms = []
    for t in tids:
        m = S.Machine()
        m.lds, m.gmem = lds, gmem
        for name, val in (("s[2:3]", G1B if g1_words else 0), ("s[4:5]", G2B if g2_words else 0), ("s[6:7]", FINB if fin_words else 0), ("s[8:9]", OUTB),
                          ("s10", n), ("s11", k), ("s[12:13]", BLOBB),
                          ("s14", 0), ("s[16:17]", STAT), ("s18", block), ("s19", 1)):
            m.sset(name, val)
        m.v[255] = t
        ms.append(m)
    pcs = [0] * len(ms)
    live = [True] * len(ms)
    rounds = 0
    while any(live):
        for i, m in enumerate(ms):
            if not live[i]:
                continue
            r = S.run(lines, m, start_pc=pcs[i], stop_label="LC_round_0")
            if r is None:
                live[i] = False
            else:
                pcs[i] = r
        rounds += 1
    return gmem, ms, rounds


def soa(elems, n_fq):
    """elems: per element a list of n_fq canonical-Montgomery 4-word tuples -> the SoA word list (plane = Fq number * 4 + word)"""
    out = []
    for fq in range(n_fq):
        for l in range(4):
            out += [e[fq][l] for e in elems]
    return out


def read_fq12(gmem, n=1, item=0):
    """the twelve output Fq of `item` as integers (Montgomery words joined)"""
    out = []
    for c in range(12):
        v = 0
        for l in range(4):
            a = OUTB + ((c * 4 + l) * n + item) * 8
            v |= (gmem[a] | (gmem[a + 4] << 32)) << (64 * l)
        out.append(v)
    return out
