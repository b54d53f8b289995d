#!/usr/bin/env python3
"""Issue-time model of a lone wave on gfx950 (tools/mad_bank_calib.hip measurements): every VALU instruction takes 4 cycles;
a maximal run of N consecutive 'slow-class' instructions (32x32/64-bit multiplies, 64-bit shifts) costs extra cycles."""
import bisect
import re

SLOW = ("v_mad_i64_i32", "v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_hi_i32", "v_ashrrev_i64", "v_lshlrev_b64", "v_lshl_add_u64")
PTS = [(4, 0.0), (5, 0.5), (6, 1.0), (7, 1.55), (8, 2.1), (11, 3.3), (15, 6.2), (23, 9.3)]


def penalty(n):
    if n <= 4:
        return 0.0
    if n >= 23:
        return 9.3 + (n - 23) * 1.0
    xs = [p[0] for p in PTS]
    i = bisect.bisect_right(xs, n) - 1
    (x0, y0), (x1, y1) = PTS[i], PTS[i + 1]
    return y0 + (y1 - y0) * (n - x0) / (x1 - x0)


def cycles(lines):
    n_inst, run, pen, runs = 0, 0, 0.0, []
    for l in lines:
        l = l.strip()
        if not l or l.endswith(":") or l.startswith(("s_nop", ";")):
            continue
        op = l.split()[0]
        n_inst += 1
        if op in SLOW:
            run += 1
        else:
            if run:
                runs.append(run)
            pen += penalty(run)
            run = 0
    pen += penalty(run)
    if run:
        runs.append(run)
    return 4 * n_inst + pen, n_inst, pen, runs


if __name__ == "__main__":
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import kgen3 as K3
    from kgen import Emitter
    for name in ("mul", "mul3", "sqr", "mulfq", "fqmul", "fqsqr", "redn"):
        e = Emitter()
        getattr(K3.L1v3(e), "r_" + name)()
        c, n, pen, runs = cycles(e.finalize())
        print(f"{name:6s} {n:5d} instr  model {c:8.1f} cycles  penalty {pen:6.1f}  runs: n={len(runs)} max={max(runs)} mean={sum(runs)/len(runs):.1f}")
