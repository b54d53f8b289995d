#!/usr/bin/env python3
"""Experiment 4: VGPR bank conflicts of v_mad_i64_i32 with operands that change every instruction (bank = register mod 4)."""
import json, sys
V = {}
acc = "v[40:41]"            # banks 0, 1
def seq(xs, ys, n=40, extra=None, every=0):
    out = []
    for i in range(n):
        out.append(f"v_mad_i64_i32 {acc}, vcc, v{xs[i % len(xs)]}, v{ys[(i * 3) % len(ys)]}, {acc}")
        if every and (i + 1) % every == 0:
            out.append(extra)
    return out
b0 = [4 * j for j in range(1, 9)]          # v4, v8, ... bank 0
b1 = [4 * j + 1 for j in range(1, 9)]
b2 = [4 * j + 2 for j in range(0, 8)]
b3 = [4 * j + 3 for j in range(0, 8)]
V["x bank2, y bank3 (no conflict)"] = seq(b2, b3)
V["x bank2, y bank2"] = seq(b2, b2[::-1])
V["x bank0, y bank1 (= acc banks)"] = seq(b0, b1)
V["x bank0, y bank3"] = seq(b0, b3)
V["x bank2, y bank1"] = seq(b2, b1)
V["x bank0, y bank0"] = seq(b0, b0[::-1])
V["x consecutive regs, y descending (fips-like)"] = [f"v_mad_i64_i32 {acc}, vcc, v{0 + i % 10}, v{29 - (i % 10)}, {acc}" for i in range(40)]
V["x = SGPR, y bank2"] = [f"v_mad_i64_i32 {acc}, vcc, v{b2[i % 8]}, s{36 + i % 10}, {acc}" for i in range(40)]
V["x = SGPR, y bank0"] = [f"v_mad_i64_i32 {acc}, vcc, v{b0[i % 8]}, s{36 + i % 10}, {acc}" for i in range(40)]
V["no conflict + add every 13"] = seq(b2, b3, 39, "v_add_u32_e32 v50, v51, v52", 13)
V["no conflict + add every 4"] = seq(b2, b3, 40, "v_add_u32_e32 v50, v51, v52", 4)
V["acc at v[42:43] (banks 2,3), x bank0, y bank1"] = [l.replace("v[40:41]", "v[42:43]") for l in seq(b0, b1)]
json.dump(V, open(sys.argv[1], "w"))
