#!/usr/bin/env python3
"""Experiment 2: cost of slot marshalling at 1 wave/SIMD (LDS with per-lane addresses, AGPR, home moves) and SALU."""
import json, sys
V = {}
mad = "v_mad_i64_i32 v[40:41], vcc, v2, v3, v[40:41]"
rd = [f"ds_read_b128 v[{60 + 4 * c}:{63 + 4 * c}], v119 offset:{4096 * c}" for c in range(5)]
wr = [f"ds_write_b128 v119, v[{60 + 4 * c}:{63 + 4 * c}] offset:{4096 * c}" for c in range(5)]
V["slot load LDS (5 ds_read_b128 + wait)"] = rd + ["s_waitcnt lgkmcnt(0)"]
V["2 slot loads LDS + wait"] = rd + [l.replace("v[6", "v[8").replace(":6", ":8").replace("v[7", "v[9").replace(":7", ":9") for l in rd] + ["s_waitcnt lgkmcnt(0)"]
V["slot store LDS (5 ds_write_b128)"] = wr
V["slot store LDS + wait"] = wr + ["s_waitcnt lgkmcnt(0)"]
V["slot load LDS + 40 mads + wait"] = rd + [mad] * 40 + ["s_waitcnt lgkmcnt(0)"]
V["40 mads (ref)"] = [mad] * 40
V["slot load home (20 v_mov)"] = [f"v_mov_b32_e32 v{60 + i}, v{90 + i}" for i in range(20)]
V["slot load AGPR (20 accvgpr_read)"] = [f"v_accvgpr_read_b32 v{60 + i}, a{i}" for i in range(20)]
V["s_call + s_setpc round trip"] = ["s_getpc_b64 s[50:51]", "s_add_u32 s50, s50, 16", "s_addc_u32 s51, s51, 0", "s_setpc_b64 s[50:51]"]
V["4 mads + s_mul_i32 + s_add_u32"] = [mad] * 4 + ["s_mul_i32 s50, s36, 7", "s_add_u32 s50, s50, s37"]
V["4 mads + s_mov_b32"] = [mad] * 4 + ["s_mov_b32 s50, s36"]
V["4 mads + 2 s_mov_b32"] = [mad] * 4 + ["s_mov_b32 s50, s36", "s_mov_b32 s51, s37"]
V["8 adds + s_mov_b32"] = ["v_add_u32_e32 v50, v51, v52"] * 8 + ["s_mov_b32 s50, s36"]
V["8 adds + s_nop 0"] = ["v_add_u32_e32 v50, v51, v52"] * 8 + ["s_nop 0"]
V["8 adds + s_waitcnt"] = ["v_add_u32_e32 v50, v51, v52"] * 8 + ["s_waitcnt lgkmcnt(0)"]
V["8 adds + s_bitcmp1 + s_cbranch (not taken)"] = ["v_add_u32_e32 v50, v51, v52"] * 8 + ["s_bitcmp1_b32 s36, 31", "s_cbranch_scc1 L_never_%="]
json.dump(V, open(sys.argv[1], "w"))
