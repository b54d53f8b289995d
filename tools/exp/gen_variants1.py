#!/usr/bin/env python3
"""Experiment 1: instruction classes at 1 wave/SIMD, and the shipped L1 routines."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import kgen3 as K3
from kgen import Emitter

V = {}
def rep(lines, n): return lines * n
mad = "v_mad_i64_i32 v[40:41], vcc, v2, v3, v[40:41]"
singles = {
 "v_sub_u32_e32": ["v_sub_u32_e32 v50, v51, v52", "v_sub_u32_e32 v53, v51, v52"],
 "v_lshlrev_b32_e32": ["v_lshlrev_b32_e32 v50, 5, v52", "v_lshlrev_b32_e32 v53, 5, v52"],
 "v_ashrrev_i32_e32": ["v_ashrrev_i32_e32 v50, 27, v52", "v_ashrrev_i32_e32 v53, 27, v52"],
 "v_lshl_add_u32": ["v_lshl_add_u32 v50, v51, 3, v52", "v_lshl_add_u32 v53, v51, 3, v52"],
 "v_accvgpr_read_b32": ["v_accvgpr_read_b32 v50, a1", "v_accvgpr_read_b32 v53, a2"],
 "v_accvgpr_write_b32": ["v_accvgpr_write_b32 a1, v50", "v_accvgpr_write_b32 a2, v53"],
 "v_alignbit_b32": ["v_alignbit_b32 v50, v51, v52, 27", "v_alignbit_b32 v53, v51, v52, 27"],
 "v_bfe_i32": ["v_bfe_i32 v50, v51, 0, 27", "v_bfe_i32 v53, v51, 0, 27"],
 "v_add3_u32": ["v_add3_u32 v50, v51, v52, v54", "v_add3_u32 v53, v51, v52, v54"],
 "v_and_or_b32": ["v_and_or_b32 v50, v51, v52, v54", "v_and_or_b32 v53, v51, v52, v54"],
 "v_mul_u32_u24_e32": ["v_mul_u32_u24_e32 v50, v51, v52", "v_mul_u32_u24_e32 v53, v51, v52"],
 "v_mul_hi_u32_u24_e32": ["v_mul_hi_u32_u24_e32 v50, v51, v52", "v_mul_hi_u32_u24_e32 v53, v51, v52"],
 "v_mul_i32_i24_e32": ["v_mul_i32_i24_e32 v50, v51, v52", "v_mul_i32_i24_e32 v53, v51, v52"],
 "v_add_co_u32+v_addc": ["v_add_co_u32_e32 v50, vcc, v51, v52", "v_addc_co_u32_e32 v53, vcc, v54, v55, vcc"],
 "v_mad_i64_i32 (ref)": [mad, mad],
 "v_add_u32_e32 (ref)": ["v_add_u32_e32 v50, v51, v52", "v_add_u32_e32 v53, v51, v52"],
 "v_xor_b32_e32": ["v_xor_b32_e32 v50, v51, v52", "v_xor_b32_e32 v53, v51, v52"],
 "v_mov_b32 literal": ["v_mov_b32_e32 v50, 0x1234567", "v_mov_b32_e32 v53, 0x7654321"],
 "v_pk_add_u16": ["v_pk_add_u16 v50, v51, v52", "v_pk_add_u16 v53, v51, v52"],
 "v_cndmask_b32_e32": ["v_cndmask_b32_e32 v50, v51, v52, vcc", "v_cndmask_b32_e32 v53, v51, v52, vcc"],
}
for k, v in singles.items():
    V["class: " + k] = rep(v, 24)
breakers = {
 "accvgpr_read": "v_accvgpr_read_b32 v50, a1", "accvgpr_write": "v_accvgpr_write_b32 a1, v50", "v_sub_e32": "v_sub_u32_e32 v50, v51, v52",
 "ashrrev_i32_e32": "v_ashrrev_i32_e32 v50, 27, v52", "lshl_add_u32": "v_lshl_add_u32 v50, v51, 3, v52", "ds_read_b128": "ds_read_b128 v[60:63], v119",
 "ds_write_b128": "ds_write_b128 v119, v[60:63]", "v_mul_u32_u24_e32": "v_mul_u32_u24_e32 v50, v51, v52",
}
for k, b in breakers.items():
    V[f"mad x4 + {k}"] = rep([mad] * 4 + [b], 10)
V["mad x4 + ds_read, wait at end"] = rep([mad] * 4 + ["ds_read_b128 v[60:63], v119"], 10) + ["s_waitcnt lgkmcnt(0)"]
V["mad x12 + s_waitcnt"] = rep([mad] * 12 + ["s_waitcnt lgkmcnt(0)"], 4)
V["mad x12 + s_nop 0"] = rep([mad] * 12 + ["s_nop 0"], 4)

# shipped L1 routines
for name in ("mul", "mul3", "sqr", "mulfq", "norm", "mulxi", "add", "sub"):
    e = Emitter()
    g = K3.L1v3(e)
    getattr(g, "r_" + name)()
    V["L1 " + name] = [l.replace("%=", "0") for l in e.finalize() if not l.strip().startswith(("s_setpc",))]
json.dump(V, open(sys.argv[1], "w"), indent=0)
print({k: len(v) for k, v in V.items()})
