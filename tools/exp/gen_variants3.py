#!/usr/bin/env python3
"""Experiment 3: how to store an 80-byte slot to LDS from one wave (ds_write_b128 measured 52 cycles each)."""
import json, sys
V = {}
V["5 ds_write_b128 (chunk layout, ref)"] = [f"ds_write_b128 v119, v[{60 + 4 * c}:{63 + 4 * c}] offset:{4096 * c}" for c in range(5)]
V["10 ds_write_b64 (same layout)"] = [f"ds_write_b64 v119, v[{60 + 2 * c}:{61 + 2 * c}] offset:{4096 * (c // 2) + 8 * (c % 2)}" for c in range(10)]
V["5 ds_write2_b64 (same layout)"] = [f"ds_write2_b64 v119, v[{60 + 4 * c}:{61 + 4 * c}], v[{62 + 4 * c}:{63 + 4 * c}] offset0:{(4096 * c) // 8} offset1:{(4096 * c) // 8 + 1}" for c in range(5) if (4096 * c) // 8 + 1 < 256]
V["20 ds_write_b32 (same layout)"] = [f"ds_write_b32 v119, v{60 + c} offset:{4096 * (c // 4) + 4 * (c % 4)}" for c in range(20)]
V["20 ds_write_b32 (limb-major, 4 B lane stride)"] = [f"ds_write_b32 v118, v{60 + c} offset:{1024 * c}" for c in range(20)]
V["10 ds_write_b64 (8 B lane stride)"] = [f"ds_write_b64 v117, v[{60 + 2 * c}:{61 + 2 * c}] offset:{2048 * c}" for c in range(10)]
V["5 ds_read_b128 + wait (ref)"] = [f"ds_read_b128 v[{60 + 4 * c}:{63 + 4 * c}], v119 offset:{4096 * c}" for c in range(5)] + ["s_waitcnt lgkmcnt(0)"]
V["10 ds_read_b64 (8 B lane stride) + wait"] = [f"ds_read_b64 v[{60 + 2 * c}:{61 + 2 * c}], v117 offset:{2048 * c}" for c in range(10)] + ["s_waitcnt lgkmcnt(0)"]
V["20 ds_read_b32 (limb-major) + wait"] = [f"ds_read_b32 v{60 + c}, v118 offset:{1024 * c}" for c in range(20)] + ["s_waitcnt lgkmcnt(0)"]
V["5 ds_write_b128 + 40 mads"] = V["5 ds_write_b128 (chunk layout, ref)"] + ["v_mad_i64_i32 v[40:41], vcc, v2, v3, v[40:41]"] * 40
V["5 ds_write_b128 spaced by 8 mads"] = sum([[V["5 ds_write_b128 (chunk layout, ref)"][c]] + ["v_mad_i64_i32 v[40:41], vcc, v2, v3, v[40:41]"] * 8 for c in range(5)], [])
V["40 mads (ref)"] = ["v_mad_i64_i32 v[40:41], vcc, v2, v3, v[40:41]"] * 40
json.dump(V, open(sys.argv[1], "w"))
