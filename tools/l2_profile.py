#!/usr/bin/env python3
"""l2_profile.py -- cycles per instruction of every L2 routine of k_pairing.

  python tools/l2_profile.py sim  out.json                 (CPU: instructions and calls per routine, inclusive, from tools/ksim.py)
  python tools/l2_profile.py gpu  lib_prof.so sim.json     (GPU box: shader cycles per routine from a KGEN_PROFILE_L2 diagnostic library,
                                                            joined with the simulator's counts)

The diagnostic library (HIPCC_EXTRA=-DBN254_DEBUG_STAMPS tools/exp/build_variant.sh prof KGEN_CLOCK_STAMP=1 KGEN_PROFILE_L2=1) brackets every
call of an L2 routine from the main program / the x-power control code with s_memtime stamps and accumulates cycles and calls per
routine in lanes of three spare VGPRs; the shipped kernels execute no stamp.  4.000 cycles per instruction = every issue slot used."""
import ctypes
import importlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests"), ROOT]


def sim(out):
    import kgen4_prog as K4P
    import test_kgen4 as T
    import helpers as H
    vec = H.load_golden("bn254_vectors.json")
    g1, g2 = T._inputs(vec, 3)
    kb = K4P.KernelBuilder(do_miller=True, do_fexp=True)
    res, m = T.run_kernel(kb, g1, g2, profile=True)
    assert res == T.HX(vec["pairing"][3])
    # calls that cannot reach their routine (+-128 KB) go through one-instruction trampolines L_hopN: name them by their targets
    lines = kb.build()
    hop = {}
    for i, l in enumerate(lines):
        mo = re.match(r"(L_hop\d+)_%=:$", l)
        if mo:
            hop[mo.group(1)] = lines[i + 1].split()[-1].replace("_%=", "")
    strip = lambda s_: hop.get(re.sub(r"_\d+$", "", s_), re.sub(r"_\d+$", "", s_))
    calls = {}
    for name in m.call_log:
        calls[strip(name)] = calls.get(strip(name), 0) + 1
    incl = {}
    for k, v in m.l2_incl.items():
        incl[strip(k)] = incl.get(strip(k), 0) + v
    json.dump({"total_instructions": m.count, "inclusive_instructions": incl, "calls": calls}, open(out, "w"), indent=1)
    print(json.dumps(incl, indent=1))


def gpu(lib_path, sim_json):
    import numpy as np
    import torch
    pkg = importlib.import_module("plonky2-bn254-pairing_amd")
    lib = pkg.load_library(os.path.abspath(lib_path))
    lib.bn254_debug_profile.restype = ctypes.c_int
    lib.bn254_debug_profile.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    lib.bn254_debug_profile_ids.restype = ctypes.c_char_p
    lib.bn254_debug_stamps.restype = ctypes.c_int
    lib.bn254_debug_stamps.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    ids = lib.bn254_debug_profile_ids().decode().split(",")
    simd = json.load(open(sim_json))
    n = 1 << 20
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    S = ctypes.c_void_p(st.cuda_stream)
    g1 = torch.zeros(8 * n, dtype=torch.int64, device=dev)
    g2 = torch.zeros(16 * n, dtype=torch.int64, device=dev)
    out = torch.zeros(48 * n, dtype=torch.int64, device=dev)
    assert lib.bn254_generate_pairs_dev(0xB2540001, P(g1), P(g2), n, 0, S) == 0
    for _ in range(12):
        assert lib.bn254_pairing_batch_dev(P(g1), P(g2), P(out), n, 0, S) == 0
    torch.cuda.synchronize()
    buf = np.zeros(768 * 512, dtype=np.uint32)
    grid = lib.bn254_debug_profile(0, S, buf.ctypes.data_as(ctypes.c_void_p), 512)
    assert grid > 0
    w = buf[: 768 * grid].reshape(grid * 4, 3, 64).astype(np.uint64)
    cyc = (w[:, 0, :] | (w[:, 1, :] << np.uint64(32))).astype(np.float64)           # [wave][id]
    calls = w[:, 2, :].astype(np.float64)
    sb = np.zeros(8 * 512, dtype=np.uint64)
    lib.bn254_debug_stamps(0, S, sb.ctypes.data_as(ctypes.c_void_p), 512)
    tot = sb[: 8 * grid].reshape(grid, 4, 2)[:, :, 0].astype(np.float64).reshape(-1)
    items = n / 256 / grid                                                          # work items per wave
    rows = []
    for i, name in enumerate(ids):
        c, k = float(np.median(cyc[:, i])) / items, float(np.median(calls[:, i])) / items
        ins = simd["inclusive_instructions"].get(name)
        rows.append((c, name, k, ins))
    rows.sort(reverse=True)
    covered = sum(r[0] for r in rows)
    res = {"what": "k_pairing, 2^20 lanes, per work item (one lane's pairing): shader cycles inside each L2 routine (inclusive of the leaf routines it calls; "
                   "median over the waves) against the simulator's instruction count of the same routine; the stamps themselves cost ~60 cycles per call",
           "wave_cycles_per_item": float(np.median(tot)) / items, "cycles_inside_l2_routines": covered, "routines": {}}
    print(f"{'routine':16s} {'calls':>7s} {'cycles':>12s} {'instr':>10s} {'cyc/instr':>9s}   share")
    for c, name, k, ins in rows:
        if k == 0:
            continue
        res["routines"][name] = {"calls": k, "cycles": c, "instructions_sim": ins, "cycles_per_instruction": (c / ins if ins else None)}
        print(f"{name:16s} {k:7.0f} {c:12.0f} {ins or 0:10d} {(c / ins if ins else 0):9.3f}   {100 * c / (float(np.median(tot)) / items):5.1f} %")
    print("wave cycles per item", res["wave_cycles_per_item"], "inside L2 routines", covered)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", os.environ.get("L2_PROFILE_OUT", "r05_l2_profile.json")), "w"), indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "sim":
        sim(sys.argv[2])
    else:
        gpu(sys.argv[2], sys.argv[3])
