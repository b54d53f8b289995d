#!/usr/bin/env python3
"""clock_stamp.py <lib_stamp.so> -- the in-kernel shader clock of k_pairing / k_mpairing (MI355X guide, 'DVFS give-back' item 6).

Needs a DIAGNOSTIC library:  HIPCC_EXTRA=-DBN254_DEBUG_STAMPS tools/exp/build_variant.sh stamp KGEN_CLOCK_STAMP=1
whose kernels stamp s_memtime (shader cycles) and s_memrealtime (100 MHz) once in front of and once behind each wave's item
loop and leave the differences in the slack at the end of the workgroup's scratch block (no output value depends on them; the
shipped kernels execute no stamp).  After >= 2 s of back-to-back launches on random (generated) inputs the stamps of the LAST launch are
read back:  clock = d(memtime) / d(memrealtime) x 100 MHz, median over the waves of all workgroups; the wall time of that launch
from HIP events and the busy share (wave time / launch time) go with it.  Run on the GPU box through gpurun."""
import ctypes
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    pkg = importlib.import_module("plonky2-bn254-pairing_amd")
    lib = pkg.load_library(os.path.abspath(sys.argv[1]))
    lib.bn254_debug_stamps.restype = ctypes.c_int
    lib.bn254_debug_stamps.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    soak_s = float(os.environ.get("STAMP_SOAK_S", "2.5"))
    n = 1 << 20
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    S = ctypes.c_void_p(st.cuda_stream)
    g1 = torch.zeros(8 * n, dtype=torch.int64, device=dev)
    g2 = torch.zeros(16 * n, dtype=torch.int64, device=dev)
    out = torch.zeros(48 * n, dtype=torch.int64, device=dev)
    assert lib.bn254_generate_pairs_dev(0xB2540001, P(g1), P(g2), n, 0, S) == 0
    torch.cuda.synchronize()
    res = {}
    for name, k in (("k_pairing (configs[2]: 2^20 pairings)", 1), ("k_mpairing (configs[3]: 2^18 groups x 4 pairs)", 4)):
        def launch():
            if k == 1:
                assert lib.bn254_pairing_batch_dev(P(g1), P(g2), P(out), n, 0, S) == 0
            else:
                assert lib.bn254_multi_pairing_batch_dev(P(g1), P(g2), P(out), n // k, k, 1, 0, S) == 0
        launch()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st)
        launch()
        b.record(st)
        torch.cuda.synchronize()
        one = a.elapsed_time(b)
        reps = max(3, int(soak_s * 1e3 / one) + 1)
        for _ in range(reps - 1):                       # back-to-back: the queue never drains
            launch()
        a.record(st)
        launch()
        b.record(st)
        torch.cuda.synchronize()
        ms = a.elapsed_time(b)
        buf = np.zeros(8 * 512, dtype=np.uint64)
        grid = lib.bn254_debug_stamps(0, S, buf.ctypes.data_as(ctypes.c_void_p), 512)
        assert grid > 0, grid
        w = buf[: 8 * grid].reshape(grid, 4, 2).astype(np.float64)
        dt_clk, dt_real = w[:, :, 0].reshape(-1), w[:, :, 1].reshape(-1)
        assert (dt_real > 0).all() and (dt_clk > 0).all()
        clk = dt_clk / dt_real * 100e6
        wave_ms = dt_real / 100e6 * 1e3
        res[name] = {"launches_back_to_back": reps, "soak_s": reps * one / 1e3, "launch_ms_hip_events": ms,
                     "in_kernel_clock_ghz_median": float(np.median(clk)) / 1e9, "in_kernel_clock_ghz_min": float(clk.min()) / 1e9,
                     "in_kernel_clock_ghz_max": float(clk.max()) / 1e9, "waves": int(clk.size),
                     "wave_ms_median": float(np.median(wave_ms)), "wave_ms_max": float(wave_ms.max()),
                     "shader_cycles_per_wave_median": float(np.median(dt_clk)),
                     "note": "clock = d(s_memtime) / d(s_memrealtime) x 100 MHz around each wave's item loop, last of the back-to-back launches"}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
