#!/usr/bin/env python3
"""Same-box A/B of kernel variants: python tools/exp/ab_bench.py libA.so libB.so [...]  (run on the GPU box through gpurun).
Each library is loaded through the package (ctypes), the 2^20-pairing launch is timed with HIP events, variants interleaved
(A B A B ...) so that clock / thermal drift hits all of them alike.  Results are spot-checked against the oracle."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import numpy as np
    import torch
    pkg = importlib.import_module("plonky2-bn254-pairing_amd")
    import helpers as H
    libs = sys.argv[1:]
    log2 = int(os.environ.get("AB_LOG2", "20"))
    k = int(os.environ.get("AB_K", "1"))               # AB_K=4: the Groth16 shape (2^(log2) pairs in groups of k, shared final exponentiation)
    reps = int(os.environ.get("AB_REPS", "4"))
    n = 1 << log2
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev)
    g1 = torch.zeros(8 * n, dtype=torch.int64, device=dev)
    g2 = torch.zeros(16 * n, dtype=torch.int64, device=dev)
    out = torch.zeros(48 * n, dtype=torch.int64, device=dev)
    handles = [pkg.load_library(os.path.abspath(p)) for p in libs]
    import ctypes
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    S = ctypes.c_void_p(st.cuda_stream)
    assert handles[0].bn254_generate_pairs_dev(0xB2540001, P(g1), P(g2), n, 0, S) == 0
    torch.cuda.synchronize()
    pos = [0, 77, n // 2, n - 1]
    g1h = g1.view(8, n)[:, pos].cpu().numpy().view(np.uint64).reshape(-1).copy()
    g2h = g2.view(16, n)[:, pos].cpu().numpy().view(np.uint64).reshape(-1).copy()
    want = H.oracle_pairing(pkg.layout.to_aos(g1h, 8), pkg.layout.to_aos(g2h, 16), len(pos), threads=4)
    groups = n // k
    if k > 1:
        gp = [0, 77, groups // 2, groups - 1]
        pairs = [g * k + j for g in gp for j in range(k)]
        g1h = g1.view(8, n)[:, pairs].cpu().numpy().view(np.uint64).reshape(-1).copy()
        g2h = g2.view(16, n)[:, pairs].cpu().numpy().view(np.uint64).reshape(-1).copy()
        want = H.oracle_multi_pairing(pkg.layout.to_aos(g1h, 8), pkg.layout.to_aos(g2h, 16), len(gp), k)
        pos = gp
    times = {p: [] for p in libs}
    for r in range(reps + 1):
        for p, h in zip(libs, handles):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(st)
            if k == 1:
                assert h.bn254_pairing_batch_dev(P(g1), P(g2), P(out), n, 0, S) == 0
            else:
                assert h.bn254_multi_pairing_batch_dev(P(g1), P(g2), P(out), groups, k, 1, 0, S) == 0
            b.record(st)
            torch.cuda.synchronize()
            if r:
                times[p].append(a.elapsed_time(b))
            else:
                got = out[:48 * groups].view(48, groups)[:, pos].cpu().numpy().view(np.uint64).reshape(-1).copy()
                if not os.environ.get("AB_NOCHECK"):      # timing-only experiments with deliberately broken variants
                    assert np.array_equal(pkg.layout.to_aos(got, 48), want), f"{p}: wrong results"
    base = None
    for p in libs:
        t = sorted(times[p])
        med = t[len(t) // 2]
        base = base or med
        print(f"{os.path.basename(p):40s} median {med:8.3f} ms  min {t[0]:8.3f}  {n / med / 1e3:7.3f} M pair(ing)s/s  ({100 * (base / med - 1):+.2f} % vs first)")


if __name__ == "__main__":
    main()
