#!/usr/bin/env python3
"""lat_after_big.py -- what a mid-size call on the lane-cooperative kernels costs right after a launch of the throughput kernel (the
alternating harness of tools/latency_bench.py) against the same call repeated: first / second / third call after the big launch."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
pk = importlib.import_module("plonky2-bn254-pairing_amd")
dev = torch.device("cuda:0")
nbig = 1 << 16
for n in (8192, 16384):
    g1 = torch.empty(8 * nbig, dtype=torch.int64, device=dev); g2 = torch.empty(16 * nbig, dtype=torch.int64, device=dev)
    out = torch.empty(48 * nbig, dtype=torch.int64, device=dev)
    pk.generate_pairs_dev(0xB2540001, g1, g2, nbig)
    pk.reserve(nbig, 1)
    res = {0: [], 1: [], 2: []}
    for rep in range(8):
        pk.set_latency_threshold(0)
        pk.pairing_batch_dev(g1, g2, out, nbig); torch.cuda.synchronize()
        pk.set_latency_threshold(1 << 30)
        for j in range(3):
            t0 = time.perf_counter(); pk.pairing_batch_dev(g1, g2, out, n); torch.cuda.synchronize(); res[j].append(time.perf_counter() - t0)
    print(n, {j: round(sorted(v[2:])[len(v[2:]) // 2] * 1e3, 4) for j, v in res.items()}, "ms (median): 1st / 2nd / 3rd call after a 2^16-pairing launch of the throughput kernel", flush=True)
