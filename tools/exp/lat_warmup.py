#!/usr/bin/env python3
"""lat_warmup.py -- six consecutive mid-size pairing calls on the lane-cooperative kernels after an idle gap, with and without launches of
the throughput kernel in front, buffers of one and of four times the size: the call time falls call by call (the shader clock ramps up
from idle: 2.22 -> 2.07 ms over six calls of 8 192 pairings, 1.89 ms in a sustained stream -- tools/exp/lat_ab.py), whatever the variant."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
pk = importlib.import_module("plonky2-bn254-pairing_amd")
dev = torch.device("cuda:0")
def run(tag, n, nmax, use_reserve, big_first):
    g1 = torch.empty(8 * nmax, dtype=torch.int64, device=dev); g2 = torch.empty(16 * nmax, dtype=torch.int64, device=dev)
    out = torch.empty(48 * nmax, dtype=torch.int64, device=dev); ref = torch.empty(48 * nmax, dtype=torch.int64, device=dev)
    pk.generate_pairs_dev(0xB2540001, g1, g2, n)
    if use_reserve: pk.reserve(nmax, 1)
    if big_first:
        pk.set_latency_threshold(0)
        for _ in range(4): pk.pairing_batch_dev(g1, g2, ref, n)
        torch.cuda.synchronize()
    pk.set_latency_threshold(1 << 30)
    for _ in range(2): pk.pairing_batch_dev(g1, g2, out, n)
    torch.cuda.synchronize()
    ts = []
    for _ in range(6):
        t0 = time.perf_counter(); pk.pairing_batch_dev(g1, g2, out, n); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(tag, n, [round(t * 1e3, 3) for t in ts], flush=True)
run("plain", 8192, 8192, False, False)
run("big first", 8192, 8192, False, True)
run("reserve", 8192, 8192, True, False)
run("nmax 4x", 8192, 32768, False, False)
run("big first, nmax 4x", 8192, 32768, False, True)
