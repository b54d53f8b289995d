"""host_latency.py -- wall time of ONE scalar call through the host-pointer entry points (what the Rust signatures bind to: staging copies,
launch, status read, synchronisation) on the throughput kernels and on the lane-cooperative path.  Run on the GPU box."""
import sys, time, numpy as np
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import helpers as H
pk = H.pkg()
P, Q = H.subgroup_points(4)
g1, g2 = H.g1_aos(P[:1]), H.g2_aos(Q[:1])
g14, g24 = H.g1_aos(P), H.g2_aos(Q)
pk.reserve(8, 4)
for name, fn in (("pairing (host pointers, n = 1)", lambda: pk.pairing_batch(g1, g2, 1)),
                 ("4-pair check (host pointers, 1 group)", lambda: pk.multi_pairing_check_batch(g14, g24, 1, 4)),
                 ("miller_loop_native (host, n = 1)", lambda: pk.miller_loop_batch(g1, g2, 1))):
    for thr in (0, 8192):
        pk.set_latency_threshold(thr)
        ts = []
        for _ in range(12):
            t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
        print(f"{name:42s} threshold {thr:5d}: min {min(ts[2:])*1e3:.3f} ms  median {sorted(ts[2:])[5]*1e3:.3f} ms")
