#!/usr/bin/env python3
"""lat_ab_multi.py <k> lib.so ... -- as lat_ab.py for the k-pair products with the final exponentiation (bn254_multi_pairing_batch_dev, do_final_exp = 1)"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
pk = importlib.import_module("plonky2-bn254-pairing_amd")
k = int(sys.argv[1])
dev = torch.device("cuda:0")
sizes = [int(x) for x in os.environ.get("LAT_SIZES", "4096,8192,16384").split(",")]
nmax = max(sizes)
g1 = torch.empty(8 * nmax * k, dtype=torch.int64, device=dev); g2 = torch.empty(16 * nmax * k, dtype=torch.int64, device=dev)
ref = torch.empty(48 * nmax, dtype=torch.int64, device=dev); out = torch.empty(48 * nmax, dtype=torch.int64, device=dev)
st = torch.cuda.current_stream(dev).cuda_stream
libs = [("shipped", pk.load_library())] + [(os.path.basename(p), pk.load_library(p)) for p in sys.argv[2:]]
for n in sizes:
    pk.generate_pairs_dev(0xB2540001, g1, g2, n * k)
    best = {name: 1e9 for name, _ in libs}; same = {name: True for name, _ in libs}
    for rnd in range(4):
        order = libs[rnd % len(libs):] + libs[:rnd % len(libs)]
        for name, lib in order:
            lib.bn254_set_latency_threshold(1 << 30)
            dst = ref if name == "shipped" else out
            ts = []
            for i in range(5):
                t0 = time.perf_counter()
                rc = lib.bn254_multi_pairing_batch_dev(g1.data_ptr(), g2.data_ptr(), dst.data_ptr(), n, k, 1, 0, st)
                torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
                assert rc == 0
            lib.bn254_last_status(0, st)
            if name != "shipped" and rnd > 0:
                same[name] = same[name] and bool(torch.equal(out[:48 * n], ref[:48 * n]))
            best[name] = min(best[name], min(ts[1:]))
    for name, _ in libs:
        print(f"k={k} n={n:6d} {name:28s} {best[name] * 1e3:9.4f} ms  same={same[name]}", flush=True)
