#!/usr/bin/env python3
"""lat_ab.py lib_a.so [lib_b.so ...] -- wall time of one bn254_pairing_batch_dev call on the lane-cooperative kernel of each library
(variants from tools/exp/lat_variant.sh), for a few batch sizes; results are compared with the shipped library's (`same`)."""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch
    pk = importlib.import_module("plonky2-bn254-pairing_amd")
    dev = torch.device("cuda:0")
    sizes = [int(x) for x in os.environ.get("LAT_SIZES", "1,1024,4096").split(",")]
    nmax = max(sizes)
    g1 = torch.empty(8 * nmax, dtype=torch.int64, device=dev)
    g2 = torch.empty(16 * nmax, dtype=torch.int64, device=dev)
    ref = torch.empty(48 * nmax, dtype=torch.int64, device=dev)
    out = torch.empty(48 * nmax, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    libs = [("shipped", pk.load_library())] + [(os.path.basename(p), pk.load_library(p)) for p in sys.argv[1:]]
    if os.environ.get("LAT_ONLY"):              # counter runs (tools/exp/lat_pmc2.sh): the named libraries alone (the package's own library only generates the inputs)
        libs = libs[1:]
    for n in sizes:
        pk.generate_pairs_dev(0xB2540001, g1, g2, n)
        best = {name: 1e9 for name, _ in libs}
        same = {name: True for name, _ in libs}
        for rnd in range(4):                    # the libraries take turns (and the order rotates): no library is always the first after an idle gap
            order = libs[rnd % len(libs):] + libs[:rnd % len(libs)]
            for name, lib in order:
                lib.bn254_set_latency_threshold(1 << 30)
                dst = ref if name == "shipped" else out
                ts = []
                for i in range(6):
                    t0 = time.perf_counter()
                    rc = lib.bn254_pairing_batch_dev(g1.data_ptr(), g2.data_ptr(), dst.data_ptr(), n, 0, st)
                    torch.cuda.synchronize()
                    ts.append(time.perf_counter() - t0)
                    assert rc == 0
                lib.bn254_last_status(0, st)
                if name != "shipped" and rnd > 0:
                    same[name] = same[name] and bool(torch.equal(out[:48 * n], ref[:48 * n]))
                best[name] = min(best[name], min(ts[2:]))
        for name, _ in libs:
            print(f"n={n:6d} {name:28s} {best[name] * 1e3:9.4f} ms  same={same[name]}", flush=True)


if __name__ == "__main__":
    main()
