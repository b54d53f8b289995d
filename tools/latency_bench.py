#!/usr/bin/env python3
"""latency_bench.py -- wall time of ONE bn254_pairing_batch_dev call (device-resident inputs, launch to completion) as a function of
the batch size, on the throughput kernel and on the lane-cooperative kernel (bn254_set_latency_threshold).  Prints a table and
the crossover; writes JSON when given a path.   python tools/latency_bench.py [out.json]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_pkg():
    import importlib
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    return importlib.import_module("plonky2-bn254-pairing_amd")


def main():
    import torch
    pk = load_pkg()
    dev = torch.device("cuda:0")
    sizes = [1, 4, 16, 64, 256, 1024, 2048, 4096, 8192, 16384, 32768, 65536]
    nmax = max(sizes)
    g1 = torch.empty(8 * nmax, dtype=torch.int64, device=dev)
    g2 = torch.empty(16 * nmax, dtype=torch.int64, device=dev)
    out = torch.empty(48 * nmax, dtype=torch.int64, device=dev)
    ref = torch.empty(48 * nmax, dtype=torch.int64, device=dev)
    rows = []
    for n in sizes:
        pk.generate_pairs_dev(0xB2540001, g1, g2, n)       # SoA planes of an n-batch (plane pitch n)
        res = {"n": n}
        for name, thr in (("throughput", 0), ("latency", 1 << 30)):
            pk.set_latency_threshold(thr)
            dst = ref if name == "throughput" else out
            for _ in range(2):
                pk.pairing_batch_dev(g1, g2, dst, n)
            torch.cuda.synchronize()
            reps = 5 if n <= 4096 else 3
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter()
                pk.pairing_batch_dev(g1, g2, dst, n)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            res[name + "_ms"] = round(min(ts) * 1e3, 4)
        pk.last_status()
        res["equal"] = bool(torch.equal(out[:48 * n], ref[:48 * n]))
        rows.append(res)
        print(f"n={n:6d}  throughput {res['throughput_ms']:9.3f} ms   latency {res['latency_ms']:9.3f} ms   equal {res['equal']}", flush=True)
    cross = max([r["n"] for r in rows if r["latency_ms"] < r["throughput_ms"]], default=0)
    print("largest measured batch on which the lane-cooperative kernel is faster:", cross)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            json.dump({"rows": rows, "crossover_n": cross, "device": torch.cuda.get_device_name(0)}, f, indent=1)


if __name__ == "__main__":
    main()
