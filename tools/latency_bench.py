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


SIGNATURES = (("pairing", 1), ("miller_loop_native", 1), ("final_exp_native", 1), ("multi_pairing k=2", 2), ("multi_pairing k=4 (Groth16 shape)", 4),
              ("multi_miller_loop_native k=2", 2), ("multi_miller_loop_native k=4", 4))


def main():
    import torch
    pk = load_pkg()
    dev = torch.device("cuda:0")
    sizes = [int(x) for x in os.environ.get("LAT_SIZES", "1,64,1024,2048,4096,8192,16384,32768").split(",")]
    nmax = max(sizes) * 4
    g1 = torch.empty(8 * nmax, dtype=torch.int64, device=dev)
    g2 = torch.empty(16 * nmax, dtype=torch.int64, device=dev)
    fin = torch.empty(48 * max(sizes), dtype=torch.int64, device=dev)
    out = torch.empty(48 * max(sizes), dtype=torch.int64, device=dev)
    ref = torch.empty(48 * max(sizes), dtype=torch.int64, device=dev)
    report = {"device": torch.cuda.get_device_name(0), "signatures": {}}
    for name, k in SIGNATURES:
        rows = []
        for n in sizes:
            pk.generate_pairs_dev(0xB2540001, g1, g2, n * k)       # SoA planes of an (n k)-batch
            if name == "final_exp_native":
                pk.set_latency_threshold(0)
                pk.miller_loop_batch_dev(g1, g2, fin, n)

            def call(dst):
                if name == "pairing":
                    pk.pairing_batch_dev(g1, g2, dst, n)
                elif name == "miller_loop_native":
                    pk.miller_loop_batch_dev(g1, g2, dst, n)
                elif name == "final_exp_native":
                    pk.final_exp_batch_dev(fin, dst, n)
                else:
                    pk.multi_pairing_batch_dev(g1, g2, dst, n, k, do_final_exp=name.startswith("multi_pairing"))
            res = {"n": n}
            for kern, thr in (("throughput", 0), ("latency", 1 << 30)):
                pk.set_latency_threshold(thr)
                dst = ref if kern == "throughput" else out
                for _ in range(2):
                    call(dst)
                torch.cuda.synchronize()
                ts = []
                for _ in range(5 if n <= 4096 else 3):
                    t0 = time.perf_counter()
                    call(dst)
                    torch.cuda.synchronize()
                    ts.append(time.perf_counter() - t0)
                res[kern + "_ms"] = round(min(ts) * 1e3, 4)
            pk.last_status()
            res["equal"] = bool(torch.equal(out[:48 * n], ref[:48 * n]))
            rows.append(res)
            print(f"{name:36s} n={n:6d}  throughput {res['throughput_ms']:9.3f} ms   latency {res['latency_ms']:9.3f} ms   equal {res['equal']}", flush=True)
        cross = max([r["n"] for r in rows if r["latency_ms"] < r["throughput_ms"]], default=0)
        report["signatures"][name] = {"rows": rows, "largest_n_where_the_lane_cooperative_kernel_is_faster": cross}
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            json.dump(report, f, indent=1)


if __name__ == "__main__":
    main()
