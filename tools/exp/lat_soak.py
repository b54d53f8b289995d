#!/usr/bin/env python3
"""lat_soak.py [iterations] -- random batch sizes, functions and program families on the lane-cooperative kernels, every result compared
with the throughput kernel's on all lanes (torch.equal).  What the simulator cannot show -- a wait that is only almost always long
enough -- would show here.  Run on the GPU box."""
import importlib
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch
    pk = importlib.import_module("plonky2-bn254-pairing_amd")
    dev = torch.device("cuda:0")
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    rng = random.Random(0xB254)
    nmax = 6000
    g1 = torch.empty(8 * nmax * 4, dtype=torch.int64, device=dev)
    g2 = torch.empty(16 * nmax * 4, dtype=torch.int64, device=dev)
    fin = torch.empty(48 * nmax, dtype=torch.int64, device=dev)
    a = torch.empty(48 * nmax, dtype=torch.int64, device=dev)
    b = torch.empty(48 * nmax, dtype=torch.int64, device=dev)
    t0 = time.time()
    seen = {}
    for it in range(iters):
        n = rng.choice([1, 2, 3, 4, 5, 7, 16, 63, 64, 65, 255, 256, 257, 511, 512, 513, 1023, 1024, 1025, 2047, 2048, 2049, rng.randrange(1, nmax)])
        fn = rng.choice(["pairing", "miller", "fexp", "multi2", "multi3", "multi4", "mmiller2", "mmiller3", "mmiller4"])
        lanes = rng.choice([0, 0, 16, 32, 64])
        k = int(fn[-1]) if fn[-1].isdigit() else 1
        pk.generate_pairs_dev(rng.randrange(1 << 40), g1, g2, n * k)
        if fn == "fexp":
            pk.set_latency_threshold(0)
            pk.miller_loop_batch_dev(g1, g2, fin, n)

        def call(dst):
            if fn == "pairing":
                pk.pairing_batch_dev(g1, g2, dst, n)
            elif fn == "miller":
                pk.miller_loop_batch_dev(g1, g2, dst, n)
            elif fn == "fexp":
                pk.final_exp_batch_dev(fin, dst, n)
            else:
                pk.multi_pairing_batch_dev(g1, g2, dst, n, k, do_final_exp=fn.startswith("multi"))
        pk.set_latency_threshold(0)
        call(a)
        pk.set_latency_threshold(1 << 30)
        pk.set_latency_lanes(lanes)
        b[:48 * n].fill_(-1)
        call(b)
        pk.last_status()
        pk.set_latency_lanes(0)
        assert torch.equal(a[:48 * n], b[:48 * n]), (it, fn, n, lanes)
        seen[(fn, lanes)] = seen.get((fn, lanes), 0) + 1
        if it % 50 == 49:
            print(f"{it + 1} calls ok, {time.time() - t0:.0f} s", flush=True)
    print("soak ok:", iters, "calls,", len(seen), "(function, family) combinations")


if __name__ == "__main__":
    main()
