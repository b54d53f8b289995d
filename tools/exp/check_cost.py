#!/usr/bin/env python3
"""check_cost.py -- what bn254_check_points_ex costs: wall ms of one `_dev` call on 2^log2 generated pairs for each flag set."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
pk = importlib.import_module("plonky2-bn254-pairing_amd")
dev = torch.device("cuda:0"); st = torch.cuda.current_stream(dev)
for log2 in (16, 20):
    n = 1 << log2
    g1 = torch.zeros(8 * n, dtype=torch.int64, device=dev); g2 = torch.zeros(16 * n, dtype=torch.int64, device=dev)
    per = torch.zeros(n, dtype=torch.uint8, device=dev)
    pk.generate_pairs_dev(0xC0571, g1, g2, n, 0, st); pk.last_status(0, st)
    for name, flags in (("infinity", 1), ("infinity + on-curve", 3), ("infinity + on-curve + G2 subgroup", 7), ("... subgroup on the portable C++ kernel", 15)):
        ts = []
        for _ in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            pk.check_points_ex_dev(g1, g2, n, flags, per, 0, st); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        pk.last_status(0, st)
        print(f"2^{log2} pairs  {name:36s} {min(ts[1:]) * 1e3:9.3f} ms   {n / min(ts[1:]) / 1e6:8.2f} M pairs/s   flagged {int(per.sum())}", flush=True)
