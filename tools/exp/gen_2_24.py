#!/usr/bin/env python3
"""gen_2_24.py -- what rank 0 of the 8-GPU run does first (bench.py, configs[4]): bn254_generate_pairs_dev on 2^24 pairs in one call (3.2 GB),
then a 2^21-lane pairing launch on a slice taken from the far end; generated points at the first / last / middle positions are checked against the
generator's stated scalars ([s]G1, [t]G2 by the big-int restatement) and the pairings against the oracle."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import helpers as H
pk = importlib.import_module("plonky2-bn254-pairing_amd")
dev = torch.device("cuda:0"); st = torch.cuda.current_stream(dev)
n = 1 << 24
g1 = torch.zeros(8 * n, dtype=torch.int64, device=dev); g2 = torch.zeros(16 * n, dtype=torch.int64, device=dev)
pk.generate_pairs_dev(0xB2540001, g1, g2, n, 0, st); pk.last_status(0, st)
pos = [0, 1, n // 2 + 3, n - 2, n - 1]
idx = torch.as_tensor(pos, device=dev)
a = pk.layout.to_aos(g1.view(8, n)[:, idx].cpu().numpy().view(np.uint64).reshape(-1).copy(), 8)
b = pk.layout.to_aos(g2.view(16, n)[:, idx].cpu().numpy().view(np.uint64).reshape(-1).copy(), 16)
R = H.R
for j, i in enumerate(pos):
    s, t = pk.generator_scalars(0xB2540001, i)
    P = R.g1_mul(R.G1_GEN, s); Q = R.g2_mul(R.G2_GEN, t)
    assert np.array_equal(a[8 * j: 8 * j + 8], H.g1_aos([P])), i
    assert np.array_equal(b[16 * j: 16 * j + 16], H.g2_aos([Q])), i
m = 1 << 21
lo = n - m
s1 = g1.view(8, n)[:, lo:].contiguous().view(-1); s2 = g2.view(16, n)[:, lo:].contiguous().view(-1)
out = torch.zeros(48 * m, dtype=torch.int64, device=dev)
pk.pairing_batch_dev(s1, s2, out, m, 0, st); pk.last_status(0, st)
p2 = [0, m // 2, m - 1]
i2 = torch.as_tensor(p2, device=dev)
got = pk.layout.to_aos(out.view(48, m)[:, i2].cpu().numpy().view(np.uint64).reshape(-1).copy(), 48)
a2 = pk.layout.to_aos(s1.view(8, m)[:, i2].cpu().numpy().view(np.uint64).reshape(-1).copy(), 8)
b2 = pk.layout.to_aos(s2.view(16, m)[:, i2].cpu().numpy().view(np.uint64).reshape(-1).copy(), 16)
assert np.array_equal(got, H.oracle_pairing(a2, b2, len(p2), threads=3))
print("ok: 2^24 pairs generated in one call (points at", pos, "equal [s]G1, [t]G2), pairings of the last 2^21-lane slice equal the oracle; peak device memory", torch.cuda.max_memory_allocated(dev) >> 20, "MiB")
