"""wide_host_check.py -- ONE group of K pairs from HOST memory (bn254_multi_pairing_batch: stage, launch, copy back) against the device-resident call."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
import __graft_entry__ as g
pk = g.build()
dev = torch.device("cuda:0"); st = torch.cuda.current_stream(dev)
for K in (1024, 65536, 1 << 20):
    g1 = torch.zeros(8 * K, dtype=torch.int64, device=dev); g2 = torch.zeros(16 * K, dtype=torch.int64, device=dev)
    pk.generate_pairs_dev(0xA66 + K, g1, g2, K, 0, st)
    torch.cuda.synchronize()
    h1, h2 = g1.cpu().numpy().view(np.uint64).copy(), g2.cpu().numpy().view(np.uint64).copy()
    out = torch.zeros(48, dtype=torch.int64, device=dev)
    pk.multi_pairing_batch_dev(g1, g2, out, 1, K, True, 0, st); torch.cuda.synchronize()
    t0 = time.perf_counter(); pk.multi_pairing_batch_dev(g1, g2, out, 1, K, True, 0, st); torch.cuda.synchronize(); t_dev = time.perf_counter() - t0
    r = pk.multi_pairing_batch(h1, h2, 1, K)
    t0 = time.perf_counter(); r = pk.multi_pairing_batch(h1, h2, 1, K); t_host = time.perf_counter() - t0
    assert np.array_equal(r, out.cpu().numpy().view(np.uint64))
    print(f"one group of {K:8d} pairs: device-resident {t_dev * 1e3:8.2f} ms   from host memory {t_host * 1e3:8.2f} ms", flush=True)
