import time, numpy as np, torch, sys
sys.path.insert(0, ".")
import __graft_entry__ as g
pk = g.build()
dev = torch.device("cuda:0")
for n in ((1 << 17) + 1000, 1 << 18, 3 << 17, 1 << 19):
    g1 = torch.zeros(8 * n, dtype=torch.int64, device=dev); g2 = torch.zeros(16 * n, dtype=torch.int64, device=dev)
    pk.generate_pairs_dev(0xB2540001, g1, g2, n, 0, torch.cuda.current_stream(dev)); torch.cuda.synchronize()
    h1 = g1.cpu().numpy().view(np.uint64).copy(); h2 = g2.cpu().numpy().view(np.uint64).copy()
    e1, e2 = pk.layout.to_aos(h1, 8), pk.layout.to_aos(h2, 16)
    ts, te = [], []
    for r in range(4):
        t = time.perf_counter(); pk.pairing_batch(h1, h2, n); ts.append((time.perf_counter() - t) * 1e3)
        t = time.perf_counter(); pk.pairing_batch_elems(e1, e2, n, out_order=1); te.append((time.perf_counter() - t) * 1e3)
    print(n, "soa", ["%.1f" % x for x in ts], "elems", ["%.1f" % x for x in te], flush=True)
