#!/bin/bash
# PCIe-inclusive rate of the host-pointer entry point (pageable numpy buffers): DESIGN.md section 5.
cd "$GRAFT_REPO_ROOT"
timeout 600 python - <<'PY'
import time, numpy as np, torch, importlib, sys
sys.path.insert(0, ".")
import __graft_entry__ as g
pk = g.build()
for lg in (16, 18, 20):
    n = 1 << lg
    dev = torch.device("cuda:0")
    g1 = torch.zeros(8 * n, dtype=torch.int64, device=dev); g2 = torch.zeros(16 * n, dtype=torch.int64, device=dev)
    pk.generate_pairs_dev(0xB2540001, g1, g2, n, 0, torch.cuda.current_stream(dev)); torch.cuda.synchronize()
    h1 = g1.cpu().numpy().view(np.uint64).copy(); h2 = g2.cpu().numpy().view(np.uint64).copy()
    pk.pairing_batch(h1, h2, n)           # warm-up (scratch allocation)
    t = time.perf_counter(); out = pk.pairing_batch(h1, h2, n); dt = time.perf_counter() - t
    t = time.perf_counter(); out2 = pk.pairing_sharded(h1, h2, n, 1); dt2 = time.perf_counter() - t
    assert np.array_equal(out, out2)
    e1, e2 = pk.layout.to_aos(h1, 8), pk.layout.to_aos(h2, 16)
    pk.pairing_batch_elems(e1, e2, n)
    t = time.perf_counter(); out3 = pk.pairing_batch_elems(e1, e2, n, out_order=pk.FQ12_ARK); dt3 = time.perf_counter() - t
    t = time.perf_counter(); pk.layout.to_soa(e1, 8); pk.layout.to_soa(e2, 16); pk.layout.to_aos(out, 48); dt4 = time.perf_counter() - t
    print(f"n=2^{lg}: bn254_pairing_batch (host pointers) {n/dt/1e6:.2f} M pairings/s ({dt*1e3:.1f} ms); bn254_pairing_sharded(1 device) {n/dt2/1e6:.2f} M/s; "
          f"bn254_pairing_batch_elems (element-major in, ark Fq12 out) {n/dt3/1e6:.2f} M/s ({dt3*1e3:.1f} ms); the same three transpositions in numpy on the host: {dt4*1e3:.1f} ms")
PY
