"""fixed_g2_bench.py -- the Groth16 shape with a fixed verifying key: 2^18 groups of 1 + 3 pairs whose last three G2 points are the same for every group
(bn254_pairing_fixed_g2_batch_dev) against four free pairs per group (bn254_multi_pairing_batch_dev, k_mpairing).  Run on the GPU box."""
import sys, time
import torch
sys.path.insert(0, ".")
import __graft_entry__ as g
pk = g.build()
dev = torch.device("cuda:0"); st = torch.cuda.current_stream(dev)
for log2, kf in ((18, 3), (18, 2), (18, 1), (16, 3)):
    n, k = 1 << log2, 1 + kf
    g1 = torch.zeros(8 * n * k, dtype=torch.int64, device=dev); g2all = torch.zeros(16 * n * k, dtype=torch.int64, device=dev)
    pk.generate_pairs_dev(0xB2540001, g1, g2all, n * k, 0, st)
    g2fix = g2all.view(16, n * k)[:, 1:1 + kf].contiguous().view(-1)
    g2var = g2all.view(16, n, k)[:, :, 0].contiguous().view(-1)
    exp = g2all.view(16, n, k).clone()
    for j in range(kf):
        exp[:, :, 1 + j] = g2fix.view(16, kf)[:, j:j + 1]
    exp = exp.contiguous().view(-1)
    table = torch.zeros(pk.g2_lines_bytes(kf) // 8, dtype=torch.int64, device=dev)
    pk.g2_lines_dev(g2fix, kf, table, 0, st); torch.cuda.synchronize()
    t0 = time.perf_counter(); pk.g2_lines_dev(g2fix, kf, table, 0, st); torch.cuda.synchronize(); t_tab = time.perf_counter() - t0      # (warm: the second call)
    a = torch.zeros(48 * n, dtype=torch.int64, device=dev); b = torch.zeros(48 * n, dtype=torch.int64, device=dev)
    def timed(fn, reps=3):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(reps): fn()
        e1.record(st); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    ms_f = timed(lambda: pk.pairing_fixed_g2_batch_dev(g1, g2var, table, kf, a, n, 0, st))
    ms_m = timed(lambda: pk.multi_pairing_batch_dev(g1, exp, b, n, k, True, 0, st))
    pk.last_status(0, st)
    print(f"2^{log2} groups of 1 + {kf} pairs: fixed-G2 kernel {ms_f:8.3f} ms = {n / ms_f / 1e3:6.3f} M groups/s   free pairs (k_mpairing, k = {k}) {ms_m:8.3f} ms = {n / ms_m / 1e3:6.3f} M groups/s"
          f"   ratio {ms_m / ms_f:5.3f}   same limbs {bool(torch.equal(a, b))}   table {t_tab * 1e3:.2f} ms", flush=True)

# groups WITHOUT a pair of their own: every G2 point is one of the table's (a KZG / PLONK opening check: e(P_1, [tau] G2) e(P_2, G2))
for log2, kf in ((18, 2), (18, 1), (16, 2)):
    n = 1 << log2
    g1 = torch.zeros(8 * n * kf, dtype=torch.int64, device=dev); g2all = torch.zeros(16 * n * kf, dtype=torch.int64, device=dev)
    pk.generate_pairs_dev(0xB2540077, g1, g2all, n * kf, 0, st)
    g2fix = g2all.view(16, n * kf)[:, :kf].contiguous().view(-1)
    exp = g2fix.view(16, 1, kf).expand(16, n, kf).contiguous().view(-1)
    table = torch.zeros(pk.g2_lines_bytes(kf) // 8, dtype=torch.int64, device=dev)
    pk.g2_lines_dev(g2fix, kf, table, 0, st)
    a = torch.zeros(48 * n, dtype=torch.int64, device=dev); b = torch.zeros(48 * n, dtype=torch.int64, device=dev)
    def timed(fn, reps=3):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(reps): fn()
        e1.record(st); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    ms_f = timed(lambda: pk.pairing_fixed_g2_batch_dev(g1, None, table, kf, a, n, 0, st))
    ms_m = timed(lambda: (pk.pairing_batch_dev(g1, exp, b, n, 0, st) if kf == 1 else pk.multi_pairing_batch_dev(g1, exp, b, n, kf, True, 0, st)))
    pk.last_status(0, st)
    print(f"2^{log2} groups of {kf} pairs, every G2 point fixed: fixed-G2 kernel {ms_f:8.3f} ms = {n / ms_f / 1e3:6.3f} M groups/s   free pairs {ms_m:8.3f} ms = {n / ms_m / 1e3:6.3f} M groups/s"
          f"   ratio {ms_m / ms_f:5.3f}   same limbs {bool(torch.equal(a, b))}", flush=True)
