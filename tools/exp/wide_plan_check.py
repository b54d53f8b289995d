"""wide_plan_check.py -- batches of G groups of k pairs (5 <= k <= 64): the spread route against one launch of the k-pair kernel (bn254_set_wide_groups(0))."""
import sys, time
import torch
sys.path.insert(0, ".")
import __graft_entry__ as g
pk = g.build()
dev = torch.device("cuda:0"); st = torch.cuda.current_stream(dev)
for G, k in ((1, 8), (4096, 8), (16384, 8), (32768, 8), (60000, 8), (1000, 64), (10000, 64), (40000, 64), (30000, 5), (65535, 5)):
    n = G * k
    g1 = torch.zeros(8 * n, dtype=torch.int64, device=dev); g2 = torch.zeros(16 * n, dtype=torch.int64, device=dev)
    pk.generate_pairs_dev(0xA66 + n, g1, g2, n, 0, st)
    out = torch.zeros(48 * G, dtype=torch.int64, device=dev)
    def wall(reps=2):
        pk.multi_pairing_batch_dev(g1, g2, out, G, k, True, 0, st); torch.cuda.synchronize()
        best = 1e9
        for _ in range(reps):
            t0 = time.perf_counter(); pk.multi_pairing_batch_dev(g1, g2, out, G, k, True, 0, st); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        return best * 1e3
    pk.set_wide_groups(1 << 30)          # (always spread, whatever the estimate says: the rule itself is bypassed only through the 65 536-group bound)
    a = wall(); ra = out.clone()
    pk.set_wide_groups(0)
    b = wall(1); same = torch.equal(ra, out)
    pk.set_wide_groups(65536)
    c = wall()
    pk.last_status(0, st)
    print(f"{G:6d} groups of {k:3d} pairs: default route {c:8.2f} ms   k-pair kernel {b:8.2f} ms   same limbs {same}", flush=True)
