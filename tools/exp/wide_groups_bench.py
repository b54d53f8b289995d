"""wide_groups_bench.py -- ONE group of K pairs (an aggregated check): final_exp_native(multi_miller_loop_native(K pairs)) through bn254_multi_pairing_batch_dev,
spread over lanes (default) against the lane-per-group walk (bn254_set_wide_groups(0)).  Run on the GPU box."""
import sys, time
import torch
sys.path.insert(0, ".")
import __graft_entry__ as g
pk = g.build()
dev = torch.device("cuda:0"); st = torch.cuda.current_stream(dev)
for K in (8, 32, 64, 256, 1024, 4096, 65536, 131072, 1 << 20):
    g1 = torch.zeros(8 * K, dtype=torch.int64, device=dev); g2 = torch.zeros(16 * K, dtype=torch.int64, device=dev)
    pk.generate_pairs_dev(0xA66 + K, g1, g2, K, 0, st)
    out = torch.zeros(48, dtype=torch.int64, device=dev)
    def wall(reps):
        pk.multi_pairing_batch_dev(g1, g2, out, 1, K, True, 0, st); torch.cuda.synchronize()
        best = 1e9
        for _ in range(reps):
            t0 = time.perf_counter(); pk.multi_pairing_batch_dev(g1, g2, out, 1, K, True, 0, st); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        return best * 1e3
    a = wall(3); ref = out.clone()
    b = None
    if K <= 1024:
        pk.set_wide_groups(0)
        b = wall(1)
        pk.set_wide_groups(65536)
        assert torch.equal(ref, out)
    pk.last_status(0, st)
    print(f"one group of {K:8d} pairs: {a:9.3f} ms = {K / a / 1e3:7.3f} M pairs/s" + (f"   lane-per-group walk {b:10.1f} ms" if b else ""), flush=True)
