"""fixed_small_lat.py -- a SMALL batch of Groth16-shaped checks through the fixed-G2 entry points: wall time of one call (launch to completion, inputs resident).
Below the latency threshold the call expands the pairs and runs the lane-cooperative k-pair program; with the threshold at 0 it is one throughput launch."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
import __graft_entry__ as g
pk = g.build()
dev = torch.device("cuda:0"); st = torch.cuda.current_stream(dev)
for kf in (2, 3):
    for n in (1, 16, 256, 1024, 4096):
        k = 1 + kf
        g1 = torch.zeros(8 * n * k, dtype=torch.int64, device=dev); g2all = torch.zeros(16 * n * k, dtype=torch.int64, device=dev)
        pk.generate_pairs_dev(0xC0DE + n, g1, g2all, n * k, 0, st)
        g2fix = g2all.view(16, n * k)[:, 1:1 + kf].contiguous().view(-1) if n * k > kf else None
        if g2fix is None or g2fix.numel() != 16 * kf:
            f1 = torch.zeros(8 * kf, dtype=torch.int64, device=dev); g2fix = torch.zeros(16 * kf, dtype=torch.int64, device=dev)
            pk.generate_pairs_dev(7, f1, g2fix, kf, 0, st)
        g2var = g2all.view(16, n, k)[:, :, 0].contiguous().view(-1)
        table = torch.zeros(pk.g2_lines_bytes(kf) // 8, dtype=torch.int64, device=dev)
        pk.g2_lines_dev(g2fix, kf, table, 0, st)
        v = torch.zeros(n, dtype=torch.uint8, device=dev)
        target = np.zeros(48, dtype=np.uint64)
        def wall(reps=20):
            for _ in range(3):
                pk.pairing_fixed_g2_check_target_batch_dev(g1, g2var, table, kf, target, v, n, 0, st)
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(reps):
                t0 = time.perf_counter()
                pk.pairing_fixed_g2_check_target_batch_dev(g1, g2var, table, kf, target, v, n, 0, st)
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            return best * 1e3
        a = wall(); ka = pk.last_kernel(0, st)
        pk.set_stream_latency(0, -1, 0, st)
        b = wall(5); kb = pk.last_kernel(0, st)
        pk.set_stream_latency(pk.LATENCY_INHERIT, -1, 0, st)
        print(f"1 + {kf} pairs, {n:5d} proofs: default route {a:7.3f} ms (kernel {ka})   throughput kernel only {b:7.3f} ms (kernel {kb})", flush=True)

# the host-pointer form (what a binding calls): element-major structs in host memory, one verdict byte back; the stream keeps the table of the last call's fixed points
for kf in (2, 3):
    for n in (1, 256, 4096, 65536):
        k = 1 + kf
        rng = np.random.default_rng(5)
        g1 = torch.zeros(8 * n * k, dtype=torch.int64, device=dev); g2all = torch.zeros(16 * (n + kf), dtype=torch.int64, device=dev)
        f1 = torch.zeros(8 * (n + kf), dtype=torch.int64, device=dev)
        pk.generate_pairs_dev(0xC0DE + n, g1, torch.zeros(16 * n * k, dtype=torch.int64, device=dev), n * k, 0, st)
        pk.generate_pairs_dev(0xBEEF + n, f1, g2all, n + kf, 0, st)
        torch.cuda.synchronize()
        e1 = pk.layout.to_aos(g1.cpu().numpy().view(np.uint64), 8)
        e2all = pk.layout.to_aos(g2all.cpu().numpy().view(np.uint64), 16)
        e2, ef = np.ascontiguousarray(e2all[: 16 * n]), np.ascontiguousarray(e2all[16 * n:])
        target = np.zeros(48, dtype=np.uint64)
        for _ in range(3):
            pk.pairing_fixed_g2_check_batch_elems(e1, e2, ef, kf, n, target=target)
        best = 1e9
        for _ in range(10):
            t0 = time.perf_counter(); pk.pairing_fixed_g2_check_batch_elems(e1, e2, ef, kf, n, target=target); best = min(best, time.perf_counter() - t0)
        print(f"host structs, 1 + {kf} pairs, {n:6d} proofs: {best * 1e3:7.3f} ms per call (copies and verdict read-back included)", flush=True)
