#!/usr/bin/env python3
"""route_soak.py [iterations] -- random shapes through the routes of round 6, every result compared on all lanes with the plain kernels':
  * groups of k pairs (k = 2 .. 300, 1 .. 3 000 groups): the default routing (lane-cooperative programs, the spread route, the k-pair kernel) against the k-pair
    kernel / the lane-per-group walk alone (bn254_set_wide_groups(0), lane-cooperative kernel off), Miller value and final value;
  * fixed-G2 groups (k_fixed = 1 .. 4, with / without a pair of their own, 1 .. 70 000 groups, limb-major / element-major, verdicts against a target): against the
    k-pair kernel on the expanded pairs.
Run on the GPU box."""
import importlib
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    pk = importlib.import_module("plonky2-bn254-pairing_amd")
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev)
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = random.Random(0xB2546)
    t0 = time.time()
    seen = {}
    idx = [pk.load_library().bn254_myfq12_to_ark_index(j) for j in range(12)]
    for it in range(iters):
        kind = rng.choice(["groups", "groups", "fixed", "fixed", "fixed"])
        if kind == "groups":
            k = rng.choice([2, 3, 4, 5, 6, 7, 8, 9, 12, 16, 31, 32, 63, 64, 65, 66, 96, 128, 130, 200, 257, 300])
            G = rng.choice([1, 1, 2, 3, 5, 17, 64, 257, rng.randrange(1, 3000)])
            if G * k > 200000:
                G = max(1, 200000 // k)
            fe = rng.random() < 0.6
            n = G * k
            g1 = torch.zeros(8 * n, dtype=torch.int64, device=dev); g2 = torch.zeros(16 * n, dtype=torch.int64, device=dev)
            pk.generate_pairs_dev(rng.randrange(1 << 40), g1, g2, n, 0, st)
            got = torch.full((48 * G + 8,), -7, dtype=torch.int64, device=dev)
            pk.multi_pairing_batch_dev(g1, g2, got, G, k, fe, 0, st)
            route = pk.last_kernel(0, st)
            pk.set_wide_groups(0); pk.set_stream_latency(0, -1, 0, st)
            try:
                ref = torch.zeros(48 * G, dtype=torch.int64, device=dev)
                pk.multi_pairing_batch_dev(g1, g2, ref, G, k, fe, 0, st)
            finally:
                pk.set_wide_groups(65536); pk.set_stream_latency(pk.LATENCY_INHERIT, -1, 0, st)
            pk.last_status(0, st)
            assert torch.equal(got[: 48 * G], ref) and bool((got[48 * G:] == -7).all()) and int(ref.abs().sum()) != 0, (it, kind, G, k, fe)
            key = ("groups", "k<=4" if k <= 4 else ("k<=64" if k <= 64 else "k>64"), "final" if fe else "miller")
        else:
            kf = rng.choice([1, 2, 3, 4]); own = rng.choice([0, 1, 1])
            n = rng.choice([1, 2, 5, 64, 300, 1025, 4096, 20000, rng.randrange(1, 70000)])
            k = kf + own
            elems = rng.random() < 0.5
            g1 = torch.zeros(8 * n * k, dtype=torch.int64, device=dev); g2v = torch.zeros(16 * n * k, dtype=torch.int64, device=dev)
            pk.generate_pairs_dev(rng.randrange(1 << 40), g1, g2v, n * k, 0, st)
            g2fix = torch.zeros(16 * kf, dtype=torch.int64, device=dev)
            pk.generate_pairs_dev(rng.randrange(1 << 40), torch.zeros(8 * kf, dtype=torch.int64, device=dev), g2fix, kf, 0, st)
            exp = g2v.view(16, n, k).clone()
            for j in range(kf):
                exp[:, :, own + j] = g2fix.view(16, kf)[:, j:j + 1]
            g2var = g2v.view(16, n, k)[:, :, 0].contiguous().view(-1) if own else None
            table = torch.zeros(pk.g2_lines_bytes(kf) // 8, dtype=torch.int64, device=dev)
            pk.g2_lines_dev(g2fix, kf, table, 0, st)
            pk.set_wide_groups(0); pk.set_stream_latency(0, -1, 0, st)
            try:
                ref = torch.zeros(48 * n, dtype=torch.int64, device=dev)
                if k == 1:
                    pk.pairing_batch_dev(g1, exp.contiguous().view(-1), ref, n, 0, st)
                else:
                    pk.multi_pairing_batch_dev(g1, exp.contiguous().view(-1), ref, n, k, True, 0, st)
            finally:
                pk.set_wide_groups(65536); pk.set_stream_latency(pk.LATENCY_INHERIT, -1, 0, st)
            got = torch.full((48 * n + 8,), -7, dtype=torch.int64, device=dev)
            if elems:
                e1 = torch.empty_like(g1); pk.soa_to_elems_dev(g1, e1, 8, n * k, 0, 0, st)
                e2 = None
                if own:
                    e2 = torch.empty_like(g2var); pk.soa_to_elems_dev(g2var, e2, 16, n, 0, 0, st)
                pk.pairing_fixed_g2_batch_elems_dev(e1, e2, table, kf, got, n, pk.FQ12_ARK, 0, st)
                want = ref.view(48, n).t().contiguous().view(n, 12, 4)[:, idx, :].contiguous().view(-1)
            else:
                pk.pairing_fixed_g2_batch_dev(g1, g2var, table, kf, got, n, 0, st)
                want = ref
            route = pk.last_kernel(0, st)
            v = torch.full((n,), 9, dtype=torch.uint8, device=dev)
            pos = rng.randrange(n)
            target = ref.view(48, n)[:, pos].contiguous().cpu().numpy().view(np.uint64)
            pk.pairing_fixed_g2_check_target_batch_dev(g1, g2var, table, kf, target, v, n, 0, st)
            pk.last_status(0, st)
            assert torch.equal(got[: 48 * n], want) and bool((got[48 * n:] == -7).all()) and int(ref.abs().sum()) != 0, (it, kind, n, kf, own, elems)
            assert int(v[pos]) == 1 and int(v.sum()) == 1, (it, "verdict", n, kf, own)
            key = ("fixed", f"kf={kf}", "own" if own else "no own pair", "elems" if elems else "planes")
        seen[key + (route,)] = seen.get(key + (route,), 0) + 1
        if (it + 1) % 25 == 0:
            print(f"{it + 1} calls ok, {time.time() - t0:.0f} s", flush=True)
    print(f"{iters} random calls, every lane equal to the plain kernels' (route = bn254_last_kernel: 1 throughput kernel, 16 / 32 / 64 lane-cooperative):")
    for kk in sorted(seen):
        print("  ", kk, seen[kk])


if __name__ == "__main__":
    main()
