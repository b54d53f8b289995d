"""one pinned 2^19 host-pointer call (limb-major) for a rocprofv3 --kernel-trace --memory-copy-trace run: do the copies overlap the kernels?"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import __graft_entry__ as g
pk = g.build()
dev = torch.device("cuda:0"); st = torch.cuda.current_stream(dev)
n = 1 << 19
g1 = torch.zeros(8 * n, dtype=torch.int64, device=dev); g2 = torch.zeros(16 * n, dtype=torch.int64, device=dev)
pk.generate_pairs_dev(0xB2540001, g1, g2, n, 0, st); torch.cuda.synchronize()
h1 = g1.cpu().numpy().view(np.uint64).copy(); h2 = g2.cpu().numpy().view(np.uint64).copy()
mode = sys.argv[1] if len(sys.argv) > 1 else "pinned"
if mode == "pinned":
    p1, p2, po = pk.alloc_pinned(8 * n), pk.alloc_pinned(16 * n), pk.alloc_pinned(48 * n)
    p1[:], p2[:] = h1, h2
else:
    p1, p2, po = h1, h2, np.empty(48 * n, dtype=np.uint64)
for _ in range(3):
    t = time.perf_counter(); pk.pairing_batch(p1, p2, n, out=po); print(mode, (time.perf_counter() - t) * 1e3, "ms", flush=True)
