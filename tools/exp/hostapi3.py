"""hostapi3.py [out.json] -- PCIe-inclusive rates of the host-pointer entry points (the path a binding of src/pairing.rs:20-22 takes: the
caller's values live in host memory), pageable against page-locked buffers (bn254_alloc_pinned; bn254_host_register on a numpy array),
limb-major and element-major, at 2^16 / 2^20 pairings and on the Groth16 shape (2^18 groups x 4 pairs).  Run on the GPU box."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import __graft_entry__ as g  # noqa: E402

pk = g.build()
dev = torch.device("cuda:0")
res = {"what": "wall time of one host-pointer call (second of two calls: buffers of the library warm), pairings/s incl. the copies both ways", "rows": []}


def pinned_copy(a):
    p = pk.alloc_pinned(a.size)
    p[:] = a
    return p


def timed(fn, reps=3):
    fn()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    return min(ts), sorted(ts)[len(ts) // 2]


for lg, k in ((16, 1), (20, 1), (18, 4)):
    units = 1 << lg
    n = units * k
    g1 = torch.zeros(8 * n, dtype=torch.int64, device=dev); g2 = torch.zeros(16 * n, dtype=torch.int64, device=dev)
    o = torch.zeros(48 * units, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream(dev)
    pk.generate_pairs_dev(0xB2540001, g1, g2, n, 0, st); torch.cuda.synchronize()
    run_dev = (lambda: (pk.pairing_batch_dev(g1, g2, o, units, 0, st), torch.cuda.synchronize())) if k == 1 else \
              (lambda: (pk.multi_pairing_batch_dev(g1, g2, o, units, k, True, 0, st), torch.cuda.synchronize()))
    t_dev, _ = timed(run_dev)
    h1 = g1.cpu().numpy().view(np.uint64).copy(); h2 = g2.cpu().numpy().view(np.uint64).copy()
    e1, e2 = pk.layout.to_aos(h1, 8), pk.layout.to_aos(h2, 16)
    ref = None
    for fmt, a1, a2 in (("limb-major", h1, h2), ("element-major in, ark Fq12 out", e1, e2)):
        elems = fmt != "limb-major"
        for mem in ("pageable", "alloc_pinned", "host_register"):
            if mem == "pageable":
                b1, b2, bo = a1, a2, np.empty(48 * units, dtype=np.uint64)
            elif mem == "alloc_pinned":
                b1, b2, bo = pinned_copy(a1), pinned_copy(a2), pk.alloc_pinned(48 * units)
            else:
                b1, b2, bo = a1.copy(), a2.copy(), np.empty(48 * units, dtype=np.uint64)
                t = time.perf_counter()
                for b in (b1, b2, bo):
                    pk.host_register(b)
                t_reg = time.perf_counter() - t
            assert pk.host_is_pinned(bo) == (mem != "pageable")
            if elems:
                call = (lambda: pk.pairing_batch_elems(b1, b2, units, out_order=pk.FQ12_ARK, out=bo)) if k == 1 else \
                       (lambda: pk.multi_pairing_batch_elems(b1, b2, units, k, True, out_order=pk.FQ12_ARK, out=bo))
            else:
                call = (lambda: pk.pairing_batch(b1, b2, units, out=bo)) if k == 1 else (lambda: pk.multi_pairing_batch(b1, b2, units, k, True, out=bo))
            best, med = timed(call)
            got = bo.copy()
            if (fmt, "ref") not in res:
                res[(fmt, "ref")] = got
            same = bool(np.array_equal(got, res[(fmt, "ref")]))
            row = {"shape": f"2^{lg} x {k}", "format": fmt, "host_memory": mem, "ms_best": best * 1e3, "ms_median": med * 1e3,
                   "pairings_per_s": n / best, "units_per_s": units / best, "resident_ms": t_dev * 1e3, "share_of_resident_rate": t_dev / best,
                   "same_limbs_as_pageable": same}
            if mem == "host_register":
                row["register_ms"] = t_reg * 1e3
                for b in (b1, b2, bo):
                    pk.host_unregister(b)
            if mem == "alloc_pinned":
                for b in (b1, b2, bo):
                    pk.free_pinned(b)
            res["rows"].append(row)
            print(json.dumps(row), flush=True)
    for key in [k_ for k_ in res if isinstance(k_, tuple)]:
        del res[key]
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
