#!/usr/bin/env python3
"""coresidency.py -- can anything run on the chip while k_pairing does?  (run on the GPU box: python tools/coresidency.py)

k_pairing is persistent: 256 workgroups (one per CU), each holding all 512 registers of every SIMD and 144 KiB of the CU's
LDS for the whole launch.  The multi-rank flow (plonky2-bn254-pairing_amd/sharded.py) has to know whether a transfer can overlap it:

  A  a kernel (64 MB device-to-device elementwise copy) enqueued on a second stream 20 ms after k_pairing started:
     when does it start / finish, and does k_pairing get slower?
  B  the same bytes as copy-engine work: device -> pinned host (hipMemcpyAsync D2H) on the second stream.
  C  device -> device hipMemcpyAsync on the second stream (whatever engine the runtime picks).
  D  a kernel that is resident FIRST (a one-workgroup spin of ~30 ms, standing in for a posted RCCL receive that waits for
     its peer): how much later does k_pairing finish?

All times from HIP events (ms), relative to the event recorded in front of k_pairing."""
import ctypes
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import __graft_entry__
    pkg = __graft_entry__.build()
    hip = ctypes.CDLL("libamdhip64.so")
    dev = torch.device("cuda:0")
    n = 1 << int(os.environ.get("CORES_LOG2", "20"))
    sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    g1 = torch.zeros(8 * n, dtype=torch.int64, device=dev)
    g2 = torch.zeros(16 * n, dtype=torch.int64, device=dev)
    out = torch.zeros(48 * n, dtype=torch.int64, device=dev)
    pkg.generate_pairs_dev(0xB2540001, g1, g2, n, 0, sa)
    pkg.pairing_batch_dev(g1, g2, out, n, 0, sa)
    pkg.last_status(0, sa)
    src = torch.ones(8 << 20, dtype=torch.int64, device=dev)          # 64 MB
    dst = torch.zeros_like(src)
    pinned = torch.zeros(8 << 20, dtype=torch.int64, pin_memory=True)
    with torch.cuda.stream(sb):
        dst.copy_(src)
        pinned.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    ev = lambda: torch.cuda.Event(enable_timing=True)

    def pairing_alone():
        a, b = ev(), ev()
        a.record(sa)
        pkg.pairing_batch_dev(g1, g2, out, n, 0, sa)
        b.record(sa)
        torch.cuda.synchronize()
        return a.elapsed_time(b)

    base = sorted(pairing_alone() for _ in range(3))
    print(f"k_pairing alone, 2^{n.bit_length() - 1} pairings: {base[0]:.2f} / {base[1]:.2f} / {base[2]:.2f} ms")

    def timed_alone(fn):
        a, b = ev(), ev()
        a.record(sb)
        with torch.cuda.stream(sb):
            fn()
        b.record(sb)
        torch.cuda.synchronize()
        return a.elapsed_time(b)

    def under(fn, label):
        alone = timed_alone(fn)
        a0, a1, b0, b1 = ev(), ev(), ev(), ev()
        a0.record(sa)
        pkg.pairing_batch_dev(g1, g2, out, n, 0, sa)
        a1.record(sa)
        time.sleep(0.020)
        b0.record(sb)
        with torch.cuda.stream(sb):
            fn()
        b1.record(sb)
        torch.cuda.synchronize()
        print(f"{label}\n    alone {alone:.3f} ms; enqueued at +{a0.elapsed_time(b0):.1f} ms, finished at +{a0.elapsed_time(b1):.1f} ms; "
              f"k_pairing {a0.elapsed_time(a1):.2f} ms (alone {base[1]:.2f})")

    under(lambda: dst.copy_(src), "A  elementwise copy KERNEL, 64 MB device -> device, second stream")
    under(lambda: pinned.copy_(src, non_blocking=True), "B  hipMemcpyAsync device -> pinned host, 64 MB, second stream")

    def d2d():
        rc = hip.hipMemcpyAsync(ctypes.c_void_p(dst.data_ptr()), ctypes.c_void_p(src.data_ptr()), ctypes.c_size_t(src.numel() * 8), 3,
                                ctypes.c_void_p(sb.cuda_stream))
        assert rc == 0
    under(d2d, "C  hipMemcpyAsync device -> device, 64 MB, second stream")

    # D: something resident first
    cyc = 1_000_000
    a, b = ev(), ev()
    a.record(sb)
    with torch.cuda.stream(sb):
        torch.cuda._sleep(cyc)
    b.record(sb)
    torch.cuda.synchronize()
    per = a.elapsed_time(b) / cyc
    spin = int(30.0 / per)
    a0, a1, b0, b1 = ev(), ev(), ev(), ev()
    b0.record(sb)
    with torch.cuda.stream(sb):
        torch.cuda._sleep(spin)
    b1.record(sb)
    time.sleep(0.002)
    a0.record(sa)
    pkg.pairing_batch_dev(g1, g2, out, n, 0, sa)
    a1.record(sa)
    torch.cuda.synchronize()
    print(f"D  one-workgroup spin kernel resident first ({b0.elapsed_time(b1):.1f} ms), k_pairing enqueued +{b0.elapsed_time(a0):.1f} ms after it\n"
          f"    k_pairing {a0.elapsed_time(a1):.2f} ms (alone {base[1]:.2f}): +{a0.elapsed_time(a1) - base[1]:.1f} ms")
    pkg.last_status(0, sa)


if __name__ == "__main__":
    main()
