"""Multi-GPU sharding of independent pairings: one process per GPU, contiguous slices of the
SoA batch per rank, NO data-path collective -- the units (`pairing(p, q)`, /root/reference/src/pairing.rs:20-22)
are independent.  The only exchanges are the north star's scatter of the G1/G2 inputs from rank 0 and the gather
of the Fq12 outputs to rank 0, as grouped point-to-point operations (`torch.distributed.batch_isend_irecv`:
one grouped RCCL launch per peer and direction under backend "nccl" -- or one per direction, or none: `p2p_group_mode` --, the same
code under "gloo").  A slice of a limb-major batch is
one contiguous run per limb plane, so the transfers go plane by plane straight out of / into the whole-batch tensors on
rank 0: no temporaries and no second copy on either side (round 4).

All buffers are torch tensors of int64 words (the u64 Montgomery limbs, SoA limb-major).  Under nccl they are device
tensors and travel over xGMI with no host bounce.  gloo moves host memory only: device tensors then go through a pinned
host buffer on either side (`_P2P`) -- that is how the configs[4] flow runs with two ranks on ONE GPU (RCCL refuses two
ranks on a device), HIP engine and device-resident shards included.

Transfers never run under the pairing kernel: it is persistent and holds every register of every SIMD and 144 KiB of each
CU's LDS for the whole launch, so a communication kernel cannot co-reside (measured: profiles/r03_coresidency.txt -- a
kernel enqueued behind it starts when the launch ends, one that is resident first delays a whole workgroup's chain).  The
exchange steps therefore sit between launches: scatter, compute, gather.  (Copy-engine transfers do overlap; the
single-process `bn254_pairing_sharded_dev` uses them.)

`compute(g1, g2, m) -> out` is the per-rank hot path on one contiguous chunk of m pairings: the HIP engine on the
rank's own device and current stream by default (`hip_compute`), anything with the same signature in the CPU tests.
"""
import importlib
import os


def shard_bounds(n, world, rank):
    """Contiguous slice [lo, hi) of rank `rank`; sizes differ by at most one."""
    lo = (n * rank) // world
    hi = (n * (rank + 1)) // world
    return lo, hi


def _cols(t, words, n, lo, hi):
    """Columns [lo, hi) of every limb plane of an SoA batch, as a contiguous flat tensor (a device-side copy)."""
    return t.view(words, n)[:, lo:hi].contiguous().view(-1)


def _moves_device_memory(dist):
    """True when the process group carries device tensors itself (RCCL).  `dist.get_backend()` of a default-initialised
    group may read "undefined" or "cpu:gloo,cuda:nccl": any backend string that names nccl qualifies."""
    try:
        name = str(dist.get_backend())
    except Exception:      # noqa: BLE001
        return False
    return "nccl" in name.lower()


def _planes(t, words, n, lo, hi):
    """Plane w of columns [lo, hi) of an SoA batch of n elements: `words` CONTIGUOUS 1-D views (no copy) -- what a
    point-to-point transfer can send from / receive into directly."""
    v = t.view(words, n)
    return [v[w, lo:hi] for w in range(words)]


class _P2P:
    """One point-to-point transfer of a contiguous int64 tensor (or view).  Device tensors under a backend that moves host
    memory only (gloo) are bounced through a pinned host buffer: filled before the send, copied to the device after the
    receive."""

    def __init__(self, dist, kind, tensor, peer):
        self.kind, self.tensor, self.bounce, self.peer = kind, tensor, None, peer
        wire = tensor
        if tensor.is_cuda and not _moves_device_memory(dist):
            import torch
            self.bounce = torch.empty(tensor.shape, dtype=tensor.dtype, device="cpu", pin_memory=True)
            if kind == "send":
                self.bounce.copy_(tensor)                # synchronous on the current stream: the data is in host memory now
            wire = self.bounce
        self.op = dist.P2POp(dist.isend if kind == "send" else dist.irecv, wire, peer)

    def finish(self):
        if self.bounce is not None and self.kind == "recv":
            self.tensor.copy_(self.bounce)


P2P_GROUP_MODES = ("peer", "all", "op")


def p2p_group_mode(mode=None):
    """How the transfers of one exchange step are grouped: "peer" (default) -- one `batch_isend_irecv` group per peer (24 plane runs in
    the scatter, 48 in the gather: one grouped RCCL launch per peer, every group posted before the first wait); "all" -- the whole step as
    ONE group (168 / 336 operations to seven peers on rank 0: fewest launches, most operations per RCCL group); "op" -- every plane run its
    own isend / irecv, posted in the same (peer, plane) order on both sides (the most conservative form).  The environment variable
    BENCH_P2P_GROUP (or BN254_P2P_GROUP) switches it without a code change; the bytes and where they land are the same in all three."""
    mode = mode or os.environ.get("BENCH_P2P_GROUP") or os.environ.get("BN254_P2P_GROUP") or "peer"
    if mode not in P2P_GROUP_MODES:
        raise ValueError(f"p2p group mode {mode!r}: expected one of {P2P_GROUP_MODES}")
    return mode


def _run(dist, xfers, mode=None):
    """Posts the transfers of one exchange step (grouped as `p2p_group_mode` says), waits for all of them, finishes the bounces."""
    if not xfers:
        return
    mode = p2p_group_mode(mode)
    if mode == "all":
        groups = [xfers]
    elif mode == "peer":
        by_peer = {}
        for x in xfers:
            by_peer.setdefault(x.peer, []).append(x)
        groups = [by_peer[r] for r in sorted(by_peer)]
    else:
        groups = [[x] for x in xfers]
    works = []
    for g in groups:
        works += dist.batch_isend_irecv([x.op for x in g])
    for w in works:
        w.wait()
    for x in xfers:
        x.finish()


def hip_compute(device_index=None):
    """The product path: bn254_pairing_batch_dev on this rank's device (LOCAL_RANK) and torch's current stream."""
    import torch
    pkg = importlib.import_module("plonky2-bn254-pairing_amd")
    if device_index is None:
        device_index = int(os.environ.get("LOCAL_RANK", "0"))

    def compute(g1, g2, m):
        assert g1.is_cuda and g1.device.index == device_index, "inputs must live on this rank's device"
        out = torch.empty(48 * m, dtype=torch.int64, device=g1.device)
        pkg.pairing_batch_dev(g1, g2, out, m, device=device_index, stream=torch.cuda.current_stream(g1.device))
        return out

    return compute


def scatter_inputs(full_g1, full_g2, n, g1_local, g2_local, dist, p2p_group=None):
    """Rank 0 holds the whole SoA batch (8n / 16n words); every rank receives its slice into g1_local / g2_local
    (8 n_local / 16 n_local words).  Empty slices are skipped on both sides.  A slice of an SoA batch is one contiguous run
    per limb plane: rank 0 sends the 8 + 16 plane runs of a peer's slice straight out of the whole-batch tensors (no
    temporaries), the peer receives them into the planes of its local tensors; the transfers are grouped per peer (`p2p_group_mode`)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    lo, hi = shard_bounds(n, world, rank)
    ops = []
    if rank == 0:
        for r in range(1, world):
            rlo, rhi = shard_bounds(n, world, r)
            if rhi == rlo:
                continue
            for full, words in ((full_g1, 8), (full_g2, 16)):
                ops += [_P2P(dist, "send", v, r) for v in _planes(full, words, n, rlo, rhi)]
        if hi > lo:
            g1_local.view(8, hi - lo).copy_(full_g1.view(8, n)[:, lo:hi])
            g2_local.view(16, hi - lo).copy_(full_g2.view(16, n)[:, lo:hi])
    elif hi > lo:
        m = hi - lo
        ops = [_P2P(dist, "recv", v, 0) for v in _planes(g1_local, 8, m, 0, m) + _planes(g2_local, 16, m, 0, m)]
    _run(dist, ops, p2p_group)


def gather_outputs(out_local, n, dist, device=None, p2p_group=None):
    """Every rank's 48 n_local output words travel to rank 0, which returns the whole SoA batch (48 n words); other
    ranks return None.  Rank 0 receives each peer's 48 plane runs STRAIGHT into the whole-batch tensor (contiguous views:
    no temporaries, no second copy); one group of 48 transfers per peer by default (`p2p_group_mode`)."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    lo, hi = shard_bounds(n, world, rank)
    if rank != 0:
        if hi > lo:
            m = hi - lo
            _run(dist, [_P2P(dist, "send", v, 0) for v in _planes(out_local, 48, m, 0, m)], p2p_group)
        return None
    full = torch.empty(48 * n, dtype=torch.int64, device=device if device is not None else out_local.device)
    ops = []
    for r in range(1, world):
        rlo, rhi = shard_bounds(n, world, r)
        if rhi == rlo:
            continue
        ops += [_P2P(dist, "recv", v, r) for v in _planes(full, 48, n, rlo, rhi)]
    if hi > lo:
        full.view(48, n)[:, lo:hi].copy_(out_local.view(48, hi - lo))
    _run(dist, ops, p2p_group)
    return full


def _default_device(dist, like, compute_is_hip=False):
    """Where a rank's shard lives when the caller did not say: next to the inputs it was handed; else -- a non-root rank
    of a scattered run has none -- this rank's GPU whenever the data must be there (RCCL moves device memory; the HIP
    engine computes on device memory whatever carries the slices), else the host."""
    import torch
    if like is not None:
        return like.device
    if _moves_device_memory(dist) or compute_is_hip:
        return torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1))
    return torch.device("cpu")


def pairing_sharded(g1, g2, n, dist=None, compute=None, scatter_from_root=False, gather_to_root=True, chunk=1 << 19, device=None):
    """n independent pairings over the ranks of `dist`.  Returns (local SoA result, gathered SoA result on rank 0 or None).

    g1 / g2: SoA tensors of the WHOLE batch -- valid on every rank, or on rank 0 only with scatter_from_root (other ranks
    may pass None).  Each rank walks its slice in launches of `chunk` lanes; the exchange steps sit between launches
    (scatter, then all of the rank's compute, then the gather): nothing is posted while a pairing kernel runs."""
    import torch
    compute_is_hip = compute is None
    compute = compute or hip_compute()
    if dist is None or not dist.is_initialized():
        out = compute(g1, g2, n) if n else torch.empty(0, dtype=torch.int64, device=g1.device)
        return out, out
    world, rank = dist.get_world_size(), dist.get_rank()
    lo, hi = shard_bounds(n, world, rank)
    n_local = hi - lo
    if device is None:
        device = _default_device(dist, g1, compute_is_hip)
    if scatter_from_root:
        l1 = torch.empty(8 * n_local, dtype=torch.int64, device=device)
        l2 = torch.empty(16 * n_local, dtype=torch.int64, device=device)
        scatter_inputs(g1, g2, n, l1, l2, dist)
    else:
        l1, l2 = _cols(g1, 8, n, lo, hi), _cols(g2, 16, n, lo, hi)
    local = torch.empty(48 * n_local, dtype=torch.int64, device=device)
    for a in range(0, n_local, chunk):
        b = min(a + chunk, n_local)
        o = compute(_cols(l1, 8, n_local, a, b), _cols(l2, 16, n_local, a, b), b - a)
        local.view(48, n_local)[:, a:b].copy_(o.view(48, b - a))
    gathered = gather_outputs(local, n, dist, device=device) if gather_to_root else None
    return local, gathered
