"""GPU tests of the host side of the C ABI added in round 3:

  * groups of more than 64 pairs (`multi_miller_loop_native` takes any Vec, /root/reference/src/miller_loop_native.rs:324-326):
    sub-group composition against the oracle at k = 65, 130, 200, shared-f Miller value, final value and the `== 1` verdict;
  * `bn254_pairing_sharded_dev` / `bn254_multi_pairing_sharded_dev`: device-resident batch, one shard per entry of the device
    list -- a device named twice / three times runs the whole slice / copy / gather path on a one-GPU box;
  * host threads: two threads on the SAME (device, stream) through the host-pointer entry points (the scalar Rust / C++
    signatures all use device 0 and the NULL stream), and two threads on two streams through the `_dev` entry points
    (`pow_native`, `frobenius_map_native`, `pairing`), none of which waits for its stream."""
import threading

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


def _pairs(n, mul=5, seed=8):
    base_P, base_Q = H.subgroup_points(seed)
    P = [base_P[(i * mul + 1) % seed] for i in range(n)]
    Q = [base_Q[(i * 3 + i // seed) % seed] for i in range(n)]
    return H.g1_aos(P), H.g2_aos(Q)


@pytest.mark.parametrize("wide", [True, False])
@pytest.mark.parametrize("k,n_groups", [(65, 3), (130, 2), (200, 1)])
def test_groups_of_more_than_64_pairs(k, n_groups, wide):
    """wide: a batch of few groups spreads each group over several lanes (one Miller launch over all chunks, a multiplication tree per group);
    not wide: every group on its own lane, sub-group after sub-group (what a batch of 65 536 groups or more does).  bn254_set_wide_groups."""
    pk = H.pkg()
    old = pk.get_wide_groups()
    pk.set_wide_groups(old if wide else 0)
    try:
        _groups_of_more_than_64_pairs(pk, k, n_groups)
    finally:
        pk.set_wide_groups(old)


def test_one_group_of_very_many_pairs():
    """One aggregated check: ONE group of 4 096 / 131 072 pairs (a lane per group would walk it for minutes).  The value must be the product of the values
    of the same pairs taken as 64-pair groups (the k-pair kernel itself, no composition), multiplied here with MyFq12 Mul -- Miller value and final value."""
    import torch
    pk = H.pkg()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev)

    def product(vals, m):                       # [48][m] planes -> [48][1]: halves multiplied until one value is left (m a power of two)
        while m > 1:
            h = m // 2
            a, b = vals.view(48, m)[:, :h].contiguous().view(-1), vals.view(48, m)[:, h:].contiguous().view(-1)
            vals = torch.zeros(48 * h, dtype=torch.int64, device=dev)
            pk.fq12_mul_batch_dev(a, b, vals, h, 0, st)
            m = h
        return vals

    assert pk.get_wide_groups() == 65536
    for K in (4096, 131072):
        g1 = torch.zeros(8 * K, dtype=torch.int64, device=dev)
        g2 = torch.zeros(16 * K, dtype=torch.int64, device=dev)
        pk.generate_pairs_dev(0xA66 + K, g1, g2, K, 0, st)
        parts = torch.zeros(48 * (K // 64), dtype=torch.int64, device=dev)
        pk.multi_pairing_batch_dev(g1, g2, parts, K // 64, 64, False, 0, st)            # the shared-f Miller values of K / 64 groups of 64 pairs
        want_m = product(parts, K // 64)
        want = torch.zeros(48, dtype=torch.int64, device=dev)
        pk.final_exp_batch_dev(want_m, want, 1, 0, st)
        got_m = torch.full((48 + 8,), -7, dtype=torch.int64, device=dev)
        pk.multi_pairing_batch_dev(g1, g2, got_m, 1, K, False, 0, st)
        got = torch.full((48 + 8,), -7, dtype=torch.int64, device=dev)
        pk.multi_pairing_batch_dev(g1, g2, got, 1, K, True, 0, st)
        pk.last_status(0, st)
        assert torch.equal(got_m[:48], want_m) and torch.equal(got[:48], want) and bool((got[48:] == -7).all()) and bool((got_m[48:] == -7).all())
        assert int(want.abs().sum()) != 0
    # three groups of 1 001 pairs (7 x 11 x 13: chunks of 13 pairs would not fill a grid -> every pair its own lane; odd levels in the tree)
    G, K = 3, 1001
    g1 = torch.zeros(8 * G * K, dtype=torch.int64, device=dev)
    g2 = torch.zeros(16 * G * K, dtype=torch.int64, device=dev)
    pk.generate_pairs_dev(0xA77, g1, g2, G * K, 0, st)
    got = torch.zeros(48 * G, dtype=torch.int64, device=dev)
    pk.multi_pairing_batch_dev(g1, g2, got, G, K, True, 0, st)
    pk.set_wide_groups(0)
    try:
        ref = torch.zeros(48 * G, dtype=torch.int64, device=dev)
        pk.multi_pairing_batch_dev(g1, g2, ref, G, K, True, 0, st)                      # the lane-per-group walk: 16 sub-groups in sequence
    finally:
        pk.set_wide_groups(65536)
    pk.last_status(0, st)
    assert torch.equal(got, ref) and int(ref.abs().sum()) != 0


def test_one_group_of_many_pairs_is_capturable_after_reserve():
    """bn254_reserve(1, k) sizes the chunk values, the tree's operand buffers and the Miller launch's scratch: the aggregated check is then a sequence of
    plain launches (Miller over the chunks, a split + a multiplication per tree level, the final exponentiation), captured and replayed on new pairs."""
    import torch
    pk = H.pkg()
    dev = torch.device("cuda:0")
    side = torch.cuda.Stream(dev)
    K = 4096 + 64
    with torch.cuda.stream(side):
        g1 = torch.zeros(8 * K, dtype=torch.int64, device=dev)
        g2 = torch.zeros(16 * K, dtype=torch.int64, device=dev)
        out = torch.zeros(48, dtype=torch.int64, device=dev)
        pk.generate_pairs_dev(0xC4B, g1, g2, K, 0, side)
        pk.reserve(1, K, 0, side)
        pk.last_status(0, side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            pk.multi_pairing_batch_dev(g1, g2, out, 1, K, True, 0, side)
        seen = []
        for seed in (0xC4C, 0xC4D):
            pk.generate_pairs_dev(seed, g1, g2, K, 0, side)
            out.zero_()
            graph.replay()
            side.synchronize()
            got = out.clone()
            out.zero_()
            pk.multi_pairing_batch_dev(g1, g2, out, 1, K, True, 0, side)
            pk.last_status(0, side)
            assert torch.equal(got, out) and int(out.abs().sum()) != 0
            seen.append(got)
        assert not torch.equal(seen[0], seen[1])
    pk.release_stream(0, side)


def _groups_of_more_than_64_pairs(pk, k, n_groups):
    g1a, g2a = _pairs(n_groups * k)
    g1, g2 = H.to_soa(g1a, 8), H.to_soa(g2a, 16)
    want_m = H.oracle_multi_miller(g1a, g2a, n_groups, k)
    got_m = H.to_aos(pk.multi_pairing_batch(g1, g2, n_groups, k, do_final_exp=False), 48)
    assert np.array_equal(got_m, want_m), f"multi_miller_loop_native k={k}"
    want = H.oracle_multi_pairing(g1a, g2a, n_groups, k)
    got = H.to_aos(pk.multi_pairing_batch(g1, g2, n_groups, k, do_final_exp=True), 48)
    assert np.array_equal(got, want), f"multi pairing k={k}"
    # element-major entry point, ark order out
    got_e = pk.multi_pairing_batch_elems(g1a, g2a, n_groups, k, do_final_exp=True, out_order=pk.FQ12_MYFQ12)
    assert np.array_equal(got_e, want)


def test_product_check_on_a_long_group():
    """66 pairs whose product is one: 33 x [e(aP, bQ) e(abP, -Q)] (the T3 pattern, final_exp_native.rs:245-263), next to a
    group with one pair exchanged."""
    pk = H.pkg()
    t3 = H.load_golden("bn254_vectors.json")["t3"]
    P3 = [tuple(int(x, 16) for x in p) for p in t3["g1"]]
    Q3 = [((int(q[0], 16), int(q[1], 16)), (int(q[2], 16), int(q[3], 16))) for q in t3["g2"]]
    good_p, good_q = P3 * 33, Q3 * 33
    bad_p, bad_q = list(good_p), list(good_q)
    bad_p[65] = P3[0]                                         # the pair in the second sub-group (64 + 2)
    g1 = H.to_soa(H.g1_aos(good_p + bad_p + good_p), 8)
    g2 = H.to_soa(H.g2_aos(good_q + bad_q + good_q), 16)
    assert pk.multi_pairing_check_batch(g1, g2, 3, 66).tolist() == [1, 0, 1]


def test_sharded_dev_one_gpu_device_list():
    import torch
    pk = H.pkg()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev)
    n = 5 * 256 + 77
    g1 = torch.zeros(8 * n, dtype=torch.int64, device=dev)
    g2 = torch.zeros(16 * n, dtype=torch.int64, device=dev)
    pk.generate_pairs_dev(0xB2540009, g1, g2, n, 0, st)
    ref = torch.zeros(48 * n, dtype=torch.int64, device=dev)
    pk.pairing_batch_dev(g1, g2, ref, n, 0, st)
    pk.last_status(0, st)
    for devices in ([0], [0, 0], [0, 0, 0]):
        out = torch.zeros(48 * n, dtype=torch.int64, device=dev)
        pk.pairing_sharded_dev(g1, g2, out, n, devices, st)
        assert torch.equal(out, ref), f"devices={devices}"
    # k-pair groups stay whole; ragged split (7 groups over 3 shards), Miller value only
    k, groups = 4, 7
    ref = torch.zeros(48 * groups, dtype=torch.int64, device=dev)
    m1, m2 = g1.view(8, n)[:, : groups * k].contiguous().view(-1), g2.view(16, n)[:, : groups * k].contiguous().view(-1)
    pk.multi_pairing_batch_dev(m1, m2, ref, groups, k, False, 0, st)
    pk.last_status(0, st)
    out = torch.zeros(48 * groups, dtype=torch.int64, device=dev)
    pk.multi_pairing_sharded_dev(m1, m2, out, groups, k, [0, 0, 0], do_final_exp=False, stream=st)
    assert torch.equal(out, ref)
    # more shards than units: empty slices are skipped
    out = torch.zeros(48 * 2, dtype=torch.int64, device=dev)
    s1, s2 = g1.view(8, n)[:, : 2 * k].contiguous().view(-1), g2.view(16, n)[:, : 2 * k].contiguous().view(-1)
    pk.multi_pairing_sharded_dev(s1, s2, out, 2, k, [0, 0, 0], do_final_exp=False, stream=st)
    assert torch.equal(out.view(48, 2), ref.view(48, groups)[:, :2])
    with pytest.raises(pk.Bn254Error) as ei:
        pk.pairing_sharded_dev(g1, g2, out, n, [0, pk.device_count()], st)
    assert ei.value.status == pk.ERR_INVALID_ARG
    # spot check against the oracle through the sharded path
    pos = [0, n // 3, n - 1]
    out = torch.zeros(48 * n, dtype=torch.int64, device=dev)
    pk.pairing_sharded_dev(g1, g2, out, n, [0, 0], st)
    g1h = g1.view(8, n)[:, pos].cpu().numpy().view(np.uint64).reshape(-1).copy()
    g2h = g2.view(16, n)[:, pos].cpu().numpy().view(np.uint64).reshape(-1).copy()
    got = out.view(48, n)[:, pos].cpu().numpy().view(np.uint64).reshape(-1).copy()
    assert np.array_equal(pk.layout.to_aos(got, 48), H.oracle_pairing(pk.layout.to_aos(g1h, 8), pk.layout.to_aos(g2h, 16), len(pos), threads=3))


def test_two_host_threads_share_the_null_stream():
    """The reference's functions are pure and `Send + Sync`: two threads calling the host-pointer entry points
    on device 0 / the NULL stream with different batch sizes (the staging buffers are per stream and grow) get their own
    results."""
    pk = H.pkg()
    sizes = (300, 2100)
    inputs, want = [], []
    for n in sizes:
        g1a, g2a = _pairs(n, mul=3 + n % 4)
        g1, g2 = H.to_soa(g1a, 8), H.to_soa(g2a, 16)
        inputs.append((g1, g2, n))
        want.append(pk.pairing_batch(g1, g2, n))
    assert np.array_equal(H.to_aos(want[0], 48)[: 48 * 40], H.oracle_pairing(*_pairs(40, mul=3 + sizes[0] % 4), 40, threads=4))
    errors = []

    def work(t):
        try:
            g1, g2, n = inputs[t]
            for rep in range(6):
                got = pk.pairing_batch(g1, g2, n)
                if not np.array_equal(got, want[t]):
                    errors.append((t, rep))
                f = pk.final_exp_batch(pk.miller_loop_batch(g1[: 8 * n], g2[: 16 * n], n), n) if rep == 0 else None
                if f is not None and not np.array_equal(f, want[t]):
                    errors.append((t, "split"))
        except Exception as e:      # noqa: BLE001
            errors.append((t, repr(e)))

    th = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors


def test_two_host_threads_two_streams_dev_calls():
    """pow_native / frobenius_map_native / pairing through the `_dev` entry points from two host threads, each on its own
    stream; pow's digits travel through pinned slots, so neither thread waits for its stream inside the call (more calls are
    queued than there are slots)."""
    import torch
    pk = H.pkg()
    lib = pk.load_library()
    dev = torch.device("cuda:0")
    n = 1 << 12
    xs = H.rand_fq12(6, seed=5)
    a_host = H.to_soa(H.fq12_aos([xs[i % 6] for i in range(n)]), 48)
    a = torch.from_numpy(a_host.view(np.int64).copy()).to(dev)
    exps = [[pk.BN_X], [0xFFFFFFFFFFFFFFF1, 0x3], [0x8000000000000001]]
    want_pow = [H.oracle_pow_native(H.fq12_aos(xs), e, 6)[1] for e in exps]
    want_frob = H.oracle_frobenius(H.fq12_aos(xs), 5, 6)
    g1 = torch.zeros(8 * n, dtype=torch.int64, device=dev)
    g2 = torch.zeros(16 * n, dtype=torch.int64, device=dev)
    pk.generate_pairs_dev(0xB254000A, g1, g2, n, 0, torch.cuda.current_stream(dev))
    ref_pair = torch.zeros(48 * n, dtype=torch.int64, device=dev)
    pk.pairing_batch_dev(g1, g2, ref_pair, n, 0, torch.cuda.current_stream(dev))
    pk.last_status(0, torch.cuda.current_stream(dev))
    errors = []

    def head(t, m=6):
        return pk.layout.to_aos(t.view(48, n)[:, :m].cpu().numpy().view(np.uint64).reshape(-1).copy(), 48)

    def work(t):
        try:
            st = torch.cuda.Stream(dev)
            import ctypes
            S = ctypes.c_void_p(st.cuda_stream)
            outs = []
            # result buffers first, on this thread's stream: torch.zeros fills on the CURRENT stream, and a fill on the default
            # stream would race the engine's writes on `st`
            with torch.cuda.stream(st):
                bufs = [torch.zeros(48 * n, dtype=torch.int64, device=dev) for _ in range(11)]
            st.synchronize()
            for rep in range(3):                                   # 9 pow calls in flight on one stream: the 4-slot ring wraps
                for e in exps:
                    o = bufs[len(outs)]
                    ev = np.array(e, dtype=np.uint64)
                    rc = lib.bn254_pow_batch_dev(ctypes.c_void_p(a.data_ptr()), ev.ctypes.data_as(ctypes.c_void_p), ev.size,
                                                 ctypes.c_void_p(o.data_ptr()), n, 0, S)
                    assert rc == 0
                    outs.append(o)
            fr = bufs[9]
            assert lib.bn254_frobenius_map_batch_dev(ctypes.c_void_p(a.data_ptr()), 5, ctypes.c_void_p(fr.data_ptr()), n, 0, S) == 0
            pr = bufs[10]
            pk.pairing_batch_dev(g1, g2, pr, n, 0, st)
            pk.last_status(0, st)
            for i, o in enumerate(outs):
                if not np.array_equal(head(o), want_pow[i % 3]):
                    errors.append((t, "pow", i))
            if not np.array_equal(head(fr), want_frob):
                errors.append((t, "frob"))
            if not torch.equal(pr, ref_pair):
                errors.append((t, "pairing"))
            pk.release_stream(0, st)
        except Exception as e:      # noqa: BLE001
            errors.append((t, repr(e)))

    th = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors


def _busy_stream(pk, torch, dev, st, n=1 << 20):
    """Inputs for a 2^20-lane launch (~110 ms of k_pairing) on stream `st`, generated and synchronised."""
    with torch.cuda.stream(st):
        g1 = torch.zeros(8 * n, dtype=torch.int64, device=dev)
        g2 = torch.zeros(16 * n, dtype=torch.int64, device=dev)
        out = torch.zeros(48 * n, dtype=torch.int64, device=dev)
    pk.generate_pairs_dev(0xB254000B, g1, g2, n, 0, st)
    pk.last_status(0, st)
    return g1, g2, out, n


def test_dev_calls_after_reserve_neither_allocate_nor_wait():
    """include/bn254_pairing.h, STREAM: after bn254_reserve(device, stream, n, k) a `_dev` call of that size returns while a
    2^20-lane launch is still running on the same stream -- including the calls that need more than the pairing did (k = 4
    scratch, the verdict's Fq12 buffer, pow_native's digit buffer and staging slot)."""
    import time
    import torch
    pk = H.pkg()
    dev = torch.device("cuda:0")
    st = torch.cuda.Stream(dev)
    g1, g2, out, n = _busy_stream(pk, torch, dev, st)
    groups, k = n // 4, 4
    pk.reserve(n, k, 0, st)
    with torch.cuda.stream(st):
        verdict = torch.zeros(groups, dtype=torch.uint8, device=dev)
        x = torch.zeros(48 * 256, dtype=torch.int64, device=dev)
        x.view(48, 256)[0] = 7
        y = torch.zeros(48 * 256, dtype=torch.int64, device=dev)
    st.synchronize()
    done = torch.cuda.Event()
    pk.pairing_batch_dev(g1, g2, out, n, 0, st)                # ~110 ms
    done.record(st)
    t0 = time.perf_counter()
    pk.multi_pairing_check_batch_dev(g1, g2, verdict, groups, k, 0, st)
    pk.pow_batch_dev(x, [pk.BN_X], y, 256, 0, st)
    pk.fq12_mul_batch_dev(x, x, y, 256, 0, st)
    dt = time.perf_counter() - t0
    still_running = not done.query()
    pk.last_status(0, st)
    assert still_running, "a `_dev` call waited for the stream"
    assert dt < 0.05, f"three `_dev` calls took {dt * 1e3:.1f} ms on the host"
    assert not bool(verdict.any())                              # generic groups: no product is one
    pk.release_stream(0, st)


def test_unreserved_growth_does_not_wait_for_the_stream():
    """Without bn254_reserve a call that needs a larger buffer allocates it on the calling thread and RETIRES the old one (freed
    in the next bn254_last_status): it still does not wait for the work queued on the stream, and the queued launch, which
    uses the old scratch, finishes with the right result."""
    import torch
    pk = H.pkg()
    dev = torch.device("cuda:0")
    st = torch.cuda.Stream(dev)
    g1, g2, out, n = _busy_stream(pk, torch, dev, st)
    ref = torch.zeros(48 * 4096, dtype=torch.int64, device=dev)
    cur = torch.cuda.current_stream(dev)
    h1, h2 = g1.view(8, n)[:, :4096].contiguous().view(-1), g2.view(16, n)[:, :4096].contiguous().view(-1)
    pk.pairing_batch_dev(h1, h2, ref, 4096, 0, cur)
    pk.last_status(0, cur)
    with torch.cuda.stream(st):
        o4 = torch.zeros(48 * 1024, dtype=torch.int64, device=dev)
    st.synchronize()
    done = torch.cuda.Event()
    pk.pairing_batch_dev(g1, g2, out, n, 0, st)                # scratch for k = 1 at a full grid
    done.record(st)
    pk.multi_pairing_batch_dev(g1, g2, o4, 1024, 4, True, 0, st)      # k = 4: the scratch must grow while the launch runs
    still_running = not done.query()
    pk.last_status(0, st)
    assert still_running, "growing a buffer waited for the stream"
    assert torch.equal(out.view(48, n)[:, :4096], ref.view(48, 4096))
    pk.release_stream(0, st)


def test_release_stream_races_calls_on_the_same_stream():
    """bn254_release_stream from one thread while another keeps calling on the same (device, stream): the context is
    reference-counted, the caller either keeps the old context until it returns or creates a fresh one -- results stay right."""
    import torch
    pk = H.pkg()
    dev = torch.device("cuda:0")
    st = torch.cuda.Stream(dev)
    n = 2048
    with torch.cuda.stream(st):
        g1 = torch.zeros(8 * n, dtype=torch.int64, device=dev)
        g2 = torch.zeros(16 * n, dtype=torch.int64, device=dev)
        outs = [torch.zeros(48 * n, dtype=torch.int64, device=dev) for _ in range(2)]
    pk.generate_pairs_dev(0xB254000C, g1, g2, n, 0, st)
    pk.pairing_batch_dev(g1, g2, outs[0], n, 0, st)
    pk.last_status(0, st)
    errors, stop = [], threading.Event()

    def caller():
        try:
            for _ in range(40):
                pk.pairing_batch_dev(g1, g2, outs[1], n, 0, st)
                pk.last_status(0, st)
                if not torch.equal(outs[0], outs[1]):
                    errors.append("mismatch")
        except Exception as e:      # noqa: BLE001
            errors.append(repr(e))
        finally:
            stop.set()

    def releaser():
        try:
            while not stop.is_set():
                pk.release_stream(0, st)
        except Exception as e:      # noqa: BLE001
            errors.append(repr(e))

    th = [threading.Thread(target=caller), threading.Thread(target=releaser)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors[:3]
    pk.release_stream(0, st)


def test_scratch_geometry():
    """Scratch geometry: the workgroup pitch is a 32-bit kernel argument (the base blockIdx * pitch is formed in 64 bits: builds with the
    split Miller loop -- KGEN_FISSION=1, measured and not adopted -- keep a line area that takes a full grid of k = 4 blocks past 4 GiB).
    The shipped build: 0.5 GiB per stream for single pairings at a full grid, 2.5 GiB for 64 pairs per lane."""
    pk = H.pkg()
    sb = pk.load_library().bn254_scratch_bytes
    assert sb(1 << 20, 1) == 512 << 20 and sb(1 << 20, 64) == 2560 << 20
    assert sb(1 << 20, 4) < sb(1 << 20, 5) < sb(1 << 20, 64)


def test_points_at_infinity_get_a_distinct_status():
    """'Infinity flags: out of contract -- return a distinct status'.  The reference never checks (its line functions read
    raw x / y, /root/reference/src/miller_loop_native.rs:10-44); bn254_check_points flags ark's affine identity (x = y = 0) in either group."""
    import torch
    pk = H.pkg()
    n = 1000
    g1a, g2a = _pairs(n)
    g1, g2 = H.to_soa(g1a, 8), H.to_soa(g2a, 16)
    pk.check_points(g1, g2, n)                                    # clean batch: no error
    bad1 = g1.copy().reshape(8, n)
    bad1[:, 777] = 0
    with pytest.raises(pk.Bn254Error) as ei:
        pk.check_points(bad1.reshape(-1), g2, n)
    assert ei.value.status == pk.ERR_INFINITY
    bad2 = g2.copy().reshape(16, n)
    bad2[:, n - 1] = 0
    with pytest.raises(pk.Bn254Error) as ei:
        pk.check_points(g1, bad2.reshape(-1), n)
    assert ei.value.status == pk.ERR_INFINITY
    bad3 = g2.copy().reshape(16, n)
    bad3[:8, 5] = 0                                               # x = 0 alone is a legitimate coordinate
    pk.check_points(g1, bad3.reshape(-1), n)
    # device form: sticky status on the stream, cleared by last_status
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev)
    d1 = torch.from_numpy(bad1.reshape(-1).view(np.int64)).to(dev)
    d2 = torch.from_numpy(g2.view(np.int64)).to(dev)
    pk.check_points_dev(d1, d2, n, 0, st)
    with pytest.raises(pk.Bn254Error) as ei:
        pk.last_status(0, st)
    assert ei.value.status == pk.ERR_INFINITY
    pk.last_status(0, st)                                         # cleared


def test_dev_calls_are_capturable_into_a_hip_graph_after_reserve():
    """After bn254_reserve a `_dev` call neither allocates, uploads nor synchronises: it is a kernel launch on the caller's stream and can be
    CAPTURED into a hipGraph (torch.cuda.CUDAGraph = hipGraph on ROCm) and replayed on new contents of the same buffers -- the throughput
    kernel on a 2^16-lane batch and the lane-cooperative kernel on a small one (pairing, and the Groth16-shape product check), replayed
    results equal to eager launches on the same inputs."""
    import torch
    pk = H.pkg()
    dev = torch.device("cuda:0")
    side = torch.cuda.Stream(dev)
    n_big, n_small, k, n_mid = 1 << 16, 300, 4, 5000
    with torch.cuda.stream(side):
        g1 = torch.zeros(8 * n_big, dtype=torch.int64, device=dev)
        g2 = torch.zeros(16 * n_big, dtype=torch.int64, device=dev)
        out_big = torch.zeros(48 * n_big, dtype=torch.int64, device=dev)
        s1 = torch.zeros(8 * n_small * k, dtype=torch.int64, device=dev)
        s2 = torch.zeros(16 * n_small * k, dtype=torch.int64, device=dev)
        out_small = torch.zeros(48 * n_small * k, dtype=torch.int64, device=dev)
        verdict = torch.zeros(n_small, dtype=torch.uint8, device=dev)
        out_mid = torch.zeros(48 * n_mid, dtype=torch.int64, device=dev)
        m1 = torch.zeros(8 * n_mid, dtype=torch.int64, device=dev)
        m2 = torch.zeros(16 * n_mid, dtype=torch.int64, device=dev)
        pk.generate_pairs_dev(0x6A01, g1, g2, n_big, 0, side)
        pk.generate_pairs_dev(0x6A02, s1, s2, n_small * k, 0, side)
        pk.reserve(n_big, k, 0, side)                       # scratch, status words, verdict buffer, every round program: nothing left to do at launch
        pk.last_status(0, side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            pk.pairing_batch_dev(g1, g2, out_big, n_big, 0, side)                         # throughput kernel
            pk.pairing_batch_dev(s1, s2, out_small, n_small * k, 0, side)                 # lane-cooperative kernel
            pk.multi_pairing_check_batch_dev(s1, s2, verdict, n_small, k, 0, side)        # k_cvm + k_is_one
            pk.pairing_batch_dev(m1, m2, out_mid, n_mid, 0, side)                         # mid-size: SEVEN launches through per-stream buffers
        want = []
        for seed in (0x6A11, 0x6A12):
            pk.generate_pairs_dev(seed, g1, g2, n_big, 0, side)
            pk.generate_pairs_dev(seed + 0x100, s1, s2, n_small * k, 0, side)
            pk.generate_pairs_dev(seed + 0x200, m1, m2, n_mid, 0, side)
            out_big.zero_(); out_small.zero_(); verdict.fill_(7); out_mid.zero_()
            graph.replay()
            side.synchronize()
            got = (out_big.clone(), out_small.clone(), verdict.clone(), out_mid.clone())
            out_big.zero_(); out_small.zero_(); verdict.fill_(7)
            pk.pairing_batch_dev(g1, g2, out_big, n_big, 0, side)
            pk.pairing_batch_dev(s1, s2, out_small, n_small * k, 0, side)
            pk.multi_pairing_check_batch_dev(s1, s2, verdict, n_small, k, 0, side)
            pk.last_status(0, side)
            assert torch.equal(got[0], out_big) and torch.equal(got[1], out_small) and torch.equal(got[2], verdict)
            pk.pairing_batch_dev(m1, m2, out_mid, n_mid, 0, side)
            pk.last_status(0, side)
            assert torch.equal(got[3], out_mid) and int(out_mid.abs().sum()) != 0
            assert int(out_big.abs().sum()) != 0 and int(verdict.max()) <= 1
            want.append(got[0][:48].clone())
        assert not torch.equal(want[0], want[1])            # the replays really ran on new inputs
    pk.release_stream(0, side)


def test_mid_size_call_is_capturable_after_a_large_reserve():
    """bn254_reserve(2^20) sizes the several-launch path's intermediates (StreamCtx::mid, fx[0..4]) for the LARGEST item count that path can
    take under the stream's threshold, not for n: an 8 192-item pairing, a 6 000-group 2-pair product and an 8 192-item final_exp_native
    (seven / seven / six launches through those buffers) are then plain launches -- capturable into a hipGraph, which a hipMalloc in
    launch_pairing / launch_fexp_pieces would break -- and the replay equals the eager results."""
    import torch
    pk = H.pkg()
    dev = torch.device("cuda:0")
    side = torch.cuda.Stream(dev)
    n_mid, n_grp = 8192, 6000
    with torch.cuda.stream(side):
        m1 = torch.zeros(8 * 2 * n_grp, dtype=torch.int64, device=dev)
        m2 = torch.zeros(16 * 2 * n_grp, dtype=torch.int64, device=dev)
        out = torch.zeros(48 * n_mid, dtype=torch.int64, device=dev)
        out_g = torch.zeros(48 * n_grp, dtype=torch.int64, device=dev)
        out_f = torch.zeros(48 * n_mid, dtype=torch.int64, device=dev)
        pk.generate_pairs_dev(0x6B01, m1, m2, 2 * n_grp, 0, side)
        pk.reserve(1 << 20, 2, 0, side)
        pk.last_status(0, side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            pk.pairing_batch_dev(m1, m2, out, n_mid, 0, side)
            pk.multi_pairing_batch_dev(m1, m2, out_g, n_grp, 2, True, 0, side)
            pk.final_exp_batch_dev(out, out_f, n_mid, 0, side)
        assert pk.last_kernel(0, side) == 16
        graph.replay()
        side.synchronize()
        got = (out.clone(), out_g.clone(), out_f.clone())
        out.zero_(); out_g.zero_(); out_f.zero_()
        pk.pairing_batch_dev(m1, m2, out, n_mid, 0, side)
        pk.multi_pairing_batch_dev(m1, m2, out_g, n_grp, 2, True, 0, side)
        pk.final_exp_batch_dev(out, out_f, n_mid, 0, side)
        pk.last_status(0, side)
        assert torch.equal(got[0], out) and torch.equal(got[1], out_g) and torch.equal(got[2], out_f)
        assert int(out.abs().sum()) != 0 and int(out_g.abs().sum()) != 0
    pk.release_stream(0, side)


def test_page_locked_host_buffers_same_limbs_and_detected():
    """bn254_alloc_pinned / bn254_host_register (include/bn254_pairing.h, PAGE-LOCKED HOST MEMORY): the host-pointer pipeline of a batch above
    one chunk takes plain asynchronous copies when all three arrays are page-locked -- the same limbs as from pageable memory, limb-major and
    element-major, pairings and 2-pair groups, ragged sizes; the library recognises both kinds of page-locked memory and forgets a
    registration that ended."""
    import torch
    pk = H.pkg()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev)
    n = (1 << 17) + 1234                       # three chunks, the last one ragged
    g1 = torch.zeros(8 * n, dtype=torch.int64, device=dev)
    g2 = torch.zeros(16 * n, dtype=torch.int64, device=dev)
    pk.generate_pairs_dev(0x6C01, g1, g2, n, 0, st)
    pk.last_status(0, st)
    h1, h2 = g1.cpu().numpy().view(np.uint64).copy(), g2.cpu().numpy().view(np.uint64).copy()
    want = pk.pairing_batch(h1, h2, n)                                           # pageable
    p1, p2, po = pk.alloc_pinned(8 * n), pk.alloc_pinned(16 * n), pk.alloc_pinned(48 * n)
    assert pk.host_is_pinned(p1) and pk.host_is_pinned(po) and not pk.host_is_pinned(h1)
    p1[:], p2[:], po[:] = h1, h2, 0
    assert pk.pairing_batch(p1, p2, n, out=po) is po and np.array_equal(po, want)
    # mixed (inputs page-locked, result pageable) falls back to the staged copies: same limbs
    assert np.array_equal(pk.pairing_batch(p1, p2, n), want)
    # element-major, ark order, registered numpy arrays
    e1, e2 = pk.layout.to_aos(h1, 8).copy(), pk.layout.to_aos(h2, 16).copy()
    want_e = pk.pairing_batch_elems(e1, e2, n, out_order=pk.FQ12_ARK)
    eo = np.zeros(48 * n, dtype=np.uint64)
    for a in (e1, e2, eo):
        pk.host_register(a)
    assert pk.host_is_pinned(e1) and pk.host_is_pinned(eo)
    pk.pairing_batch_elems(e1, e2, n, out_order=pk.FQ12_ARK, out=eo)
    assert np.array_equal(eo, want_e)
    # 2-pair groups through the same page-locked arrays (n // 2 groups: one plane stride for inputs, another for the result)
    groups = n // 2
    m1, m2 = pk.layout.to_soa(e1[: 8 * 2 * groups], 8), pk.layout.to_soa(e2[: 16 * 2 * groups], 16)
    want_m = pk.multi_pairing_batch(m1, m2, groups, 2)
    go = pk.alloc_pinned(48 * groups)
    q1, q2 = pk.alloc_pinned(8 * 2 * groups), pk.alloc_pinned(16 * 2 * groups)
    q1[:], q2[:] = m1, m2
    pk.multi_pairing_batch(q1, q2, groups, 2, out=go)
    assert np.array_equal(go, want_m)
    for a in (e1, e2, eo):
        pk.host_unregister(a)
    assert not pk.host_is_pinned(eo)
    for a in (p1, p2, po, go, q1, q2):
        pk.free_pinned(a)
    with pytest.raises(pk.Bn254Error):
        pk.pairing_batch(h1, h2, n, out=np.zeros(3, dtype=np.uint64))
