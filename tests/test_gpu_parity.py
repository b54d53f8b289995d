"""GPU parity tests (bit-exact, u64 Montgomery limbs) through the C ABI against the CPU oracle and
the committed golden fixtures.  Integer work: the bar is identical limbs, no tolerance."""
import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu

HX = lambda xs: [int(x, 16) for x in xs]


def _golden_points(vec):
    P = [tuple(HX(p)) for p in vec["g1"]]
    Q = [((int(q[0], 16), int(q[1], 16)), (int(q[2], 16), int(q[3], 16))) for q in vec["g2"]]
    return P, Q


@pytest.fixture(scope="module")
def vec():
    return H.load_golden("bn254_vectors.json")


def test_golden_pairing_and_miller(vec):
    """configs[0] (e(G1gen, G2gen)) and 12 random pairs: miller_loop_native, final_exp_native, pairing."""
    pk = H.pkg()
    P, Q = _golden_points(vec)
    n = len(P)
    g1, g2 = H.to_soa(H.g1_aos(P), 8), H.to_soa(H.g2_aos(Q), 16)
    got_m = H.fq12_from_aos(H.to_aos(pk.miller_loop_batch(g1, g2, n), 48), n)
    got_p = H.fq12_from_aos(H.to_aos(pk.pairing_batch(g1, g2, n), 48), n)
    for i in range(n):
        assert got_m[i] == HX(vec["miller"][i]), f"miller_loop_native mismatch at {i}"
        assert got_p[i] == HX(vec["pairing"][i]), f"pairing mismatch at {i}"
    # final_exp_native on the Miller values
    f = H.to_soa(H.fq12_aos([HX(m) for m in vec["miller"]]), 48)
    got_f = H.fq12_from_aos(H.to_aos(pk.final_exp_batch(f, n), 48), n)
    assert got_f == [HX(p) for p in vec["pairing"]]


def test_golden_final_exp_arbitrary_fq12(vec):
    """T4 shape (final_exp_native.rs:274-285): arbitrary, non-unitary Fq12 inputs (and the identity)."""
    pk = H.pkg()
    xs = [HX(x) for x in vec["fq12_in"]]
    n = len(xs)
    a = H.to_soa(H.fq12_aos(xs), 48)
    assert H.fq12_from_aos(H.to_aos(pk.final_exp_batch(a, n), 48), n) == [HX(x) for x in vec["final_exp"]]
    assert H.fq12_from_aos(H.to_aos(pk.pow_batch(a, [pk.BN_X], n), 48), n) == [HX(x) for x in vec["pow_x"]]
    for k, want in vec["frobenius"].items():
        got = H.fq12_from_aos(H.to_aos(pk.frobenius_map_batch(a, int(k), n), 48), n)
        assert got == [HX(x) for x in want], f"frobenius_map_native power {k}"
    b = H.to_soa(H.fq12_aos(xs[1:] + xs[:1]), 48)
    assert H.fq12_from_aos(H.to_aos(pk.fq12_mul_batch(a, b, n), 48), n) == [HX(x) for x in vec["fq12_mul"]]


def test_golden_multi_miller(vec):
    pk = H.pkg()
    P, Q = _golden_points(vec)
    for g in vec["groups"]:
        k, idx = g["k"], g["idx"]
        g1 = H.to_soa(H.g1_aos([P[i] for i in idx]), 8)
        g2 = H.to_soa(H.g2_aos([Q[i] for i in idx]), 16)
        m = H.fq12_from_aos(pk.multi_pairing_batch(g1, g2, 1, k, do_final_exp=False), 1)[0]
        assert m == HX(g["miller"]), f"multi_miller_loop_native k={k}"
        e = H.fq12_from_aos(pk.multi_pairing_batch(g1, g2, 1, k, do_final_exp=True), 1)[0]
        assert e == HX(g["pairing"])
    t3 = vec["t3"]
    P3 = [tuple(HX(p)) for p in t3["g1"]]
    Q3 = [((int(q[0], 16), int(q[1], 16)), (int(q[2], 16), int(q[3], 16))) for q in t3["g2"]]
    m = H.fq12_from_aos(pk.multi_pairing_batch(H.to_soa(H.g1_aos(P3), 8), H.to_soa(H.g2_aos(Q3), 16), 1, 2, do_final_exp=False), 1)[0]
    assert m == HX(t3["miller"])


def test_oracle_parity_ragged_batch():
    """HIP vs C oracle on seeded subgroup points, ragged size (not a multiple of the 256-lane work item)."""
    pk = H.pkg()
    n = 300
    base_P, base_Q = H.subgroup_points(8)
    P = [base_P[i % 8] for i in range(n)]
    Q = [base_Q[(i * 3 + i // 8) % 8] for i in range(n)]
    g1a, g2a = H.g1_aos(P), H.g2_aos(Q)
    want = H.oracle_pairing(g1a, g2a, n, threads=8)
    got = H.to_aos(pk.pairing_batch(H.to_soa(g1a, 8), H.to_soa(g2a, 16), n), 48)
    assert np.array_equal(got, want)
    want_m = H.oracle_miller(g1a[: 8 * 40], g2a[: 16 * 40], 40)
    got_m = H.to_aos(pk.miller_loop_batch(H.to_soa(g1a[: 8 * 40], 8), H.to_soa(g2a[: 16 * 40], 16), 40), 48)
    assert np.array_equal(got_m, want_m)


def test_multi_pairing_batch_vs_oracle():
    """Groth16 shape (k = 4), k = 3, and long groups (k = 9, and the maximum k = 64) over ragged batches of groups;
    shared-f Miller value and final value."""
    pk = H.pkg()
    base_P, base_Q = H.subgroup_points(8)
    # (a handful of groups of 5 .. 64 pairs is spread over the lanes by default -- bn254_set_wide_groups; with 0 the k-pair kernel itself takes them)
    for wide, k, n_groups in ((None, 4, 70), (None, 3, 5), (None, 9, 3), (None, 64, 2), (0, 9, 3), (0, 64, 2), (0, 5, 70)):
        old = pk.get_wide_groups()
        if wide is not None:
            pk.set_wide_groups(wide)
        try:
            _multi_vs_oracle(pk, base_P, base_Q, k, n_groups)
        finally:
            pk.set_wide_groups(old)


def _multi_vs_oracle(pk, base_P, base_Q, k, n_groups):
    if True:
        n = n_groups * k
        P = [base_P[(i * 5 + 1) % 8] for i in range(n)]
        Q = [base_Q[(i * 3 + i // 8) % 8] for i in range(n)]
        g1a, g2a = H.g1_aos(P), H.g2_aos(Q)
        g1, g2 = H.to_soa(g1a, 8), H.to_soa(g2a, 16)
        want_m = H.oracle_multi_miller(g1a, g2a, n_groups, k)
        got_m = H.to_aos(pk.multi_pairing_batch(g1, g2, n_groups, k, do_final_exp=False), 48)
        assert np.array_equal(got_m, want_m), f"multi_miller_loop_native k={k}"
        want = H.oracle_multi_pairing(g1a, g2a, n_groups, k)
        got = H.to_aos(pk.multi_pairing_batch(g1, g2, n_groups, k, do_final_exp=True), 48)
        assert np.array_equal(got, want), f"multi pairing k={k}"


def test_groth16_style_product_is_one():
    """e(aP, bQ) e(abP, -Q) = 1 for every group (T3 pattern, final_exp_native.rs:245-263), on device-generated points."""
    import torch
    pk = H.pkg()
    vec = H.load_golden("bn254_vectors.json")
    t3 = vec["t3"]
    P3 = [tuple(int(x, 16) for x in p) for p in t3["g1"]]
    Q3 = [((int(q[0], 16), int(q[1], 16)), (int(q[2], 16), int(q[3], 16))) for q in t3["g2"]]
    n_groups = 300
    g1 = H.to_soa(H.g1_aos(P3 * n_groups), 8)
    g2 = H.to_soa(H.g2_aos(Q3 * n_groups), 16)
    out = H.fq12_from_aos(H.to_aos(pk.multi_pairing_batch(g1, g2, n_groups, 2, do_final_exp=True), 48), n_groups)
    one = [1] + [0] * 11
    assert all(o == one for o in out)


def test_final_exp_zero_is_an_error():
    """final_exp_native(0): the reference panics (division by zero, final_exp_native.rs:200)."""
    pk = H.pkg()
    with pytest.raises(pk.Bn254Error) as ei:
        pk.final_exp_batch(np.zeros(48 * 3, dtype=np.uint64), 3)
    assert ei.value.status == pk.ERR_ZERO_DIVISOR
    # and the status word is cleared afterwards
    x = H.to_soa(H.fq12_aos([[1] + [0] * 11]), 48)
    assert H.fq12_from_aos(pk.final_exp_batch(x, 1), 1)[0] == [1] + [0] * 11


def test_large_batch_properties():
    """BASELINE configs[1] size (2^16): on-device inputs, spot-check vs oracle + bilinearity-free invariants:
    every output is in the order-r subgroup's image under x -> x^r == 1 is too slow on CPU for all, so check
    (a) 2048 random positions against the oracle, (b) determinism of two runs, (c) no lane wrote outside its slot."""
    import torch
    pk = H.pkg()
    n = 1 << 16
    dev = torch.device("cuda:0")
    g1 = torch.zeros(8 * n, dtype=torch.int64, device=dev)
    g2 = torch.zeros(16 * n, dtype=torch.int64, device=dev)
    out = torch.full((48 * n + 64,), -1, dtype=torch.int64, device=dev)      # guard words behind the output
    out2 = torch.zeros(48 * n, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream(dev)
    pk.generate_pairs_dev(0xB2540001, g1, g2, n, 0, st)
    pk.pairing_batch_dev(g1, g2, out, n, 0, st)
    pk.pairing_batch_dev(g1, g2, out2, n, 0, st)
    pk.last_status(0, st)
    assert torch.equal(out[:48 * n], out2)
    assert bool((out[48 * n:] == -1).all())
    rng = np.random.default_rng(5)
    pos = np.sort(rng.choice(n, size=2048, replace=False))
    g1h = g1.cpu().numpy().view(np.uint64).reshape(8, n)[:, pos].reshape(-1).copy()
    g2h = g2.cpu().numpy().view(np.uint64).reshape(16, n)[:, pos].reshape(-1).copy()
    got = out[:48 * n].cpu().numpy().view(np.uint64).reshape(48, n)[:, pos].reshape(-1).copy()
    want = H.oracle_pairing(pk.layout.to_aos(g1h, 8), pk.layout.to_aos(g2h, 16), len(pos), threads=min(64, len(__import__("os").sched_getaffinity(0))))
    assert np.array_equal(pk.layout.to_aos(got, 48), want)


def test_multi_pairing_check_verdicts(vec):
    """On-device `== MyFq12::one` verdict (final_exp_native.rs:245-263): product-one groups interleaved with groups that
    are not; the byte verdict equals the comparison done on the full Fq12 outputs."""
    pk = H.pkg()
    t3 = vec["t3"]
    P3 = [tuple(int(x, 16) for x in p) for p in t3["g1"]]
    Q3 = [((int(q[0], 16), int(q[1], 16)), (int(q[2], 16), int(q[3], 16))) for q in t3["g2"]]
    P, Q = _golden_points(vec)
    n_groups = 259
    ps, qs, want = [], [], []
    for g in range(n_groups):
        if g % 3 == 1:                                   # a group whose product is not one
            i, j = g % len(P), (g + 1) % len(P)
            ps += [P[i], P[j]]
            qs += [Q[i], Q[j]]
            want.append(0)
        else:
            ps += P3
            qs += Q3
            want.append(1)
    g1, g2 = H.to_soa(H.g1_aos(ps), 8), H.to_soa(H.g2_aos(qs), 16)
    verdict = pk.multi_pairing_check_batch(g1, g2, n_groups, 2)
    assert verdict.dtype == np.uint8 and verdict.tolist() == want
    full = H.fq12_from_aos(H.to_aos(pk.multi_pairing_batch(g1, g2, n_groups, 2, do_final_exp=True), 48), n_groups)
    assert [int(o == [1] + [0] * 11) for o in full] == want
    # k = 1: e(P, Q) is never one for subgroup points
    v1 = pk.multi_pairing_check_batch(H.to_soa(H.g1_aos(P), 8), H.to_soa(H.g2_aos(Q), 16), len(P), 1)
    assert not v1.any()


def test_sharded_entry_point_single_process(vec):
    """bn254_pairing_sharded / bn254_multi_pairing_sharded (one process, several devices): slices are staged with 2-D
    copies out of the caller's SoA planes; with every visible device the result equals the single-device call."""
    pk = H.pkg()
    n = 300
    P, Q = H.subgroup_points(n, seed=77)
    g1, g2 = H.to_soa(H.g1_aos(P), 8), H.to_soa(H.g2_aos(Q), 16)
    want = pk.pairing_batch(g1, g2, n)
    n_dev = pk.device_count()
    for d in sorted({1, n_dev}):
        assert np.array_equal(pk.pairing_sharded(g1, g2, n, d), want)
    k, n_groups = 3, 100
    assert np.array_equal(pk.multi_pairing_sharded(g1, g2, n_groups, k, n_dev), pk.multi_pairing_batch(g1, g2, n_groups, k))
    with pytest.raises(pk.Bn254Error) as ei:
        pk.pairing_sharded(g1, g2, n, n_dev + 1)
    assert ei.value.status == pk.ERR_INVALID_ARG
    # element-major host arrays through the same per-device pipeline
    e1, e2 = H.g1_aos(P), H.g2_aos(Q)
    assert np.array_equal(pk.pairing_sharded_elems(e1, e2, n, n_dev), H.to_aos(want, 48))
    assert np.array_equal(pk.multi_pairing_sharded_elems(e1, e2, n_groups, k, n_dev, do_final_exp=False),
                          H.to_aos(pk.multi_pairing_batch(g1, g2, n_groups, k, do_final_exp=False), 48))


def test_streams_are_independent():
    """Two streams of one device run concurrently on private scratch and status words."""
    import torch
    pk = H.pkg()
    n = 1 << 14
    dev = torch.device("cuda:0")
    bufs = []
    for seed in (11, 12):
        g1 = torch.zeros(8 * n, dtype=torch.int64, device=dev)
        g2 = torch.zeros(16 * n, dtype=torch.int64, device=dev)
        pk.generate_pairs_dev(seed, g1, g2, n, 0, torch.cuda.current_stream(dev))
        bufs.append((g1, g2, torch.zeros(48 * n, dtype=torch.int64, device=dev), torch.zeros(48 * n, dtype=torch.int64, device=dev)))
    torch.cuda.synchronize()
    for g1, g2, ref, _ in bufs:                           # reference: one after the other on the current stream
        pk.pairing_batch_dev(g1, g2, ref, n, 0, torch.cuda.current_stream(dev))
    pk.last_status(0, torch.cuda.current_stream(dev))
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    for rep in range(3):
        for (g1, g2, _, out), st in zip(bufs, streams):
            pk.pairing_batch_dev(g1, g2, out, n, 0, st)
    for st in streams:
        pk.last_status(0, st)
    for _, _, ref, out in bufs:
        assert torch.equal(ref, out)
    for st in streams:
        pk.release_stream(0, st)


def test_empty_and_single_element_batches(vec):
    pk = H.pkg()
    assert pk.pairing_batch(np.zeros(0, np.uint64), np.zeros(0, np.uint64), 0).size == 0
    assert pk.multi_pairing_check_batch(np.zeros(0, np.uint64), np.zeros(0, np.uint64), 0, 4).size == 0
    P, Q = _golden_points(vec)
    g1, g2 = H.to_soa(H.g1_aos(P[:1]), 8), H.to_soa(H.g2_aos(Q[:1]), 16)
    assert H.fq12_from_aos(pk.pairing_batch(g1, g2, 1), 1)[0] == HX(vec["pairing"][0])
    with pytest.raises(pk.Bn254Error):
        pk.pairing_batch(g1, g2, 2)                       # buffer length does not match n


def test_host_pipeline_matches_device_path():
    """Host-pointer calls above 2^16 lanes run chunked on private streams (copies under compute): same limbs as the
    single-launch device path, ragged tail included; also for k-pair groups."""
    import torch
    pk = H.pkg()
    n = (1 << 17) * 2 + 300
    dev = torch.device("cuda:0")
    g1 = torch.zeros(8 * n, dtype=torch.int64, device=dev)
    g2 = torch.zeros(16 * n, dtype=torch.int64, device=dev)
    out = torch.zeros(48 * n, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream(dev)
    pk.generate_pairs_dev(0xB2540003, g1, g2, n, 0, st)
    pk.pairing_batch_dev(g1, g2, out, n, 0, st)
    pk.last_status(0, st)
    h1, h2 = g1.cpu().numpy().view(np.uint64).copy(), g2.cpu().numpy().view(np.uint64).copy()
    assert np.array_equal(pk.pairing_batch(h1, h2, n), out.cpu().numpy().view(np.uint64))
    # lanes of the second and third work items of a wave (persistent grid loop) against the oracle
    pos = np.array([65536, 65537 + 255, 99999, 131071, 131072, 200000, 262143, 262144, n - 1])
    g1s = h1.reshape(8, n)[:, pos].reshape(-1).copy()
    g2s = h2.reshape(16, n)[:, pos].reshape(-1).copy()
    want = H.oracle_pairing(pk.layout.to_aos(g1s, 8), pk.layout.to_aos(g2s, 16), len(pos), threads=9)
    got = out.cpu().numpy().view(np.uint64).reshape(48, n)[:, pos].reshape(-1).copy()
    assert np.array_equal(pk.layout.to_aos(got, 48), want)
    k = 2
    groups = n // k
    og = torch.zeros(48 * groups, dtype=torch.int64, device=dev)
    # the first groups*k pairs of every plane, re-packed to batch length groups*k
    sel = lambda t, planes: t.view(planes, n)[:, :groups * k].contiguous().view(-1)
    g1k, g2k = sel(g1, 8), sel(g2, 16)
    pk.multi_pairing_batch_dev(g1k, g2k, og, groups, k, True, 0, st)
    pk.last_status(0, st)
    got = pk.multi_pairing_batch(g1k.cpu().numpy().view(np.uint64).copy(), g2k.cpu().numpy().view(np.uint64).copy(), groups, k)
    assert np.array_equal(got, og.cpu().numpy().view(np.uint64))


def test_full_size_product_check():
    """Size-independent property at BASELINE configs[1] size: e(P_i, Q_i) e(P_i, -Q_i) = 1 for all 2^16 generated pairs
    (T3 pattern, final_exp_native.rs:245-263), through the shared-f multi-pairing kernel and the on-device verdict."""
    import torch
    pk = H.pkg()
    n = 1 << 16
    dev = torch.device("cuda:0")
    g1 = torch.zeros(8 * n, dtype=torch.int64, device=dev)
    g2 = torch.zeros(16 * n, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream(dev)
    pk.generate_pairs_dev(0xB2540004, g1, g2, n, 0, st)
    pk.last_status(0, st)
    h1 = g1.cpu().numpy().view(np.uint64).reshape(8, n)
    h2 = g2.cpu().numpy().view(np.uint64).reshape(16, n)
    # -Q: y -> p - y on both components (Montgomery limbs negate like canonical ones); y != 0 on these curves
    P_LIMBS = [0x3c208c16d87cfd47, 0x97816a916871ca8d, 0xb85045b68181585d, 0x30644e72e131a029]
    neg = h2.copy()
    for comp in (2, 3):
        borrow = np.zeros(n, dtype=np.uint64)
        for l in range(4):
            y = h2[comp * 4 + l]
            pl = np.uint64(P_LIMBS[l])
            d = pl - y - borrow
            borrow = ((y + borrow > pl) | ((borrow == 1) & (y == np.uint64(0xFFFFFFFFFFFFFFFF)))).astype(np.uint64)
            neg[comp * 4 + l] = d
    # groups of k = 2: (P_i, Q_i), (P_i, -Q_i); pair j of group i is element 2 i + j
    g1k = np.repeat(h1, 2, axis=1)
    g2k = np.empty((16, 2 * n), dtype=np.uint64)
    g2k[:, 0::2], g2k[:, 1::2] = h2, neg
    verdict = pk.multi_pairing_check_batch(g1k.reshape(-1), g2k.reshape(-1), n, 2)
    assert verdict.shape == (n,) and bool(verdict.all())
    # and flipping one pair of one group breaks exactly that group
    g2k[:, 2 * 777 + 1] = h2[:, 778]
    v2 = pk.multi_pairing_check_batch(g1k.reshape(-1), g2k.reshape(-1), n, 2)
    assert int(v2.sum()) == n - 1 and v2[777] == 0


def test_helpers_vs_oracle_multi_limb_exponent():
    """pow_native with a three-limb exponent (NAF with -1 digits and a carry across limbs), frobenius_map_native for
    every power 0..11 and MyFq12 Mul on a ragged batch of arbitrary (non-unitary) elements, against the oracle."""
    pk = H.pkg()
    n = 300
    xs = H.rand_fq12(n, seed=11)
    a = H.fq12_aos(xs)
    a_soa = H.to_soa(a, 48)
    exp = [0xFFFFFFFFFFFFFFF7, 0x0123456789ABCDEF, 0x00000000DEADBEEF]
    rc, want = H.oracle_pow_native(a, exp, n)
    assert rc == 0 and np.array_equal(H.to_aos(pk.pow_batch(a_soa, exp, n), 48), want)
    for power in range(12):
        assert np.array_equal(H.to_aos(pk.frobenius_map_batch(a_soa, power, n), 48), H.oracle_frobenius(a, power, n)), power
    b = H.fq12_aos(xs[7:] + xs[:7])
    assert np.array_equal(H.to_aos(pk.fq12_mul_batch(a_soa, H.to_soa(b, 48), n), 48), H.oracle_fq12_mul(a, b, n))
    # pow_native(0, e): the reference divides (and panics) only on a -1 digit
    z = np.zeros(48 * 2, dtype=np.uint64)
    assert not pk.pow_batch(z, [5], 2).any()                  # NAF(5) = 101
    with pytest.raises(pk.Bn254Error) as ei:
        pk.pow_batch(z, [7], 2)                               # NAF(7) = 100(-1)
    assert ei.value.status == pk.ERR_ZERO_DIVISOR


def test_host_pipeline_chunk_edges():
    """Chunk boundaries of the host-pointer pipeline (chunks of 2^16 lanes): one lane more than a chunk (a 1-lane tail chunk
    for the second worker), exactly one chunk (single-launch path), two chunks plus one lane, against the device path."""
    import torch
    pk = H.pkg()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev)
    for n in ((1 << 16) + 1, 1 << 16, (1 << 17) + 1):
        g1 = torch.zeros(8 * n, dtype=torch.int64, device=dev)
        g2 = torch.zeros(16 * n, dtype=torch.int64, device=dev)
        out = torch.zeros(48 * n, dtype=torch.int64, device=dev)
        pk.generate_pairs_dev(0xB2540005 + n, g1, g2, n, 0, st)
        pk.pairing_batch_dev(g1, g2, out, n, 0, st)
        pk.last_status(0, st)
        got = pk.pairing_batch(g1.cpu().numpy().view(np.uint64).copy(), g2.cpu().numpy().view(np.uint64).copy(), n)
        assert np.array_equal(got, out.cpu().numpy().view(np.uint64)), n


# ------------------------------------------------------------------------------------------------ full-size runs per entry point
P_TOP = 0x30644e72e131a029


def _rand_fq12_dev(n, seed, dev):
    """n arbitrary (non-unitary) Fq12 elements on the device, SoA: every coefficient a uniform 256-bit pattern below p
    (top limb below p's top limb), i.e. a valid Montgomery representation of some field element."""
    import torch
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    t = torch.randint(-(1 << 63), (1 << 63) - 1, (48, n), dtype=torch.int64, device=dev, generator=g)
    top = torch.randint(0, P_TOP, (12, n), dtype=torch.int64, device=dev, generator=g)
    t[3::4] = top
    return t.view(-1)


def _take(t, words, n, pos):
    return t.view(words, n)[:, pos].cpu().numpy().view(np.uint64).reshape(-1).copy()


def test_miller_loop_full_size_vs_oracle():
    """bn254_miller_loop_batch_dev (the bit-exact miller_loop_native value, miller_loop_native.rs:320-322, with the running line
    scale divided out) over 3 x 2^16 + 300 lanes: every workgroup walks three or four 256-lane items of the persistent loop.
    640 oracle spot checks spread over the first, second, third and ragged last items; determinism; guard words."""
    import torch
    pk = H.pkg()
    n = 3 * (1 << 16) + 300
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev)
    g1 = torch.zeros(8 * n, dtype=torch.int64, device=dev)
    g2 = torch.zeros(16 * n, dtype=torch.int64, device=dev)
    pk.generate_pairs_dev(0xB2540011, g1, g2, n, 0, st)
    out = torch.full((48 * n + 64,), -1, dtype=torch.int64, device=dev)
    out2 = torch.zeros(48 * n, dtype=torch.int64, device=dev)
    pk.miller_loop_batch_dev(g1, g2, out, n, 0, st)
    pk.miller_loop_batch_dev(g1, g2, out2, n, 0, st)
    pk.last_status(0, st)
    assert torch.equal(out[:48 * n], out2) and bool((out[48 * n:] == -1).all())
    rng = np.random.default_rng(21)
    pos = np.unique(np.concatenate([rng.choice(n, size=600, replace=False), [0, 255, 256, 65535, 65536, 65791, 131071, 131072, 196607, 196608, n - 301, n - 300, n - 1]]))
    want = H.oracle_miller(pk.layout.to_aos(_take(g1, 8, n, pos), 8), pk.layout.to_aos(_take(g2, 16, n, pos), 16), len(pos))
    got = pk.layout.to_aos(_take(out[:48 * n], 48, n, pos), 48)
    assert np.array_equal(got, want)


def test_final_exp_full_size_vs_oracle():
    """bn254_final_exp_batch_dev (final_exp_native, final_exp_native.rs:209-213) on 2 x 2^16 + 77 arbitrary non-unitary Fq12
    elements (the T4 shape, :274-285): 520 oracle spot checks incl. lanes of the second and third items; determinism."""
    import torch
    pk = H.pkg()
    n = 2 * (1 << 16) + 77
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev)
    a = _rand_fq12_dev(n, 5, dev)
    out = torch.full((48 * n + 64,), -1, dtype=torch.int64, device=dev)
    out2 = torch.zeros(48 * n, dtype=torch.int64, device=dev)
    pk.final_exp_batch_dev(a, out, n, 0, st)
    pk.final_exp_batch_dev(a, out2, n, 0, st)
    pk.last_status(0, st)
    assert torch.equal(out[:48 * n], out2) and bool((out[48 * n:] == -1).all())
    rng = np.random.default_rng(22)
    pos = np.unique(np.concatenate([rng.choice(n, size=500, replace=False), [0, 255, 256, 65535, 65536, 70000, 131071, 131072, n - 1]]))
    rc, want = H.oracle_final_exp(pk.layout.to_aos(_take(a, 48, n, pos), 48), len(pos))
    assert rc == 0
    assert np.array_equal(pk.layout.to_aos(_take(out[:48 * n], 48, n, pos), 48), want)


def test_pairing_2_20_vs_oracle():
    """BASELINE.json configs[2]: 2^20 independent pairings in one launch (16 items per workgroup): determinism, guard words,
    512 oracle spot checks over the whole index range."""
    import os
    import torch
    pk = H.pkg()
    n = 1 << 20
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev)
    g1 = torch.zeros(8 * n, dtype=torch.int64, device=dev)
    g2 = torch.zeros(16 * n, dtype=torch.int64, device=dev)
    pk.generate_pairs_dev(0xB2540001, g1, g2, n, 0, st)
    out = torch.full((48 * n + 64,), -1, dtype=torch.int64, device=dev)
    out2 = torch.zeros(48 * n, dtype=torch.int64, device=dev)
    pk.pairing_batch_dev(g1, g2, out, n, 0, st)
    pk.pairing_batch_dev(g1, g2, out2, n, 0, st)
    pk.last_status(0, st)
    assert torch.equal(out[:48 * n], out2) and bool((out[48 * n:] == -1).all())
    rng = np.random.default_rng(23)
    pos = np.unique(np.concatenate([rng.choice(n, size=500, replace=False), [0, 65535, 65536, 524287, 524288, 983040, n - 256, n - 1]]))
    want = H.oracle_pairing(pk.layout.to_aos(_take(g1, 8, n, pos), 8), pk.layout.to_aos(_take(g2, 16, n, pos), 16), len(pos),
                            threads=min(32, len(os.sched_getaffinity(0))))
    assert np.array_equal(pk.layout.to_aos(_take(out[:48 * n], 48, n, pos), 48), want)


def _neg_fq_planes(y):
    """p - y on u64 limb planes y[4][m] (Montgomery limbs negate like canonical ones; y != 0 for curve points)."""
    P_LIMBS = [0x3c208c16d87cfd47, 0x97816a916871ca8d, 0xb85045b68181585d, 0x30644e72e131a029]
    out = np.empty_like(y)
    borrow = np.zeros(y.shape[1], dtype=np.uint64)
    for l in range(4):
        pl = np.uint64(P_LIMBS[l])
        out[l] = pl - y[l] - borrow
        borrow = ((y[l] + borrow > pl) | ((borrow == 1) & (y[l] == np.uint64(0xFFFFFFFFFFFFFFFF)))).astype(np.uint64)
    return out


def test_groth16_shape_2_18_groups():
    """BASELINE.json configs[3]: 2^18 groups x 4 pairs, one shared-f multi-Miller loop + one final exponentiation per group
    (multi_miller_loop_native, miller_loop_native.rs:324-326).  (a) generic groups: 96 oracle spot checks of the Fq12 value;
    (b) size-independent property over ALL groups (T3 pattern, final_exp_native.rs:245-263): groups built as
    (P, Q), (P, -Q), (P', Q'), (P', -Q') have product one -- checked with the on-device verdict."""
    import torch
    pk = H.pkg()
    groups, k = 1 << 18, 4
    n = groups * k
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev)
    g1 = torch.zeros(8 * n, dtype=torch.int64, device=dev)
    g2 = torch.zeros(16 * n, dtype=torch.int64, device=dev)
    pk.generate_pairs_dev(0xB2540013, g1, g2, n, 0, st)
    out = torch.full((48 * groups + 64,), -1, dtype=torch.int64, device=dev)
    pk.multi_pairing_batch_dev(g1, g2, out, groups, k, True, 0, st)
    pk.last_status(0, st)
    assert bool((out[48 * groups:] == -1).all())
    rng = np.random.default_rng(24)
    gp = np.unique(np.concatenate([rng.choice(groups, size=90, replace=False), [0, 255, 256, 65535, 65536, groups - 1]]))
    pairs = (gp[:, None] * k + np.arange(k)[None, :]).reshape(-1)
    want = H.oracle_multi_pairing(pk.layout.to_aos(_take(g1, 8, n, pairs), 8), pk.layout.to_aos(_take(g2, 16, n, pairs), 16), len(gp), k)
    assert np.array_equal(pk.layout.to_aos(_take(out[:48 * groups], 48, groups, gp), 48), want)
    # (b) product-one groups over the whole batch
    h1 = g1.cpu().numpy().view(np.uint64).reshape(8, n)[:, :n // 2]
    h2 = g2.cpu().numpy().view(np.uint64).reshape(16, n)[:, :n // 2]
    neg = h2.copy()
    neg[8:12] = _neg_fq_planes(h2[8:12])
    neg[12:16] = _neg_fq_planes(h2[12:16])
    g1k = np.repeat(h1, 2, axis=1)                                  # pairs 2i, 2i+1 share P_i
    g2k = np.empty((16, n), dtype=np.uint64)
    g2k[:, 0::2], g2k[:, 1::2] = h2, neg
    d1 = torch.from_numpy(g1k.reshape(-1).view(np.int64)).to(dev)
    d2 = torch.from_numpy(g2k.reshape(-1).view(np.int64)).to(dev)
    verdict = torch.zeros(groups, dtype=torch.uint8, device=dev)
    pk.multi_pairing_check_batch_dev(d1, d2, verdict, groups, k, 0, st)
    pk.last_status(0, st)
    assert bool(verdict.all())
    # breaking one pair of one group breaks exactly that group
    d2.view(16, n)[:, 4 * 4321 + 3] = d2.view(16, n)[:, 4 * 4321 + 2]
    pk.multi_pairing_check_batch_dev(d1, d2, verdict, groups, k, 0, st)
    pk.last_status(0, st)
    assert int(verdict.sum()) == groups - 1 and int(verdict[4321]) == 0


def test_generated_pairs_are_the_stated_subgroup_points():
    """bn254_generate_pairs_dev (stands in for G1Affine::rand / G2Affine::rand, /root/reference/src/pairing.rs:65-66):
    P_i = [s_i] G1, Q_i = [t_i] G2 with the SplitMix64-derived scalars the header states -- the first 64 pairs (and a few at
    the far end) against the big-integer restatement; every point of a full 2^16 batch is on its curve."""
    import torch
    pk = H.pkg()
    n = 1 << 16
    seed = 0xB2540001
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev)
    g1 = torch.zeros(8 * n, dtype=torch.int64, device=dev)
    g2 = torch.zeros(16 * n, dtype=torch.int64, device=dev)
    pk.generate_pairs_dev(seed, g1, g2, n, 0, st)
    pk.last_status(0, st)
    h1 = pk.layout.to_aos(g1.cpu().numpy().view(np.uint64), 8).reshape(n, 2, 4)
    h2 = pk.layout.to_aos(g2.cpu().numpy().view(np.uint64), 16).reshape(n, 4, 4)
    val = lambda limbs: H.R.from_mont(sum(int(limbs[l]) << (64 * l) for l in range(4)))
    for i in list(range(64)) + [255, 256, 40000, n - 1]:
        s, t = pk.generator_scalars(seed, i)
        P = H.R.g1_mul(H.R.G1_GEN, s % H.R.R_ORDER)
        Q = H.R.g2_mul(H.R.G2_GEN, t % H.R.R_ORDER)
        assert (val(h1[i, 0]), val(h1[i, 1])) == tuple(P), i
        assert ((val(h2[i, 0]), val(h2[i, 1])), (val(h2[i, 2]), val(h2[i, 3]))) == (tuple(Q[0]), tuple(Q[1])), i
    lib = H.oracle()
    g1a, g2a = h1.reshape(-1), h2.reshape(-1)
    for i in range(n):
        assert lib.oracle_g1_on_curve(H.ptr(g1a[8 * i: 8 * i + 8])) == 1, i
        assert lib.oracle_g2_on_curve(H.ptr(g2a[16 * i: 16 * i + 16])) == 1, i


def test_pow_native_zero_and_empty_exponent():
    """pow_native(a, [0]) and pow_native(a, []) return a (all-zero NAF: the loop never starts, final_exp_native.rs:56-84)."""
    pk = H.pkg()
    n = 5
    a = H.to_soa(H.fq12_aos(H.rand_fq12(n, seed=3)), 48)
    assert np.array_equal(pk.pow_batch(a, [0], n), a)
    assert np.array_equal(pk.pow_batch(a, [0, 0], n), a)
    assert np.array_equal(pk.pow_batch(a, [], n), a)
    rc, want = H.oracle_pow_native(H.to_aos(a, 48), [0], n)
    assert rc == 0 and np.array_equal(H.to_soa(want, 48), a)


def test_fused_and_split_kernels_agree_on_every_lane():
    """pairing = final_exp_native(miller_loop_native) (/root/reference/src/pairing.rs:20-22) over ALL 2^20 lanes, through
    different kernels: k_pairing (untracked projective lines: any Fq2 factor dies in the easy part) must equal k_fexp applied
    to k_miller's exact miller_loop_native value (tracked line scale, divided out by an Fq2 inversion) limb for limb on every
    lane -- a size-independent cross-check of the whole 2^20 batch (the oracle only spot-checks 512 lanes of it)."""
    import torch
    pk = H.pkg()
    n = 1 << 20
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev)
    g1 = torch.zeros(8 * n, dtype=torch.int64, device=dev)
    g2 = torch.zeros(16 * n, dtype=torch.int64, device=dev)
    pk.generate_pairs_dev(0xB2540021, g1, g2, n, 0, st)
    a = torch.zeros(48 * n, dtype=torch.int64, device=dev)
    m = torch.zeros(48 * n, dtype=torch.int64, device=dev)
    b = torch.zeros(48 * n, dtype=torch.int64, device=dev)
    pk.pairing_batch_dev(g1, g2, a, n, 0, st)
    pk.miller_loop_batch_dev(g1, g2, m, n, 0, st)
    pk.final_exp_batch_dev(m, b, n, 0, st)
    pk.last_status(0, st)
    assert torch.equal(a, b)
    assert not torch.equal(a, m)


def test_configs4_shard_size_2_21_lanes():
    """BASELINE.json configs[4] per-GPU shard: 2^21 independent pairings in ONE launch (32 work items per workgroup -- twice the
    largest launch of configs[2]).  (a) fused == split on every lane: k_pairing equals k_fexp(k_miller), (b) the first half equals
    a separate 2^20 launch of the same inputs (a lane's result does not depend on the launch it ran in), (c) 256 oracle spot
    checks over the whole index range, guard words."""
    import os
    import torch
    pk = H.pkg()
    n = 1 << 21
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev)
    g1 = torch.zeros(8 * n, dtype=torch.int64, device=dev)
    g2 = torch.zeros(16 * n, dtype=torch.int64, device=dev)
    pk.generate_pairs_dev(0xB2540041, g1, g2, n, 0, st)
    a = torch.full((48 * n + 64,), -1, dtype=torch.int64, device=dev)
    pk.pairing_batch_dev(g1, g2, a, n, 0, st)
    m = torch.zeros(48 * n, dtype=torch.int64, device=dev)
    b = torch.zeros(48 * n, dtype=torch.int64, device=dev)
    pk.miller_loop_batch_dev(g1, g2, m, n, 0, st)
    pk.final_exp_batch_dev(m, b, n, 0, st)
    pk.last_status(0, st)
    assert bool((a[48 * n:] == -1).all())
    assert torch.equal(a[:48 * n], b)
    del m, b
    half = n // 2
    h1, h2 = g1.view(8, n)[:, :half].contiguous().view(-1), g2.view(16, n)[:, :half].contiguous().view(-1)
    c = torch.zeros(48 * half, dtype=torch.int64, device=dev)
    pk.pairing_batch_dev(h1, h2, c, half, 0, st)
    pk.last_status(0, st)
    assert torch.equal(a[:48 * n].view(48, n)[:, :half], c.view(48, half))
    rng = np.random.default_rng(41)
    pos = np.unique(np.concatenate([rng.choice(n, size=248, replace=False), [0, 65535, 65536, (1 << 20) - 1, 1 << 20, (1 << 20) + 65536, n - 256, n - 1]]))
    want = H.oracle_pairing(pk.layout.to_aos(_take(g1, 8, n, pos), 8), pk.layout.to_aos(_take(g2, 16, n, pos), 16), len(pos),
                            threads=min(32, len(os.sched_getaffinity(0))))
    assert np.array_equal(pk.layout.to_aos(_take(a[:48 * n], 48, n, pos), 48), want)
