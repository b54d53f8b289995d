"""The lane-cooperative (latency) kernel of pairing() -- sixteen lanes per pairing, tools/cvm.py -- against the golden vectors,
the C oracle and the throughput kernel: identical limbs on every lane, for batch sizes around the group / wave / grid edges.
Which kernel a call takes is the stream's own setting (bn254_set_stream_latency) where it has one, else the process-wide default
(bn254_set_latency_threshold / _lanes): the tests pin the NULL stream's setting per call and never touch the defaults."""
import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu

HX = lambda xs: [int(x, 16) for x in xs]


class _Pinned:
    """The package with the kernel selection of ONE stream -- the NULL stream of device 0, which every call of this file uses -- in the
    test's hands (bn254_set_stream_latency): `set_latency_threshold` here pins that stream's threshold and the fixture's program family;
    the process-wide defaults are never touched (round 4 flipped them around every call)."""

    def __init__(self, p, lanes):
        self._p, self._lanes = p, lanes

    def __getattr__(self, k):
        return getattr(self._p, k)

    def set_latency_threshold(self, thr):
        self._p.set_stream_latency(thr, self._lanes, 0, None)


@pytest.fixture(params=[16, 32, 64], ids=["16-lanes", "32-lanes", "64-lanes"])
def pk(request):
    """the package with the lane-cooperative program family pinned on the NULL stream (sixteen / thirty-two / sixty-four lanes per
    item; functions without a program of the family take the next smaller one); the stream returns to the defaults afterwards"""
    p = H.pkg()
    keep = (p.get_latency_threshold(), p.get_latency_lanes())
    yield _Pinned(p, request.param)
    p.set_stream_latency(p.LATENCY_INHERIT, -1, 0, None)
    assert (p.get_latency_threshold(), p.get_latency_lanes()) == keep


def test_threshold_is_settable():
    """the process-wide DEFAULTS (what a stream without a setting of its own follows)"""
    pk = H.pkg()
    keep = (pk.get_latency_threshold(), pk.get_latency_lanes())
    try:
        pk.set_latency_threshold(12345)
        assert pk.get_latency_threshold() == 12345
        pk.set_latency_threshold(0)
        assert pk.get_latency_threshold() == 0
        for lanes, want in ((16, 16), (32, 32), (64, 64), (0, 0), (7, 0)):
            pk.set_latency_lanes(lanes)
            assert pk.get_latency_lanes() == want
    finally:
        pk.set_latency_threshold(keep[0])
        pk.set_latency_lanes(keep[1])


def test_program_family_follows_the_launch_size():
    """lanes = 0: sixty-four / thirty-two lanes per item up to one wave per SIMD, sixteen beyond -- same limbs either way, on both sides of the edge"""
    import torch
    pk = H.pkg()
    assert pk.get_latency_lanes() == 0
    try:
        seen = {}
        for n in (1024, 1025, 2047, 2048, 2049, 3000):
            g1, g2 = _dev_pairs(pk, n, 0x5EED + n)
            outs = []
            for thr in (0, 1 << 20):
                o = torch.empty(48 * n, dtype=torch.int64, device=torch.device("cuda:0"))
                pk.set_stream_latency(thr, 0, 0, None)
                pk.pairing_batch_dev(g1, g2, o, n)
                outs.append(o)
                seen[(n, thr)] = pk.last_kernel(0, None)
            pk.last_status()
            assert torch.equal(outs[0], outs[1]), n
        assert all(seen[(n, 0)] == 1 for n in (1024, 2048, 3000))
        assert seen[(1024, 1 << 20)] == 64 and seen[(1025, 1 << 20)] == 32 and seen[(2048, 1 << 20)] == 32
        assert seen[(2049, 1 << 20)] == 16 and seen[(3000, 1 << 20)] == 16
        # final_exp_native has a sixty-four-lane program of its own since round 5 (single products: its chain is shorter there too)
        n = 700
        g1, g2 = _dev_pairs(pk, n, 0xFE70)
        f = torch.empty(48 * n, dtype=torch.int64, device=torch.device("cuda:0"))
        pk.set_stream_latency(0, 0, 0, None)
        pk.miller_loop_batch_dev(g1, g2, f, n)
        outs = []
        for thr, want in ((0, 1), (1 << 20, 64)):
            o = torch.empty(48 * n, dtype=torch.int64, device=torch.device("cuda:0"))
            pk.set_stream_latency(thr, 0, 0, None)
            pk.final_exp_batch_dev(f, o, n)
            assert pk.last_kernel(0, None) == want
            outs.append(o)
        pk.last_status()
        assert torch.equal(outs[0], outs[1])
    finally:
        pk.set_stream_latency(pk.LATENCY_INHERIT, -1, 0, None)


def test_golden_vectors_on_the_latency_kernel(pk):
    vec = H.load_golden("bn254_vectors.json")
    P = [tuple(HX(p)) for p in vec["g1"]]
    Q = [((int(q[0], 16), int(q[1], 16)), (int(q[2], 16), int(q[3], 16))) for q in vec["g2"]]
    n = len(P)
    g1, g2 = H.to_soa(H.g1_aos(P), 8), H.to_soa(H.g2_aos(Q), 16)
    pk.set_latency_threshold(1 << 20)
    got = H.fq12_from_aos(H.to_aos(pk.pairing_batch(g1, g2, n), 48), n)
    for i in range(n):
        assert got[i] == HX(vec["pairing"][i]), f"pairing mismatch at {i}"
    # the scalar signature: one pairing (configs[0]: e(G1, G2))
    one = H.fq12_from_aos(pk.pairing_batch(H.g1_aos(P[:1]), H.g2_aos(Q[:1]), 1), 1)
    assert one[0] == HX(vec["pairing"][0])


@pytest.mark.parametrize("thr", [0, 1 << 20], ids=["throughput-kernel", "latency-kernel"])
@pytest.mark.parametrize("n", [1, 3, 4, 5, 63, 64, 65, 300])
def test_oracle_parity_small_batches(pk, n, thr):
    base_P, base_Q = H.subgroup_points(8)
    P = [base_P[(i * 5 + 1) % 8] for i in range(n)]
    Q = [base_Q[(i * 3 + i // 8) % 8] for i in range(n)]
    g1a, g2a = H.g1_aos(P), H.g2_aos(Q)
    want = H.oracle_pairing(g1a, g2a, n, threads=8)
    pk.set_latency_threshold(thr)
    got = H.to_aos(pk.pairing_batch(H.to_soa(g1a, 8), H.to_soa(g2a, 16), n), 48)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("thr", [0, 1 << 20], ids=["throughput-kernel", "latency-kernel"])
def test_zero_divisor_status(pk, thr):
    """pairing of the all-zero encodings: the Miller value is 0 and the easy part divides by it -- the reference panics
    (final_exp_native.rs:200), both kernels raise the status word; a valid lane beside it is unaffected, and the word is cleared"""
    base_P, base_Q = H.subgroup_points(2)
    g1a, g2a = H.g1_aos(base_P), H.g2_aos(base_Q)
    want = H.oracle_pairing(g1a, g2a, 2, threads=1)
    g1a[:8] = 0
    g2a[:16] = 0
    pk.set_latency_threshold(thr)
    with pytest.raises(pk.Bn254Error) as ei:
        pk.pairing_batch(H.to_soa(g1a, 8), H.to_soa(g2a, 16), 2)
    assert ei.value.status == pk.ERR_ZERO_DIVISOR
    got = H.to_aos(pk.pairing_batch(H.to_soa(g1a[8:], 8), H.to_soa(g2a[16:], 16), 1), 48)
    assert np.array_equal(got, want[48:])


def test_both_kernels_agree_on_every_lane(pk):
    """10 000 generated subgroup pairs, device resident: latency kernel == throughput kernel (torch.equal on the limb planes)"""
    import torch
    dev = torch.device("cuda:0")
    n = 10000
    g1 = torch.empty(8 * n, dtype=torch.int64, device=dev)
    g2 = torch.empty(16 * n, dtype=torch.int64, device=dev)
    pk.generate_pairs_dev(0xB2540001, g1, g2, n)
    a = torch.empty(48 * n, dtype=torch.int64, device=dev)
    b = torch.empty(48 * n, dtype=torch.int64, device=dev)
    pk.set_latency_threshold(0)
    pk.pairing_batch_dev(g1, g2, a, n)
    pk.set_latency_threshold(1 << 20)
    pk.pairing_batch_dev(g1, g2, b, n)
    pk.last_status()
    assert torch.equal(a, b)


def test_threshold_selects_the_kernel(pk):
    """above the threshold the throughput kernel runs (same values; the dispatch is by batch size only)"""
    import torch
    dev = torch.device("cuda:0")
    n = 8
    g1 = torch.empty(8 * n, dtype=torch.int64, device=dev)
    g2 = torch.empty(16 * n, dtype=torch.int64, device=dev)
    pk.generate_pairs_dev(7, g1, g2, n)
    outs = []
    for thr in (0, 7, 8):
        o = torch.empty(48 * n, dtype=torch.int64, device=dev)
        pk.set_latency_threshold(thr)
        pk.pairing_batch_dev(g1, g2, o, n)
        outs.append(o)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


def _dev_pairs(pk, n, seed):
    import torch
    dev = torch.device("cuda:0")
    g1 = torch.empty(8 * n, dtype=torch.int64, device=dev)
    g2 = torch.empty(16 * n, dtype=torch.int64, device=dev)
    pk.generate_pairs_dev(seed, g1, g2, n)
    return g1, g2


def _both(pk, fn, words, n):
    """fn(out) under both kernels -> (throughput result, latency result)"""
    import torch
    outs = []
    for thr in (0, 1 << 20):
        o = torch.empty(words * n, dtype=torch.int64, device=torch.device("cuda:0"))
        pk.set_latency_threshold(thr)
        fn(o)
        pk.last_status()
        outs.append(o)
    return outs


@pytest.mark.parametrize("n", [1, 5, 257])
def test_miller_loop_and_final_exp_on_the_latency_kernel(pk, n):
    """miller_loop_native (the exact value) and final_exp_native: lane-cooperative programs == throughput kernels, every lane"""
    import torch
    g1, g2 = _dev_pairs(pk, n, 0xB2540001)
    a, b = _both(pk, lambda o: pk.miller_loop_batch_dev(g1, g2, o, n), 48, n)
    assert torch.equal(a, b)
    c, d = _both(pk, lambda o: pk.final_exp_batch_dev(a, o, n), 48, n)
    assert torch.equal(c, d)
    e, f = _both(pk, lambda o: pk.pairing_batch_dev(g1, g2, o, n), 48, n)
    assert torch.equal(e, f) and torch.equal(c, e)                 # pairing = final_exp(miller)


@pytest.mark.parametrize("n", [4097, 9000])
def test_mid_size_batches_take_the_multi_launch_form(n):
    """Above 4 096 items (lanes left to the launch size) `pairing` runs as seven launches of the lane-cooperative kernel -- the Miller loop
    without the line scale, then final_exp_native in six pieces -- and `final_exp_native` alone as those six: every lane equals the
    throughput kernels (round 5)."""
    import torch
    p = H.pkg()
    g1, g2 = _dev_pairs(p, n, 0xB25400AA + n)
    try:
        outs = {}
        for thr in (0, 1 << 20):
            p.set_stream_latency(thr, 0, 0, None)
            mil = torch.empty(48 * n, dtype=torch.int64, device=torch.device("cuda:0"))
            fe = torch.empty_like(mil)
            pa = torch.empty_like(mil)
            p.miller_loop_batch_dev(g1, g2, mil, n)
            p.final_exp_batch_dev(mil, fe, n)
            p.pairing_batch_dev(g1, g2, pa, n)
            p.last_status()
            assert p.last_kernel(0, None) == (1 if thr == 0 else 16)
            outs[thr] = (mil, fe, pa)
        a, b = outs[0], outs[1 << 20]
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(b[1], b[2])
        # a zero input among them: the easy part's inversion raises the status word from its own launch
        z = a[0].clone()
        z.view(48, n)[:, 5] = 0
        p.final_exp_batch_dev(z, torch.empty_like(z), n)
        with pytest.raises(p.Bn254Error) as e:
            p.last_status()
        assert e.value.status == p.ERR_ZERO_DIVISOR
    finally:
        p.set_stream_latency(p.LATENCY_INHERIT, -1, 0, None)


@pytest.mark.parametrize("k", [2, 3, 4])
def test_mid_size_products_take_the_multi_launch_form(k):
    """k-pair products with the final exponentiation above 4 096 groups: the Miller half in one launch, final_exp_native in six pieces -- every
    group equals the throughput kernel's value, the `== one` verdicts likewise"""
    import torch
    p = H.pkg()
    n = 4200
    g1, g2 = _dev_pairs(p, n * k, 0xB25400CC + k)
    dev = torch.device("cuda:0")
    try:
        outs = {}
        for thr in (0, 1 << 20):
            p.set_stream_latency(thr, 0, 0, None)
            o = torch.empty(48 * n, dtype=torch.int64, device=dev)
            v = torch.empty(n, dtype=torch.uint8, device=dev)
            p.multi_pairing_batch_dev(g1, g2, o, n, k, do_final_exp=True)
            p.multi_pairing_check_batch_dev(g1, g2, v, n, k)
            p.last_status()
            assert p.last_kernel(0, None) == (1 if thr == 0 else 16)
            outs[thr] = (o, v)
        assert torch.equal(outs[0][0], outs[1 << 20][0]) and torch.equal(outs[0][1], outs[1 << 20][1])
    finally:
        p.set_stream_latency(p.LATENCY_INHERIT, -1, 0, None)


def test_golden_miller_and_final_exp_on_the_latency_kernel(pk):
    vec = H.load_golden("bn254_vectors.json")
    P = [tuple(HX(p)) for p in vec["g1"]]
    Q = [((int(q[0], 16), int(q[1], 16)), (int(q[2], 16), int(q[3], 16))) for q in vec["g2"]]
    n = len(P)
    pk.set_latency_threshold(1 << 20)
    got_m = H.fq12_from_aos(H.to_aos(pk.miller_loop_batch(H.to_soa(H.g1_aos(P), 8), H.to_soa(H.g2_aos(Q), 16), n), 48), n)
    assert got_m == [HX(m) for m in vec["miller"]]
    f = H.to_soa(H.fq12_aos([HX(m) for m in vec["miller"]]), 48)
    assert H.fq12_from_aos(H.to_aos(pk.final_exp_batch(f, n), 48), n) == [HX(p) for p in vec["pairing"]]
    # final_exp_native on arbitrary (non-Miller) Fq12 values
    x = H.rand_fq12(5, seed=11)
    rc, want = H.oracle_final_exp(H.fq12_aos(x), 5)
    assert rc == 0
    assert np.array_equal(H.to_aos(pk.final_exp_batch(H.to_soa(H.fq12_aos(x), 48), 5), 48), want)


@pytest.mark.parametrize("k", [2, 3, 4])
def test_multi_pairing_on_the_latency_kernel(pk, k):
    """multi_miller_loop_native over k pairs (shared f), exact value and with final_exp_native: == throughput kernels on every
    lane, == the C oracle on a few groups; the == one verdict of a Groth16-style product"""
    import torch
    n = 37
    g1, g2 = _dev_pairs(pk, n * k, 0xC0FFEE + k)
    for fe in (False, True):
        a, b = _both(pk, lambda o: pk.multi_pairing_batch_dev(g1, g2, o, n, k, do_final_exp=fe), 48, n)
        assert torch.equal(a, b), (k, fe)
    # oracle, host path
    base_P, base_Q = H.subgroup_points(8)
    ng = 3
    P = [base_P[(g * k + j) % 8] for g in range(ng) for j in range(k)]
    Q = [base_Q[(3 * g + j + 1) % 8] for g in range(ng) for j in range(k)]
    g1a, g2a = H.g1_aos(P), H.g2_aos(Q)
    pk.set_latency_threshold(1 << 20)
    got = H.to_aos(pk.multi_pairing_batch(H.to_soa(g1a, 8), H.to_soa(g2a, 16), ng, k, do_final_exp=False), 48)
    assert np.array_equal(got, H.oracle_multi_miller(g1a, g2a, ng, k))
    got = H.to_aos(pk.multi_pairing_batch(H.to_soa(g1a, 8), H.to_soa(g2a, 16), ng, k, do_final_exp=True), 48)
    assert np.array_equal(got, H.oracle_multi_pairing(g1a, g2a, ng, k))


def test_groth16_style_check_on_the_latency_kernel(pk):
    """e(aP, bQ) e(-abP, Q) e(cP, dQ) e(-cdP, Q) == 1 (final_exp_native.rs:245-263 pattern) and a broken group, verdict bytes"""
    R_ = H.R
    a, b, c, d = 1234567, 7654321, 424242, 99991
    G1, G2 = R_.G1_GEN, R_.G2_GEN
    good = [(R_.g1_mul(G1, a), R_.g2_mul(G2, b)), (R_.g1_neg(R_.g1_mul(G1, a * b % R_.R_ORDER)), G2),
            (R_.g1_mul(G1, c), R_.g2_mul(G2, d)), (R_.g1_neg(R_.g1_mul(G1, c * d % R_.R_ORDER)), G2)]
    bad = list(good)
    bad[2] = (R_.g1_mul(G1, c + 1), good[2][1])
    groups = [good, bad, good]
    P = [p for g in groups for p, _ in g]
    Q = [q for g in groups for _, q in g]
    for thr in (0, 1 << 20):
        pk.set_latency_threshold(thr)
        v = pk.multi_pairing_check_batch(H.to_soa(H.g1_aos(P), 8), H.to_soa(H.g2_aos(Q), 16), 3, 4)
        assert list(v) == [1, 0, 1], thr
