"""The lane-cooperative (latency) kernel of pairing() -- sixteen lanes per pairing, tools/cvm.py -- against the golden vectors,
the C oracle and the throughput kernel: identical limbs on every lane, for batch sizes around the group / wave / grid edges.
Which kernel a call takes is the library's threshold (bn254_set_latency_threshold): the tests pin it per call."""
import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu

HX = lambda xs: [int(x, 16) for x in xs]


@pytest.fixture()
def pk():
    p = H.pkg()
    old = p.get_latency_threshold()
    yield p
    p.set_latency_threshold(old)


def test_threshold_is_settable(pk):
    pk.set_latency_threshold(12345)
    assert pk.get_latency_threshold() == 12345
    pk.set_latency_threshold(0)
    assert pk.get_latency_threshold() == 0


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
