"""Element-major data (the order the reference's callers hold `&[G1Affine]`, `Vec<(&G1Affine, &G2Affine)>`, `Vec<MyFq12>`,
`Vec<Fq12>` in -- src/pairing.rs:20-22, miller_loop_native.rs:324) through the C ABI: the on-device layout kernels against
numpy, the `*_elems` entry points against the limb-major ones, the MyFq12 -> ark Fq12 coefficient order (`.into()` at
pairing.rs:21) against the golden vectors.  Byte / index work: the bar is identical words."""
import numpy as np
import pytest

import helpers as H
from helpers import R

pytestmark = pytest.mark.gpu

HX = lambda xs: [int(x, 16) for x in xs]


def _plane(pk, w, words, order):
    """SoA plane of word w of an element (include/bn254_pairing.h)."""
    if words != 48 or order == pk.FQ12_MYFQ12:
        return w
    return pk.load_library().bn254_myfq12_to_ark_index(w // 4) * 4 + w % 4


def _soa_of(pk, elems, words, n, order):
    e = elems.reshape(n, words)
    soa = np.empty((words, n), dtype=np.uint64)
    for w in range(words):
        soa[_plane(pk, w, words, order)] = e[:, w]
    return soa.reshape(-1)


@pytest.mark.parametrize("words", [8, 16, 48])
def test_layout_kernels_match_numpy(words):
    import torch
    pk = H.pkg()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev)
    rng = np.random.default_rng(words)
    for n in (1, 63, 64, 65, 257, 1000, 70001):
        for order in ((pk.FQ12_MYFQ12, pk.FQ12_ARK) if words == 48 else (pk.FQ12_MYFQ12,)):
            elems = rng.integers(0, 1 << 63, size=words * n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=words * n, dtype=np.uint64)
            want = _soa_of(pk, elems, words, n, order)
            guard = 64                                              # words after the end must stay untouched
            d_e = torch.from_numpy(elems.view(np.int64)).to(dev)
            d_s = torch.full((words * n + guard,), -7, dtype=torch.int64, device=dev)
            pk.soa_from_elems_dev(d_e, d_s, words, n, order, 0, st)
            got = d_s.cpu().numpy().view(np.uint64)
            assert np.array_equal(got[:words * n], want), (words, n, order)
            assert np.all(got[words * n:] == np.uint64(-7 & (2**64 - 1)))
            d_b = torch.full((words * n + guard,), -7, dtype=torch.int64, device=dev)
            pk.soa_to_elems_dev(d_s, d_b, words, n, order, 0, st)
            back = d_b.cpu().numpy().view(np.uint64)
            assert np.array_equal(back[:words * n], elems), (words, n, order)
            assert np.all(back[words * n:] == np.uint64(-7 & (2**64 - 1)))


def test_layout_invalid_arguments():
    import torch
    pk = H.pkg()
    dev = torch.device("cuda:0")
    a = torch.zeros(48 * 4, dtype=torch.int64, device=dev)
    b = torch.zeros(48 * 4, dtype=torch.int64, device=dev)
    for words, order, src, dst in ((12, 0, a, b), (48, 2, a, b), (48, 0, a, a)):
        with pytest.raises(pk.Bn254Error) as e:
            pk.soa_from_elems_dev(src, dst, words, 4, order, 0, None)
        assert e.value.status == pk.ERR_INVALID_ARG
    pk.soa_from_elems_dev(a, b, 48, 0, 0, 0, None)                # empty batch: nothing to do


def test_layout_round_trip_full_size():
    """2^20 Fq12 values (403 MB each way), both coefficient orders: to_elems(from_elems(x)) = x and every plane is a
    permutation-free copy of its element words (column checksums)."""
    import torch
    pk = H.pkg()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev)
    n = 1 << 20
    g = torch.Generator(device=dev)
    g.manual_seed(1234)
    x = torch.randint(-(1 << 62), 1 << 62, (48 * n,), dtype=torch.int64, device=dev, generator=g)
    s = torch.empty_like(x)
    y = torch.empty_like(x)
    for order in (pk.FQ12_MYFQ12, pk.FQ12_ARK):
        pk.soa_from_elems_dev(x, s, 48, n, order, 0, st)
        pk.soa_to_elems_dev(s, y, 48, n, order, 0, st)
        assert torch.equal(x, y)
        col = x.view(n, 48).sum(dim=0)                              # wrap-around sums: order-independent checksums
        planes = s.view(48, n).sum(dim=1)
        perm = torch.tensor([_plane(pk, w, 48, order) for w in range(48)], device=dev)
        assert torch.equal(planes[perm], col)


def test_elems_entry_points_match_limb_major_and_golden():
    pk = H.pkg()
    vec = H.load_golden("bn254_vectors.json")
    P = [tuple(HX(p)) for p in vec["g1"]]
    Q = [((int(q[0], 16), int(q[1], 16)), (int(q[2], 16), int(q[3], 16))) for q in vec["g2"]]
    n = len(P)
    e1, e2 = H.g1_aos(P), H.g2_aos(Q)
    # pairing(): MyFq12 order and ark order (the value pairing.rs:20-22 returns) against the golden vectors
    got = H.fq12_from_aos(pk.pairing_batch_elems(e1, e2, n), n)
    assert got == [HX(p) for p in vec["pairing"]]
    got = H.fq12_from_aos(pk.pairing_batch_elems(e1, e2, n, out_order=pk.FQ12_ARK), n)
    assert got == [R.myfq12_to_ark(HX(p)) for p in vec["pairing"]]
    assert H.fq12_from_aos(pk.miller_loop_batch_elems(e1, e2, n), n) == [HX(m) for m in vec["miller"]]
    # final_exp_native on element-major Fq12 in either order
    f_my = H.fq12_aos([HX(m) for m in vec["fq12_in"]])
    f_ark = H.fq12_aos([R.myfq12_to_ark(HX(m)) for m in vec["fq12_in"]])
    m = len(vec["fq12_in"])
    want = [HX(p) for p in vec["final_exp"]]
    assert H.fq12_from_aos(pk.final_exp_batch_elems(f_my, m), m) == want
    assert H.fq12_from_aos(pk.final_exp_batch_elems(f_ark, m, in_order=pk.FQ12_ARK), m) == want
    assert H.fq12_from_aos(pk.final_exp_batch_elems(f_ark, m, in_order=pk.FQ12_ARK, out_order=pk.FQ12_ARK), m) == [R.myfq12_to_ark(w) for w in want]
    # groups of pairs (multi_miller_loop_native, miller_loop_native.rs:324)
    for g in vec["groups"]:
        k = g["k"]
        ge1, ge2 = H.g1_aos([P[i] for i in g["idx"]]), H.g2_aos([Q[i] for i in g["idx"]])
        assert H.fq12_from_aos(pk.multi_pairing_batch_elems(ge1, ge2, 1, k, do_final_exp=False), 1)[0] == HX(g["miller"])


def test_elems_ragged_batch_vs_limb_major():
    """300 pairs / 100 groups of 3: element-major entry points = limb-major entry points, word for word."""
    pk = H.pkg()
    n = 300
    Ps, Qs = H.subgroup_points(24, seed=77)
    P = [Ps[i % 24] for i in range(n)]
    Q = [Qs[(7 * i + i // 24) % 24] for i in range(n)]
    e1, e2 = H.g1_aos(P), H.g2_aos(Q)
    s1, s2 = H.to_soa(e1, 8), H.to_soa(e2, 16)
    assert np.array_equal(pk.pairing_batch_elems(e1, e2, n), H.to_aos(pk.pairing_batch(s1, s2, n), 48))
    assert np.array_equal(pk.miller_loop_batch_elems(e1, e2, n), H.to_aos(pk.miller_loop_batch(s1, s2, n), 48))
    for fe in (True, False):
        assert np.array_equal(pk.multi_pairing_batch_elems(e1, e2, 100, 3, do_final_exp=fe),
                              H.to_aos(pk.multi_pairing_batch(s1, s2, 100, 3, do_final_exp=fe), 48))
    assert np.array_equal(pk.multi_pairing_check_batch_elems(e1, e2, 100, 3), pk.multi_pairing_check_batch(s1, s2, 100, 3))
    assert pk.pairing_batch_elems(e1[:0], e2[:0], 0).size == 0


def test_elems_product_check_verdicts():
    """e(aP, Q) e(-P, aQ) = 1 (final_exp_native.rs:245-263) on element-major pairs; a broken group gives 0."""
    pk = H.pkg()
    P, Q = R.G1_GEN, R.G2_GEN
    pts1, pts2 = [], []
    for a in (3, 0x1234567, 2**100 + 9):
        pts1 += [R.g1_mul(P, a), R.g1_neg(P)]
        pts2 += [Q, R.g2_mul(Q, a)]
    pts1 += [R.g1_mul(P, 5), P]
    pts2 += [Q, R.g2_mul(Q, 5)]
    v = pk.multi_pairing_check_batch_elems(H.g1_aos(pts1), H.g2_aos(pts2), 4, 2)
    assert v.tolist() == [1, 1, 1, 0]


def test_elems_pipeline_large_batch():
    """Above 2^16 lanes the element-major host entry point runs the chunked two-worker pipeline (contiguous chunk copies,
    planes made on the device): same words as one device launch, ragged tail included, ark order on the way out."""
    import torch
    pk = H.pkg()
    n = (1 << 17) * 2 + 300
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev)
    g1 = torch.zeros(8 * n, dtype=torch.int64, device=dev)
    g2 = torch.zeros(16 * n, dtype=torch.int64, device=dev)
    out = torch.zeros(48 * n, dtype=torch.int64, device=dev)
    pk.generate_pairs_dev(0xB2540005, g1, g2, n, 0, st)
    pk.pairing_batch_dev(g1, g2, out, n, 0, st)
    pk.last_status(0, st)
    e1 = H.to_aos(g1.cpu().numpy().view(np.uint64), 8)
    e2 = H.to_aos(g2.cpu().numpy().view(np.uint64), 16)
    want = H.to_aos(out.cpu().numpy().view(np.uint64), 48).reshape(n, 12, 4)
    got = pk.pairing_batch_elems(e1, e2, n, out_order=pk.FQ12_ARK).reshape(n, 12, 4)
    idx = [pk.load_library().bn254_myfq12_to_ark_index(j) for j in range(12)]
    assert np.array_equal(got, want[:, idx, :])
    k = 2
    groups = n // k
    og = torch.zeros(48 * groups, dtype=torch.int64, device=dev)
    sel = lambda t, planes: t.view(planes, n)[:, :groups * k].contiguous().view(-1)
    pk.multi_pairing_batch_dev(sel(g1, 8), sel(g2, 16), og, groups, k, True, 0, st)
    pk.last_status(0, st)
    got = pk.multi_pairing_batch_elems(e1[:8 * groups * k], e2[:16 * groups * k], groups, k)
    assert np.array_equal(got, H.to_aos(og.cpu().numpy().view(np.uint64), 48))


@pytest.mark.parametrize("threshold", [0, 1 << 20])
def test_elems_direct_kernel_io_equals_the_transposition_route(threshold):
    """The throughput kernels read and write element-major arrays themselves (I/O mode bits of their k argument, tools/kgen4_prog.py:
    io_walk_begin): with the latency path switched off for the NULL stream every `_elems` call below launches k_pairing / k_miller / k_fexp /
    k_mpairing / k_mmiller straight on the element arrays; with it switched on the same calls take the lane-cooperative programs behind the
    transposition kernels.  Both give the limb-major entry points' words: ragged sizes, groups of 1 - 3 and 5 pairs, both Fq12 orders."""
    pk = H.pkg()
    pk.set_stream_latency(threshold, -1, 0, None)
    try:
        n = 301
        Ps, Qs = H.subgroup_points(24, seed=78)
        P = [Ps[i % 24] for i in range(n)]
        Q = [Qs[(5 * i + i // 24) % 24] for i in range(n)]
        e1, e2 = H.g1_aos(P), H.g2_aos(Q)
        s1, s2 = H.to_soa(e1, 8), H.to_soa(e2, 16)
        idx = [pk.load_library().bn254_myfq12_to_ark_index(j) for j in range(12)]
        want = H.to_aos(pk.pairing_batch(s1, s2, n), 48)
        assert np.array_equal(pk.pairing_batch_elems(e1, e2, n), want)
        assert np.array_equal(pk.pairing_batch_elems(e1, e2, n, out_order=pk.FQ12_ARK).reshape(n, 12, 4), want.reshape(n, 12, 4)[:, idx, :])
        mil = H.to_aos(pk.miller_loop_batch(s1, s2, n), 48)
        assert np.array_equal(pk.miller_loop_batch_elems(e1, e2, n), mil)
        for k in (2, 3, 5):
            g = n // k
            a1, a2 = e1[: 8 * g * k], e2[: 16 * g * k]
            b1, b2 = H.to_soa(a1, 8), H.to_soa(a2, 16)
            for fe in (True, False):
                w = H.to_aos(pk.multi_pairing_batch(b1, b2, g, k, do_final_exp=fe), 48)
                assert np.array_equal(pk.multi_pairing_batch_elems(a1, a2, g, k, do_final_exp=fe), w), (k, fe)
                assert np.array_equal(pk.multi_pairing_batch_elems(a1, a2, g, k, do_final_exp=fe, out_order=pk.FQ12_ARK).reshape(g, 12, 4),
                                      w.reshape(g, 12, 4)[:, idx, :]), (k, fe)
        fe_want = H.to_aos(pk.final_exp_batch(H.to_soa(mil, 48), n), 48)
        assert np.array_equal(pk.final_exp_batch_elems(mil, n), fe_want)
        assert np.array_equal(pk.final_exp_batch_elems(mil, n, out_order=pk.FQ12_ARK).reshape(n, 12, 4), fe_want.reshape(n, 12, 4)[:, idx, :])
        ark_in = mil.reshape(n, 12, 4)[:, idx, :].reshape(-1).copy()
        assert np.array_equal(pk.final_exp_batch_elems(ark_in, n, in_order=pk.FQ12_ARK), fe_want)
    finally:
        pk.set_stream_latency(pk.LATENCY_INHERIT, -1, 0, None)


def test_elems_dev_entry_points():
    """bn254_pairing_batch_elems_dev / bn254_multi_pairing_batch_elems_dev on device-resident element-major arrays: a 2^16 + 77 batch (the
    throughput kernel, straight on the element arrays) and a 500-item one (the lane-cooperative programs behind the transposition kernels)
    give the words of the limb-major `_dev` launch, in either Fq12 order; groups of two pairs likewise; words behind the result stay untouched."""
    import torch
    pk = H.pkg()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev)
    idx = [pk.load_library().bn254_myfq12_to_ark_index(j) for j in range(12)]
    for n in ((1 << 16) + 77, 500):
        g1 = torch.zeros(8 * n, dtype=torch.int64, device=dev)
        g2 = torch.zeros(16 * n, dtype=torch.int64, device=dev)
        out = torch.zeros(48 * n, dtype=torch.int64, device=dev)
        pk.generate_pairs_dev(0xB2540009 + n, g1, g2, n, 0, st)
        pk.pairing_batch_dev(g1, g2, out, n, 0, st)
        e1, e2 = torch.empty_like(g1), torch.empty_like(g2)
        pk.soa_to_elems_dev(g1, e1, 8, n, 0, 0, st)
        pk.soa_to_elems_dev(g2, e2, 16, n, 0, 0, st)
        want = out.view(48, n).t().contiguous().view(n, 12, 4)
        for order in (pk.FQ12_MYFQ12, pk.FQ12_ARK):
            eo = torch.full((48 * n + 64,), -7, dtype=torch.int64, device=dev)
            pk.pairing_batch_elems_dev(e1, e2, eo, n, order, 0, st)
            pk.last_status(0, st)
            got = eo[: 48 * n].view(n, 12, 4)
            assert torch.equal(got, want if order == pk.FQ12_MYFQ12 else want[:, idx, :]), (n, order)
            assert bool((eo[48 * n:] == -7).all())
        k, g = 2, n // 2
        og = torch.zeros(48 * g, dtype=torch.int64, device=dev)
        sel = lambda t, planes: t.view(planes, n)[:, : g * k].contiguous().view(-1)
        pk.multi_pairing_batch_dev(sel(g1, 8), sel(g2, 16), og, g, k, False, 0, st)
        eg = torch.zeros(48 * g, dtype=torch.int64, device=dev)
        pk.multi_pairing_batch_elems_dev(e1[: 8 * g * k], e2[: 16 * g * k], eg, g, k, False, pk.FQ12_MYFQ12, 0, st)
        pk.last_status(0, st)
        assert torch.equal(eg.view(g, 48), og.view(48, g).t().contiguous())
