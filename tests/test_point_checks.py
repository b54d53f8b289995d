"""bn254_check_points_ex (round 5): the reference's IMPLICIT input contract.

`twisted_frobenius` / `neg_twisted_frobenius` build their results with `G2Affine::new(out_x, out_y)`
(/root/reference/src/miller_loop_native.rs:303,311); ark-ec's `Affine::new` asserts on-curve and in-subgroup, so the reference
PANICS on a G2 point outside the r-torsion, and `G1Affine::rand` / `G2Affine::rand` (src/pairing.rs:65-66) never produce one.  The
engine computes a value for any coordinates; callers with untrusted points run the optional check.

CPU: the one-scalar-multiplication criterion the kernel uses ([x+1]Q + psi([x]Q) + psi^2([x]Q) == psi^3([2x]Q), ePrint 2022/348)
agrees with the definition [r]Q == O on subgroup points, random twist points and a cofactor-cleared point (big-int restatement).
GPU: every flag, precedence, the per-point bytes, and the documented async flow (check, compute, ONE status read at the end)."""
import os
import random

import numpy as np
import pytest

import helpers as H

R = H.R
P = R.P


def _psi(Q):
    c2, c3 = R._end_constants()
    return R.twisted_frobenius(Q, c2, c3)


def _sqrt_fq2(a):
    a0, a1 = a
    if a1 == 0:
        if pow(a0, (P - 1) // 2, P) == 1:
            return (pow(a0, (P + 1) // 4, P), 0)
        return (0, pow(-a0 % P, (P + 1) // 4, P))
    n = (a0 * a0 + a1 * a1) % P
    if pow(n, (P - 1) // 2, P) != 1:
        return None
    s = pow(n, (P + 1) // 4, P)
    for sg in (s, -s % P):
        t = (a0 + sg) * pow(2, -1, P) % P
        if pow(t, (P - 1) // 2, P) == 1:
            x0 = pow(t, (P + 1) // 4, P)
            x1 = a1 * pow(2 * x0, -1, P) % P
            if R.fq2_mul((x0, x1), (x0, x1)) == (a0 % P, a1 % P):
                return (x0, x1)
    return None


def twist_point(rng):
    """a random point of E'(Fp2): y^2 = x^3 + 3/xi -- in the r-torsion with probability ~ 2^-254 (cofactor 2p - r)"""
    while True:
        x = (rng.randrange(P), rng.randrange(P))
        y = _sqrt_fq2(R.fq2_add(R.fq2_mul(R.fq2_mul(x, x), x), R.TWIST_B))
        if y is not None:
            assert R.g2_on_curve((x, y))
            return (x, y)


def criterion(Q):
    x = R.BN_X
    a = R.g2_mul(Q, x)
    b = _psi(a)
    lhs = R.g2_add(R.g2_add(_psi(b), b), R.g2_add(a, Q))
    return lhs == _psi(_psi(_psi(R.g2_mul(Q, 2 * x))))


def test_subgroup_criterion_equals_the_definition():
    rng = random.Random(5)
    for _ in range(3):
        Q = R.g2_mul(R.G2_GEN, rng.randrange(1, R.R_ORDER))
        assert R.g2_mul(Q, R.R_ORDER) is None and criterion(Q)
    for _ in range(4):
        Q = twist_point(rng)
        assert (R.g2_mul(Q, R.R_ORDER) is None) == criterion(Q) == False      # noqa: E712
    Q = R.g2_mul(twist_point(rng), 2 * P - R.R_ORDER)                        # cofactor cleared: in the subgroup
    assert R.g2_mul(Q, R.R_ORDER) is None and criterion(Q)


def _batch(Ps, Qs):
    pk = H.pkg()
    return pk.layout.to_soa(H.g1_aos(Ps), 8), pk.layout.to_soa(H.g2_aos(Qs), 16)


@pytest.mark.gpu
def test_check_points_ex_every_verdict():
    pk = H.pkg()
    rng = random.Random(77)
    n = 70
    Ps, Qs = H.subgroup_points(n, seed=91)
    Ps, Qs = list(Ps), list(Qs)
    g1, g2 = _batch(Ps, Qs)
    ALL = pk.CHECK_INFINITY | pk.CHECK_ON_CURVE | pk.CHECK_SUBGROUP
    rc, per = pk.check_points_ex(g1, g2, n, ALL, want_per_point=True)
    assert rc == 0 and not per.any()
    pk.check_points_ex(g1, g2, n)                                        # raising form: nothing to raise
    # a twist point outside the r-torsion at lane 13, one more at the last lane
    bad = list(Qs)
    bad[13] = twist_point(rng)
    bad[n - 1] = twist_point(rng)
    g1b, g2b = _batch(Ps, bad)
    rc, per = pk.check_points_ex(g1b, g2b, n, ALL, want_per_point=True)
    assert rc == pk.ERR_NOT_IN_SUBGROUP
    assert [i for i in range(n) if per[i]] == [13, n - 1] and per[13] == pk.PT_NOT_IN_SUBGROUP
    with pytest.raises(pk.Bn254Error) as e:
        pk.check_points_ex(g1b, g2b, n)
    assert e.value.status == pk.ERR_NOT_IN_SUBGROUP
    # ... which the on-curve check alone accepts (it IS on the twist)
    rc, per = pk.check_points_ex(g1b, g2b, n, pk.CHECK_ON_CURVE, want_per_point=True)
    assert rc == 0 and not per.any()
    # off the curve: G2 (y + 1), G1 (x + 1), and a coordinate that is not below p
    q = Qs[5]
    offq = list(Qs)
    offq[5] = (q[0], ((q[1][0] + 1) % P, q[1][1]))
    offp = list(Ps)
    offp[64] = ((Ps[64][0] + 1) % P, Ps[64][1])
    g1c, g2c = _batch(offp, offq)
    rc, per = pk.check_points_ex(g1c, g2c, n, ALL, want_per_point=True)
    assert rc == pk.ERR_NOT_ON_CURVE and [i for i in range(n) if per[i]] == [5, 64] and per[5] == per[64] == pk.PT_NOT_ON_CURVE
    g1d = g1.copy()
    xm = sum(int(w) << (64 * l) for l, w in enumerate(g1d.reshape(8, n)[:4, 2]))
    g1d.reshape(8, n)[:4, 2] = np.array(R.limbs4(xm + P), dtype=np.uint64)   # x + p: same residue, non-canonical limbs (x + p < 2^255)
    rc, per = pk.check_points_ex(g1d, g2, n, pk.CHECK_ON_CURVE, want_per_point=True)
    assert rc == pk.ERR_NOT_ON_CURVE and [i for i in range(n) if per[i]] == [2]
    # infinity (ark's affine identity x = y = 0) wins over everything else in the batch
    g2e = g2b.copy()
    g2e.reshape(16, n)[:, 40] = 0
    rc, per = pk.check_points_ex(g1c, g2e, n, ALL, want_per_point=True)
    assert rc == pk.ERR_INFINITY and per[40] == pk.PT_INFINITY and per[13] == pk.PT_NOT_IN_SUBGROUP and per[64] == pk.PT_NOT_ON_CURVE
    # without the infinity flag an all-zero point is simply skipped by the curve checks
    rc, per = pk.check_points_ex(g1, g2e, n, pk.CHECK_SUBGROUP, want_per_point=True)
    assert rc == pk.ERR_NOT_IN_SUBGROUP and per[40] == 0
    # e(G1, G2)'s own generator and the golden vectors' points pass
    vec = H.load_golden("bn254_vectors.json")
    Pg = [tuple(int(x, 16) for x in v) for v in vec["g1"]]
    Qg = [((int(v[0], 16), int(v[1], 16)), (int(v[2], 16), int(v[3], 16))) for v in vec["g2"]]
    a, b = _batch(Pg, Qg)
    pk.check_points_ex(a, b, len(Pg), ALL)


@pytest.mark.gpu
def test_generated_batch_is_in_the_subgroup_and_the_engine_agrees_with_the_oracle_there():
    """2^14 on-device generated pairs all pass the full check (per-point bytes all zero); the pairing of those points is the oracle's."""
    import torch
    pk = H.pkg()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev)
    n = 1 << 14
    g1 = torch.zeros(8 * n, dtype=torch.int64, device=dev)
    g2 = torch.zeros(16 * n, dtype=torch.int64, device=dev)
    per = torch.full((n,), 255, dtype=torch.uint8, device=dev)
    pk.generate_pairs_dev(0xB2540077, g1, g2, n, 0, st)
    pk.check_points_ex_dev(g1, g2, n, 7, per, 0, st)
    out = torch.zeros(48 * n, dtype=torch.int64, device=dev)
    pk.pairing_batch_dev(g1, g2, out, n, 0, st)
    pk.last_status(0, st)
    assert int(per.sum()) == 0
    pos = [0, 1, n // 2, n - 1]
    idx = torch.as_tensor(pos, device=dev)
    g1h = g1.view(8, n)[:, idx].cpu().numpy().view(np.uint64).reshape(-1).copy()
    g2h = g2.view(16, n)[:, idx].cpu().numpy().view(np.uint64).reshape(-1).copy()
    got = out.view(48, n)[:, idx].cpu().numpy().view(np.uint64).reshape(-1).copy()
    want = H.oracle_pairing(pk.layout.to_aos(g1h, 8), pk.layout.to_aos(g2h, 16), len(pos), threads=4)
    assert np.array_equal(pk.layout.to_aos(got, 48), want)


@pytest.mark.gpu
@pytest.mark.parametrize("threshold", [0, 1 << 20])
def test_async_flow_keeps_the_point_verdict(threshold):
    """The documented asynchronous flow -- check_points_dev, pairing_batch_dev, ONE bn254_last_status at the end -- on a batch with an
    all-zero pair: the pairing kernels trip their zero-divisor flag on that lane, and the status read must still say INFINITY (round 4
    kept both flags in one word and the kernels' plain store overwrote the check's verdict).  Both kernel families."""
    import torch
    pk = H.pkg()
    dev = torch.device("cuda:0")
    st = torch.cuda.Stream(dev)
    n = 64
    with torch.cuda.stream(st):
        g1 = torch.zeros(8 * n, dtype=torch.int64, device=dev)
        g2 = torch.zeros(16 * n, dtype=torch.int64, device=dev)
        out = torch.zeros(48 * n, dtype=torch.int64, device=dev)
        pk.generate_pairs_dev(0xB2540078, g1, g2, n, 0, st)
        pk.last_status(0, st)
        g1.view(8, n)[:, 7] = 0
        g2.view(16, n)[:, 7] = 0
        pk.set_stream_latency(threshold, -1, 0, st)
        pk.check_points_dev(g1, g2, n, 0, st)
        pk.pairing_batch_dev(g1, g2, out, n, 0, st)
        with pytest.raises(pk.Bn254Error) as e:
            pk.last_status(0, st)
        assert e.value.status == pk.ERR_INFINITY
        # without the check the same launch reports what the kernels themselves see
        pk.pairing_batch_dev(g1, g2, out, n, 0, st)
        with pytest.raises(pk.Bn254Error) as e:
            pk.last_status(0, st)
        assert e.value.status == pk.ERR_ZERO_DIVISOR
        pk.last_status(0, st)                           # sticky words were cleared
        # full check + compute on a non-subgroup point: NOT_IN_SUBGROUP at the end, whatever the pairing kernel made of it
        rng = random.Random(3)
        Q = twist_point(rng)
        w = torch.from_numpy(H.g2_aos([Q]).view(np.int64).copy()).to(dev)
        g1b = g1.clone()
        g2b = g2.clone()
        g1b.view(8, n)[:, 7] = g1b.view(8, n)[:, 8]
        g2b.view(16, n)[:, 7] = w
        pk.check_points_ex_dev(g1b, g2b, n, 7, None, 0, st)
        pk.pairing_batch_dev(g1b, g2b, out, n, 0, st)
        with pytest.raises(pk.Bn254Error) as e:
            pk.last_status(0, st)
        assert e.value.status == pk.ERR_NOT_IN_SUBGROUP
    pk.release_stream(0, st)


@pytest.mark.gpu
def test_two_threads_two_streams_two_kernel_selections():
    """Kernel selection per (device, stream): thread A wants the throughput kernel for its 300-item batches, thread B the
    lane-cooperative kernel for the same size, concurrently, neither touching the process-wide default; bn254_last_kernel says which
    family each launch took, and all results are the same limbs."""
    import threading
    import torch
    pk = H.pkg()
    dev = torch.device("cuda:0")
    n = 300
    g1 = torch.zeros(8 * n, dtype=torch.int64, device=dev)
    g2 = torch.zeros(16 * n, dtype=torch.int64, device=dev)
    st0 = torch.cuda.current_stream(dev)
    pk.generate_pairs_dev(0xB2540079, g1, g2, n, 0, st0)
    pk.last_status(0, st0)
    keep = (pk.get_latency_threshold(), pk.get_latency_lanes())
    streams = [torch.cuda.Stream(dev) for _ in range(3)]
    outs = [torch.zeros(48 * n, dtype=torch.int64, device=dev) for _ in range(3)]
    want_kernel = [1, 16, 32]
    pk.set_stream_latency(0, -1, 0, streams[0])              # never the latency kernel
    pk.set_stream_latency(1 << 20, 16, 0, streams[1])        # always, sixteen lanes per item
    pk.set_stream_latency(1 << 62, 32, 0, streams[2])        # always (a threshold whose product with 1000 overflows 64 bits), thirty-two
    seen = [[] for _ in range(3)]
    errs = []

    def work(i):
        try:
            for _ in range(6):
                pk.pairing_batch_dev(g1, g2, outs[i], n, 0, streams[i])
                seen[i].append(pk.last_kernel(0, streams[i]))
            pk.last_status(0, streams[i])
        except Exception as e:      # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=work, args=(i,)) for i in range(3)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for i in range(3):
        assert seen[i] == [want_kernel[i]] * 6, (i, seen[i])
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert (pk.get_latency_threshold(), pk.get_latency_lanes()) == keep
    # back to the defaults: the stream follows the process-wide setting again
    pk.set_stream_latency(pk.LATENCY_INHERIT, -1, 0, streams[0])
    pk.pairing_batch_dev(g1, g2, outs[0], n, 0, streams[0])
    pk.last_status(0, streams[0])
    assert pk.last_kernel(0, streams[0]) != 1 if keep[0] >= n else pk.last_kernel(0, streams[0]) == 1
    for s in streams:
        pk.release_stream(0, s)


def test_subgroup_check_walks_the_naf_of_x():
    """csrc/bn254_point_checks.h computes [x]Q by the non-adjacent form of BN_X: its two digit masks are that number's NAF"""
    import re
    txt = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "plonky2-bn254-pairing_amd", "csrc", "bn254_point_checks.h")).read()
    nz, neg = (int(x, 16) for x in re.search(r"X_NZ = (0x[0-9a-f]+)ull, X_NEG = (0x[0-9a-f]+)ull", txt).groups())
    assert neg & ~nz == 0 and nz >> 62 == 1 and not (neg >> 62) & 1          # 63 digits, the top one is +1 (the loop starts from Q)
    assert sum((-1 if (neg >> i) & 1 else 1) << i for i in range(63) if (nz >> i) & 1) == R.BN_X
    assert nz & (nz >> 1) == 0 and bin(nz).count("1") == 24                  # non-adjacent, 24 digits


@pytest.mark.gpu
def test_generated_subgroup_kernel_equals_the_portable_kernel_point_by_point():
    """The subgroup criterion runs on the generated kernel k_subcheck (tools/kgen4_prog.py: Jacobian steps without the group law's exceptional
    branches, a final Z = 0 counted as a failure); BN254_CHECK_SUBGROUP_PORTABLE keeps it on the HIP C++ kernel that spells every exceptional
    case out.  Same per-point bytes and the same status on a ragged batch that mixes r-torsion points, random twist points, cofactor-cleared
    twist points, points off the twist, non-canonical coordinates and the point at infinity -- under every flag combination that asks for
    the subgroup check -- and both say what the big-int definition [r]Q == O says."""
    pk = H.pkg()
    rng = random.Random(2025)
    n = 333
    Ps, Qs = H.subgroup_points(n, seed=93)
    Ps, Qs = list(Ps), list(Qs)
    expect = [0] * n
    for i in range(0, n, 7):                                   # random twist points: not in the subgroup
        Qs[i] = twist_point(rng)
        expect[i] = pk.PT_NOT_IN_SUBGROUP
    for i in (3, 150, n - 1):                                  # cofactor-cleared: in the subgroup again
        Qs[i] = R.g2_mul(twist_point(rng), 2 * P - R.R_ORDER)
    for i in (5, 200):                                         # off the twist
        q = Qs[i]
        Qs[i] = (q[0], ((q[1][0] + 1) % P, q[1][1]))
        expect[i] = pk.PT_NOT_ON_CURVE
    g1, g2 = _batch(Ps, Qs)
    g2.reshape(16, n)[:, 41] = 0                               # the point at infinity (ark's x = y = 0)
    expect[41] = pk.PT_INFINITY
    for i in range(n):
        if expect[i] in (0, pk.PT_NOT_IN_SUBGROUP) and i != 41:
            assert (R.g2_mul(Qs[i], R.R_ORDER) is None) == (expect[i] == 0), i
    ALL = pk.CHECK_INFINITY | pk.CHECK_ON_CURVE | pk.CHECK_SUBGROUP
    for flags in (ALL, pk.CHECK_SUBGROUP, pk.CHECK_SUBGROUP | pk.CHECK_ON_CURVE, pk.CHECK_SUBGROUP | pk.CHECK_INFINITY):
        rc_g, per_g = pk.check_points_ex(g1, g2, n, flags, want_per_point=True)
        rc_p, per_p = pk.check_points_ex(g1, g2, n, flags | pk.CHECK_SUBGROUP_PORTABLE, want_per_point=True)
        assert rc_g == rc_p and np.array_equal(per_g, per_p), flags
        want = [e if (e != pk.PT_INFINITY or (flags & pk.CHECK_INFINITY)) else 0 for e in expect]
        assert list(per_g) == want, flags
    # the portable flag alone selects the subgroup check too
    rc, per = pk.check_points_ex(g1, g2, n, pk.CHECK_SUBGROUP_PORTABLE, want_per_point=True)
    assert rc == pk.ERR_NOT_ON_CURVE and list(per) == [e if e != pk.PT_INFINITY else 0 for e in expect]
