"""Fixed G2 points (round 6): `multi_miller_loop_native` (/root/reference/src/miller_loop_native.rs:192-282, :324-326) with some of the G2 points the
same for every group of the batch -- a Groth16 verifier's beta, gamma, delta.  `bn254_g2_lines_dev` makes the line table of the fixed points once;
`bn254_pairing_fixed_g2_batch_dev` must give, limb for limb, what `bn254_multi_pairing_batch_dev(do_final_exp = 1)` gives on the expanded pairs
(and the oracle's final_exp_native(multi_miller_loop_native(...)) on spot-checked groups)."""
import numpy as np
import pytest

import helpers as H
from helpers import R

pytestmark = pytest.mark.gpu


def _setup(pk, torch, dev, st, n, kf, seed):
    k = 1 + kf
    g1 = torch.zeros(8 * n * k, dtype=torch.int64, device=dev)
    g2all = torch.zeros(16 * n * k, dtype=torch.int64, device=dev)
    pk.generate_pairs_dev(seed, g1, g2all, n * k, 0, st)
    f1 = torch.zeros(8 * kf, dtype=torch.int64, device=dev)
    g2fix = torch.zeros(16 * kf, dtype=torch.int64, device=dev)
    pk.generate_pairs_dev(seed ^ 0x5555, f1, g2fix, kf, 0, st)
    # the group's own Q = the generated Q of its first pair; the expanded batch has the fixed points in the other places
    g2var = g2all.view(16, n, k)[:, :, 0].contiguous().view(-1)
    exp = g2all.view(16, n, k).clone()
    for j in range(kf):
        exp[:, :, 1 + j] = g2fix.view(16, kf)[:, j:j + 1]
    table = torch.zeros(pk.g2_lines_bytes(kf) // 8, dtype=torch.int64, device=dev)
    pk.g2_lines_dev(g2fix, kf, table, 0, st)
    return g1, g2var, exp.contiguous().view(-1), table


@pytest.mark.parametrize("kf", [1, 2, 3, 4])
def test_fixed_g2_equals_the_expanded_multi_pairing_on_every_lane(kf):
    import torch
    pk = H.pkg()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev)
    pk.set_stream_latency(0, -1, 0, st)          # the reference value from the throughput kernel as well (any size)
    try:
        n, k = 1000 + 77 * kf, 1 + kf
        g1, g2var, g2exp, table = _setup(pk, torch, dev, st, n, kf, 0xF1D0 + kf)
        want = torch.zeros(48 * n, dtype=torch.int64, device=dev)
        pk.multi_pairing_batch_dev(g1, g2exp, want, n, k, True, 0, st)
        got = torch.full((48 * n + 64,), -7, dtype=torch.int64, device=dev)
        pk.pairing_fixed_g2_batch_dev(g1, g2var, table, kf, got, n, 0, st)
        pk.last_status(0, st)
        assert torch.equal(got[: 48 * n], want) and bool((got[48 * n:] == -7).all()) and int(want.abs().sum()) != 0
        # element-major in, ark order out
        e1 = torch.empty_like(g1)
        e2 = torch.empty_like(g2var)
        pk.soa_to_elems_dev(g1, e1, 8, n * k, 0, 0, st)
        pk.soa_to_elems_dev(g2var, e2, 16, n, 0, 0, st)
        eo = torch.zeros(48 * n, dtype=torch.int64, device=dev)
        pk.pairing_fixed_g2_batch_elems_dev(e1, e2, table, kf, eo, n, pk.FQ12_ARK, 0, st)
        pk.last_status(0, st)
        idx = [pk.load_library().bn254_myfq12_to_ark_index(j) for j in range(12)]
        assert torch.equal(eo.view(n, 12, 4), want.view(48, n).t().contiguous().view(n, 12, 4)[:, idx, :])
        # the oracle on three groups
        pos = [0, n // 2, n - 1]
        sel = torch.as_tensor([p * k + j for p in pos for j in range(k)], device=dev)
        g1h = g1.view(8, n * k)[:, sel].cpu().numpy().view(np.uint64).reshape(-1).copy()
        g2h = g2exp.view(16, n * k)[:, sel].cpu().numpy().view(np.uint64).reshape(-1).copy()
        ora = H.oracle_multi_pairing(pk.layout.to_aos(g1h, 8), pk.layout.to_aos(g2h, 16), len(pos), k)
        mine = got[: 48 * n].view(48, n)[:, torch.as_tensor(pos, device=dev)].cpu().numpy().view(np.uint64).reshape(-1).copy()
        assert np.array_equal(pk.layout.to_aos(mine, 48), ora)
    finally:
        pk.set_stream_latency(pk.LATENCY_INHERIT, -1, 0, st)


def test_fixed_g2_full_grid_and_verdicts():
    """2^16 + 300 groups of 1 + 3 pairs (the Groth16 shape with a fixed verifying key) against k_mpairing on the expanded batch; the verdict bytes
    on crafted groups: e([6]G1, G2) e(-[2]G1, [3]G2) == 1 with [3]G2 fixed, a generic group != 1."""
    import torch
    pk = H.pkg()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev)
    n, kf = (1 << 16) + 300, 3
    g1, g2var, g2exp, table = _setup(pk, torch, dev, st, n, kf, 0xF1DF)
    want = torch.zeros(48 * n, dtype=torch.int64, device=dev)
    pk.multi_pairing_batch_dev(g1, g2exp, want, n, 1 + kf, True, 0, st)
    got = torch.zeros(48 * n, dtype=torch.int64, device=dev)
    pk.pairing_fixed_g2_batch_dev(g1, g2var, table, kf, got, n, 0, st)
    pk.last_status(0, st)
    assert torch.equal(got, want)
    # verdicts
    G1, G2 = R.G1_GEN, R.G2_GEN
    Qf = R.g2_mul(G2, 3)
    grp = [(R.g1_mul(G1, 6), R.g1_neg(R.g1_mul(G1, 2))), (R.g1_mul(G1, 6), R.g1_mul(G1, 2)), (R.g1_mul(G1, 9), R.g1_neg(R.g1_mul(G1, 3)))]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64).copy()).to(dev)
    g1c = t(pk.layout.to_soa(H.g1_aos([p for g in grp for p in g]), 8))
    g2c = t(pk.layout.to_soa(H.g2_aos([G2] * len(grp)), 16))
    qf = t(pk.layout.to_soa(H.g2_aos([Qf]), 16))
    tab = torch.zeros(pk.g2_lines_bytes(1) // 8, dtype=torch.int64, device=dev)
    pk.g2_lines_dev(qf, 1, tab, 0, st)
    v = torch.full((len(grp),), 9, dtype=torch.uint8, device=dev)
    pk.pairing_fixed_g2_check_batch_dev(g1c, g2c, tab, 1, v, len(grp), 0, st)
    pk.last_status(0, st)
    assert v.tolist() == [1, 0, 1]
    # ... against a target the verifier holds (e(alpha, beta) of a verifying key): e(P0, G2) e(P1, 3 G2) == e(4 G1, G2)
    target = pk.pairing_batch(pk.layout.to_soa(H.g1_aos([R.g1_mul(G1, 4)]), 8), pk.layout.to_soa(H.g2_aos([G2]), 16), 1)
    grp = [(R.g1_mul(G1, 7), R.g1_neg(G1)), (R.g1_mul(G1, 6), R.g1_neg(R.g1_mul(G1, 2))), (R.g1_mul(G1, 10), R.g1_neg(R.g1_mul(G1, 2)))]
    g1c = t(pk.layout.to_soa(H.g1_aos([p for g in grp for p in g]), 8))
    v = torch.full((len(grp),), 9, dtype=torch.uint8, device=dev)
    pk.pairing_fixed_g2_check_target_batch_dev(g1c, g2c, tab, 1, target, v, len(grp), 0, st)
    pk.last_status(0, st)
    assert v.tolist() == [1, 0, 1]
    g2x = t(pk.layout.to_soa(H.g2_aos([G2, Qf] * len(grp)), 16))                       # the same groups as free pairs
    v2 = torch.full((len(grp),), 9, dtype=torch.uint8, device=dev)
    pk.multi_pairing_check_target_batch_dev(g1c, g2x, target, v2, len(grp), 2, 0, st)
    v3 = torch.full((len(grp),), 9, dtype=torch.uint8, device=dev)
    pk.multi_pairing_check_batch_dev(g1c, g2x, v3, len(grp), 2, 0, st)                 # == one: only the middle group (6 - 6)
    pk.last_status(0, st)
    assert v2.tolist() == [1, 0, 1] and v3.tolist() == [0, 1, 0]
    with pytest.raises(ValueError):
        pk.pairing_fixed_g2_check_target_batch_dev(g1c, g2c, tab, 1, target[:47], v, len(grp), 0, st)


def test_fixed_g2_argument_checks():
    import torch
    pk = H.pkg()
    dev = torch.device("cuda:0")
    z = torch.zeros(64, dtype=torch.int64, device=dev)
    for kf in (0, 5):
        with pytest.raises(pk.Bn254Error) as e:
            pk.pairing_fixed_g2_batch_dev(z, z, z, kf, z, 1)
        assert e.value.status == pk.ERR_INVALID_ARG
    with pytest.raises(pk.Bn254Error):
        pk.g2_lines_dev(z, 5, z)
    assert pk.g2_lines_bytes(3) == 3 * 87 * 4 * 72 + 3 * 128              # the lines, then the points themselves


@pytest.mark.parametrize("route", ["throughput", "lane-cooperative"])
def test_fixed_g2_host_pointer_forms(route):
    """bn254_pairing_fixed_g2_batch / _elems on host arrays (the table is made inside the call) against bn254_multi_pairing_batch on the expanded pairs"""
    pk = H.pkg()
    n, kf = 130, 2
    k = 1 + kf
    old = pk.get_latency_threshold()
    pk.set_latency_threshold(0 if route == "throughput" else old)       # 130 groups: below the default threshold the call expands the pairs for the k-pair program
    try:
        _host_forms(pk, n, kf, k)
    finally:
        pk.set_latency_threshold(old)


def _host_forms(pk, n, kf, k):
    Ps, Qs = H.subgroup_points(n * k + kf, seed=95)
    P, Qv, Qf = list(Ps[: n * k]), [Qs[g * k] for g in range(n)], list(Qs[n * k:])
    e1, e2, ef = H.g1_aos(P), H.g2_aos(Qv), H.g2_aos(Qf)
    exp = H.g2_aos([Qv[g] if j == 0 else Qf[j - 1] for g in range(n) for j in range(k)])
    want = pk.multi_pairing_batch(H.to_soa(e1, 8), H.to_soa(exp, 16), n, k)
    got = pk.pairing_fixed_g2_batch(H.to_soa(e1, 8), H.to_soa(e2, 16), H.to_soa(ef, 16), kf, n)
    assert np.array_equal(got, want)
    got_e = pk.pairing_fixed_g2_batch(e1, e2, ef, kf, n, elems=True, out_order=pk.FQ12_ARK)
    idx = [pk.load_library().bn254_myfq12_to_ark_index(j) for j in range(12)]
    assert np.array_equal(got_e.reshape(n, 12, 4), H.to_aos(want, 48).reshape(n, 12, 4)[:, idx, :])
    # the verdict form on the same host structs: target = one group's own value marks exactly the groups that share it
    w = H.to_aos(want, 48).reshape(n, 48)
    v = pk.pairing_fixed_g2_check_batch_elems(e1, e2, ef, kf, n, target=w[5])
    assert v.dtype == np.uint8 and v.tolist() == [1 if np.array_equal(w[g], w[5]) else 0 for g in range(n)] and v[5] == 1 and v.sum() == 1
    assert pk.pairing_fixed_g2_check_batch_elems(e1, e2, ef, kf, n).sum() == 0                  # none of the random products is one
    with pytest.raises(pk.Bn254Error):
        pk.pairing_fixed_g2_check_batch_elems(e1, e2, ef[: 16 * 1], 5, n)
    # the stream keeps the table of its last host-pointer call's fixed points: other points give their own values, the first ones theirs again
    Qf2 = [R.g2_mul(R.G2_GEN, 11 + j) for j in range(kf)]
    ef2 = H.g2_aos(Qf2)
    exp2 = H.g2_aos([Qv[g] if j == 0 else Qf2[j - 1] for g in range(n) for j in range(k)])
    want2 = pk.multi_pairing_batch(H.to_soa(e1, 8), H.to_soa(exp2, 16), n, k)
    assert not np.array_equal(want2, want)
    assert np.array_equal(pk.pairing_fixed_g2_batch(H.to_soa(e1, 8), H.to_soa(e2, 16), H.to_soa(ef2, 16), kf, n), want2)
    assert np.array_equal(pk.pairing_fixed_g2_batch(H.to_soa(e1, 8), H.to_soa(e2, 16), H.to_soa(ef, 16), kf, n), want)
    assert np.array_equal(pk.pairing_fixed_g2_batch(H.to_soa(e1, 8), H.to_soa(e2, 16), H.to_soa(ef, 16), kf, n), want)          # (a hit)
    assert np.array_equal(pk.pairing_fixed_g2_batch(e1, e2, ef, kf, n, elems=True), H.to_aos(want, 48))                           # (the same points, the other layout)
    # six keys in turn, twice: more than the four tables a stream keeps -- every call still gives its own key's values
    sets = []
    for t_ in range(6):
        Qt = [R.g2_mul(R.G2_GEN, 100 + 7 * t_ + j) for j in range(kf)]
        ex = H.g2_aos([Qv[g] if j == 0 else Qt[j - 1] for g in range(n) for j in range(k)])
        sets.append((H.to_soa(H.g2_aos(Qt), 16), pk.multi_pairing_batch(H.to_soa(e1, 8), H.to_soa(ex, 16), n, k)))
    for _ in range(2):
        for fx, wt in sets:
            assert np.array_equal(pk.pairing_fixed_g2_batch(H.to_soa(e1, 8), H.to_soa(e2, 16), fx, kf, n), wt)
    assert len({w.tobytes() for _, w in sets}) == 6


@pytest.mark.parametrize("kf", [1, 2, 3, 4])
def test_fixed_g2_small_batches_take_the_lane_cooperative_programs(kf):
    """Below the latency threshold a fixed-G2 call expands its pairs (the table carries the points) and runs the k-pair lane-cooperative program: the same
    limbs as k_fpairing, 0.8 ms instead of 8; k_fixed = 4 (five pairs: no such program) stays on the throughput kernel."""
    import torch
    pk = H.pkg()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev)
    k = 1 + kf
    idx = [pk.load_library().bn254_myfq12_to_ark_index(j) for j in range(12)]
    for n in (1, 5, 300):
        g1, g2var, g2exp, table = _setup(pk, torch, dev, st, n, kf, 0x5A11 + 16 * kf + n)
        pk.set_stream_latency(0, -1, 0, st)
        try:
            want = torch.zeros(48 * n, dtype=torch.int64, device=dev)
            pk.pairing_fixed_g2_batch_dev(g1, g2var, table, kf, want, n, 0, st)
            assert pk.last_kernel(0, st) == 1
        finally:
            pk.set_stream_latency(pk.LATENCY_INHERIT, -1, 0, st)
        got = torch.full((48 * n + 8,), -7, dtype=torch.int64, device=dev)
        pk.pairing_fixed_g2_batch_dev(g1, g2var, table, kf, got, n, 0, st)
        assert pk.last_kernel(0, st) == 1 if kf == 4 else pk.last_kernel(0, st) in (16, 32, 64)
        e1, e2 = torch.empty_like(g1), torch.empty_like(g2var)
        pk.soa_to_elems_dev(g1, e1, 8, n * k, 0, 0, st)
        pk.soa_to_elems_dev(g2var, e2, 16, n, 0, 0, st)
        eo = torch.full((48 * n + 8,), -7, dtype=torch.int64, device=dev)
        pk.pairing_fixed_g2_batch_elems_dev(e1, e2, table, kf, eo, n, pk.FQ12_ARK, 0, st)
        v = torch.full((n,), 9, dtype=torch.uint8, device=dev)
        target = want.view(48, n)[:, n - 1].contiguous().cpu().numpy().view(np.uint64)
        pk.pairing_fixed_g2_check_target_batch_dev(g1, g2var, table, kf, target, v, n, 0, st)
        pk.last_status(0, st)
        assert torch.equal(got[: 48 * n], want) and bool((got[48 * n:] == -7).all()) and int(want.abs().sum()) != 0
        assert torch.equal(eo[: 48 * n].view(n, 12, 4), want.view(48, n).t().contiguous().view(n, 12, 4)[:, idx, :]) and bool((eo[48 * n:] == -7).all())
        assert v.tolist() == [0] * (n - 1) + [1]


def test_fixed_g2_host_forms_above_one_chunk_take_the_pipeline():
    """More than 65 536 groups from host memory: the two-worker chunked pipeline (the table made once, a small last chunk on the lane-cooperative
    program) must give what one device-resident launch gives -- limb-major, element-major / ark order, and the verdict bytes."""
    import torch
    pk = H.pkg()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev)
    kf, n = 2, 2 * 65536 + 777
    k = 1 + kf
    g1, g2var, g2exp, table = _setup(pk, torch, dev, st, n, kf, 0x91FE)
    f1 = torch.zeros(8 * kf, dtype=torch.int64, device=dev)
    g2fix = torch.zeros(16 * kf, dtype=torch.int64, device=dev)
    pk.generate_pairs_dev(0x91FE ^ 0x5555, f1, g2fix, kf, 0, st)              # (the fixed points _setup made the table from)
    want = torch.zeros(48 * n, dtype=torch.int64, device=dev)
    pk.set_stream_latency(0, -1, 0, st)
    try:
        pk.pairing_fixed_g2_batch_dev(g1, g2var, table, kf, want, n, 0, st)
    finally:
        pk.set_stream_latency(pk.LATENCY_INHERIT, -1, 0, st)
    pk.last_status(0, st)
    h = lambda t: t.cpu().numpy().view(np.uint64).copy()
    h1, h2, hf, hw = h(g1), h(g2var), h(g2fix), h(want)
    got = pk.pairing_fixed_g2_batch(h1, h2, hf, kf, n)
    assert np.array_equal(got, hw)
    e1, e2, ef = H.to_aos(h1, 8), H.to_aos(h2, 16), H.to_aos(hf, 16)
    got_e = pk.pairing_fixed_g2_batch(e1, e2, ef, kf, n, elems=True, out_order=pk.FQ12_ARK)
    idx = [pk.load_library().bn254_myfq12_to_ark_index(j) for j in range(12)]
    w = H.to_aos(hw, 48).reshape(n, 12, 4)
    assert np.array_equal(got_e.reshape(n, 12, 4), w[:, idx, :])
    pos = [0, 65535, 65536, 2 * 65536 + 776]
    v = pk.pairing_fixed_g2_check_batch_elems(e1, e2, ef, kf, n, target=w[pos[2]].reshape(-1))
    assert v.sum() == 1 and v[pos[2]] == 1
    vs = pk.pairing_fixed_g2_check_sharded_elems(e1, e2, ef, kf, n, pk.device_count() if hasattr(pk, "device_count") else 1, target=w[pos[2]].reshape(-1))
    assert np.array_equal(vs, v)                              # (every GPU of the process; one here)
    assert np.array_equal(pk.pairing_fixed_g2_check_sharded_elems(e1[: 8 * k * 300], e2[: 16 * 300], ef, kf, 300, 1, target=w[7].reshape(-1)), v[:300] * 0 + (np.arange(300) == 7))
    ora = H.oracle_multi_pairing(pk.layout.to_aos(h(g1.view(8, n * k)[:, torch.as_tensor([p * k + j for p in pos[:2] for j in range(k)], device=dev)].contiguous()), 8),
                                 pk.layout.to_aos(h(g2exp.view(16, n * k)[:, torch.as_tensor([p * k + j for p in pos[:2] for j in range(k)], device=dev)].contiguous()), 16), 2, k)
    assert np.array_equal(np.concatenate([w[p].reshape(-1) for p in pos[:2]]), ora)
    # the pipeline keeps the table of the last call's fixed points: other points must give their own values, the first ones theirs again
    f1b = torch.zeros(8 * kf, dtype=torch.int64, device=dev)
    g2fix_b = torch.zeros(16 * kf, dtype=torch.int64, device=dev)
    pk.generate_pairs_dev(0xB0B, f1b, g2fix_b, kf, 0, st)
    table_b = torch.zeros(pk.g2_lines_bytes(kf) // 8, dtype=torch.int64, device=dev)
    pk.g2_lines_dev(g2fix_b, kf, table_b, 0, st)
    want_b = torch.zeros(48 * n, dtype=torch.int64, device=dev)
    pk.pairing_fixed_g2_batch_dev(g1, g2var, table_b, kf, want_b, n, 0, st)
    pk.last_status(0, st)
    assert not torch.equal(want_b, want)
    assert np.array_equal(pk.pairing_fixed_g2_batch(h1, h2, h(g2fix_b), kf, n), h(want_b))
    assert np.array_equal(pk.pairing_fixed_g2_batch(h1, h2, hf, kf, n), hw)
    assert np.array_equal(pk.pairing_fixed_g2_batch(h1, h2, hf, kf, n), hw)                      # (a hit)


def test_fixed_g2_calls_are_capturable_after_reserve():
    """After bn254_reserve(n, 1 + k_fixed) the fixed-G2 `_dev` calls are plain launches: the throughput kernel on a large batch, and -- for a batch below
    the latency threshold -- the pair expansion + the lane-cooperative k-pair program through per-stream buffers; captured into a hipGraph and replayed
    on new contents of the same buffers (the verdict against a target included: the target travels in the kernel's arguments)."""
    import torch
    pk = H.pkg()
    dev = torch.device("cuda:0")
    side = torch.cuda.Stream(dev)
    kf, n_big, n_small = 2, 70000, 300
    k = 1 + kf
    with torch.cuda.stream(side):
        g1, g2var, _, table = _setup(pk, torch, dev, side, n_big, kf, 0x77A0)
        s1, s2 = g1[: 8 * n_small * k].clone(), g2var[: 16 * n_small].clone()
        out_big = torch.zeros(48 * n_big, dtype=torch.int64, device=dev)
        out_small = torch.zeros(48 * n_small, dtype=torch.int64, device=dev)
        verdict = torch.zeros(n_small, dtype=torch.uint8, device=dev)
        target = np.zeros(48, dtype=np.uint64)
        pk.reserve(n_big, k, 0, side)
        pk.last_status(0, side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            pk.pairing_fixed_g2_batch_dev(g1, g2var, table, kf, out_big, n_big, 0, side)
            pk.pairing_fixed_g2_batch_dev(s1, s2, table, kf, out_small, n_small, 0, side)
            pk.pairing_fixed_g2_check_target_batch_dev(s1, s2, table, kf, target, verdict, n_small, 0, side)
        first = None
        for seed in (0x77B1, 0x77B2):
            pk.generate_pairs_dev(seed, g1, torch.zeros(16 * n_big * k, dtype=torch.int64, device=dev), n_big * k, 0, side)
            s1.copy_(g1[8 * 5: 8 * 5 + 8 * n_small * k])
            out_big.zero_(); out_small.zero_(); verdict.fill_(7)
            graph.replay()
            side.synchronize()
            got = (out_big.clone(), out_small.clone(), verdict.clone())
            out_big.zero_(); out_small.zero_(); verdict.fill_(7)
            pk.pairing_fixed_g2_batch_dev(g1, g2var, table, kf, out_big, n_big, 0, side)
            pk.pairing_fixed_g2_batch_dev(s1, s2, table, kf, out_small, n_small, 0, side)
            assert pk.last_kernel(0, side) in (16, 32, 64)
            pk.pairing_fixed_g2_check_target_batch_dev(s1, s2, table, kf, target, verdict, n_small, 0, side)
            pk.last_status(0, side)
            assert torch.equal(got[0], out_big) and torch.equal(got[1], out_small) and torch.equal(got[2], verdict)
            assert int(out_big.abs().sum()) != 0 and int(out_small.abs().sum()) != 0 and int(verdict.max()) == 0
            if first is not None:
                assert not torch.equal(first, got[1])
            first = got[1]
    pk.release_stream(0, side)


@pytest.mark.parametrize("kf", [1, 2, 3, 4])
def test_fixed_g2_groups_without_an_own_pair(kf):
    """g2_var = NULL: every G2 point of a group is one of the table's (a KZG-style opening check): final_exp_native(multi_miller_loop_native) of the
    k_fixed pairs -- the same limbs as the k-pair kernel on the expanded pairs, on every lane; small batches through the lane-cooperative programs,
    element-major / ark order, verdicts, host forms (single shot and pipeline)."""
    import torch
    pk = H.pkg()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev)
    idx = [pk.load_library().bn254_myfq12_to_ark_index(j) for j in range(12)]
    g2fix = torch.zeros(16 * kf, dtype=torch.int64, device=dev)
    pk.generate_pairs_dev(0x4B5A + kf, torch.zeros(8 * kf, dtype=torch.int64, device=dev), g2fix, kf, 0, st)
    table = torch.zeros(pk.g2_lines_bytes(kf) // 8, dtype=torch.int64, device=dev)
    pk.g2_lines_dev(g2fix, kf, table, 0, st)
    for n in (3, 1500 + 13 * kf, 65536 + 300):
        g1 = torch.zeros(8 * n * kf, dtype=torch.int64, device=dev)
        pk.generate_pairs_dev(0x4B00 + n, g1, torch.zeros(16 * n * kf, dtype=torch.int64, device=dev), n * kf, 0, st)
        exp = g2fix.view(16, 1, kf).expand(16, n, kf).contiguous().view(-1)
        want = torch.zeros(48 * n, dtype=torch.int64, device=dev)
        if kf == 1:
            pk.pairing_batch_dev(g1, exp, want, n, 0, st)
        else:
            pk.multi_pairing_batch_dev(g1, exp, want, n, kf, True, 0, st)
        for thr in (0, None):                                   # the throughput kernel at every size; then the default route (small batches: lane-cooperative)
            if thr is not None:
                pk.set_stream_latency(thr, -1, 0, st)
            try:
                got = torch.full((48 * n + 8,), -7, dtype=torch.int64, device=dev)
                pk.pairing_fixed_g2_batch_dev(g1, None, table, kf, got, n, 0, st)
                kern = pk.last_kernel(0, st)
                e1 = torch.empty_like(g1)
                pk.soa_to_elems_dev(g1, e1, 8, n * kf, 0, 0, st)
                eo = torch.full((48 * n + 8,), -7, dtype=torch.int64, device=dev)
                pk.pairing_fixed_g2_batch_elems_dev(e1, None, table, kf, eo, n, pk.FQ12_ARK, 0, st)
                v = torch.full((n,), 9, dtype=torch.uint8, device=dev)
                target = want.view(48, n)[:, n - 1].contiguous().cpu().numpy().view(np.uint64)
                pk.pairing_fixed_g2_check_target_batch_dev(g1, None, table, kf, target, v, n, 0, st)
                pk.last_status(0, st)
            finally:
                pk.set_stream_latency(pk.LATENCY_INHERIT, -1, 0, st)
            assert kern == 1 if (thr == 0 or n > 60000) else kern in (16, 32, 64)
            assert torch.equal(got[: 48 * n], want) and bool((got[48 * n:] == -7).all()) and int(want.abs().sum()) != 0
            assert torch.equal(eo[: 48 * n].view(n, 12, 4), want.view(48, n).t().contiguous().view(n, 12, 4)[:, idx, :]) and bool((eo[48 * n:] == -7).all())
            assert v.tolist() == [0] * (n - 1) + [1]
        h = lambda t: t.cpu().numpy().view(np.uint64).copy()
        h1, hf, hw = h(g1), h(g2fix), h(want)
        assert np.array_equal(pk.pairing_fixed_g2_batch(h1, None, hf, kf, n), hw)
        e1h, efh = H.to_aos(h1, 8), H.to_aos(hf, 16)
        assert np.array_equal(pk.pairing_fixed_g2_batch(e1h, None, efh, kf, n, elems=True), H.to_aos(hw, 48))
        vh = pk.pairing_fixed_g2_check_batch_elems(e1h, None, efh, kf, n, target=H.to_aos(hw, 48).reshape(n, 48)[0])
        assert vh[0] == 1 and vh.sum() == 1
    # the oracle on two groups of the last batch
    pos = [1, n - 1]
    sel = torch.as_tensor([p * kf + j for p in pos for j in range(kf)], device=dev)
    g1h = g1.view(8, n * kf)[:, sel].cpu().numpy().view(np.uint64).reshape(-1).copy()
    g2h = exp.view(16, n * kf)[:, sel].cpu().numpy().view(np.uint64).reshape(-1).copy()
    ora = H.oracle_multi_pairing(pk.layout.to_aos(g1h, 8), pk.layout.to_aos(g2h, 16), len(pos), kf)
    mine = want.view(48, n)[:, torch.as_tensor(pos, device=dev)].cpu().numpy().view(np.uint64).reshape(-1).copy()
    assert np.array_equal(pk.layout.to_aos(mine, 48), ora)


def test_fixed_g2_and_spread_calls_from_three_host_threads():
    """Three host threads on the null stream at once: a small fixed-G2 check (one key), a pipeline-sized one (another key), one group of 300 pairs through the
    spread route -- each call must give what it gives alone (per-stream tables and buffers are guarded by the stream's lock, the pipeline by the device's)."""
    import threading
    import torch
    pk = H.pkg()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev)
    h = lambda t: t.cpu().numpy().view(np.uint64).copy()

    def fixed_case(n, kf, seed):
        k = 1 + kf
        g1 = torch.zeros(8 * n * k, dtype=torch.int64, device=dev); g2a = torch.zeros(16 * n * k, dtype=torch.int64, device=dev)
        pk.generate_pairs_dev(seed, g1, g2a, n * k, 0, st)
        fx = g2a.view(16, n * k)[:, 1:1 + kf].contiguous().view(-1)
        var = g2a.view(16, n, k)[:, :, 0].contiguous().view(-1)
        e1, e2, ef = H.to_aos(h(g1), 8), H.to_aos(h(var), 16), H.to_aos(h(fx), 16)
        ref = pk.pairing_fixed_g2_batch(e1, e2, ef, kf, n, elems=True).reshape(n, 48)
        pos = n // 3
        return (e1, e2, ef, kf, n, ref[pos].copy(), pos)

    small, big = fixed_case(200, 2, 0x7E01), fixed_case(70000, 1, 0x7E02)
    K = 300
    g1 = torch.zeros(8 * K, dtype=torch.int64, device=dev); g2 = torch.zeros(16 * K, dtype=torch.int64, device=dev)
    pk.generate_pairs_dev(0x7E03, g1, g2, K, 0, st)
    w1, w2 = h(g1), h(g2)
    want_w = pk.multi_pairing_batch(w1, w2, 1, K)
    errors = []

    def fixed_worker(case):
        e1, e2, ef, kf, n, target, pos = case
        try:
            for _ in range(4):
                v = pk.pairing_fixed_g2_check_batch_elems(e1, e2, ef, kf, n, target=target)
                assert v[pos] == 1 and v.sum() == 1
        except BaseException as ex:      # noqa: BLE001
            errors.append(repr(ex))

    def wide_worker():
        try:
            for _ in range(6):
                assert np.array_equal(pk.multi_pairing_batch(w1, w2, 1, K), want_w)
        except BaseException as ex:      # noqa: BLE001
            errors.append(repr(ex))

    th = [threading.Thread(target=fixed_worker, args=(small,)), threading.Thread(target=fixed_worker, args=(big,)), threading.Thread(target=wide_worker)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
