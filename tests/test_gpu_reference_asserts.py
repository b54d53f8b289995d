"""The reference's OWN test assertions, executed on the HIP engine at batch size, entirely on the device, with NO oracle in
the loop: no value the CPU restatement computes is ever compared with.  (`helpers` supplies constants, limb packing and -- for
the two tests that need points with KNOWN scalars -- the big-integer group law to build INPUTS; expected values always come
from the device.)

  T1  miller_loop_native.rs:336-348   multi_miller_loop_native([(P0,Q0),(P1,Q1)]) == miller(Q0,P0) * miller(Q1,P1)
  T3  final_exp_native.rs:240-264     m == m0 * m1  and  final_exp(m0) * final_exp(m1) == final_exp(m)   (+ the product is one)
  T4  final_exp_native.rs:266-286     pow_native(x, [BN_X]) == x.pow([BN_X])   and
                                      final_exp_native(x) == x.pow((p^12 - 1) / r)   for a random (non-unitary) Fq12
  bilinearity   e([s]G1, [t]G2) == e(G1, G2)^(s t mod r),  e(G1,G2)^r == 1 != e(G1,G2)

ark's `Field::pow` is the plain left-to-right binary square-and-multiply; `plain_pow` below restates it from
`bn254_fq12_mul_batch_dev` ALONE (MyFq12 `Mul`), so the right-hand sides share nothing with the kernels under test
(k_fexp: easy part with a true inversion, cyclotomic squarings, fixed-set recoded x-powers, Frobenius constants;
k_op pow: NAF digits with true divisions).  Every comparison is over ALL lanes (torch.equal on the limb planes): integer
work, identical limbs, no tolerance."""
import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu

P_MOD = H.R.P
R_ORD = H.R.R_ORDER
FINAL_EXP = (P_MOD ** 12 - 1) // R_ORD            # final_exp_native.rs:279 `(p.pow(12) - 1u32) / r`
P_TOP = 0x30644e72e131a029


def limbs64(x):
    out = []
    while x:
        out.append(x & 0xFFFFFFFFFFFFFFFF)
        x >>= 64
    return out or [0]


def rand_fq12_dev(n, seed, dev):
    """n arbitrary Fq12 elements, SoA: every coefficient a uniform bit pattern whose top limb is below p's (a valid
    Montgomery representative of some field element) -- the `Fq12::rand` of final_exp_native.rs:269."""
    import torch
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    t = torch.randint(-(1 << 63), (1 << 63) - 1, (48, n), dtype=torch.int64, device=dev, generator=g)
    t[3::4] = torch.randint(0, P_TOP, (12, n), dtype=torch.int64, device=dev, generator=g)
    return t.view(-1)


class Dev:
    def __init__(self):
        import torch
        self.torch = torch
        self.pk = H.pkg()
        self.dev = torch.device("cuda:0")
        self.st = torch.cuda.current_stream(self.dev)

    def empty(self, words, n):
        return self.torch.empty(words * n, dtype=self.torch.int64, device=self.dev)

    def mul(self, a, b, n, out=None):
        out = self.empty(48, n) if out is None else out
        self.pk.fq12_mul_batch_dev(a, b, out, n, 0, self.st)
        return out

    def plain_pow(self, x, e, n):
        """ark `Field::pow`: res = 1; for each bit from the top: res = res^2; if bit: res *= x (leading zeros skipped: the
        first set bit gives res = x).  Built from MyFq12 Mul launches only; two ping-pong buffers."""
        assert e > 0
        bits = bin(e)[2:]
        res, tmp = x.clone(), self.empty(48, n)
        for b in bits[1:]:
            self.mul(res, res, n, out=tmp)
            res, tmp = tmp, res
            if b == "1":
                self.mul(res, x, n, out=tmp)
                res, tmp = tmp, res
        return res

    def sync(self):
        self.pk.last_status(0, self.st)


@pytest.fixture(scope="module")
def d():
    return Dev()


def test_T4_pow_native_equals_plain_pow_on_every_lane(d):
    """final_exp_native.rs:270-273: `pow_native(x, vec![BN_X]) == x.pow(&[BN_X])`, 2^16 random non-unitary x."""
    n = 1 << 16
    x = rand_fq12_dev(n, 41, d.dev)
    got = d.empty(48, n)
    d.pk.pow_batch_dev(x, [d.pk.BN_X], got, n, 0, d.st)
    want = d.plain_pow(x, d.pk.BN_X, n)
    d.sync()
    assert d.torch.equal(got, want)
    assert not d.torch.equal(got, x)


def test_T4_final_exp_equals_the_exact_exponent_on_every_lane(d):
    """final_exp_native.rs:275-285: `final_exp_native(x) == x.pow((p^12 - 1) / r)` for random (non-unitary) x on every lane of
    a 2^16 batch.  Right-hand side twice: (a) plain binary square-and-multiply over the 2 790-bit exponent from MyFq12 Mul
    launches only (4 190 launches), (b) `pow_native` with the 44-limb exponent (NAF with carries across limbs and true
    Fq12 divisions on the -1 digits)."""
    n = 1 << 16
    x = rand_fq12_dev(n, 42, d.dev)
    assert FINAL_EXP.bit_length() == 2790 and len(limbs64(FINAL_EXP)) == 44
    got = d.empty(48, n)
    d.pk.final_exp_batch_dev(x, got, n, 0, d.st)
    want = d.plain_pow(x, FINAL_EXP, n)
    d.sync()
    assert d.torch.equal(got, want)
    via_naf = d.empty(48, n)
    d.pk.pow_batch_dev(x, limbs64(FINAL_EXP), via_naf, n, 0, d.st)
    d.sync()
    assert d.torch.equal(via_naf, want)
    # the result has order dividing r: (x^((p^12-1)/r))^r == 1 on every lane (plain pow again)
    one = d.plain_pow(got, R_ORD, n)
    d.sync()
    one_limbs = d.torch.zeros(48, n, dtype=d.torch.int64, device=d.dev)
    mont_one = H.R.limbs4(H.R.to_mont(1))
    for l in range(4):
        one_limbs[l] = np.array(mont_one[l], dtype=np.uint64).astype(np.int64).item()
    assert d.torch.equal(one.view(48, n), one_limbs)


def _pairs(d, n_groups, seed):
    """2 n_groups generated pairs; group g = pairs (2g, 2g+1) -- the layout bn254_multi_pairing_batch_dev expects -- plus the
    even / odd pairs as two contiguous n_groups batches."""
    t = d.torch
    n = 2 * n_groups
    g1, g2 = d.empty(8, n), d.empty(16, n)
    d.pk.generate_pairs_dev(seed, g1, g2, n, 0, d.st)
    split = lambda buf, words, j: buf.view(words, n_groups, 2)[:, :, j].contiguous().view(-1)
    return g1, g2, (split(g1, 8, 0), split(g2, 16, 0)), (split(g1, 8, 1), split(g2, 16, 1))


def test_T1_multi_miller_equals_product_on_every_lane(d):
    """miller_loop_native.rs:336-348: r0 = miller(Q0, P0), r1 = miller(Q1, P1), multi([(P0,Q0),(P1,Q1)]) == r0 * r1, on 2^16
    random groups (three kernels: k_mmiller with the shared f and two tracked line scales, k_miller, k_op Mul)."""
    n = 1 << 16
    g1, g2, (p0, q0), (p1, q1) = _pairs(d, n, 0xB2540101)
    multi = d.empty(48, n)
    d.pk.multi_pairing_batch_dev(g1, g2, multi, n, 2, False, 0, d.st)
    r0, r1 = d.empty(48, n), d.empty(48, n)
    d.pk.miller_loop_batch_dev(p0, q0, r0, n, 0, d.st)
    d.pk.miller_loop_batch_dev(p1, q1, r1, n, 0, d.st)
    want = d.mul(r0, r1, n)
    d.sync()
    assert d.torch.equal(multi, want)
    assert not d.torch.equal(r0, r1)


def test_T3_final_exp_is_multiplicative_on_every_lane(d):
    """final_exp_native.rs:253-263: m = multi([(P0,Q0),(P1,Q1)]), m0, m1 the single Miller values:
    `m == m0 * m1` and `final_exp(m0) * final_exp(m1) == final_exp(m)` -- 2^16 random groups, plus the fused kernels:
    k_mpairing(group) and k_pairing(P0,Q0) * k_pairing(P1,Q1) give the same limbs."""
    n = 1 << 16
    g1, g2, (p0, q0), (p1, q1) = _pairs(d, n, 0xB2540102)
    m, m0, m1 = d.empty(48, n), d.empty(48, n), d.empty(48, n)
    d.pk.multi_pairing_batch_dev(g1, g2, m, n, 2, False, 0, d.st)
    d.pk.miller_loop_batch_dev(p0, q0, m0, n, 0, d.st)
    d.pk.miller_loop_batch_dev(p1, q1, m1, n, 0, d.st)
    assert d.torch.equal(m, d.mul(m0, m1, n))
    r0, r1, r_mul = d.empty(48, n), d.empty(48, n), d.empty(48, n)
    d.pk.final_exp_batch_dev(m0, r0, n, 0, d.st)
    d.pk.final_exp_batch_dev(m1, r1, n, 0, d.st)
    d.pk.final_exp_batch_dev(m, r_mul, n, 0, d.st)
    r_sep = d.mul(r0, r1, n)
    d.sync()
    assert d.torch.equal(r_sep, r_mul)
    fused = d.empty(48, n)
    d.pk.multi_pairing_batch_dev(g1, g2, fused, n, 2, True, 0, d.st)
    e0, e1 = d.empty(48, n), d.empty(48, n)
    d.pk.pairing_batch_dev(p0, q0, e0, n, 0, d.st)
    d.pk.pairing_batch_dev(p1, q1, e1, n, 0, d.st)
    d.sync()
    assert d.torch.equal(fused, r_mul) and d.torch.equal(d.mul(e0, e1, n), r_mul)


def test_T3_exact_construction_of_the_reference(d):
    """final_exp_native.rs:240-263 with its own scalars: s = 5, t = 6, P0 = [s]G1, Q0 = [t]G2, P1 = [s t]G1, Q1 = -G2.
    Points from the big-integer group law (inputs only); every assertion of the test on the device, and -- what the
    construction is for -- the product of the two pairings is MyFq12::one."""
    R = H.R
    s, t = 5, 6
    P0, Q0 = R.g1_mul(R.G1_GEN, s), R.g2_mul(R.G2_GEN, t)
    P1 = R.g1_mul(R.G1_GEN, s * t)
    gx, gy = R.G2_GEN
    Q1 = (gx, tuple((P_MOD - c) % P_MOD for c in gy))
    pk = d.pk
    g1 = H.to_soa(H.g1_aos([P0, P1]), 8)
    g2 = H.to_soa(H.g2_aos([Q0, Q1]), 16)
    m = pk.multi_pairing_batch(g1, g2, 1, 2, do_final_exp=False)
    ms = pk.miller_loop_batch(g1, g2, 2).reshape(48, 2)
    m0, m1 = ms[:, 0].copy(), ms[:, 1].copy()
    assert np.array_equal(m, pk.fq12_mul_batch(m0, m1, 1))                      # assert_eq!(m, m0 * m1)
    r0, r1, r_mul = pk.final_exp_batch(m0, 1), pk.final_exp_batch(m1, 1), pk.final_exp_batch(m, 1)
    assert np.array_equal(pk.fq12_mul_batch(r0, r1, 1), r_mul)                  # assert_eq!(r_sep, r_mul)
    one = H.fq12_aos([[1] + [0] * 11])
    assert np.array_equal(r_mul, one) and not np.array_equal(r0, one)
    assert pk.multi_pairing_check_batch(g1, g2, 1, 2).tolist() == [1]


def test_bilinearity_on_generated_scalars(d):
    """e([s_i]G1, [t_i]G2) == e(G1, G2)^(s_i t_i mod r) for the first 32 pairs of bn254_generate_pairs_dev (whose scalars the
    header states: generator_scalars), e(G1,G2)^r == 1 and e(G1,G2) != 1.  Left: k_pairing on generated points; right:
    k_pairing on the generators, then pow_native with a four-limb exponent AND the plain binary pow from Mul launches."""
    R = H.R
    pk = d.pk
    seed, n = 0xB2540001, 32
    g1, g2, out = d.empty(8, n), d.empty(16, n), d.empty(48, n)
    pk.generate_pairs_dev(seed, g1, g2, n, 0, d.st)
    pk.pairing_batch_dev(g1, g2, out, n, 0, d.st)
    d.sync()
    lhs = out.cpu().numpy().view(np.uint64).reshape(48, n)
    e = pk.pairing_batch(H.to_soa(H.g1_aos([tuple(R.G1_GEN)]), 8), H.to_soa(H.g2_aos([R.G2_GEN]), 16), 1)
    one = H.fq12_aos([[1] + [0] * 11])
    assert not np.array_equal(e, one)
    assert np.array_equal(pk.pow_batch(e, limbs64(R_ORD), 1), one)
    e_dev = d.torch.from_numpy(e.view(np.int64)).to(d.dev)
    for i in range(n):
        s, t = pk.generator_scalars(seed, i)
        ex = (s * t) % R_ORD
        assert np.array_equal(pk.pow_batch(e, limbs64(ex), 1), lhs[:, i]), i
        if i < 4:
            plain = d.plain_pow(e_dev, ex, 1)
            d.sync()
            assert np.array_equal(plain.cpu().numpy().view(np.uint64), lhs[:, i]), i
