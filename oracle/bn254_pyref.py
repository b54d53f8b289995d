"""Pure-Python (big-int) restatement of the reference's native pairing path.

TEST INFRASTRUCTURE ONLY.  Nothing outside tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this file.  It is the *independent*
second restatement used to cross-check the C oracle (oracle/bn254_oracle.c) and
to generate the committed golden fixtures (tests/golden/gen_golden.py).

PARITY UNPINNED (known-answer): the reference (qope/plonky2-bn254-pairing) ships
no golden vectors and cannot be built here (no Rust toolchain, arithmetic crates
ark-bn254 0.4.0 / ark-ff 0.4.2 / ark-ec 0.4.2 / plonky2-bn254@d616d57 are not
vendored).  What pins this restatement is the set of algebraic identities the
reference's own tests assert (T1/T3/T4 of SURVEY.md section 4) plus bilinearity;
see tests/test_oracle.py.

Every function cites the reference file:line it follows (paths relative to
/root/reference).  Control flow is kept identical to the reference, including
its quirks (affine G2 stepping with one inversion per step, NAF pow with true
division on -1 digits, Frobenius coefficients recomputed by exponentiation).
Values are canonical integers in [0,p); Montgomery conversion happens only at
the fixture boundary (to_mont / from_mont).
"""

# --------------------------------------------------------------------------
# Field constants (ark-bn254 0.4.0 Fq / Fr; published curve parameters)
# --------------------------------------------------------------------------
P = 21888242871839275222246405745257275088696311157297823662689037894645226208583
R_ORDER = 21888242871839275222246405745257275088548364400416034343698204186575808495617
BN_X = 4965661367192848881  # src/final_exp_native.rs:15
MONT_R = 1 << 256  # ark-ff MontBackend<_, 4>: R = 2^(64*4)
MONT_R_INV = pow(MONT_R, -1, P)

# src/miller_loop_native.rs:314-318
SIX_U_PLUS_2_NAF = [
    0, 0, 0, 1, 0, 1, 0, -1, 0, 0, 1, -1, 0, 0, 1, 0, 0, 1, 1, 0, -1, 0, 0, 1, 0, -1, 0, 0, 0, 0,
    1, 1, 1, 0, 0, -1, 0, 0, 1, 0, 0, 0, 0, 0, -1, 0, 0, 1, 1, 0, 0, -1, 0, 0, 0, 1, 1, 0, -1, 0,
    0, 1, 0, 1, 1,
]

G1_GEN = (1, 2)
G2_GEN = (
    (10857046999023057135944570762232829481370756359578518086990519993285655852781,
     11559732032986387107991004021392285783925812861821192530917403151452391805634),
    (8495653923123431417604973247489272438418190587263600148770280649306958101930,
     4082367875863433681332203403145435568316851327593401208105741076214120093531),
)


def to_mont(a):
    return (a * MONT_R) % P


def from_mont(a):
    return (a * MONT_R_INV) % P


def limbs4(a):
    """u64 little-endian limbs (ark `Fp.0.0`)."""
    return [(a >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]


# --------------------------------------------------------------------------
# Fq2 = Fq[u]/(u^2+1), elements are (c0, c1)
# --------------------------------------------------------------------------
XI = (9, 1)  # src/miller_loop_native.rs:38,74
FQ2_ZERO = (0, 0)
FQ2_ONE = (1, 0)


def fq2_add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def fq2_sub(a, b):
    return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def fq2_neg(a):
    return ((-a[0]) % P, (-a[1]) % P)


def fq2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def fq2_inv(a):
    n = pow((a[0] * a[0] + a[1] * a[1]) % P, -1, P)
    return ((a[0] * n) % P, (-a[1] * n) % P)


def fq2_pow(a, e):
    r = FQ2_ONE
    for bit in bin(e)[2:]:
        r = fq2_mul(r, r)
        if bit == "1":
            r = fq2_mul(r, a)
    return r


def fq2_from_int(k):
    return (k % P, 0)


def conjugate_fp2(x):  # src/miller_loop_native.rs:284-289
    return (x[0], (-x[1]) % P)


def neg_conjugate_fp2(x):  # src/miller_loop_native.rs:291-296
    return ((-x[0]) % P, x[1])


# --------------------------------------------------------------------------
# MyFq12 (plonky2-bn254 fields::native::MyFq12): coeffs[12] of Fq,
# element = sum_{i<6} (coeffs[i] + coeffs[i+6] u) w^i,  w^6 = XI
# (layout evidenced by src/miller_loop_native.rs:47-51,86-92)
# --------------------------------------------------------------------------
def fq12_to_fp2s(a):
    return [(a[i], a[i + 6]) for i in range(6)]


def fq12_from_fp2s(c):
    return [x[0] for x in c] + [x[1] for x in c]


def fq12_one():
    return [1] + [0] * 11


def fq12_mul(a, b):
    """MyFq12 `Mul` (dense product in the w-basis, w^6 = 9+u)."""
    af, bf = fq12_to_fp2s(a), fq12_to_fp2s(b)
    prod = [FQ2_ZERO] * 11
    for i in range(6):
        for j in range(6):
            prod[i + j] = fq2_add(prod[i + j], fq2_mul(af[i], bf[j]))
    out = []
    for i in range(6):
        if i < 5:
            out.append(fq2_add(prod[i], fq2_mul(prod[i + 6], XI)))
        else:
            out.append(prod[5])
    return fq12_from_fp2s(out)


def fq12_conjugate(a):  # src/final_exp_native.rs:171-181 (conjugate_fp12)
    return [c if i % 2 == 0 else (-c) % P for i, c in enumerate(a)]


# Fq12 inversion through the tower Fq12 = Fq6[w]/(w^2 - v), Fq6 = Fq2[v]/(v^3 - XI).
# MyFq12 coefficient i (power w^i): even i -> c0 part (v^(i/2)), odd i -> c1 part.
def _fq6_mul(a, b):
    a0, a1, a2 = a
    b0, b1, b2 = b
    t0 = fq2_mul(a0, b0)
    t1 = fq2_mul(a1, b1)
    t2 = fq2_mul(a2, b2)
    c0 = fq2_add(t0, fq2_mul(XI, fq2_add(fq2_mul(a1, b2), fq2_mul(a2, b1))))
    c1 = fq2_add(fq2_add(fq2_mul(a0, b1), fq2_mul(a1, b0)), fq2_mul(XI, t2))
    c2 = fq2_add(fq2_add(fq2_mul(a0, b2), fq2_mul(a2, b0)), t1)
    return (c0, c1, c2)


def _fq6_mul_by_v(a):
    return (fq2_mul(XI, a[2]), a[0], a[1])


def _fq6_sub(a, b):
    return tuple(fq2_sub(x, y) for x, y in zip(a, b))


def _fq6_neg(a):
    return tuple(fq2_neg(x) for x in a)


def _fq6_inv(a):
    a0, a1, a2 = a
    t0 = fq2_sub(fq2_mul(a0, a0), fq2_mul(XI, fq2_mul(a1, a2)))
    t1 = fq2_sub(fq2_mul(XI, fq2_mul(a2, a2)), fq2_mul(a0, a1))
    t2 = fq2_sub(fq2_mul(a1, a1), fq2_mul(a0, a2))
    n = fq2_add(fq2_mul(a0, t0), fq2_mul(XI, fq2_add(fq2_mul(a2, t1), fq2_mul(a1, t2))))
    ni = fq2_inv(n)
    return (fq2_mul(t0, ni), fq2_mul(t1, ni), fq2_mul(t2, ni))


def fq12_inv(a):
    f = fq12_to_fp2s(a)
    c0 = (f[0], f[2], f[4])
    c1 = (f[1], f[3], f[5])
    d = _fq6_sub(_fq6_mul(c0, c0), _fq6_mul_by_v(_fq6_mul(c1, c1)))
    di = _fq6_inv(d)
    r0 = _fq6_mul(c0, di)
    r1 = _fq6_neg(_fq6_mul(c1, di))
    return fq12_from_fp2s([r0[0], r1[0], r0[1], r1[1], r0[2], r1[2]])


def fq12_div(a, b):  # ark `Fq12 / Fq12` (src/final_exp_native.rs:74,200)
    return fq12_mul(a, fq12_inv(b))


def fq12_pow(a, e):  # ark `Field::pow` (square-and-multiply from 1)
    r = fq12_one()
    for bit in bin(e)[2:]:
        r = fq12_mul(r, r)
        if bit == "1":
            r = fq12_mul(r, a)
    return r


def myfq12_to_ark(a):
    """MyFq12 -> ark Fq12 flat order [c0.c0.c0, c0.c0.c1, c0.c1.c0, ... c1.c2.c1]
    (`.into()` at src/pairing.rs:21; w^2 = v)."""
    out = []
    for h in range(2):
        for k in range(3):
            for e in range(2):
                out.append(a[(2 * k + h) + 6 * e])
    return out


def ark_to_myfq12(f):
    a = [0] * 12
    idx = 0
    for h in range(2):
        for k in range(3):
            for e in range(2):
                a[(2 * k + h) + 6 * e] = f[idx]
                idx += 1
    return a


# --------------------------------------------------------------------------
# G1 / G2 affine group law (ark-ec short-Weierstrass; result normalised to affine,
# i.e. the mathematically unique affine point -- call sites miller_loop_native.rs
# :157,167,186).  Points are (x, y); identity is None.
# --------------------------------------------------------------------------
def g1_add(a, b):
    if a is None:
        return b
    if b is None:
        return a
    if a[0] == b[0]:
        if (a[1] + b[1]) % P == 0:
            return None
        lam = (3 * a[0] * a[0]) * pow(2 * a[1], -1, P) % P
    else:
        lam = (b[1] - a[1]) * pow(b[0] - a[0], -1, P) % P
    x3 = (lam * lam - a[0] - b[0]) % P
    y3 = (lam * (a[0] - x3) - a[1]) % P
    return (x3, y3)


def g1_neg(a):
    return None if a is None else (a[0], (-a[1]) % P)


def g1_mul(a, k):
    r = None
    for bit in bin(k)[2:] if k > 0 else "":
        r = g1_add(r, r)
        if bit == "1":
            r = g1_add(r, a)
    return r


def g2_add(a, b):
    if a is None:
        return b
    if b is None:
        return a
    if a[0] == b[0]:
        if fq2_add(a[1], b[1]) == FQ2_ZERO:
            return None
        num = fq2_mul(fq2_from_int(3), fq2_mul(a[0], a[0]))
        lam = fq2_mul(num, fq2_inv(fq2_add(a[1], a[1])))
    else:
        lam = fq2_mul(fq2_sub(b[1], a[1]), fq2_inv(fq2_sub(b[0], a[0])))
    x3 = fq2_sub(fq2_sub(fq2_mul(lam, lam), a[0]), b[0])
    y3 = fq2_sub(fq2_mul(lam, fq2_sub(a[0], x3)), a[1])
    return (x3, y3)


def g2_neg(a):
    return None if a is None else (a[0], fq2_neg(a[1]))


def g2_mul(a, k):
    r = None
    for bit in bin(k)[2:] if k > 0 else "":
        r = g2_add(r, r)
        if bit == "1":
            r = g2_add(r, a)
    return r


TWIST_B = fq2_mul(fq2_from_int(3), fq2_inv(XI))  # y^2 = x^3 + 3/xi


def g1_on_curve(a):
    return (a[1] * a[1] - a[0] ** 3 - 3) % P == 0


def g2_on_curve(a):
    lhs = fq2_mul(a[1], a[1])
    rhs = fq2_add(fq2_mul(fq2_mul(a[0], a[0]), a[0]), TWIST_B)
    return lhs == rhs


# --------------------------------------------------------------------------
# Miller loop  (src/miller_loop_native.rs)
# --------------------------------------------------------------------------
def sparse_line_function_unequal_native(Q, Pt):  # :10-28
    (x_1, y_1), (x_2, y_2) = Q
    x, y = Pt
    y1_minus_y2 = fq2_sub(y_1, y_2)
    x2_minus_x1 = fq2_sub(x_2, x_1)
    x1y2 = fq2_mul(x_1, y_2)
    x2y1 = fq2_mul(x_2, y_1)
    out3 = fq2_mul(y1_minus_y2, (x, 0))
    out2 = fq2_mul(x2_minus_x1, (y, 0))
    out5 = fq2_sub(x1y2, x2y1)
    return [None, None, out2, out3, None, out5]


def sparse_line_function_equal_native(Q, Pt):  # :30-44
    x, y = Q
    x_sq = fq2_mul(x, x)
    x_cube = fq2_mul(x_sq, x)
    three_x_cu = fq2_mul(x_cube, fq2_from_int(3))
    y_sq = fq2_mul(y, y)
    two_y_sq = fq2_mul(y_sq, fq2_from_int(2))
    out0_left = fq2_sub(three_x_cu, two_y_sq)
    out0 = fq2_mul(out0_left, XI)
    x_sq_px = fq2_mul(x_sq, (Pt[0], 0))
    out4 = fq2_mul(x_sq_px, fq2_from_int(-3))
    y_py = fq2_mul(y, (Pt[1], 0))
    out3 = fq2_mul(y_py, fq2_from_int(2))
    return [out0, None, None, out3, out4, None]


def sparse_fp12_multiply_native(a, b):  # :46-96
    a_fp2 = fq12_to_fp2s(a)
    prod_2d = [None] * 11
    for i in range(6):
        for j in range(6):
            if b[j] is None:
                continue
            ab = fq2_mul(a_fp2[i], b[j])
            prod_2d[i + j] = ab if prod_2d[i + j] is None else fq2_add(prod_2d[i + j], ab)
    out_fp2 = []
    for i in range(6):
        if i != 5:
            eval_w6 = None if prod_2d[i + 6] is None else fq2_mul(prod_2d[i + 6], XI)
            if prod_2d[i] is None:
                assert eval_w6 is not None  # reference: `b.unwrap()`
                prod = eval_w6
            elif eval_w6 is None:
                prod = prod_2d[i]
            else:
                prod = fq2_add(prod_2d[i], eval_w6)
        else:
            assert prod_2d[i] is not None
            prod = prod_2d[i]
        out_fp2.append(prod)
    return fq12_from_fp2s(out_fp2)


def fp12_multiply_with_line_unequal_native(g, Q, Pt):  # :98-105
    return sparse_fp12_multiply_native(g, sparse_line_function_unequal_native(Q, Pt))


def fp12_multiply_with_line_equal_native(g, Q, Pt):  # :107-110
    return sparse_fp12_multiply_native(g, sparse_line_function_equal_native(Q, Pt))


def _sparse_to_dense(sparse_f):  # :130-149
    c = [x if x is not None else FQ2_ZERO for x in sparse_f]
    return fq12_from_fp2s(c)


def _end_constants():  # :176-181
    k = (P - 1) // 6
    expected_c = fq2_pow(XI, k)
    c2 = fq2_mul(expected_c, expected_c)
    c3 = fq2_mul(c2, expected_c)
    return c2, c3


def twisted_frobenius(Q, c2, c3):  # :298-304
    return (fq2_mul(c2, conjugate_fp2(Q[0])), fq2_mul(c3, conjugate_fp2(Q[1])))


def neg_twisted_frobenius(Q, c2, c3):  # :306-312
    return (fq2_mul(c2, conjugate_fp2(Q[0])), fq2_mul(c3, neg_conjugate_fp2(Q[1])))


def miller_loop_BN_native(Q, Pt, enc):  # :112-190
    i = len(enc) - 1
    while enc[i] == 0:
        i -= 1
    last_index = i
    assert enc[i] in (1, -1)
    R = Q if enc[i] == 1 else g2_neg(Q)
    i -= 1
    f = _sparse_to_dense(sparse_line_function_equal_native(R, Pt))
    while True:
        if i != last_index - 1:
            f_sq = fq12_mul(f, f)
            f = fp12_multiply_with_line_equal_native(f_sq, R, Pt)
        R = g2_add(R, R)
        assert -1 <= enc[i] <= 1
        if enc[i] != 0:
            sign_Q = Q if enc[i] == 1 else g2_neg(Q)
            f = fp12_multiply_with_line_unequal_native(f, (R, sign_Q), Pt)
            R = g2_add(R, sign_Q)
        if i == 0:
            break
        i -= 1
    c2, c3 = _end_constants()
    Q_1 = twisted_frobenius(Q, c2, c3)
    neg_Q_2 = neg_twisted_frobenius(Q_1, c2, c3)
    f = fp12_multiply_with_line_unequal_native(f, (R, Q_1), Pt)
    R = g2_add(R, Q_1)
    f = fp12_multiply_with_line_unequal_native(f, (R, neg_Q_2), Pt)
    return f


def multi_miller_loop_BN_native(pairs, enc):  # :192-282 ; pairs = [(G1, G2), ...]
    i = len(enc) - 1
    while enc[i] == 0:
        i -= 1
    last_index = i
    assert enc[last_index] == 1
    neg_b = [g2_neg(b) for (_, b) in pairs]
    f = _sparse_to_dense(sparse_line_function_equal_native(pairs[0][1], pairs[0][0]))
    for (a, b) in pairs[1:]:
        f = fp12_multiply_with_line_equal_native(f, b, a)
    i -= 1
    r = [b for (_, b) in pairs]
    while True:
        if i != last_index - 1:
            f = fq12_mul(f, f)
            for rr, (a, _) in zip(r, pairs):
                f = fp12_multiply_with_line_equal_native(f, rr, a)
        r = [g2_add(rr, rr) for rr in r]
        assert -1 <= enc[i] <= 1
        if enc[i] != 0:
            for idx, (a, b) in enumerate(pairs):
                sign_b = b if enc[i] == 1 else neg_b[idx]
                f = fp12_multiply_with_line_unequal_native(f, (r[idx], sign_b), a)
                r[idx] = g2_add(r[idx], sign_b)
        if i == 0:
            break
        i -= 1
    c2, c3 = _end_constants()
    for idx, (a, b) in enumerate(pairs):
        b_1 = twisted_frobenius(b, c2, c3)
        neg_b_2 = neg_twisted_frobenius(b_1, c2, c3)
        f = fp12_multiply_with_line_unequal_native(f, (r[idx], b_1), a)
        r[idx] = g2_add(r[idx], b_1)
        f = fp12_multiply_with_line_unequal_native(f, (r[idx], neg_b_2), a)
    return f


def miller_loop_native(Q, Pt):  # :320-322
    return miller_loop_BN_native(Q, Pt, SIX_U_PLUS_2_NAF)


def multi_miller_loop_native(pairs):  # :324-326
    return multi_miller_loop_BN_native(pairs, SIX_U_PLUS_2_NAF)


# --------------------------------------------------------------------------
# Final exponentiation  (src/final_exp_native.rs)
# --------------------------------------------------------------------------
def frob_coeffs(index):  # :183-192
    k = (P ** index - 1) // 6
    return fq2_pow(XI, k)


def frobenius_map_native(a, power):  # :17-54
    assert P % 4 == 3 and P % 6 == 1
    pw = power % 12
    out_fp2 = []
    fc = frob_coeffs(pw)
    for i in range(6):
        frob_coeff = fq2_pow(fc, i)
        a_fp2 = (a[i], a[i + 6])
        if pw % 2 != 0:
            a_fp2 = conjugate_fp2(a_fp2)
        if frob_coeff == FQ2_ONE:
            out_fp2.append(a_fp2)
        elif frob_coeff[1] == 0:
            out_fp2.append(fq2_mul(a_fp2, (frob_coeff[0], 0)))
        else:
            out_fp2.append(fq2_mul(a_fp2, frob_coeff))
    return fq12_from_fp2s(out_fp2)


def get_naf(exp):  # :86-128 ; exp = list of u64 limbs, LSB limb first
    exp = list(exp)
    naf = []
    ln = len(exp)
    for idx in range(ln):
        e = exp[idx]
        for _ in range(64):
            if e & 1 == 1:
                z = 2 - (e % 4)
                e //= 2
                if z == -1:
                    e += 1
                naf.append(z)
            else:
                naf.append(0)
                e //= 2
        if e != 0:
            assert e == 1
            j = idx + 1
            while j < len(exp) and exp[j] == 0xFFFFFFFFFFFFFFFF:
                exp[j] = 0
                j += 1
            if j < len(exp):
                exp[j] += 1
            else:
                exp.append(1)
    if len(exp) != ln:
        # reference :123 is `assert_eq!(len, exp.len() + 1)`, which can never hold after the
        # push above: a carry out of the top limb PANICS in the reference.  Mirrored.
        assert ln == len(exp) + 1, "get_naf: carry out of the top limb (reference panics here)"
        assert exp[ln] == 1
        naf.append(1)
    return naf


def pow_native(a, exp):  # :56-84
    res = list(a)
    is_started = False
    naf = get_naf(exp)
    for z in reversed(naf):
        if is_started:
            res = fq12_mul(res, res)
        if z != 0:
            assert z in (1, -1)
            if is_started:
                res = fq12_mul(res, a) if z == 1 else fq12_div(res, a)
            else:
                assert z == 1
                is_started = True
    return res


def hard_part_BN_native(m):  # :130-169
    mp = frobenius_map_native(m, 1)
    mp2 = frobenius_map_native(m, 2)
    mp3 = frobenius_map_native(m, 3)
    mp2_mp3 = fq12_mul(mp2, mp3)
    y0 = fq12_mul(mp, mp2_mp3)
    y1 = fq12_conjugate(m)
    mx = pow_native(m, [BN_X])
    mxp = frobenius_map_native(mx, 1)
    mx2 = pow_native(mx, [BN_X])
    mx2p = frobenius_map_native(mx2, 1)
    y2 = frobenius_map_native(mx2, 2)
    y5 = fq12_conjugate(mx2)
    mx3 = pow_native(mx2, [BN_X])
    mx3p = frobenius_map_native(mx3, 1)
    y3 = fq12_conjugate(mxp)
    mx_mx2p = fq12_mul(mx, mx2p)
    y4 = fq12_conjugate(mx_mx2p)
    mx3_mx3p = fq12_mul(mx3, mx3p)
    y6 = fq12_conjugate(mx3_mx3p)
    T0 = fq12_mul(y6, y6)
    T0 = fq12_mul(T0, y4)
    T0 = fq12_mul(T0, y5)
    T1 = fq12_mul(y3, y5)
    T1 = fq12_mul(T1, T0)
    T0 = fq12_mul(y2, T0)
    T1 = fq12_mul(T1, T1)
    T1 = fq12_mul(T1, T0)
    T1 = fq12_mul(T1, T1)
    T0 = fq12_mul(T1, y1)
    T1 = fq12_mul(T1, y0)
    T0 = fq12_mul(T0, T0)
    T0 = fq12_mul(T0, T1)
    return T0


def easy_part(a):  # :195-206
    f1 = fq12_conjugate(a)
    f2 = fq12_div(f1, a)
    f3 = frobenius_map_native(f2, 2)
    return fq12_mul(f3, f2)


def final_exp_native(a):  # :209-213
    return hard_part_BN_native(easy_part(a))


def pairing(p, q):  # src/pairing.rs:20-22 (result in ark Fq12 flat order)
    return myfq12_to_ark(final_exp_native(miller_loop_native(q, p)))


def pairing_myfq12(p, q):
    return final_exp_native(miller_loop_native(q, p))


# --------------------------------------------------------------------------
# Deterministic synthetic inputs: SplitMix64 scalars
# --------------------------------------------------------------------------
def splitmix64(state):
    state = (state + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    z = state
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return state, z ^ (z >> 31)


def rand_scalar(state):
    """256 bits of SplitMix64 reduced mod r; never 0."""
    v = 0
    for _ in range(4):
        state, w = splitmix64(state)
        v = (v << 64) | w
    v %= R_ORDER
    return state, (v if v != 0 else 1)
