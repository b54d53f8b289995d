"""Big-int model of the *GPU schedule* (what the HIP kernels compute), checked against the
reference restatement (oracle/bn254_pyref.py) in tests/test_sched_model.py.

The HIP kernels do not follow the reference's operation order: they use inversion-free
homogeneous projective G2 stepping with a tracked Fq2 scale (so that the un-normalised
affine line values of `miller_loop_native` are reproduced exactly), tower/Karatsuba Fq12
arithmetic, cyclotomic squarings and conjugate-for-inverse in the hard part.  This file
states that schedule in plain Python so the algebra is testable without a GPU; the
device code in plonky2-bn254-pairing_amd/csrc/ is a transliteration of it.

Test infrastructure only (imports the oracle).
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import bn254_pyref as R  # noqa: E402

P = R.P
XI = R.XI
add, sub, mul, neg = R.fq2_add, R.fq2_sub, R.fq2_mul, R.fq2_neg


def sqr(a):
    return mul(a, a)


def mul_fq(a, k):
    return ((a[0] * k) % P, (a[1] * k) % P)


def mul_xi(a):
    return mul(a, XI)


def small(a, k):
    return ((a[0] * k) % P, (a[1] * k) % P)


THREE_B = mul(R.fq2_from_int(3), R.TWIST_B)  # 3 b' = 9/xi


# ------------------------------------------------------------------ G2 steps
def dbl_step(Rp, Pt):
    """R=(X,Y,Z) homogeneous projective.  Returns (2R, (L0,L3,L4), lam) with
    line_affine(R) * lam == L, lam = Z^2  (reference line: miller_loop_native.rs:30-44)."""
    X, Y, Z = Rp
    B = sqr(Y)
    C = sqr(Z)
    E = mul(THREE_B, C)
    F = small(E, 3)
    H = small(mul(Y, Z), 2)          # 2YZ
    XX = sqr(X)
    X3 = mul(small(mul(X, Y), 2), sub(B, F))
    BF = add(B, F)
    Y3 = sub(sqr(BF), small(sqr(E), 12))
    Z3 = small(mul(B, H), 4)
    # The kernels return this point scaled by xi^2 and never form E = 3 b' Z^2 (a multiplication by a full-size constant):
    # with N = 9 Z^2, T = xi B - 3 N, S = xi B + 3 N:  xi^2 (X3, Y3, Z3) = (2 xi X Y T, S^2 - 12 N^2, 4 (xi B)(xi H))
    # (tools/kgen4.py: L1v4.r_dblstep).  Any representative of the projective point serves the following steps.
    N = small(C, 9)
    xB = mul_xi(B)
    T, S = sub(xB, small(N, 3)), add(xB, small(N, 3))
    scaled = (small(mul_xi(mul(mul(X, Y), T)), 2), sub(sqr(S), small(sqr(N), 12)), small(mul(xB, mul_xi(H)), 4))
    assert scaled == tuple(mul_xi(mul_xi(c)) for c in (X3, Y3, Z3))
    X3, Y3, Z3 = scaled
    L0 = sub(mul_xi(B), small(C, 9))     # (B - E) * xi = xi*B - 9*C
    L3 = mul_fq(H, Pt[1])
    L4 = neg(mul_fq(small(XX, 3), Pt[0]))
    return (X3, Y3, Z3), (L0, L3, L4), C


def add_step(Rp, Q, Pt):
    """Mixed addition R + Q (Q affine).  Returns (R+Q, (L2,L3,L5), lam) with
    line_affine(R,Q) * lam == L, lam = Z  (reference line: miller_loop_native.rs:10-28)."""
    X, Y, Z = Rp
    x2, y2 = Q
    theta = sub(Y, mul(y2, Z))
    mu = sub(X, mul(x2, Z))
    L2 = neg(mul_fq(mu, Pt[1]))
    L3 = mul_fq(theta, Pt[0])
    L5 = sub(mul(X, y2), mul(x2, Y))
    C = sqr(theta)
    D = sqr(mu)
    E = mul(mu, D)
    F = mul(Z, C)
    G = mul(X, D)
    H = sub(add(E, F), small(G, 2))
    X3 = mul(mu, H)
    Y3 = sub(mul(theta, sub(G, H)), mul(E, Y))
    Z3 = mul(Z, E)
    return (X3, Y3, Z3), (L2, L3, L5), Z


# ------------------------------------------------------------------ Fq12 (w-basis, 6 Fq2 coefficients)
def fq6_mul(a, b):
    a0, a1, a2 = a
    b0, b1, b2 = b
    t0, t1, t2 = mul(a0, b0), mul(a1, b1), mul(a2, b2)
    c0 = add(t0, mul_xi(sub(sub(mul(add(a1, a2), add(b1, b2)), t1), t2)))
    c1 = add(sub(sub(mul(add(a0, a1), add(b0, b1)), t0), t1), mul_xi(t2))
    c2 = add(sub(sub(mul(add(a0, a2), add(b0, b2)), t0), t2), t1)
    return (c0, c1, c2)


def fq6_add(a, b):
    return tuple(add(x, y) for x, y in zip(a, b))


def fq6_sub(a, b):
    return tuple(sub(x, y) for x, y in zip(a, b))


def fq6_mul_v(a):
    return (mul_xi(a[2]), a[0], a[1])


def split(f):  # f = list of 6 Fq2 (w-basis) -> (A0, A1) with f = A0 + A1 w, v = w^2
    return (f[0], f[2], f[4]), (f[1], f[3], f[5])


def join(A0, A1):
    return [A0[0], A1[0], A0[1], A1[1], A0[2], A1[2]]


def fq12_mul(f, g):
    A0, A1 = split(f)
    B0, B1 = split(g)
    t0 = fq6_mul(A0, B0)
    t1 = fq6_mul(A1, B1)
    m = fq6_mul(fq6_add(A0, A1), fq6_add(B0, B1))
    return join(fq6_add(t0, fq6_mul_v(t1)), fq6_sub(fq6_sub(m, t0), t1))


def fq12_sqr(f):
    A0, A1 = split(f)
    t = fq6_mul(A0, A1)
    u = fq6_mul(fq6_add(A0, A1), fq6_add(A0, fq6_mul_v(A1)))
    return join(fq6_sub(fq6_sub(u, t), fq6_mul_v(t)), fq6_add(t, t))


def fq12_conj(f):
    return [f[0], neg(f[1]), f[2], neg(f[3]), f[4], neg(f[5])]


def cyclotomic_sqr(f):
    """Granger-Scott squaring for f in the cyclotomic subgroup (f^(p^6+1) = 1)."""
    z0, z4, z3, z2, z1, z5 = f[0], f[2], f[4], f[1], f[3], f[5]

    def fq4_sqr(a, b):  # (a + b y)^2, y^2 = xi
        t = mul(a, b)
        return sub(sub(mul(add(a, b), add(a, mul_xi(b))), t), mul_xi(t)), add(t, t)

    t0, t1 = fq4_sqr(z0, z1)
    t2, t3 = fq4_sqr(z2, z3)
    t4, t5 = fq4_sqr(z4, z5)
    z0 = add(small(sub(t0, z0), 2), t0)
    z1 = add(small(add(t1, z1), 2), t1)
    tmp = mul_xi(t5)
    z2 = add(small(add(tmp, z2), 2), tmp)
    z3 = add(small(sub(t4, z3), 2), t4)
    z4 = add(small(sub(t2, z4), 2), t2)
    z5 = add(small(add(t3, z5), 2), t3)
    return [z0, z2, z4, z1, z3, z5]


def mul_by_034(f, L):
    b0, b3, b4 = L
    a = f
    return [
        add(mul(a[0], b0), mul_xi(add(mul(a[3], b3), mul(a[2], b4)))),
        add(mul(a[1], b0), mul_xi(add(mul(a[4], b3), mul(a[3], b4)))),
        add(mul(a[2], b0), mul_xi(add(mul(a[5], b3), mul(a[4], b4)))),
        add(add(mul(a[3], b0), mul(a[0], b3)), mul_xi(mul(a[5], b4))),
        add(add(mul(a[4], b0), mul(a[1], b3)), mul(a[0], b4)),
        add(add(mul(a[5], b0), mul(a[2], b3)), mul(a[1], b4)),
    ]


def mul_by_235(f, L):
    b2, b3, b5 = L
    a = f
    return [
        mul_xi(add(add(mul(a[4], b2), mul(a[3], b3)), mul(a[1], b5))),
        mul_xi(add(add(mul(a[5], b2), mul(a[4], b3)), mul(a[2], b5))),
        add(mul(a[0], b2), mul_xi(add(mul(a[5], b3), mul(a[3], b5)))),
        add(add(mul(a[1], b2), mul(a[0], b3)), mul_xi(mul(a[4], b5))),
        add(add(mul(a[2], b2), mul(a[1], b3)), mul_xi(mul(a[5], b5))),
        add(add(mul(a[3], b2), mul(a[2], b3)), mul(a[0], b5)),
    ]


def to_list(f):   # 6 Fq2 -> MyFq12 coeffs[12]
    return R.fq12_from_fp2s(f)


def from_list(a):
    return R.fq12_to_fp2s(a)


# ------------------------------------------------------------------ Miller loop (GPU schedule)
C2, C3 = R._end_constants()


def miller_projective(pairs, track_scale=True):
    """Shared-f multi Miller loop over pairs [(P, Q)], k >= 1 (k = 1: miller_loop_native).
    Returns (f_proj, s) with f_proj = s * f_ref, s in Fq2 (s is None if not tracked)."""
    enc = R.SIX_U_PLUS_2_NAF
    k = len(pairs)
    Rs = [(Q[0], Q[1], R.FQ2_ONE) for (_, Q) in pairs]
    s = R.FQ2_ONE if track_scale else None
    f = None
    # i = 63: f = product of tangent lines at Q_j (Z = 1 -> lam = 1)
    for j, (Pt, Q) in enumerate(pairs):
        Rs[j], L, lam = dbl_step(Rs[j], Pt)
        f = [L[0], R.FQ2_ZERO, R.FQ2_ZERO, L[1], L[2], R.FQ2_ZERO] if f is None else mul_by_034(f, L)
    for i in range(63, -1, -1):
        if i != 63:
            f = fq12_sqr(f)
            if track_scale:
                s = sqr(s)
            for j, (Pt, Q) in enumerate(pairs):
                Rs[j], L, lam = dbl_step(Rs[j], Pt)
                f = mul_by_034(f, L)
                if track_scale:
                    s = mul(s, lam)
        if enc[i] != 0:
            for j, (Pt, Q) in enumerate(pairs):
                Qs = Q if enc[i] == 1 else (Q[0], neg(Q[1]))
                Rs[j], L, lam = add_step(Rs[j], Qs, Pt)
                f = mul_by_235(f, L)
                if track_scale:
                    s = mul(s, lam)
    for j, (Pt, Q) in enumerate(pairs):
        Q1 = (mul(C2, R.conjugate_fp2(Q[0])), mul(C3, R.conjugate_fp2(Q[1])))
        nQ2 = (mul(C2, R.conjugate_fp2(Q1[0])), mul(C3, R.neg_conjugate_fp2(Q1[1])))
        Rs[j], L, lam = add_step(Rs[j], Q1, Pt)
        f = mul_by_235(f, L)
        if track_scale:
            s = mul(s, lam)
        _, L, lam = add_step(Rs[j], nQ2, Pt)
        f = mul_by_235(f, L)
        if track_scale:
            s = mul(s, lam)
    return f, s


def miller_exact(pairs):
    f, s = miller_projective(pairs, True)
    si = R.fq2_inv(s)
    return to_list([mul(c, si) for c in f])


# ------------------------------------------------------------------ final exponentiation (GPU schedule)
FROB = {k: [R.fq2_pow(R.frob_coeffs(k), i) for i in range(6)] for k in (1, 2, 3)}


def frobenius(f, k):
    out = []
    for i in range(6):
        a = f[i]
        if k % 2:
            a = R.conjugate_fp2(a)
        out.append(mul(a, FROB[k][i]))
    return out


def fq6_inv(a):
    a0, a1, a2 = a
    t0 = sub(sqr(a0), mul_xi(mul(a1, a2)))
    t1 = sub(mul_xi(sqr(a2)), mul(a0, a1))
    t2 = sub(sqr(a1), mul(a0, a2))
    n = add(mul(a0, t0), mul_xi(add(mul(a2, t1), mul(a1, t2))))
    ni = R.fq2_inv(n)
    return (mul(t0, ni), mul(t1, ni), mul(t2, ni))


def fq12_inv(f):
    A0, A1 = split(f)
    d = fq6_sub(fq6_mul(A0, A0), fq6_mul_v(fq6_mul(A1, A1)))
    di = fq6_inv(d)
    r0 = fq6_mul(A0, di)
    r1 = fq6_mul(A1, di)
    return join(r0, tuple(neg(x) for x in r1))


BN_X_NAF = R.get_naf([R.BN_X])


def pow_x_cyclotomic(a):
    naf = BN_X_NAF
    top = len(naf) - 1
    while naf[top] == 0:
        top -= 1
    assert naf[top] == 1
    res = a
    ac = fq12_conj(a)
    for i in range(top - 1, -1, -1):
        res = cyclotomic_sqr(res)
        if naf[i] == 1:
            res = fq12_mul(res, a)
        elif naf[i] == -1:
            res = fq12_mul(res, ac)
    return res


def final_exp_gpu(fl):
    f = from_list(fl)
    # easy part: (conj(f)/f)^(p^2) * (conj(f)/f)
    f2 = fq12_mul(fq12_conj(f), fq12_inv(f))
    m = fq12_mul(frobenius(f2, 2), f2)
    # hard part (src/final_exp_native.rs:130-169) with cyclotomic pow
    mp, mp2, mp3 = frobenius(m, 1), frobenius(m, 2), frobenius(m, 3)
    y0 = fq12_mul(mp, fq12_mul(mp2, mp3))
    y1 = fq12_conj(m)
    mx = pow_x_cyclotomic(m)
    mxp = frobenius(mx, 1)
    mx2 = pow_x_cyclotomic(mx)
    mx2p = frobenius(mx2, 1)
    y2 = frobenius(mx2, 2)
    y5 = fq12_conj(mx2)
    mx3 = pow_x_cyclotomic(mx2)
    mx3p = frobenius(mx3, 1)
    y3 = fq12_conj(mxp)
    y4 = fq12_conj(fq12_mul(mx, mx2p))
    y6 = fq12_conj(fq12_mul(mx3, mx3p))
    T0 = cyclotomic_sqr(y6)
    T0 = fq12_mul(T0, y4)
    T0 = fq12_mul(T0, y5)
    T1 = fq12_mul(y3, y5)
    T1 = fq12_mul(T1, T0)
    T0 = fq12_mul(y2, T0)
    T1 = cyclotomic_sqr(T1)
    T1 = fq12_mul(T1, T0)
    T1 = cyclotomic_sqr(T1)
    T0 = fq12_mul(T1, y1)
    T1 = fq12_mul(T1, y0)
    T0 = cyclotomic_sqr(T0)
    T0 = fq12_mul(T0, T1)
    return to_list(T0)


def pairing_gpu(Pt, Q):
    f, _ = miller_projective([(Pt, Q)], track_scale=False)
    return final_exp_gpu(to_list(f))
