#!/usr/bin/env python3
"""independent_pairing.py -- a structurally independent statement of the BN254 optimal ate pairing (test infrastructure).

Everything the product, the C oracle and oracle/bn254_pyref.py share with the reference's control flow is ABSENT here:
  * no tower: Fp12 = Fp[w] / (w^12 - 18 w^6 + 82), one flat polynomial basis (w^6 = 9 + u, u^2 = -1);
  * no twist arithmetic: Q in E'(Fp2) is mapped once to E(Fp12): (x', y') -> (x' w^2, y' w^3), and every point operation
    is the textbook affine chord / tangent rule on y^2 = x^3 + 3 over Fp12;
  * no sparse lines: a line is the generic function  l(P) = (y_P - y_T) - lambda (x_P - x_T)  in Fp12;
  * no NAF: plain binary double-and-add over 6u + 2, then the two Frobenius steps with Q1 = pi(Q), Q2 = pi^2(Q) computed as
    honest p-th powers of the Fp12 coordinates;
  * the final exponentiation is one plain power by (p^12 - 1) / r.
The optimal ate pairing is unique as a function, so this must equal the reference's `pairing()` EXACTLY (exponent relation 1):
the line functions of /root/reference/src/miller_loop_native.rs:10-44 differ from the generic lines only by factors in proper
subfields of Fp12, which the final exponentiation kills.  Agreement therefore pins the conventions SURVEY.md marks [memory]:
the twist map, the MyFq12 <-> w-basis coefficient order and the twisted-Frobenius constants.

tests/test_independent_pin.py compares it with the restatement (and thereby with every golden `pairing` vector).
"""
P = 21888242871839275222246405745257275088696311157297823662689037894645226208583
R_ORDER = 21888242871839275222246405745257275088548364400416034343698204186575808495617
U = 4965661367192848881
LOOP = 6 * U + 2
DEG = 12


def pmul(a, b):
    """product in Fp[w] / (w^12 - 18 w^6 + 82)"""
    t = [0] * (2 * DEG - 1)
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                t[i + j] += x * y
    for k in range(2 * DEG - 2, DEG - 1, -1):          # w^k = 18 w^(k-6) - 82 w^(k-12)
        c = t[k]
        if c:
            t[k - 6] += 18 * c
            t[k - 12] -= 82 * c
    return [x % P for x in t[:DEG]]


def padd(a, b):
    return [(x + y) % P for x, y in zip(a, b)]


def psub(a, b):
    return [(x - y) % P for x, y in zip(a, b)]


def pscal(a, k):
    return [x * k % P for x in a]


ONE = [1] + [0] * (DEG - 1)
ZERO = [0] * DEG


def ppow(a, e):
    r, base = ONE, a
    while e:
        if e & 1:
            r = pmul(r, base)
        base = pmul(base, base)
        e >>= 1
    return r


def pinv(a):
    """inverse by solving the 12 x 12 linear system  a * x = 1  over Fp (Gauss-Jordan)"""
    cols = []
    for j in range(DEG):                                 # column j = a * w^j
        e = [0] * DEG
        e[j] = 1
        cols.append(pmul(a, e))
    m = [[cols[j][i] for j in range(DEG)] + [1 if i == 0 else 0] for i in range(DEG)]
    for c in range(DEG):
        piv = next(r for r in range(c, DEG) if m[r][c])
        m[c], m[piv] = m[piv], m[c]
        inv = pow(m[c][c], -1, P)
        m[c] = [x * inv % P for x in m[c]]
        for r in range(DEG):
            if r != c and m[r][c]:
                f = m[r][c]
                m[r] = [(x - f * y) % P for x, y in zip(m[r], m[c])]
    return [m[i][DEG] for i in range(DEG)]


def fp2_embed(c0, c1):
    """c0 + c1 u  with  u = w^6 - 9"""
    e = [0] * DEG
    e[0] = (c0 - 9 * c1) % P
    e[6] = c1 % P
    return e


def untwist(q):
    """E'(Fp2): y^2 = x^3 + 3/(9+u)  ->  E(Fp12): y^2 = x^3 + 3 ;  (x, y) -> (x w^2, y w^3)"""
    (x0, x1), (y0, y1) = q
    w2 = [0, 0, 1] + [0] * 9
    w3 = [0, 0, 0, 1] + [0] * 8
    return (pmul(fp2_embed(x0, x1), w2), pmul(fp2_embed(y0, y1), w3))


def on_curve(pt):
    x, y = pt
    return psub(pmul(y, y), pmul(pmul(x, x), x)) == [3] + [0] * (DEG - 1)


def slope(p1, p2):
    (x1, y1), (x2, y2) = p1, p2
    if x1 != x2:
        return pmul(psub(y2, y1), pinv(psub(x2, x1)))
    assert y1 == y2
    return pmul(pscal(pmul(x1, x1), 3), pinv(pscal(y1, 2)))


def add(p1, p2):
    lam = slope(p1, p2)
    x3 = psub(psub(pmul(lam, lam), p1[0]), p2[0])
    return (x3, psub(pmul(lam, psub(p1[0], x3)), p1[1]))


def line(p1, p2, t):
    """the line through p1 and p2 (tangent if equal), evaluated at t"""
    lam = slope(p1, p2)
    return psub(psub(t[1], p1[1]), pmul(lam, psub(t[0], p1[0])))


def frob_point(pt):
    return (ppow(pt[0], P), ppow(pt[1], P))


def optimal_ate(g1, g2):
    """g1 = (x, y) in E(Fp), g2 = ((x0, x1), (y0, y1)) in E'(Fp2)  ->  12 coefficients of the w-basis"""
    Pt = ([g1[0] % P] + [0] * 11, [g1[1] % P] + [0] * 11)
    Q = untwist(g2)
    assert on_curve(Pt) and on_curve(Q)
    T, f = Q, ONE
    for bit in bin(LOOP)[3:]:
        f = pmul(pmul(f, f), line(T, T, Pt))
        T = add(T, T)
        if bit == "1":
            f = pmul(f, line(T, Q, Pt))
            T = add(T, Q)
    Q1 = frob_point(Q)
    Q2 = frob_point(Q1)
    nQ2 = (Q2[0], [(-c) % P for c in Q2[1]])
    f = pmul(f, line(T, Q1, Pt))
    T = add(T, Q1)
    f = pmul(f, line(T, nQ2, Pt))
    return ppow(f, (P ** 12 - 1) // R_ORDER)


def to_myfq12(a):
    """w-basis coefficients a_0..a_11 over Fp  ->  MyFq12.coeffs (coeffs[i] + coeffs[i+6] u is the coefficient of w^i)"""
    return [(a[i] + 9 * a[i + 6]) % P for i in range(6)] + [a[i + 6] for i in range(6)]


if __name__ == "__main__":
    G2 = ((10857046999023057135944570762232829481370756359578518086990519993285655852781,
           11559732032986387107991004021392285783925812861821192530917403151452391805634),
          (8495653923123431417604973247489272438418190587263600148770280649306958101930,
           4082367875863433681332203403145435568316851327593401208105741076214120093531))
    for c in to_myfq12(optimal_ate((1, 2), G2)):
        print(c)
