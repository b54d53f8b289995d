// bn254_point_checks.h -- optional input validation (bn254_check_points_ex): the contract the reference gets from ark for free.
//
// `twisted_frobenius` / `neg_twisted_frobenius` build their results with `G2Affine::new(out_x, out_y)`
// (/root/reference/src/miller_loop_native.rs:303,311); ark-ec's `Affine::new` asserts that the point is on the curve AND in the
// prime-order subgroup, and `G1Affine::rand` / `G2Affine::rand` (src/pairing.rs:65-66) only ever produce such points.  So a caller of
// the reference cannot get a VALUE for a G2 point outside the r-torsion: the reference panics.  The pairing kernels never look (their
// projective steps assume the curve equation); a caller that wants the reference's behaviour for untrusted points runs this check
// first.  It is NOT part of the hot path and is written in plain HIP C++ on ark's own 4 x u64 Montgomery limbs (no conversion):
//
//   G1, G2 on the curve         y^2 = x^3 + 3          /   y^2 = x^3 + 3/(9+u)          (coordinates must also be < p)
//   G2 in the r-torsion         [x+1]Q + psi([x]Q) + psi^2([x]Q) == psi^3([2x]Q)        (El Housni - Guillevic - Piellard,
//                               ePrint 2022/348: one scalar multiplication by the 63-bit BN parameter x instead of [r]Q;
//                               psi = the untwist-Frobenius-twist endomorphism, the reference's `twisted_frobenius`, :298-304.
//                               Checked against [r]Q == O on subgroup and non-subgroup twist points by tests/test_point_checks.py
//                               through the big-int restatement.)   G1 has cofactor one: on the curve is in the subgroup.
//
// Cost (stated, not hidden): per G2 point 63 doublings + 24 mixed and 2 general additions in Jacobian coordinates = ~0.75 k Fq2 products,
// i.e. about an eighth of a pairing's field work, in compiler-scheduled 64-bit arithmetic (several times slower per product than the generated
// kernels).  MEASURED on MI355X (tools/exp/check_cost.py, 2^20 pairs resident): infinity 0.07 ms, + on-curve 0.29 ms, + subgroup
// 54.8 ms = 19.1 M pairs/s.  Since round 6 the subgroup criterion of bn254_check_points_ex runs on the generated kernel k_subcheck
// (tools/kgen4_prog.py: _subcheck_routines); this kernel keeps infinity / on-curve, marks the points the criterion does not apply to
// (CHECK_DEFER_SUBGROUP) and remains the whole check's portable form (BN254_CHECK_SUBGROUP_PORTABLE): the generated kernel's cross-check.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bn254_chk {

typedef unsigned __int128 u128;

struct Fq { uint64_t v[4]; };
struct Fq2 { Fq a, b; };               // a + b u
struct Jac { Fq2 x, y, z; };           // z == 0: the point at infinity

// p, -p^-1 mod 2^64 (SURVEY.md section 8 'common data facts'), Montgomery R = 2^256
#define CHK_P0 0x3c208c16d87cfd47ull
#define CHK_P1 0x97816a916871ca8dull
#define CHK_P2 0xb85045b68181585dull
#define CHK_P3 0x30644e72e131a029ull
#define CHK_INV 0x87d20782e4866389ull

__device__ __forceinline__ uint64_t plimb(int i) { return i == 0 ? CHK_P0 : i == 1 ? CHK_P1 : i == 2 ? CHK_P2 : CHK_P3; }

__device__ __forceinline__ bool fq_is_zero(const Fq& a) { return (a.v[0] | a.v[1] | a.v[2] | a.v[3]) == 0; }
__device__ __forceinline__ bool fq_eq(const Fq& a, const Fq& b) {
    return ((a.v[0] ^ b.v[0]) | (a.v[1] ^ b.v[1]) | (a.v[2] ^ b.v[2]) | (a.v[3] ^ b.v[3])) == 0;
}
__device__ __forceinline__ bool fq_geq_p(const Fq& a) {           // a >= p  (a non-canonical encoding)
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a.v[i] - plimb(i) - br;
        br = (uint64_t)(d >> 64) & 1;
    }
    return br == 0;
}
__device__ __forceinline__ Fq fq_csub_p(const Fq& a) {            // a - p if a >= p
    Fq r;
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a.v[i] - plimb(i) - br;
        r.v[i] = (uint64_t)d;
        br = (uint64_t)(d >> 64) & 1;
    }
    return br ? a : r;
}
__device__ __forceinline__ Fq fq_add(const Fq& a, const Fq& b) {  // p < 2^254: the sum of two reduced values fits 256 bits
    Fq r;
    u128 c = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        c += (u128)a.v[i] + b.v[i];
        r.v[i] = (uint64_t)c;
        c >>= 64;
    }
    return fq_csub_p(r);
}
__device__ __forceinline__ Fq fq_sub(const Fq& a, const Fq& b) {
    Fq r;
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a.v[i] - b.v[i] - br;
        r.v[i] = (uint64_t)d;
        br = (uint64_t)(d >> 64) & 1;
    }
    if (br) {
        u128 c = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            c += (u128)r.v[i] + plimb(i);
            r.v[i] = (uint64_t)c;
            c >>= 64;
        }
    }
    return r;
}
__device__ __forceinline__ Fq fq_neg(const Fq& a) {
    Fq z = {{0, 0, 0, 0}};
    return fq_is_zero(a) ? a : fq_sub(z, a);
}
// Montgomery product (CIOS), inputs < p, result < p
__device__ __noinline__ Fq fq_mul(const Fq& a, const Fq& b) {
    uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        u128 c = (u128)a.v[0] * b.v[i] + t0;
        t0 = (uint64_t)c; c >>= 64;
        c += (u128)a.v[1] * b.v[i] + t1; t1 = (uint64_t)c; c >>= 64;
        c += (u128)a.v[2] * b.v[i] + t2; t2 = (uint64_t)c; c >>= 64;
        c += (u128)a.v[3] * b.v[i] + t3; t3 = (uint64_t)c; c >>= 64;
        t4 += (uint64_t)c;
        uint64_t m = t0 * CHK_INV;
        c = (u128)m * CHK_P0 + t0; c >>= 64;
        c += (u128)m * CHK_P1 + t1; t0 = (uint64_t)c; c >>= 64;
        c += (u128)m * CHK_P2 + t2; t1 = (uint64_t)c; c >>= 64;
        c += (u128)m * CHK_P3 + t3; t2 = (uint64_t)c; c >>= 64;
        c += t4; t3 = (uint64_t)c; t4 = (uint64_t)(c >> 64);
    }
    Fq r = {{t0, t1, t2, t3}};
    return fq_csub_p(r);
}

__device__ __forceinline__ Fq2 f2_add(const Fq2& x, const Fq2& y) { return {fq_add(x.a, y.a), fq_add(x.b, y.b)}; }
__device__ __forceinline__ Fq2 f2_sub(const Fq2& x, const Fq2& y) { return {fq_sub(x.a, y.a), fq_sub(x.b, y.b)}; }
__device__ __forceinline__ Fq2 f2_dbl(const Fq2& x) { return f2_add(x, x); }
__device__ __forceinline__ Fq2 f2_conj(const Fq2& x) { return {x.a, fq_neg(x.b)}; }
__device__ __forceinline__ bool f2_is_zero(const Fq2& x) { return fq_is_zero(x.a) && fq_is_zero(x.b); }
__device__ __forceinline__ bool f2_eq(const Fq2& x, const Fq2& y) { return fq_eq(x.a, y.a) && fq_eq(x.b, y.b); }
__device__ __noinline__ Fq2 f2_mul(const Fq2& x, const Fq2& y) {
    Fq aa = fq_mul(x.a, y.a), bb = fq_mul(x.b, y.b);
    Fq s = fq_mul(fq_add(x.a, x.b), fq_add(y.a, y.b));
    return {fq_sub(aa, bb), fq_sub(fq_sub(s, aa), bb)};
}
__device__ __noinline__ Fq2 f2_sqr(const Fq2& x) {
    Fq t = fq_mul(x.a, x.b);
    return {fq_mul(fq_add(x.a, x.b), fq_sub(x.a, x.b)), fq_add(t, t)};
}

// Montgomery forms of the constants (tools/gen_consts.py writes them into bn254_consts_gen.h)
__device__ __forceinline__ Fq fq_const(const uint64_t (&w)[4]) { return {{w[0], w[1], w[2], w[3]}}; }

__device__ __noinline__ Jac jac_dbl(const Jac& p) {                 // dbl-2009-l (a = 0); infinity (z = 0) stays infinity
    Fq2 A = f2_sqr(p.x), B = f2_sqr(p.y), C = f2_sqr(B);
    Fq2 D = f2_dbl(f2_sub(f2_sub(f2_sqr(f2_add(p.x, B)), A), C));
    Fq2 E = f2_add(f2_dbl(A), A), F = f2_sqr(E);
    Jac r;
    r.x = f2_sub(F, f2_dbl(D));
    Fq2 C8 = f2_dbl(f2_dbl(f2_dbl(C)));
    r.y = f2_sub(f2_mul(E, f2_sub(D, r.x)), C8);
    r.z = f2_dbl(f2_mul(p.y, p.z));
    return r;
}
// general addition (add-2007-bl) with every exceptional case: either operand infinite, equal points (-> doubling), opposite points
__device__ __noinline__ Jac jac_add(const Jac& p, const Jac& q) {
    if (f2_is_zero(p.z)) return q;
    if (f2_is_zero(q.z)) return p;
    Fq2 Z1Z1 = f2_sqr(p.z), Z2Z2 = f2_sqr(q.z);
    Fq2 U1 = f2_mul(p.x, Z2Z2), U2 = f2_mul(q.x, Z1Z1);
    Fq2 S1 = f2_mul(f2_mul(p.y, q.z), Z2Z2), S2 = f2_mul(f2_mul(q.y, p.z), Z1Z1);
    Fq2 H = f2_sub(U2, U1), rr = f2_dbl(f2_sub(S2, S1));
    if (f2_is_zero(H)) {
        if (f2_is_zero(rr)) return jac_dbl(p);
        Jac inf = {p.x, p.y, {{{0, 0, 0, 0}}, {{0, 0, 0, 0}}}};
        return inf;
    }
    Fq2 I = f2_sqr(f2_dbl(H)), J = f2_mul(H, I), V = f2_mul(U1, I);
    Jac r;
    r.x = f2_sub(f2_sub(f2_sqr(rr), J), f2_dbl(V));
    r.y = f2_sub(f2_mul(rr, f2_sub(V, r.x)), f2_dbl(f2_mul(S1, J)));
    r.z = f2_mul(f2_sub(f2_sub(f2_sqr(f2_add(p.z, q.z)), Z1Z1), Z2Z2), H);
    return r;
}
// mixed addition p + (qx, qy, 1) (madd-2007-bl: seven products and four squarings instead of eleven and five), same exceptional cases
__device__ __noinline__ Jac jac_madd(const Jac& p, const Fq2& qx, const Fq2& qy, const Fq2& one) {
    if (f2_is_zero(p.z)) return {qx, qy, one};
    Fq2 Z1Z1 = f2_sqr(p.z);
    Fq2 U2 = f2_mul(qx, Z1Z1), S2 = f2_mul(f2_mul(qy, p.z), Z1Z1);
    Fq2 H = f2_sub(U2, p.x), rr = f2_dbl(f2_sub(S2, p.y));
    if (f2_is_zero(H)) {
        if (f2_is_zero(rr)) return jac_dbl(p);
        Jac inf = {p.x, p.y, {{{0, 0, 0, 0}}, {{0, 0, 0, 0}}}};
        return inf;
    }
    Fq2 HH = f2_sqr(H), I = f2_dbl(f2_dbl(HH)), J = f2_mul(H, I), V = f2_mul(p.x, I);
    Jac r;
    r.x = f2_sub(f2_sub(f2_sqr(rr), J), f2_dbl(V));
    r.y = f2_sub(f2_mul(rr, f2_sub(V, r.x)), f2_dbl(f2_mul(p.y, J)));
    r.z = f2_sub(f2_sub(f2_sqr(f2_add(p.z, H)), Z1Z1), HH);
    return r;
}
// psi on Jacobian coordinates: (c2 conj(X), c3 conj(Y), conj(Z)),  c2 = xi^((p-1)/3), c3 = xi^((p-1)/2)
// (the reference's twisted_frobenius, miller_loop_native.rs:298-304, on x = X/Z^2, y = Y/Z^3: conjugation is a field automorphism)
__device__ __forceinline__ Jac jac_psi(const Jac& p, const Fq2& c2, const Fq2& c3) {
    return {f2_mul(f2_conj(p.x), c2), f2_mul(f2_conj(p.y), c3), f2_conj(p.z)};
}
__device__ __noinline__ bool jac_eq(const Jac& p, const Jac& q) {
    bool pi = f2_is_zero(p.z), qi = f2_is_zero(q.z);
    if (pi || qi) return pi && qi;
    Fq2 Z1Z1 = f2_sqr(p.z), Z2Z2 = f2_sqr(q.z);
    if (!f2_eq(f2_mul(p.x, Z2Z2), f2_mul(q.x, Z1Z1))) return false;
    return f2_eq(f2_mul(f2_mul(p.y, q.z), Z2Z2), f2_mul(f2_mul(q.y, p.z), Z1Z1));
}

// flag bits of the per-point verdict byte and of the stream's point-check status word
enum { PT_INFINITY = 2, PT_NOT_ON_CURVE = 4, PT_NOT_IN_SUBGROUP = 8,
       PT_SKIP_SUBGROUP = 0x80 };       // internal (never leaves the library): the G2 point is infinite or off the twist -- no subgroup verdict for it

// what to check (bn254_pairing.h: BN254_CHECK_*)
enum { CHECK_INFINITY = 1, CHECK_ON_CURVE = 2, CHECK_SUBGROUP = 4,
       CHECK_DEFER_SUBGROUP = 0x100 };  // internal: the criterion itself runs on the generated kernel (k_subcheck); mark the points it does not apply to

struct Consts {
    uint64_t one[4];        // R mod p
    uint64_t three[4];      // 3 R mod p                     (G1: b = 3)
    uint64_t twist_b[8];    // 3/(9+u)                        (G2: b' = 3/xi)
    uint64_t c2[8];         // xi^((p-1)/3)
    uint64_t c3[8];         // xi^((p-1)/2)
};

__device__ __forceinline__ Fq2 f2_const(const uint64_t (&w)[8]) {
    return {{{w[0], w[1], w[2], w[3]}}, {{w[4], w[5], w[6], w[7]}}};
}

__global__ void __launch_bounds__(64) k_check_points_ex(const uint64_t* __restrict__ g1, const uint64_t* __restrict__ g2, size_t n, int flags,
                                                        Consts K, uint8_t* __restrict__ per_point, int* __restrict__ status) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        Fq px, py;
        Fq2 qx, qy;
#pragma unroll
        for (int l = 0; l < 4; l++) {
            px.v[l] = g1[(size_t)(0 + l) * n + i];
            py.v[l] = g1[(size_t)(4 + l) * n + i];
            qx.a.v[l] = g2[(size_t)(0 + l) * n + i];
            qx.b.v[l] = g2[(size_t)(4 + l) * n + i];
            qy.a.v[l] = g2[(size_t)(8 + l) * n + i];
            qy.b.v[l] = g2[(size_t)(12 + l) * n + i];
        }
        int bad = 0;
        bool inf1 = fq_is_zero(px) && fq_is_zero(py), inf2 = f2_is_zero(qx) && f2_is_zero(qy);
        if ((flags & CHECK_INFINITY) && (inf1 || inf2)) bad |= PT_INFINITY;
        if (flags & (CHECK_ON_CURVE | CHECK_SUBGROUP)) {
            // an infinite point has been reported (or the caller does not ask): the curve equations are for finite points
            if (!inf1) {
                bool ok = !fq_geq_p(px) && !fq_geq_p(py);
                if (ok) ok = fq_eq(fq_mul(py, py), fq_add(fq_mul(fq_mul(px, px), px), fq_const(K.three)));
                if (!ok) bad |= PT_NOT_ON_CURVE;
            }
            if (!inf2) {
                bool ok = !fq_geq_p(qx.a) && !fq_geq_p(qx.b) && !fq_geq_p(qy.a) && !fq_geq_p(qy.b);
                if (ok) ok = f2_eq(f2_sqr(qy), f2_add(f2_mul(f2_sqr(qx), qx), f2_const(K.twist_b)));
                if (!ok) bad |= PT_NOT_ON_CURVE;
                else if (flags & CHECK_DEFER_SUBGROUP) {
                } else if (flags & CHECK_SUBGROUP) {
                    Fq2 c2 = f2_const(K.c2), c3 = f2_const(K.c3);
                    Jac Q = {qx, qy, {fq_const(K.one), {{0, 0, 0, 0}}}};
                    Jac a = Q;                                   // [x]Q, x = BN_X (final_exp_native.rs:15) by its non-adjacent form, top digit first:
                    // 63 digits, 24 of them non-zero (the binary form has 28 ones) -- 62 doublings, 23 MIXED additions of +-Q (Q is affine);
                    // tests/test_point_checks.py checks the masks against the number
                    const uint64_t X_NZ = 0x452a95544aa90a11ull, X_NEG = 0x0020815000200010ull;
                    const Fq2 nqy = {fq_neg(qy.a), fq_neg(qy.b)};
                    for (int bit = 61; bit >= 0; bit--) {
                        a = jac_dbl(a);
                        if ((X_NZ >> bit) & 1) a = jac_madd(a, qx, ((X_NEG >> bit) & 1) ? nqy : qy, Q.z);
                    }
                    Jac b = jac_psi(a, c2, c3);                  // psi([x]Q)
                    Jac lhs = jac_add(jac_add(jac_psi(b, c2, c3), b), jac_madd(a, qx, qy, Q.z));
                    Jac rhs = jac_dbl(jac_psi(jac_psi(b, c2, c3), c2, c3));          // psi^3([2x]Q)
                    if (!jac_eq(lhs, rhs)) bad |= PT_NOT_IN_SUBGROUP;
                }
            }
        }
        if (bad) atomicOr(status, bad);
        if ((flags & CHECK_DEFER_SUBGROUP) && (inf2 || (bad & PT_NOT_ON_CURVE))) bad |= PT_SKIP_SUBGROUP;
        if (per_point) per_point[i] = (uint8_t)bad;
    }
}

}  // namespace bn254_chk
