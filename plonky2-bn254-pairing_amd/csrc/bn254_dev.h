// bn254_dev.h -- gfx950 device code for the batched BN254 optimal-ate pairing
// (one pairing per lane).  Field layer: Fq = 8 x u32 Montgomery limbs (R = 2^256, the same
// bits as ark-ff's 4 x u64 `Fp.0.0`), Fq2 = Fq[u]/(u^2+1), Fq12 = Fq2[w]/(w^6 - (9+u)) held
// as six Fq2 coefficients of w^0..w^5 (the `MyFq12` layout of the reference:
// coeffs[i] + coeffs[i+6] u, src/miller_loop_native.rs:47-51,86-92).
//
// Storage model (per lane): "slots" of one Fq2 (64 B).  Slot ids below NLDS live in LDS
// ([slot][quarter][thread] uint4 -> conflict-free ds_read_b128 / ds_write_b128), ids >= NLDS
// live in a global scratch buffer with the same shape over the whole launch (coalesced
// dwordx4).  The running Fq12 accumulator f, the G2 point R and the line scale stay in LDS;
// cold Fq12 temporaries of the final exponentiation live in scratch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fq_asm_gen.h"
#include "bn254_consts_gen.h"

#define DEV __device__ __forceinline__
#define DEVNI __device__ __attribute__((noinline))

namespace bn254 {

constexpr int BLOCK = 256;   // 4 waves, one per SIMD; one workgroup per CU (LDS-limited)
constexpr int NLDS = 10;     // Fq2 slots per lane in LDS: 10 * 64 B * 256 lanes = 160 KiB
constexpr size_t LDS_BYTES = (size_t)NLDS * 64 * BLOCK;

struct Fq2 { u32x8 c0, c1; };

// ------------------------------------------------------------------ Fq (add/sub/neg; multiply is asm)
__device__ constexpr uint32_t PL[8] = BN254_FQ_P;
__device__ constexpr uint32_t ONE_L[8] = BN254_FQ_ONE;
__device__ constexpr uint32_t R2_L[8] = BN254_FQ_R2;

DEV u32x8 fq_zero() { u32x8 r = {0, 0, 0, 0, 0, 0, 0, 0}; return r; }
DEV u32x8 fq_one() {
    u32x8 r = {ONE_L[0], ONE_L[1], ONE_L[2], ONE_L[3], ONE_L[4], ONE_L[5], ONE_L[6], ONE_L[7]};
    return r;
}
DEV u32x8 fq_r2() {
    u32x8 r = {R2_L[0], R2_L[1], R2_L[2], R2_L[3], R2_L[4], R2_L[5], R2_L[6], R2_L[7]};
    return r;
}
DEV u32x8 fq_const(const uint32_t* c) {  // c in __constant__ memory (uniform -> scalar loads)
    u32x8 r = {c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]};
    return r;
}

// (a + b) mod p, a,b in [0,p)
DEV u32x8 fq_add(u32x8 a, u32x8 b) {
    uint32_t t[8], d[8];
    unsigned c = 0, br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { unsigned co; t[i] = __builtin_addc(a[i], b[i], c, &co); c = co; }
#pragma unroll
    for (int i = 0; i < 8; i++) { unsigned bo; d[i] = __builtin_subc(t[i], PL[i], br, &bo); br = bo; }
    u32x8 o;
#pragma unroll
    for (int i = 0; i < 8; i++) o[i] = br ? t[i] : d[i];
    return o;
}
// (a - b) mod p
DEV u32x8 fq_sub(u32x8 a, u32x8 b) {
    uint32_t t[8], d[8];
    unsigned c = 0, br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { unsigned bo; t[i] = __builtin_subc(a[i], b[i], br, &bo); br = bo; }
#pragma unroll
    for (int i = 0; i < 8; i++) { unsigned co; d[i] = __builtin_addc(t[i], PL[i], c, &co); c = co; }
    u32x8 o;
#pragma unroll
    for (int i = 0; i < 8; i++) o[i] = br ? d[i] : t[i];
    return o;
}
DEV bool fq_is_zero(u32x8 a) { return (a[0] | a[1] | a[2] | a[3] | a[4] | a[5] | a[6] | a[7]) == 0; }
DEV u32x8 fq_neg(u32x8 a) {  // p - a, and 0 -> 0
    uint32_t d[8];
    unsigned br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { unsigned bo; d[i] = __builtin_subc(PL[i], a[i], br, &bo); br = bo; }
    bool z = fq_is_zero(a);
    u32x8 o;
#pragma unroll
    for (int i = 0; i < 8; i++) o[i] = z ? 0u : d[i];
    return o;
}
DEV u32x8 fq_dbl(u32x8 a) { return fq_add(a, a); }
DEV u32x8 fq_mul(u32x8 a, u32x8 b) { return fq_mul_asm(a, b); }
DEV u32x8 fq_sqr(u32x8 a) { return fq_mul_asm(a, a); }

// a^(p-2): Fermat inversion, fixed exponent (no data-dependent branches).  0 -> 0.
DEVNI u32x8 fq_inv(u32x8 a) {
    u32x8 r = a;  // top bit of p-2 (bit 253) is set
    for (int i = 252; i >= 0; --i) {
        r = fq_mul_asm(r, r);
        if ((BN254_P_MINUS_2[i >> 5] >> (i & 31)) & 1) r = fq_mul_asm(r, a);
    }
    return r;
}

// ------------------------------------------------------------------ Fq2
DEV Fq2 fq2_zero() { Fq2 r; r.c0 = fq_zero(); r.c1 = fq_zero(); return r; }
DEV Fq2 fq2_one() { Fq2 r; r.c0 = fq_one(); r.c1 = fq_zero(); return r; }
DEV Fq2 fq2_const(const uint32_t (*c)[8]) { Fq2 r; r.c0 = fq_const(c[0]); r.c1 = fq_const(c[1]); return r; }
DEV Fq2 fq2_add(Fq2 a, Fq2 b) { Fq2 r; r.c0 = fq_add(a.c0, b.c0); r.c1 = fq_add(a.c1, b.c1); return r; }
DEV Fq2 fq2_sub(Fq2 a, Fq2 b) { Fq2 r; r.c0 = fq_sub(a.c0, b.c0); r.c1 = fq_sub(a.c1, b.c1); return r; }
DEV Fq2 fq2_neg(Fq2 a) { Fq2 r; r.c0 = fq_neg(a.c0); r.c1 = fq_neg(a.c1); return r; }
DEV Fq2 fq2_dbl(Fq2 a) { return fq2_add(a, a); }
DEV Fq2 fq2_conj(Fq2 a) { Fq2 r; r.c0 = a.c0; r.c1 = fq_neg(a.c1); return r; }       // conjugate_fp2, miller_loop_native.rs:284
DEV Fq2 fq2_neg_conj(Fq2 a) { Fq2 r; r.c0 = fq_neg(a.c0); r.c1 = a.c1; return r; }   // neg_conjugate_fp2, :291
DEV Fq2 fq2_mul(Fq2 a, Fq2 b) { Fq2V v = fq2_mul_asm(a.c0, a.c1, b.c0, b.c1); Fq2 r; r.c0 = v.c0; r.c1 = v.c1; return r; }
DEV Fq2 fq2_sqr(Fq2 a) { Fq2V v = fq2_sqr_asm(a.c0, a.c1); Fq2 r; r.c0 = v.c0; r.c1 = v.c1; return r; }
DEV Fq2 fq2_mul_fq(Fq2 a, u32x8 k) { Fq2V v = fq2_mul_fq_asm(a.c0, a.c1, k); Fq2 r; r.c0 = v.c0; r.c1 = v.c1; return r; }
// (9 + u) * a = (9 a0 - a1) + (a0 + 9 a1) u
DEV Fq2 fq2_mul_xi(Fq2 a) {
    u32x8 t0 = fq_dbl(fq_dbl(fq_dbl(a.c0)));
    u32x8 t1 = fq_dbl(fq_dbl(fq_dbl(a.c1)));
    Fq2 r;
    r.c0 = fq_sub(fq_add(t0, a.c0), a.c1);
    r.c1 = fq_add(fq_add(t1, a.c1), a.c0);
    return r;
}
DEV bool fq2_is_zero(Fq2 a) { return fq_is_zero(a.c0) && fq_is_zero(a.c1); }
// 1/a = conj(a) / (a0^2 + a1^2).  0 -> 0 (callers flag the zero divisor).
DEV Fq2 fq2_inv(Fq2 a) {
    u32x8 n = fq_add(fq_sqr(a.c0), fq_sqr(a.c1));
    u32x8 ni = fq_inv(n);
    Fq2 r;
    r.c0 = fq_mul(a.c0, ni);
    r.c1 = fq_neg(fq_mul(a.c1, ni));
    return r;
}

// ------------------------------------------------------------------ slots
struct Slots {
    uint4* lds;        // this lane's LDS base (lds_mem + threadIdx.x); quarter stride = BLOCK
    uint4* g;          // this lane's scratch base (scratch + lane id); quarter stride = gstride
    uint32_t gstride;  // lanes in the launch (padded to BLOCK)
};

DEV uint4 pack4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { uint4 r; r.x = a; r.y = b; r.z = c; r.w = d; return r; }

DEV Fq2 ld(const Slots& S, int s) {
    uint4 q0, q1, q2, q3;
    if (s < NLDS) {
        const uint4* p = S.lds + (size_t)s * 4 * BLOCK;
        q0 = p[0]; q1 = p[BLOCK]; q2 = p[2 * BLOCK]; q3 = p[3 * BLOCK];
    } else {
        const uint4* p = S.g + (size_t)(s - NLDS) * 4 * S.gstride;
        q0 = p[0]; q1 = p[S.gstride]; q2 = p[2 * (size_t)S.gstride]; q3 = p[3 * (size_t)S.gstride];
    }
    Fq2 r;
    r.c0 = (u32x8){q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
    r.c1 = (u32x8){q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
    return r;
}
DEV void st(const Slots& S, int s, Fq2 v) {
    uint4 q0 = pack4(v.c0[0], v.c0[1], v.c0[2], v.c0[3]), q1 = pack4(v.c0[4], v.c0[5], v.c0[6], v.c0[7]);
    uint4 q2 = pack4(v.c1[0], v.c1[1], v.c1[2], v.c1[3]), q3 = pack4(v.c1[4], v.c1[5], v.c1[6], v.c1[7]);
    if (s < NLDS) {
        uint4* p = S.lds + (size_t)s * 4 * BLOCK;
        p[0] = q0; p[BLOCK] = q1; p[2 * BLOCK] = q2; p[3 * BLOCK] = q3;
    } else {
        uint4* p = S.g + (size_t)(s - NLDS) * 4 * S.gstride;
        p[0] = q0; p[S.gstride] = q1; p[2 * (size_t)S.gstride] = q2; p[3 * (size_t)S.gstride] = q3;
    }
}

// ------------------------------------------------------------------ global I/O (u64 SoA, limb-major)
// elem(c, l, i) = buf[(c*4 + l)*n + i]   (include/bn254_pairing.h)
DEV u32x8 load_fq_soa(const uint64_t* buf, size_t n, size_t i, int c) {
    u32x8 r;
#pragma unroll
    for (int l = 0; l < 4; l++) {
        uint64_t w = buf[((size_t)c * 4 + l) * n + i];
        r[2 * l] = (uint32_t)w; r[2 * l + 1] = (uint32_t)(w >> 32);
    }
    return r;
}
DEV void store_fq_soa(uint64_t* buf, size_t n, size_t i, int c, u32x8 v) {
#pragma unroll
    for (int l = 0; l < 4; l++) buf[((size_t)c * 4 + l) * n + i] = (uint64_t)v[2 * l] | ((uint64_t)v[2 * l + 1] << 32);
}
DEV Fq2 load_fq2_soa(const uint64_t* buf, size_t n, size_t i, int c) {
    Fq2 r; r.c0 = load_fq_soa(buf, n, i, c); r.c1 = load_fq_soa(buf, n, i, c + 1); return r;
}
// MyFq12: coefficient k (w^k) = coeffs[k] + coeffs[k+6] u
DEV Fq2 load_fq12_coeff(const uint64_t* buf, size_t n, size_t i, int k) {
    Fq2 r; r.c0 = load_fq_soa(buf, n, i, k); r.c1 = load_fq_soa(buf, n, i, k + 6); return r;
}
DEV void store_fq12_coeff(uint64_t* buf, size_t n, size_t i, int k, Fq2 v) {
    store_fq_soa(buf, n, i, k, v.c0); store_fq_soa(buf, n, i, k + 6, v.c1);
}

// ------------------------------------------------------------------ Fq6 = Fq2[v]/(v^3 - xi): Karatsuba, operands by functor
template <class FA, class FB, class FC>
DEV void fq6_mul_t(FA A, FB B, FC C) {
    Fq2 v0 = fq2_mul(A(0), B(0));
    Fq2 v1 = fq2_mul(A(1), B(1));
    Fq2 v2 = fq2_mul(A(2), B(2));
    Fq2 t = fq2_mul(fq2_add(A(1), A(2)), fq2_add(B(1), B(2)));
    t = fq2_sub(fq2_sub(t, v1), v2);
    C(0, fq2_add(v0, fq2_mul_xi(t)));
    t = fq2_mul(fq2_add(A(0), A(1)), fq2_add(B(0), B(1)));
    t = fq2_sub(fq2_sub(t, v0), v1);
    C(1, fq2_add(t, fq2_mul_xi(v2)));
    t = fq2_mul(fq2_add(A(0), A(2)), fq2_add(B(0), B(2)));
    t = fq2_sub(fq2_sub(t, v0), v2);
    C(2, fq2_add(t, v1));
}

// ------------------------------------------------------------------ Fq12 over slots
// An Fq12 occupies six consecutive slots (w^0..w^5).  Even powers form A0 = (f0,f2,f4), odd
// powers A1 = (f1,f3,f5) of the tower Fq12 = Fq6[w]/(w^2 - v), v = w^2.
// `cj` conjugates on load (conjugate_fp12, final_exp_native.rs:171-181: odd coefficients negated).
DEV Fq2 ld12(const Slots& S, int base, int k, bool cj) {
    Fq2 v = ld(S, base + k);
    if (cj && (k & 1)) v = fq2_neg(v);
    return v;
}

// dst = a * b.  T: six temp slots, disjoint from a, b, dst.  dst may alias a or b.
DEVNI void fq12_mul(Slots S, int dst, int a, int b, int T, bool cja, bool cjb) {
    fq6_mul_t([&](int i) { return ld12(S, a, 2 * i, cja); }, [&](int i) { return ld12(S, b, 2 * i, cjb); },
              [&](int i, Fq2 v) { st(S, T + i, v); });
    fq6_mul_t([&](int i) { return ld12(S, a, 2 * i + 1, cja); }, [&](int i) { return ld12(S, b, 2 * i + 1, cjb); },
              [&](int i, Fq2 v) { st(S, T + 3 + i, v); });
    Fq2 m0, m1, m2;
    fq6_mul_t([&](int i) { return fq2_add(ld12(S, a, 2 * i, cja), ld12(S, a, 2 * i + 1, cja)); },
              [&](int i) { return fq2_add(ld12(S, b, 2 * i, cjb), ld12(S, b, 2 * i + 1, cjb)); },
              [&](int i, Fq2 v) { if (i == 0) m0 = v; else if (i == 1) m1 = v; else m2 = v; });
    Fq2 t00 = ld(S, T + 0), t01 = ld(S, T + 1), t02 = ld(S, T + 2);
    Fq2 t10 = ld(S, T + 3), t11 = ld(S, T + 4), t12 = ld(S, T + 5);
    st(S, dst + 0, fq2_add(t00, fq2_mul_xi(t12)));
    st(S, dst + 2, fq2_add(t01, t10));
    st(S, dst + 4, fq2_add(t02, t11));
    st(S, dst + 1, fq2_sub(fq2_sub(m0, t00), t10));
    st(S, dst + 3, fq2_sub(fq2_sub(m1, t01), t11));
    st(S, dst + 5, fq2_sub(fq2_sub(m2, t02), t12));
}

// f = f^2 in place (complex squaring over Fq6).  T: three temp slots.
DEVNI void fq12_sqr(Slots S, int f, int T) {
    fq6_mul_t([&](int i) { return ld(S, f + 2 * i); }, [&](int i) { return ld(S, f + 2 * i + 1); },
              [&](int i, Fq2 v) { st(S, T + i, v); });
    Fq2 u0, u1, u2;
    fq6_mul_t([&](int i) { return fq2_add(ld(S, f + 2 * i), ld(S, f + 2 * i + 1)); },
              [&](int i) {  // A0 + v*A1 = (a0 + xi a5, a2 + a1, a4 + a3)
                  if (i == 0) return fq2_add(ld(S, f + 0), fq2_mul_xi(ld(S, f + 5)));
                  return fq2_add(ld(S, f + 2 * i), ld(S, f + 2 * i - 1));
              },
              [&](int i, Fq2 v) { if (i == 0) u0 = v; else if (i == 1) u1 = v; else u2 = v; });
    Fq2 t0 = ld(S, T + 0), t1 = ld(S, T + 1), t2 = ld(S, T + 2);
    st(S, f + 0, fq2_sub(fq2_sub(u0, t0), fq2_mul_xi(t2)));
    st(S, f + 2, fq2_sub(fq2_sub(u1, t1), t0));
    st(S, f + 4, fq2_sub(fq2_sub(u2, t2), t1));
    st(S, f + 1, fq2_dbl(t0));
    st(S, f + 3, fq2_dbl(t1));
    st(S, f + 5, fq2_dbl(t2));
}

// (a + b y)^2 in Fq4 = Fq2[y]/(y^2 - xi): returns (a^2 + xi b^2, 2ab)
DEV void fq4_sqr(Fq2 a, Fq2 b, Fq2& r0, Fq2& r1) {
    Fq2 t = fq2_mul(a, b);
    Fq2 s = fq2_mul(fq2_add(a, b), fq2_add(a, fq2_mul_xi(b)));
    r0 = fq2_sub(fq2_sub(s, t), fq2_mul_xi(t));
    r1 = fq2_dbl(t);
}

// Granger-Scott squaring, valid for f in the cyclotomic subgroup (f^(p^6+1) = 1).  In place.
DEVNI void fq12_cyc_sqr(Slots S, int f) {
    // z0=f0 z4=f2 z3=f4 z2=f1 z1=f3 z5=f5
    Fq2 t0, t1, t2, t3, t4, t5;
    fq4_sqr(ld(S, f + 0), ld(S, f + 3), t0, t1);
    fq4_sqr(ld(S, f + 1), ld(S, f + 4), t2, t3);
    fq4_sqr(ld(S, f + 2), ld(S, f + 5), t4, t5);
    Fq2 z;
    z = fq2_sub(t0, ld(S, f + 0)); st(S, f + 0, fq2_add(fq2_dbl(z), t0));          // z0 = 3 t0 - 2 z0
    z = fq2_add(t1, ld(S, f + 3)); st(S, f + 3, fq2_add(fq2_dbl(z), t1));          // z1 = 3 t1 + 2 z1
    Fq2 x5 = fq2_mul_xi(t5);
    z = fq2_add(x5, ld(S, f + 1)); st(S, f + 1, fq2_add(fq2_dbl(z), x5));          // z2 = 3 xi t5 + 2 z2
    z = fq2_sub(t4, ld(S, f + 4)); st(S, f + 4, fq2_add(fq2_dbl(z), t4));          // z3 = 3 t4 - 2 z3
    z = fq2_sub(t2, ld(S, f + 2)); st(S, f + 2, fq2_add(fq2_dbl(z), t2));          // z4 = 3 t2 - 2 z4
    z = fq2_add(t3, ld(S, f + 5)); st(S, f + 5, fq2_add(fq2_dbl(z), t3));          // z5 = 3 t3 + 2 z5
}

DEV void fq12_copy(const Slots& S, int dst, int src, bool cj) {
    for (int k = 0; k < 6; k++) st(S, dst + k, ld12(S, src, k, cj));
}

// frobenius_map_native (final_exp_native.rs:17-54): dst_i = conj^power(a_i) * frob_coeffs(power)^i
DEVNI void fq12_frobenius(Slots S, int dst, int src, int power) {
    int pw = power % 12;
    for (int i = 0; i < 6; i++) {
        Fq2 a = ld(S, src + i);
        if (pw & 1) a = fq2_conj(a);
        int kind = BN254_FROB_KIND[pw][i];
        if (kind == 1) a = fq2_mul_fq(a, fq_const(BN254_FROB[pw][i][0]));
        else if (kind == 2) a = fq2_mul(a, fq2_const(BN254_FROB[pw][i]));
        st(S, dst + i, a);
    }
}

// Fq6 inverse (norm to Fq2), operands/outputs in registers
DEV void fq6_inv(Fq2 a0, Fq2 a1, Fq2 a2, Fq2& r0, Fq2& r1, Fq2& r2, bool& zero_div) {
    Fq2 t0 = fq2_sub(fq2_sqr(a0), fq2_mul_xi(fq2_mul(a1, a2)));
    Fq2 t1 = fq2_sub(fq2_mul_xi(fq2_sqr(a2)), fq2_mul(a0, a1));
    Fq2 t2 = fq2_sub(fq2_sqr(a1), fq2_mul(a0, a2));
    Fq2 n = fq2_add(fq2_mul(a0, t0), fq2_mul_xi(fq2_add(fq2_mul(a2, t1), fq2_mul(a1, t2))));
    zero_div = fq2_is_zero(n);
    Fq2 ni = fq2_inv(n);
    r0 = fq2_mul(t0, ni); r1 = fq2_mul(t1, ni); r2 = fq2_mul(t2, ni);
}

// dst = 1/a (ark `Fq12` inverse used by `/`, final_exp_native.rs:74,200).  T: six temp slots.
// Returns true when a == 0 (the reference panics there).  dst must not alias a.
DEVNI bool fq12_inv(Slots S, int dst, int a, int T) {
    // d = A0^2 - v A1^2
    fq6_mul_t([&](int i) { return ld(S, a + 2 * i); }, [&](int i) { return ld(S, a + 2 * i); },
              [&](int i, Fq2 v) { st(S, T + i, v); });
    fq6_mul_t([&](int i) { return ld(S, a + 2 * i + 1); }, [&](int i) { return ld(S, a + 2 * i + 1); },
              [&](int i, Fq2 v) { st(S, T + 3 + i, v); });
    Fq2 d0 = fq2_sub(ld(S, T + 0), fq2_mul_xi(ld(S, T + 5)));
    Fq2 d1 = fq2_sub(ld(S, T + 1), ld(S, T + 3));
    Fq2 d2 = fq2_sub(ld(S, T + 2), ld(S, T + 4));
    Fq2 i0, i1, i2;
    bool zero_div;
    fq6_inv(d0, d1, d2, i0, i1, i2, zero_div);
    st(S, T + 0, i0); st(S, T + 1, i1); st(S, T + 2, i2);
    fq6_mul_t([&](int i) { return ld(S, a + 2 * i); }, [&](int i) { return ld(S, T + i); },
              [&](int i, Fq2 v) { st(S, dst + 2 * i, v); });
    fq6_mul_t([&](int i) { return ld(S, a + 2 * i + 1); }, [&](int i) { return ld(S, T + i); },
              [&](int i, Fq2 v) { st(S, dst + 2 * i + 1, fq2_neg(v)); });
    return zero_div;
}

// ------------------------------------------------------------------ sparse line multiplications
// f *= L0 + L3 w^3 + L4 w^4   (tangent line shape, sparse_line_function_equal_native :30-44)
// T: three temp slots.
DEV void mul_by_034(const Slots& S, int f, int T, Fq2 b0, Fq2 b3, Fq2 b4) {
    Fq2 c;
    c = fq2_add(fq2_mul(ld(S, f + 0), b0), fq2_mul_xi(fq2_add(fq2_mul(ld(S, f + 3), b3), fq2_mul(ld(S, f + 2), b4)))); st(S, T + 0, c);
    c = fq2_add(fq2_mul(ld(S, f + 1), b0), fq2_mul_xi(fq2_add(fq2_mul(ld(S, f + 4), b3), fq2_mul(ld(S, f + 3), b4)))); st(S, T + 1, c);
    c = fq2_add(fq2_mul(ld(S, f + 2), b0), fq2_mul_xi(fq2_add(fq2_mul(ld(S, f + 5), b3), fq2_mul(ld(S, f + 4), b4)))); st(S, T + 2, c);
    Fq2 c3 = fq2_add(fq2_add(fq2_mul(ld(S, f + 3), b0), fq2_mul(ld(S, f + 0), b3)), fq2_mul_xi(fq2_mul(ld(S, f + 5), b4)));
    Fq2 c4 = fq2_add(fq2_add(fq2_mul(ld(S, f + 4), b0), fq2_mul(ld(S, f + 1), b3)), fq2_mul(ld(S, f + 0), b4));
    Fq2 c5 = fq2_add(fq2_add(fq2_mul(ld(S, f + 5), b0), fq2_mul(ld(S, f + 2), b3)), fq2_mul(ld(S, f + 1), b4));
    st(S, f + 3, c3); st(S, f + 4, c4); st(S, f + 5, c5);
    st(S, f + 0, ld(S, T + 0)); st(S, f + 1, ld(S, T + 1)); st(S, f + 2, ld(S, T + 2));
}
// f *= L2 w^2 + L3 w^3 + L5 w^5   (chord line shape, sparse_line_function_unequal_native :10-28)
DEV void mul_by_235(const Slots& S, int f, int T, Fq2 b2, Fq2 b3, Fq2 b5) {
    Fq2 c;
    c = fq2_mul_xi(fq2_add(fq2_add(fq2_mul(ld(S, f + 4), b2), fq2_mul(ld(S, f + 3), b3)), fq2_mul(ld(S, f + 1), b5))); st(S, T + 0, c);
    c = fq2_mul_xi(fq2_add(fq2_add(fq2_mul(ld(S, f + 5), b2), fq2_mul(ld(S, f + 4), b3)), fq2_mul(ld(S, f + 2), b5))); st(S, T + 1, c);
    c = fq2_add(fq2_mul(ld(S, f + 0), b2), fq2_mul_xi(fq2_add(fq2_mul(ld(S, f + 5), b3), fq2_mul(ld(S, f + 3), b5)))); st(S, T + 2, c);
    Fq2 c3 = fq2_add(fq2_add(fq2_mul(ld(S, f + 1), b2), fq2_mul(ld(S, f + 0), b3)), fq2_mul_xi(fq2_mul(ld(S, f + 4), b5)));
    Fq2 c4 = fq2_add(fq2_add(fq2_mul(ld(S, f + 2), b2), fq2_mul(ld(S, f + 1), b3)), fq2_mul_xi(fq2_mul(ld(S, f + 5), b5)));
    Fq2 c5 = fq2_add(fq2_add(fq2_mul(ld(S, f + 3), b2), fq2_mul(ld(S, f + 2), b3)), fq2_mul(ld(S, f + 0), b5));
    st(S, f + 3, c3); st(S, f + 4, c4); st(S, f + 5, c5);
    st(S, f + 0, ld(S, T + 0)); st(S, f + 1, ld(S, T + 1)); st(S, f + 2, ld(S, T + 2));
}

// ------------------------------------------------------------------ G2 steps (homogeneous projective, inversion-free)
// R = (X,Y,Z) in slots r..r+2.  Line values are the reference's un-normalised affine line values
// times a known Fq2 factor lam (tangent: Z^2, chord: Z); when `sc` >= 0 the running product of
// those factors is kept in slot sc (s <- s * lam), so miller_loop_native's exact value can be
// recovered with one Fq2 inversion at the end (tests/sched_model.py states the algebra).

// Doubling step: R <- 2R; returns the tangent line (L0, L3, L4) at the OLD R evaluated at P.
DEV void dbl_step(const Slots& S, int r, int sc, bool sq_scale, u32x8 px, u32x8 py, Fq2& L0, Fq2& L3, Fq2& L4) {
    Fq2 X = ld(S, r), Y = ld(S, r + 1), Z = ld(S, r + 2);
    Fq2 B = fq2_sqr(Y);
    Fq2 C = fq2_sqr(Z);
    if (sc >= 0) {
        Fq2 s = ld(S, sc);
        if (sq_scale) s = fq2_sqr(s);
        st(S, sc, fq2_mul(s, C));
    }
    Fq2 E = fq2_mul(fq2_const(BN254_THREE_B), C);          // 3 b' Z^2
    Fq2 F = fq2_add(fq2_dbl(E), E);                        // 9 b' Z^2
    Fq2 H = fq2_dbl(fq2_mul(Y, Z));                        // 2 Y Z
    Fq2 XX = fq2_sqr(X);
    // line: L0 = xi*B - 9*C (= (B - E) xi), L3 = H * Py, L4 = -3 X^2 * Px
    Fq2 C8 = fq2_dbl(fq2_dbl(fq2_dbl(C)));
    L0 = fq2_sub(fq2_mul_xi(B), fq2_add(C8, C));
    L3 = fq2_mul_fq(H, py);
    L4 = fq2_neg(fq2_mul_fq(fq2_add(fq2_dbl(XX), XX), px));
    // point
    Fq2 X3 = fq2_mul(fq2_dbl(fq2_mul(X, Y)), fq2_sub(B, F));
    Fq2 BF = fq2_add(B, F);
    Fq2 E2 = fq2_sqr(E);
    Fq2 E2x3 = fq2_add(fq2_dbl(E2), E2);
    Fq2 Y3 = fq2_sub(fq2_sqr(BF), fq2_dbl(fq2_dbl(E2x3)));  // (B+F)^2 - 12 E^2
    Fq2 Z3 = fq2_dbl(fq2_dbl(fq2_mul(B, H)));               // 4 B H
    st(S, r, X3); st(S, r + 1, Y3); st(S, r + 2, Z3);
}

// Mixed addition R <- R + Q (Q affine); returns the chord line (L2, L3, L5) through the OLD R and Q.
// `update` false: line only (last step of the Miller loop).
DEV void add_step(const Slots& S, int r, int sc, Fq2 x2, Fq2 y2, u32x8 px, u32x8 py, bool update, Fq2& L2, Fq2& L3, Fq2& L5) {
    Fq2 X = ld(S, r), Y = ld(S, r + 1), Z = ld(S, r + 2);
    if (sc >= 0) st(S, sc, fq2_mul(ld(S, sc), Z));
    Fq2 theta = fq2_sub(Y, fq2_mul(y2, Z));
    Fq2 mu = fq2_sub(X, fq2_mul(x2, Z));
    L2 = fq2_neg(fq2_mul_fq(mu, py));
    L3 = fq2_mul_fq(theta, px);
    L5 = fq2_sub(fq2_mul(X, y2), fq2_mul(x2, Y));
    if (!update) return;
    Fq2 Cc = fq2_sqr(theta);
    Fq2 D = fq2_sqr(mu);
    Fq2 E = fq2_mul(mu, D);
    Fq2 F = fq2_mul(Z, Cc);
    Fq2 G = fq2_mul(X, D);
    Fq2 Hh = fq2_sub(fq2_add(E, F), fq2_dbl(G));
    Fq2 X3 = fq2_mul(mu, Hh);
    Fq2 Y3 = fq2_sub(fq2_mul(theta, fq2_sub(G, Hh)), fq2_mul(E, Y));
    Fq2 Z3 = fq2_mul(Z, E);
    st(S, r, X3); st(S, r + 1, Y3); st(S, r + 2, Z3);
}

}  // namespace bn254
