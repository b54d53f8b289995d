// bn254_kernels.hip -- kernels + extern "C" ABI (include/bn254_pairing.h).
//
// One pairing per lane; 256-thread workgroups (4 waves, one per SIMD), one workgroup per CU
// (the per-lane Fq12 accumulator, the G2 point and the line scale fill the CU's 160 KiB LDS).
// Kernels are persistent over 256-lane work items: grid = min(#items, #CUs), each workgroup
// walks items blockIdx.x, blockIdx.x + gridDim.x, ... so the global scratch is sized by the
// grid, not by the batch.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <map>
#include <mutex>
#include <thread>
#include <vector>
#include "bn254_dev.h"
#include "pairing_asm_gen.h"
#include "../../include/bn254_pairing.h"

using namespace bn254;

// ------------------------------------------------------------------ slot map
namespace {
constexpr int SL_F = 0;                 // LDS 0..5  : running Fq12 accumulator
constexpr int SL_R = 6;                 // LDS 6..8  : G2 point R (single-pair Miller loop)
constexpr int SL_SC = 9;                // LDS 9     : line scale (exact miller_loop_native value)
constexpr int SL_GT = NLDS + 0;         // scratch 0..5   : temporaries of fq12_mul / sqr / inv
constexpr int SL_GA0 = NLDS + 6;        // scratch 6..53  : eight Fq12 registers
constexpr int SL_GR = NLDS + 54;        // scratch 54..   : R_j of the multi-pair Miller loop
constexpr int N_GSLOTS_BASE = 54;
DEV int SL_GA(int j) { return SL_GA0 + 6 * j; }

struct Lane {
    Slots S;
    size_t idx;    // element index (clamped)
    bool valid;
};

extern __shared__ uint4 lds_mem[];

DEV Slots make_slots(uint4* scratch, uint32_t gstride) {
    Slots S;
    S.lds = lds_mem + threadIdx.x;
    S.g = scratch + (size_t)blockIdx.x * BLOCK + threadIdx.x;
    S.gstride = gstride;
    return S;
}

// ------------------------------------------------------------------ Miller loop pieces (noinline: one copy of each in the code object)
DEVNI void miller_dbl_mul(Slots S, int r, int sc, bool sq_scale, u32x8 px, u32x8 py, bool first) {
    Fq2 L0, L3, L4;
    dbl_step(S, r, sc, sq_scale, px, py, L0, L3, L4);
    if (first) {
        st(S, SL_F + 0, L0); st(S, SL_F + 1, fq2_zero()); st(S, SL_F + 2, fq2_zero());
        st(S, SL_F + 3, L3); st(S, SL_F + 4, L4); st(S, SL_F + 5, fq2_zero());
    } else {
        mul_by_034(S, SL_F, SL_GT, L0, L3, L4);
    }
}
DEVNI void miller_add_mul(Slots S, int r, int sc, u32x8 x2c0, u32x8 x2c1, u32x8 y2c0, u32x8 y2c1, u32x8 px, u32x8 py, bool update) {
    Fq2 x2, y2, L2, L3, L5;
    x2.c0 = x2c0; x2.c1 = x2c1; y2.c0 = y2c0; y2.c1 = y2c1;
    add_step(S, r, sc, x2, y2, px, py, update, L2, L3, L5);
    mul_by_235(S, SL_F, SL_GT, L2, L3, L5);
}

// Shared-f Miller loop over k pairs (k = 1: miller_loop_native, :112-190; k > 1:
// multi_miller_loop_native, :192-282).  Result in LDS slots SL_F..SL_F+5.
// TRACK: keep the line scale and divide it out at the end (exact reference value).
template <bool TRACK>
DEV void miller_loop(const Slots& S, const uint64_t* g1, const uint64_t* g2, size_t n_pairs, size_t first_pair, int k) {
    const int sc = TRACK ? SL_SC : -1;
    auto rslot = [&](int j) { return k == 1 ? SL_R : SL_GR + 3 * j; };
    for (int j = 0; j < k; j++) {
        size_t e = first_pair + j;
        int r = rslot(j);
        st(S, r, load_fq2_soa(g2, n_pairs, e, 0));
        st(S, r + 1, load_fq2_soa(g2, n_pairs, e, 2));
        st(S, r + 2, fq2_one());
    }
    if (TRACK) st(S, SL_SC, fq2_one());
    // top NAF digit (index 64) is +1: R = Q; f = product of the tangent lines at Q_j (Z = 1 -> scale 1)
    for (int j = 0; j < k; j++) {
        size_t e = first_pair + j;
        miller_dbl_mul(S, rslot(j), -1, false, load_fq_soa(g1, n_pairs, e, 0), load_fq_soa(g1, n_pairs, e, 1), j == 0);
    }
    for (int i = 63; i >= 0; --i) {
        if (i != 63) {
            fq12_sqr(S, SL_F, SL_GT);
            for (int j = 0; j < k; j++) {
                size_t e = first_pair + j;
                miller_dbl_mul(S, rslot(j), sc, j == 0, load_fq_soa(g1, n_pairs, e, 0), load_fq_soa(g1, n_pairs, e, 1), false);
            }
        }
        int d = BN254_SIX_U_PLUS_2_NAF[i];
        if (d != 0) {
            for (int j = 0; j < k; j++) {
                size_t e = first_pair + j;
                Fq2 qx = load_fq2_soa(g2, n_pairs, e, 0), qy = load_fq2_soa(g2, n_pairs, e, 2);
                if (d < 0) qy = fq2_neg(qy);
                miller_add_mul(S, rslot(j), sc, qx.c0, qx.c1, qy.c0, qy.c1, load_fq_soa(g1, n_pairs, e, 0), load_fq_soa(g1, n_pairs, e, 1), true);
            }
        }
    }
    // Q1 = pi(Q), -Q2 = -pi^2(Q)   (twisted_frobenius / neg_twisted_frobenius, :298-312)
    for (int j = 0; j < k; j++) {
        size_t e = first_pair + j;
        Fq2 qx = load_fq2_soa(g2, n_pairs, e, 0), qy = load_fq2_soa(g2, n_pairs, e, 2);
        Fq2 c2 = fq2_const(BN254_TWIST_C2), c3 = fq2_const(BN254_TWIST_C3);
        Fq2 q1x = fq2_mul(c2, fq2_conj(qx)), q1y = fq2_mul(c3, fq2_conj(qy));
        u32x8 px = load_fq_soa(g1, n_pairs, e, 0), py = load_fq_soa(g1, n_pairs, e, 1);
        miller_add_mul(S, rslot(j), sc, q1x.c0, q1x.c1, q1y.c0, q1y.c1, px, py, true);
        Fq2 q2x = fq2_mul(c2, fq2_conj(q1x)), q2y = fq2_mul(c3, fq2_neg_conj(q1y));
        miller_add_mul(S, rslot(j), sc, q2x.c0, q2x.c1, q2y.c0, q2y.c1, px, py, false);
    }
    if (TRACK) {
        Fq2 si = fq2_inv(ld(S, SL_SC));
        for (int c = 0; c < 6; c++) st(S, SL_F + c, fq2_mul(ld(S, SL_F + c), si));
    }
}

// res (LDS SL_F) <- res^x, x = BN_X, for res in the cyclotomic subgroup; `base` holds the input.
// Same digits as pow_native (final_exp_native.rs:56-84); division by a unitary element = multiply
// by its conjugate, squarings are Granger-Scott: identical field elements.
DEV void pow_x_cyclotomic(const Slots& S, int base) {
    for (int i = BN254_X_NAF_LEN - 2; i >= 0; --i) {
        fq12_cyc_sqr(S, SL_F);
        int d = BN254_X_NAF[i];
        if (d != 0) fq12_mul(S, SL_F, SL_F, base, SL_GT, false, d < 0);
    }
}

// final_exp_native (final_exp_native.rs:209-213) on the Fq12 in LDS SL_F; result in SL_F.
// Returns true on a zero divisor (reference panics in easy_part, :200).
DEV bool final_exp(const Slots& S) {
    const int F = SL_F, T = SL_GT;
    const int G0 = SL_GA(0), GM = SL_GA(1), G2 = SL_GA(2), G3 = SL_GA(3), G4 = SL_GA(4), G5 = SL_GA(5), G6 = SL_GA(6), G7 = SL_GA(7);
    // easy part (:195-206): f2 = conj(a)/a ; m = frob(f2, 2) * f2
    bool zero_div = fq12_inv(S, G0, F, T);
    fq12_mul(S, F, F, G0, T, true, false);
    fq12_frobenius(S, G0, F, 2);
    fq12_mul(S, F, G0, F, T, false, false);
    // hard part (:130-169)
    fq12_copy(S, GM, F, false);                       // m
    fq12_frobenius(S, G2, F, 1);                      // mp
    fq12_frobenius(S, G3, F, 2);                      // mp2
    fq12_frobenius(S, G4, F, 3);                      // mp3
    fq12_mul(S, G3, G3, G4, T, false, false);         // mp2 * mp3
    fq12_mul(S, G2, G2, G3, T, false, false);         // y0 = mp * mp2_mp3
    pow_x_cyclotomic(S, GM);                          // F = mx
    fq12_copy(S, G3, F, false);                       // G3 = mx
    pow_x_cyclotomic(S, G3);                          // F = mx2
    fq12_copy(S, G4, F, false);                       // G4 = mx2
    pow_x_cyclotomic(S, G4);                          // F = mx3
    fq12_copy(S, G5, F, false);                       // G5 = mx3
    fq12_frobenius(S, G6, G3, 1);                     // G6 = mxp       (y3 = conj)
    fq12_frobenius(S, G7, G4, 1);                     // G7 = mx2p
    fq12_mul(S, G7, G3, G7, T, false, false);         // G7 = mx * mx2p (y4 = conj)
    fq12_frobenius(S, G3, G4, 2);                     // G3 = y2 = frob(mx2, 2)
    fq12_frobenius(S, F, G5, 1);                      // F = mx3p
    fq12_mul(S, F, G5, F, T, false, false);           // F = mx3 * mx3p (y6 = conj)
    fq12_copy(S, F, F, true);                         // F = y6
    fq12_cyc_sqr(S, F);                               // T0 = y6^2
    fq12_mul(S, F, F, G7, T, false, true);            // T0 *= y4
    fq12_mul(S, F, F, G4, T, false, true);            // T0 *= y5 (= conj mx2)
    fq12_mul(S, G5, G6, G4, T, true, true);           // T1 = y3 * y5
    fq12_mul(S, G5, G5, F, T, false, false);          // T1 *= T0
    fq12_mul(S, F, G3, F, T, false, false);           // T0 = y2 * T0
    fq12_cyc_sqr(S, G5);                              // T1 = T1^2
    fq12_mul(S, G5, G5, F, T, false, false);          // T1 *= T0
    fq12_cyc_sqr(S, G5);                              // T1 = T1^2
    fq12_mul(S, F, G5, GM, T, false, true);           // T0 = T1 * y1 (= conj m)
    fq12_mul(S, G5, G5, G2, T, false, false);         // T1 *= y0
    fq12_cyc_sqr(S, F);                               // T0 = T0^2
    fq12_mul(S, F, F, G5, T, false, false);           // T0 *= T1
    return zero_div;
}

DEV void load_fq12(const Slots& S, int base, const uint64_t* buf, size_t n, size_t i) {
    for (int k = 0; k < 6; k++) st(S, base + k, load_fq12_coeff(buf, n, i, k));
}
DEV void store_fq12(const Slots& S, int base, uint64_t* buf, size_t n, size_t i) {
    for (int k = 0; k < 6; k++) store_fq12_coeff(buf, n, i, k, ld(S, base + k));
}

// ------------------------------------------------------------------ kernels
// mode bit 0: Miller loop; bit 1: final exponentiation; TRACK only for Miller-only output.
template <bool DO_MILLER, bool DO_FEXP>
__global__ void __launch_bounds__(BLOCK, 1)
k_pairing(const uint64_t* g1, const uint64_t* g2, const uint64_t* f_in, uint64_t* out, size_t n_groups, int k,
          uint4* scratch, uint32_t gstride, int* status) {
    Slots S = make_slots(scratch, gstride);
    size_t n_items = (n_groups + BLOCK - 1) / BLOCK;
    for (size_t item = blockIdx.x; item < n_items; item += gridDim.x) {
        size_t e = item * BLOCK + threadIdx.x;
        bool valid = e < n_groups;
        size_t idx = valid ? e : n_groups - 1;
        if (DO_MILLER) {
            if (DO_FEXP) miller_loop<false>(S, g1, g2, n_groups * (size_t)k, idx * (size_t)k, k);
            else miller_loop<true>(S, g1, g2, n_groups * (size_t)k, idx * (size_t)k, k);
        } else {
            load_fq12(S, SL_F, f_in, n_groups, idx);
        }
        if (DO_FEXP) {
            bool zd = final_exp(S);
            if (zd && valid) atomicOr(status, 1);
        }
        if (valid) store_fq12(S, SL_F, out, n_groups, idx);
    }
}

// ------------------------------------------------------------------ v2: generated whole-kernel assembly (tools/kgen*.py)
// hipcc contributes the kernel descriptor and the argument SGPRs; the body is one asm statement.
#define BN254_ASM_KERNEL(NAME, BLOB)                                                                                       \
    __global__ void __launch_bounds__(BLOCK, 1)                                                                            \
    NAME(const uint64_t* g1, const uint64_t* g2, const uint64_t* f_in, uint64_t* out, uint32_t n, uint32_t k, uint4* scratch, \
         uint32_t gslot_stride, int* status) {                                                                             \
        uint32_t tid = threadIdx.x, bid = blockIdx.x, grid = gridDim.x;                                                    \
        asm volatile(BLOB                                                                                                  \
                     :                                                                                                     \
                     : "s"(g1), "s"(g2), "s"(f_in), "s"(out), "s"(n), "s"(k), "s"(scratch), "s"(gslot_stride), "s"(status), \
                       "v"(tid), "s"(bid), "s"(grid)                                                                       \
                     : BN254_ASM_CLOBBERS);                                                                                \
    }
// balanced signed radix-2^29 limbs, nine limbs (tools/kgen4*.py); 72-byte slots
BN254_ASM_KERNEL(k3_pairing, BN254_ASM_PAIRING)
BN254_ASM_KERNEL(k3_miller, BN254_ASM_MILLER)
BN254_ASM_KERNEL(k3_fexp, BN254_ASM_FEXP)
BN254_ASM_KERNEL(k3_mpairing, BN254_ASM_MPAIRING)   // k pairs per lane, shared f (multi_miller_loop_native) + final exp
BN254_ASM_KERNEL(k3_mmiller, BN254_ASM_MMILLER)     // k pairs per lane, exact multi_miller_loop_native value
BN254_ASM_KERNEL(k3_op, BN254_ASM_OP)               // batched helpers: MyFq12 Mul / frobenius_map_native / pow_native (k = op | power << 8 | naf_len << 16)
constexpr int V3_GSLOTS = BN254_GSLOTS;         // twelve Fq12 registers + eight overflow temporaries (+ 7 per pair in the multi kernels)
constexpr int V3_SLOT_BYTES = BN254_SLOT_BYTES;

enum { OP_MUL = 0, OP_FROB = 1, OP_POW = 2, OP_INV = 3, OP_SQR = 4, OP_CYC_SQR = 5 };

// Batched MyFq12 helpers: Mul, frobenius_map_native, pow_native (general: true inverse on -1 digits).
__global__ void __launch_bounds__(BLOCK, 1)
k_fq12_op(int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, size_t power, const int8_t* naf, int naf_len,
          uint4* scratch, uint32_t gstride, int* status) {
    Slots S = make_slots(scratch, gstride);
    size_t n_items = (n + BLOCK - 1) / BLOCK;
    for (size_t item = blockIdx.x; item < n_items; item += gridDim.x) {
        size_t e = item * BLOCK + threadIdx.x;
        bool valid = e < n;
        size_t idx = valid ? e : n - 1;
        load_fq12(S, SL_F, a, n, idx);
        if (op == OP_MUL) {
            load_fq12(S, SL_GA(0), b, n, idx);
            fq12_mul(S, SL_F, SL_F, SL_GA(0), SL_GT, false, false);
        } else if (op == OP_SQR) {
            fq12_sqr(S, SL_F, SL_GT);
        } else if (op == OP_CYC_SQR) {
            fq12_cyc_sqr(S, SL_F);
        } else if (op == OP_FROB) {
            fq12_frobenius(S, SL_GA(0), SL_F, (int)(power % 12));
            fq12_copy(S, SL_F, SL_GA(0), false);
        } else if (op == OP_INV) {
            bool zd = fq12_inv(S, SL_GA(0), SL_F, SL_GT);
            if (zd && valid) atomicOr(status, 1);
            fq12_copy(S, SL_F, SL_GA(0), false);
        } else if (op == OP_POW) {
            // pow_native (final_exp_native.rs:56-84): res = a; MSB-first over the NAF digits
            fq12_copy(S, SL_GA(0), SL_F, false);             // a
            bool need_inv = false;
            for (int t = 0; t < naf_len; t++) need_inv |= (naf[t] < 0);
            if (need_inv) {
                bool zd = fq12_inv(S, SL_GA(1), SL_GA(0), SL_GT);  // 1/a  (res / a == res * a^-1)
                if (zd && valid) atomicOr(status, 1);
            }
            bool started = false;
            for (int t = naf_len - 1; t >= 0; --t) {
                int z = naf[t];
                if (started) fq12_sqr(S, SL_F, SL_GT);
                if (z != 0) {
                    if (started) fq12_mul(S, SL_F, SL_F, z > 0 ? SL_GA(0) : SL_GA(1), SL_GT, false, false);
                    else started = true;
                }
            }
        }
        if (valid) store_fq12(S, SL_F, out, n, idx);
    }
}

// ------------------------------------------------------------------ synthetic subgroup points
DEV uint64_t splitmix64(uint64_t& st_) {
    st_ += 0x9E3779B97F4A7C15ull;
    uint64_t z = st_;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// y^2 = x^3 + 3 over Fq, homogeneous projective; same formulas as dbl_step/add_step (3b = 9)
DEV void g1_dbl(u32x8& X, u32x8& Y, u32x8& Z) {
    u32x8 B = fq_sqr(Y), C = fq_sqr(Z);
    u32x8 C8 = fq_dbl(fq_dbl(fq_dbl(C)));
    u32x8 E = fq_add(C8, C);                 // 9 C
    u32x8 F = fq_add(fq_dbl(E), E);
    u32x8 H = fq_dbl(fq_mul(Y, Z));
    u32x8 X3 = fq_mul(fq_dbl(fq_mul(X, Y)), fq_sub(B, F));
    u32x8 BF = fq_add(B, F), E2 = fq_sqr(E);
    u32x8 E2x3 = fq_add(fq_dbl(E2), E2);
    u32x8 Y3 = fq_sub(fq_sqr(BF), fq_dbl(fq_dbl(E2x3)));
    u32x8 Z3 = fq_dbl(fq_dbl(fq_mul(B, H)));
    X = X3; Y = Y3; Z = Z3;
}
DEV void g1_add_mixed(u32x8& X, u32x8& Y, u32x8& Z, u32x8 x2, u32x8 y2) {
    u32x8 theta = fq_sub(Y, fq_mul(y2, Z)), mu = fq_sub(X, fq_mul(x2, Z));
    u32x8 Cc = fq_sqr(theta), D = fq_sqr(mu), E = fq_mul(mu, D), F = fq_mul(Z, Cc), G = fq_mul(X, D);
    u32x8 Hh = fq_sub(fq_add(E, F), fq_dbl(G));
    u32x8 X3 = fq_mul(mu, Hh), Y3 = fq_sub(fq_mul(theta, fq_sub(G, Hh)), fq_mul(E, Y)), Z3 = fq_mul(Z, E);
    X = X3; Y = Y3; Z = Z3;
}

__global__ void __launch_bounds__(BLOCK, 1)
k_generate(uint64_t seed, uint64_t* g1_out, uint64_t* g2_out, size_t n, uint4* scratch, uint32_t gstride) {
    Slots S = make_slots(scratch, gstride);
    size_t n_items = (n + BLOCK - 1) / BLOCK;
    for (size_t item = blockIdx.x; item < n_items; item += gridDim.x) {
        size_t e = item * BLOCK + threadIdx.x;
        bool valid = e < n;
        size_t idx = valid ? e : n - 1;
        uint64_t sm = seed ^ (0xD1B54A32D192ED03ull * (idx + 1));
        uint64_t s_lo = splitmix64(sm), s_hi = splitmix64(sm) | (1ull << 63);
        uint64_t t_lo = splitmix64(sm), t_hi = splitmix64(sm) | (1ull << 63);
        // G1: [s] (1, 2)
        u32x8 gx = fq_const(BN254_G1_GEN[0]), gy = fq_const(BN254_G1_GEN[1]);
        u32x8 X = gx, Y = gy, Z = fq_one();
        for (int bit = 126; bit >= 0; --bit) {
            g1_dbl(X, Y, Z);
            uint64_t w = bit >= 64 ? s_hi : s_lo;
            if ((w >> (bit & 63)) & 1) g1_add_mixed(X, Y, Z, gx, gy);   // uniform trip count, divergent add
        }
        u32x8 zi = fq_inv(Z);
        if (valid) {
            store_fq_soa(g1_out, n, idx, 0, fq_mul(X, zi));
            store_fq_soa(g1_out, n, idx, 1, fq_mul(Y, zi));
        }
        // G2: [t] G2gen, via the Miller-loop point steps (lines discarded)
        Fq2 qx, qy;
        qx.c0 = fq_const(BN254_G2_GEN[0]); qx.c1 = fq_const(BN254_G2_GEN[1]);
        qy.c0 = fq_const(BN254_G2_GEN[2]); qy.c1 = fq_const(BN254_G2_GEN[3]);
        st(S, SL_R, qx); st(S, SL_R + 1, qy); st(S, SL_R + 2, fq2_one());
        u32x8 one = fq_one();
        for (int bit = 126; bit >= 0; --bit) {
            Fq2 l0, l1, l2;
            dbl_step(S, SL_R, -1, false, one, one, l0, l1, l2);
            uint64_t w = bit >= 64 ? t_hi : t_lo;
            if ((w >> (bit & 63)) & 1) add_step(S, SL_R, -1, qx, qy, one, one, true, l0, l1, l2);
        }
        Fq2 zi2 = fq2_inv(ld(S, SL_R + 2));
        Fq2 ax = fq2_mul(ld(S, SL_R), zi2), ay = fq2_mul(ld(S, SL_R + 1), zi2);
        if (valid) {
            store_fq_soa(g2_out, n, idx, 0, ax.c0); store_fq_soa(g2_out, n, idx, 1, ax.c1);
            store_fq_soa(g2_out, n, idx, 2, ay.c0); store_fq_soa(g2_out, n, idx, 3, ay.c1);
        }
    }
}

// verdict[i] = 1 iff Fq12 element i equals MyFq12::one (coeffs[0] = R mod p in ark's Montgomery limbs, the rest 0):
// the check pattern of final_exp_native.rs:245-263 (a Groth16-style product of pairings == 1), one byte per group.
__global__ void __launch_bounds__(256) k_is_one(const uint64_t* __restrict__ f, uint8_t* __restrict__ verdict, size_t n) {
    const uint64_t one[4] = {0xd35d438dc58f0d9dull, 0x0a78eb28f5c70b3dull, 0x666ea36f7879462cull, 0x0e0a77c19a07df2full};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t diff = 0;
        for (int c = 0; c < 12; c++)
            for (int l = 0; l < 4; l++) diff |= f[((size_t)c * 4 + l) * n + i] ^ (c == 0 ? one[l] : 0ull);
        verdict[i] = diff == 0 ? 1 : 0;
    }
}

// ------------------------------------------------------------------ host side
// Scratch and the status word are per (device, stream): calls on different streams of one device are independent
// ("library is re-entrant, one HIP stream per call").
struct StreamCtx {
    uint4* scratch = nullptr;
    size_t scratch_bytes = 0;
    int* status = nullptr;
    int8_t* naf = nullptr;     // device copy of NAF digits for pow
    size_t naf_cap = 0;
    uint64_t* tmp = nullptr;   // Fq12 staging of the == 1 verdict path
    size_t tmp_bytes = 0;
};
struct DeviceCtx {
    bool init = false;
    int n_cu = 0;
    std::map<hipStream_t, StreamCtx> streams;
};
std::mutex g_mu;
DeviceCtx g_ctx[64];

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { return BN254_ERR_HIP; } } while (0)

struct LaunchCtx {
    StreamCtx* s;
    int n_cu;
    uint4* scratch;
    int* status;
};

int ctx_get(int device, void* stream, size_t k, LaunchCtx* out, uint32_t* grid_out, size_t n_items) {
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) return BN254_ERR_NO_DEVICE;
    if (device < 0 || device >= cnt || device >= 64) return BN254_ERR_INVALID_ARG;
    HIPCHK(hipSetDevice(device));
    std::lock_guard<std::mutex> lk(g_mu);
    DeviceCtx& c = g_ctx[device];
    if (!c.init) {
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, device));
        c.n_cu = prop.multiProcessorCount;
        const void* kernels[] = {(const void*)k_pairing<true, true>, (const void*)k_pairing<true, false>, (const void*)k_pairing<false, true>,
                                 (const void*)k_fq12_op, (const void*)k_generate, (const void*)k3_pairing, (const void*)k3_miller, (const void*)k3_fexp,
                                 (const void*)k3_mpairing, (const void*)k3_mmiller, (const void*)k3_op};
        for (const void* f : kernels) HIPCHK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
        c.init = true;
    }
    StreamCtx& sc = c.streams[(hipStream_t)stream];
    if (!sc.status) {
        HIPCHK(hipMalloc(&sc.status, sizeof(int)));
        HIPCHK(hipMemsetAsync(sc.status, 0, sizeof(int), (hipStream_t)stream));
    }
    uint32_t grid = (uint32_t)(n_items < (size_t)c.n_cu ? n_items : (size_t)c.n_cu);
    if (grid == 0) grid = 1;
    size_t slots = N_GSLOTS_BASE + 3 * (k > 1 ? k : 0);
    size_t need = slots * 64 * (size_t)c.n_cu * BLOCK;   // sized for a full grid so the buffer is stable
    size_t need3 = (size_t)(V3_GSLOTS + 7 * (k > 1 ? k : 0)) * V3_SLOT_BYTES * (size_t)c.n_cu * BLOCK;
    if (need3 > need) need = need3;
    if (need > sc.scratch_bytes) {
        if (sc.scratch) { HIPCHK(hipStreamSynchronize((hipStream_t)stream)); HIPCHK(hipFree(sc.scratch)); sc.scratch = nullptr; sc.scratch_bytes = 0; }
        if (hipMalloc(&sc.scratch, need) != hipSuccess) return BN254_ERR_ALLOC;
        sc.scratch_bytes = need;
    }
    out->s = &sc;
    out->n_cu = c.n_cu;
    out->scratch = sc.scratch;
    out->status = sc.status;
    *grid_out = grid;
    return BN254_OK;
}

template <bool M, bool F>
int launch_pairing(const uint64_t* g1, const uint64_t* g2, const uint64_t* f_in, uint64_t* out, size_t n_groups, size_t k, int device, void* stream) {
    if (n_groups == 0) return BN254_OK;
    if (!out || (M && (!g1 || !g2)) || (!M && !f_in) || k == 0 || k > 64) return BN254_ERR_INVALID_ARG;
    LaunchCtx lc; LaunchCtx* c = &lc; uint32_t grid;
    size_t n_items = (n_groups + BLOCK - 1) / BLOCK;
    int rc = ctx_get(device, stream, k, &lc, &grid, n_items);
    if (rc) return rc;
    static const bool use_v1 = (getenv("BN254_FORCE_V1") != nullptr);
    static const bool use_v2 = (getenv("BN254_FORCE_V2") != nullptr);
    if (k == 1 && !use_v1 && !use_v2) {
        if (n_groups >= (1ull << 29)) return BN254_ERR_INVALID_ARG;   // 32-bit element offsets in the asm kernels
        uint32_t stride = grid * BLOCK * V3_SLOT_BYTES;                 // bytes between scratch slots
        if (M && F) hipLaunchKernelGGL(k3_pairing, dim3(grid), dim3(BLOCK), LDS_BYTES, (hipStream_t)stream, g1, g2, f_in, out,
                                       (uint32_t)n_groups, 1u, c->scratch, stride, c->status);
        else if (M) hipLaunchKernelGGL(k3_miller, dim3(grid), dim3(BLOCK), LDS_BYTES, (hipStream_t)stream, g1, g2, f_in, out,
                                       (uint32_t)n_groups, 1u, c->scratch, stride, c->status);
        else hipLaunchKernelGGL(k3_fexp, dim3(grid), dim3(BLOCK), LDS_BYTES, (hipStream_t)stream, g1, g2, f_in, out,
                                (uint32_t)n_groups, 1u, c->scratch, stride, c->status);
        HIPCHK(hipGetLastError());
        return BN254_OK;
    }
    if (k > 1 && M && !use_v1 && !use_v2) {
        if (n_groups * k >= (1ull << 29)) return BN254_ERR_INVALID_ARG;
        uint32_t stride = grid * BLOCK * V3_SLOT_BYTES;
        if (F) hipLaunchKernelGGL(k3_mpairing, dim3(grid), dim3(BLOCK), LDS_BYTES, (hipStream_t)stream, g1, g2, f_in, out,
                                  (uint32_t)n_groups, (uint32_t)k, c->scratch, stride, c->status);
        else hipLaunchKernelGGL(k3_mmiller, dim3(grid), dim3(BLOCK), LDS_BYTES, (hipStream_t)stream, g1, g2, f_in, out,
                                (uint32_t)n_groups, (uint32_t)k, c->scratch, stride, c->status);
        HIPCHK(hipGetLastError());
        return BN254_OK;
    }
    hipLaunchKernelGGL((k_pairing<M, F>), dim3(grid), dim3(BLOCK), LDS_BYTES, (hipStream_t)stream,
                       g1, g2, f_in, out, n_groups, (int)k, c->scratch, (uint32_t)(c->n_cu * BLOCK), c->status);
    HIPCHK(hipGetLastError());
    return BN254_OK;
}

int launch_op(int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, size_t power, const int8_t* naf_host, int naf_len,
              int device, void* stream) {
    if (n == 0) return BN254_OK;
    if (!a || !out || (op == OP_MUL && !b)) return BN254_ERR_INVALID_ARG;
    LaunchCtx lc; LaunchCtx* c = &lc; uint32_t grid;
    size_t n_items = (n + BLOCK - 1) / BLOCK;
    int rc = ctx_get(device, stream, 1, &lc, &grid, n_items);
    if (rc) return rc;
    const int8_t* naf_dev = nullptr;
    static const bool use_v1 = (getenv("BN254_FORCE_V1") != nullptr) || (getenv("BN254_FORCE_V2") != nullptr);
    if (op == OP_POW) {
        while (naf_len > 0 && naf_host[naf_len - 1] == 0) naf_len--;        // the top digit of a NAF is +1
        StreamCtx* sc = c->s;
        if ((size_t)naf_len + 64 > sc->naf_cap) {
            if (sc->naf) { HIPCHK(hipStreamSynchronize((hipStream_t)stream)); HIPCHK(hipFree(sc->naf)); sc->naf = nullptr; }
            HIPCHK(hipMalloc(&sc->naf, (size_t)naf_len + 64));
            sc->naf_cap = (size_t)naf_len + 64;
        }
        HIPCHK(hipMemcpyAsync(sc->naf, naf_host, (size_t)naf_len, hipMemcpyHostToDevice, (hipStream_t)stream));
        HIPCHK(hipStreamSynchronize((hipStream_t)stream));   // naf_host may be a caller temporary
        naf_dev = sc->naf;
    }
    // v3 assembly (signed radix-2^27 limbs): Mul, frobenius_map_native, pow_native.  k = op | power << 8 | naf_len << 16
    if (!use_v1 && (op == OP_MUL || op == OP_FROB || (op == OP_POW && naf_len >= 1 && naf_len < 65536)) && n < (1ull << 29)) {
        uint32_t kk = op == OP_MUL ? 0u : op == OP_FROB ? (1u | ((uint32_t)(power % 12) << 8)) : (2u | ((uint32_t)naf_len << 16));
        if (op == OP_POW) for (int t = 0; t < naf_len; t++) if (naf_host[t] < 0) { kk |= 1u << 8; break; }   // 1/a is needed
        uint32_t stride = grid * BLOCK * V3_SLOT_BYTES;
        hipLaunchKernelGGL(k3_op, dim3(grid), dim3(BLOCK), LDS_BYTES, (hipStream_t)stream, b, (const uint64_t*)naf_dev, a, out,
                           (uint32_t)n, kk, c->scratch, stride, c->status);
        HIPCHK(hipGetLastError());
        return BN254_OK;
    }
    hipLaunchKernelGGL(k_fq12_op, dim3(grid), dim3(BLOCK), LDS_BYTES, (hipStream_t)stream,
                       op, a, b, out, n, power, naf_dev, naf_len, c->scratch, (uint32_t)(c->n_cu * BLOCK), c->status);
    HIPCHK(hipGetLastError());
    return BN254_OK;
}

// host-pointer staging helper
struct Staged {
    std::vector<void*> bufs;
    ~Staged() { for (void* p : bufs) if (p) (void)hipFree(p); }
    int up(const uint64_t* h, size_t words, uint64_t** d, hipStream_t s) {
        if (hipMalloc((void**)d, words * 8) != hipSuccess) return BN254_ERR_ALLOC;
        bufs.push_back(*d);
        if (h && hipMemcpyAsync(*d, h, words * 8, hipMemcpyHostToDevice, s) != hipSuccess) return BN254_ERR_HIP;
        return BN254_OK;
    }
};

int finish_host(uint64_t* h_out, const uint64_t* d_out, size_t words, int device, void* stream) {
    if (hipMemcpyAsync(h_out, d_out, words * 8, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return BN254_ERR_HIP;
    return bn254_last_status(device, stream);
}

// get_naf -- final_exp_native.rs:86-128 (host logic, identical control flow)
long get_naf_host(const uint64_t* exp_in, size_t n, int8_t* naf) {
    std::vector<uint64_t> exp(exp_in, exp_in + n);
    size_t len = n, k = 0;
    for (size_t idx = 0; idx < len; idx++) {
        uint64_t e = exp[idx];
        for (int b = 0; b < 64; b++) {
            if (e & 1) {
                int8_t z = (int8_t)(2 - (int)(e % 4));
                e /= 2;
                if (z == -1) e += 1;
                naf[k++] = z;
            } else { naf[k++] = 0; e /= 2; }
        }
        if (e != 0) {
            size_t j = idx + 1;
            while (j < exp.size() && exp[j] == UINT64_MAX) { exp[j] = 0; j++; }
            if (j < exp.size()) exp[j] += 1; else exp.push_back(1);
        }
    }
    if (exp.size() != len) return BN254_ERR_NAF_CARRY;  // the reference's assert at :123 cannot hold: it panics
    return (long)k;
}

}  // namespace

// ------------------------------------------------------------------ extern "C"
extern "C" {

constexpr size_t PIPE_CHUNK = 2 * 65536;      // lanes per chunk of the host-pointer pipeline (two full grids)
static int run_pipeline(const int* devices, int n_dev, const uint64_t* g1, const uint64_t* g2, uint64_t* out, size_t n_units, size_t k,
                        int do_final_exp);

int bn254_device_count(void) {
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess) return 0;
    return cnt;
}

const char* bn254_strerror(int status) {
    switch (status) {
        case BN254_OK: return "ok";
        case BN254_ERR_INVALID_ARG: return "invalid argument";
        case BN254_ERR_NO_DEVICE: return "no HIP device";
        case BN254_ERR_HIP: return "HIP runtime error";
        case BN254_ERR_ZERO_DIVISOR: return "division by zero in Fq12 (the reference panics here)";
        case BN254_ERR_NAF_CARRY: return "get_naf: carry out of the top limb (the reference panics here)";
        case BN254_ERR_ALLOC: return "device allocation failed";
        default: return "unknown status";
    }
}

int bn254_last_status(int device, void* stream) {
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) return BN254_ERR_NO_DEVICE;
    if (device < 0 || device >= cnt || device >= 64) return BN254_ERR_INVALID_ARG;
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    int* status = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        DeviceCtx& c = g_ctx[device];
        auto it = c.streams.find((hipStream_t)stream);
        if (!c.init || it == c.streams.end()) return BN254_OK;
        status = it->second.status;
    }
    int h = 0;
    HIPCHK(hipMemcpy(&h, status, sizeof(int), hipMemcpyDeviceToHost));
    if (h) { HIPCHK(hipMemset(status, 0, sizeof(int))); return BN254_ERR_ZERO_DIVISOR; }
    return BN254_OK;
}

size_t bn254_scratch_bytes(size_t n, size_t k) {
    (void)n;
    size_t v1 = (size_t)(N_GSLOTS_BASE + 3 * (k > 1 ? k : 0)) * 64 * 256 * BLOCK;
    size_t v3 = (size_t)(V3_GSLOTS + 7 * (k > 1 ? k : 0)) * V3_SLOT_BYTES * 256 * BLOCK;
    return v1 > v3 ? v1 : v3;
}

int bn254_pairing_batch_dev(const uint64_t* g1, const uint64_t* g2, uint64_t* out, size_t n, int device, void* stream) {
    return launch_pairing<true, true>(g1, g2, nullptr, out, n, 1, device, stream);
}
int bn254_miller_loop_batch_dev(const uint64_t* g1, const uint64_t* g2, uint64_t* f_out, size_t n, int device, void* stream) {
    return launch_pairing<true, false>(g1, g2, nullptr, f_out, n, 1, device, stream);
}
int bn254_final_exp_batch_dev(const uint64_t* f_in, uint64_t* out, size_t n, int device, void* stream) {
    return launch_pairing<false, true>(nullptr, nullptr, f_in, out, n, 1, device, stream);
}
int bn254_multi_pairing_batch_dev(const uint64_t* g1, const uint64_t* g2, uint64_t* out, size_t n_groups, size_t k, int do_final_exp,
                                  int device, void* stream) {
    if (do_final_exp) return launch_pairing<true, true>(g1, g2, nullptr, out, n_groups, k, device, stream);
    return launch_pairing<true, false>(g1, g2, nullptr, out, n_groups, k, device, stream);
}
int bn254_fq12_mul_batch_dev(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, int device, void* stream) {
    return launch_op(OP_MUL, a, b, out, n, 0, nullptr, 0, device, stream);
}
int bn254_frobenius_map_batch_dev(const uint64_t* a, size_t power, uint64_t* out, size_t n, int device, void* stream) {
    return launch_op(OP_FROB, a, nullptr, out, n, power, nullptr, 0, device, stream);
}
int bn254_pow_batch_dev(const uint64_t* a, const uint64_t* exp, size_t exp_limbs, uint64_t* out, size_t n, int device, void* stream) {
    if (exp_limbs && !exp) return BN254_ERR_INVALID_ARG;
    std::vector<int8_t> naf(64 * exp_limbs + 1);
    long len = exp_limbs ? get_naf_host(exp, exp_limbs, naf.data()) : 0;
    if (len < 0) return (int)len;
    bool any = false;
    for (long t = 0; t < len; t++) any |= (naf[t] != 0);
    if (!any) {
        // pow_native (final_exp_native.rs:56-84) with an all-zero NAF (exp == 0 or an empty vector): `is_started` never
        // becomes true, the loop body is skipped and the function returns `res`, which was initialised to `a`.
        if (n == 0) return BN254_OK;
        if (!a || !out) return BN254_ERR_INVALID_ARG;
        int cnt = 0;
        if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) return BN254_ERR_NO_DEVICE;
        if (device < 0 || device >= cnt) return BN254_ERR_INVALID_ARG;
        HIPCHK(hipSetDevice(device));
        if (out != a) HIPCHK(hipMemcpyAsync(out, a, 48 * n * sizeof(uint64_t), hipMemcpyDeviceToDevice, (hipStream_t)stream));
        return BN254_OK;
    }
    return launch_op(OP_POW, a, nullptr, out, n, 0, naf.data(), (int)len, device, stream);
}
/* test hooks for the field layer (not part of the reference's surface) */
int bn254_fq12_unary_batch_dev(int op, const uint64_t* a, uint64_t* out, size_t n, int device, void* stream) {
    if (op != OP_INV && op != OP_SQR && op != OP_CYC_SQR) return BN254_ERR_INVALID_ARG;
    return launch_op(op, a, nullptr, out, n, 0, nullptr, 0, device, stream);
}

long bn254_get_naf(const uint64_t* exp, size_t exp_limbs, int8_t* naf) {
    if (!exp || !naf) return BN254_ERR_INVALID_ARG;
    return get_naf_host(exp, exp_limbs, naf);
}

int bn254_frob_coeffs(size_t index, uint64_t* out8) {
    if (index >= 12 || !out8) return BN254_ERR_INVALID_ARG;
    memcpy(out8, BN254_FROB_COEFFS_HOST[index], 8 * sizeof(uint64_t));      // host table: no device involved
    return BN254_OK;
}
const int8_t* bn254_six_u_plus_2_naf(void) {
    static const int8_t naf[65] = {0, 0, 0, 1, 0, 1, 0, -1, 0, 0, 1, -1, 0, 0, 1, 0, 0, 1, 1, 0, -1, 0, 0, 1, 0, -1, 0, 0, 0, 0,
                                   1, 1, 1, 0, 0, -1, 0, 0, 1, 0, 0, 0, 0, 0, -1, 0, 0, 1, 1, 0, 0, -1, 0, 0, 0, 1, 1, 0, -1, 0,
                                   0, 1, 0, 1, 1};
    return naf;
}
uint64_t bn254_bn_x(void) { return BN254_BN_X; }
int bn254_myfq12_to_ark_index(int j) {
    if (j < 0 || j > 11) return -1;
    int h = j / 6, k = (j % 6) / 2, e = j % 2;   // ark flat index j = (h*3 + k)*2 + e
    return (2 * k + h) + 6 * e;
}

int bn254_generate_pairs_dev(uint64_t seed, uint64_t* g1_out, uint64_t* g2_out, size_t n, int device, void* stream) {
    if (n == 0) return BN254_OK;
    if (!g1_out || !g2_out) return BN254_ERR_INVALID_ARG;
    LaunchCtx lc; LaunchCtx* c = &lc; uint32_t grid;
    size_t n_items = (n + BLOCK - 1) / BLOCK;
    int rc = ctx_get(device, stream, 1, &lc, &grid, n_items);
    if (rc) return rc;
    hipLaunchKernelGGL(k_generate, dim3(grid), dim3(BLOCK), LDS_BYTES, (hipStream_t)stream, seed, g1_out, g2_out, n, c->scratch,
                       (uint32_t)(c->n_cu * BLOCK));
    HIPCHK(hipGetLastError());
    return BN254_OK;
}

int bn254_multi_pairing_check_batch_dev(const uint64_t* g1, const uint64_t* g2, uint8_t* verdict, size_t n_groups, size_t k, int device,
                                        void* stream) {
    if (n_groups == 0) return BN254_OK;
    if (!g1 || !g2 || !verdict || k == 0 || k > 64) return BN254_ERR_INVALID_ARG;
    LaunchCtx lc; uint32_t grid;
    int rc = ctx_get(device, stream, k, &lc, &grid, (n_groups + BLOCK - 1) / BLOCK);
    if (rc) return rc;
    StreamCtx* sc = lc.s;
    size_t need = 384 * n_groups;
    if (need > sc->tmp_bytes) {
        if (sc->tmp) { HIPCHK(hipStreamSynchronize((hipStream_t)stream)); HIPCHK(hipFree(sc->tmp)); sc->tmp = nullptr; sc->tmp_bytes = 0; }
        if (hipMalloc(&sc->tmp, need) != hipSuccess) return BN254_ERR_ALLOC;
        sc->tmp_bytes = need;
    }
    if ((rc = launch_pairing<true, true>(g1, g2, nullptr, sc->tmp, n_groups, k, device, stream))) return rc;
    size_t blocks = (n_groups + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_is_one, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)stream, sc->tmp, verdict, n_groups);
    HIPCHK(hipGetLastError());
    return BN254_OK;
}

int bn254_release_stream(int device, void* stream) {
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) return BN254_ERR_NO_DEVICE;
    if (device < 0 || device >= cnt || device >= 64) return BN254_ERR_INVALID_ARG;
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    std::lock_guard<std::mutex> lk(g_mu);
    DeviceCtx& c = g_ctx[device];
    auto it = c.streams.find((hipStream_t)stream);
    if (it == c.streams.end()) return BN254_OK;
    StreamCtx& sc = it->second;
    if (sc.scratch) (void)hipFree(sc.scratch);
    if (sc.status) (void)hipFree(sc.status);
    if (sc.naf) (void)hipFree(sc.naf);
    if (sc.tmp) (void)hipFree(sc.tmp);
    c.streams.erase(it);
    return BN254_OK;
}

// ---- host-pointer variants: stage through the device, synchronise, report the status word
int bn254_multi_pairing_check_batch(const uint64_t* g1, const uint64_t* g2, uint8_t* verdict, size_t n_groups, size_t k, int device, void* stream) {
    if (n_groups == 0) return BN254_OK;
    if (!g1 || !g2 || !verdict || k == 0) return BN254_ERR_INVALID_ARG;
    if (bn254_device_count() <= 0) return BN254_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return BN254_ERR_INVALID_ARG;
    Staged s; uint64_t *d1, *d2, *d3; int rc; size_t np = n_groups * k;
    if ((rc = s.up(g1, 8 * np, &d1, (hipStream_t)stream)) || (rc = s.up(g2, 16 * np, &d2, (hipStream_t)stream)) ||
        (rc = s.up(nullptr, (n_groups + 7) / 8, &d3, (hipStream_t)stream))) return rc;
    if ((rc = bn254_multi_pairing_check_batch_dev(d1, d2, (uint8_t*)d3, n_groups, k, device, stream))) return rc;
    if (hipMemcpyAsync(verdict, d3, n_groups, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return BN254_ERR_HIP;
    return bn254_last_status(device, stream);
}

int bn254_pairing_batch(const uint64_t* g1, const uint64_t* g2, uint64_t* out, size_t n, int device, void* stream) {
    if (n == 0) return BN254_OK;
    if (!g1 || !g2 || !out) return BN254_ERR_INVALID_ARG;
    if (bn254_device_count() <= 0) return BN254_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return BN254_ERR_INVALID_ARG;
    if (n > PIPE_CHUNK) {                 // large batch: chunked, copies overlapped with compute (private streams)
        if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return BN254_ERR_HIP;
        return run_pipeline(&device, 1, g1, g2, out, n, 1, 1);
    }
    Staged s; uint64_t *d1, *d2, *d3; int rc;
    if ((rc = s.up(g1, 8 * n, &d1, (hipStream_t)stream)) || (rc = s.up(g2, 16 * n, &d2, (hipStream_t)stream)) || (rc = s.up(nullptr, 48 * n, &d3, (hipStream_t)stream))) return rc;
    if ((rc = bn254_pairing_batch_dev(d1, d2, d3, n, device, stream))) return rc;
    return finish_host(out, d3, 48 * n, device, stream);
}
int bn254_miller_loop_batch(const uint64_t* g1, const uint64_t* g2, uint64_t* f_out, size_t n, int device, void* stream) {
    if (n == 0) return BN254_OK;
    if (!g1 || !g2 || !f_out) return BN254_ERR_INVALID_ARG;
    if (bn254_device_count() <= 0) return BN254_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return BN254_ERR_INVALID_ARG;
    Staged s; uint64_t *d1, *d2, *d3; int rc;
    if ((rc = s.up(g1, 8 * n, &d1, (hipStream_t)stream)) || (rc = s.up(g2, 16 * n, &d2, (hipStream_t)stream)) || (rc = s.up(nullptr, 48 * n, &d3, (hipStream_t)stream))) return rc;
    if ((rc = bn254_miller_loop_batch_dev(d1, d2, d3, n, device, stream))) return rc;
    return finish_host(f_out, d3, 48 * n, device, stream);
}
int bn254_final_exp_batch(const uint64_t* f_in, uint64_t* out, size_t n, int device, void* stream) {
    if (n == 0) return BN254_OK;
    if (!f_in || !out) return BN254_ERR_INVALID_ARG;
    if (bn254_device_count() <= 0) return BN254_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return BN254_ERR_INVALID_ARG;
    Staged s; uint64_t *d1, *d3; int rc;
    if ((rc = s.up(f_in, 48 * n, &d1, (hipStream_t)stream)) || (rc = s.up(nullptr, 48 * n, &d3, (hipStream_t)stream))) return rc;
    if ((rc = bn254_final_exp_batch_dev(d1, d3, n, device, stream))) return rc;
    return finish_host(out, d3, 48 * n, device, stream);
}
int bn254_multi_pairing_batch(const uint64_t* g1, const uint64_t* g2, uint64_t* out, size_t n_groups, size_t k, int do_final_exp,
                              int device, void* stream) {
    if (n_groups == 0) return BN254_OK;
    if (!g1 || !g2 || !out || k == 0) return BN254_ERR_INVALID_ARG;
    if (bn254_device_count() <= 0) return BN254_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return BN254_ERR_INVALID_ARG;
    if (n_groups > PIPE_CHUNK && k <= 64) {
        if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return BN254_ERR_HIP;
        return run_pipeline(&device, 1, g1, g2, out, n_groups, k, do_final_exp);
    }
    Staged s; uint64_t *d1, *d2, *d3; int rc; size_t np = n_groups * k;
    if ((rc = s.up(g1, 8 * np, &d1, (hipStream_t)stream)) || (rc = s.up(g2, 16 * np, &d2, (hipStream_t)stream)) || (rc = s.up(nullptr, 48 * n_groups, &d3, (hipStream_t)stream))) return rc;
    if ((rc = bn254_multi_pairing_batch_dev(d1, d2, d3, n_groups, k, do_final_exp, device, stream))) return rc;
    return finish_host(out, d3, 48 * n_groups, device, stream);
}
int bn254_fq12_mul_batch(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, int device, void* stream) {
    if (n == 0) return BN254_OK;
    if (!a || !b || !out) return BN254_ERR_INVALID_ARG;
    if (bn254_device_count() <= 0) return BN254_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return BN254_ERR_INVALID_ARG;
    Staged s; uint64_t *d1, *d2, *d3; int rc;
    if ((rc = s.up(a, 48 * n, &d1, (hipStream_t)stream)) || (rc = s.up(b, 48 * n, &d2, (hipStream_t)stream)) || (rc = s.up(nullptr, 48 * n, &d3, (hipStream_t)stream))) return rc;
    if ((rc = bn254_fq12_mul_batch_dev(d1, d2, d3, n, device, stream))) return rc;
    return finish_host(out, d3, 48 * n, device, stream);
}
int bn254_frobenius_map_batch(const uint64_t* a, size_t power, uint64_t* out, size_t n, int device, void* stream) {
    if (n == 0) return BN254_OK;
    if (!a || !out) return BN254_ERR_INVALID_ARG;
    if (bn254_device_count() <= 0) return BN254_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return BN254_ERR_INVALID_ARG;
    Staged s; uint64_t *d1, *d3; int rc;
    if ((rc = s.up(a, 48 * n, &d1, (hipStream_t)stream)) || (rc = s.up(nullptr, 48 * n, &d3, (hipStream_t)stream))) return rc;
    if ((rc = bn254_frobenius_map_batch_dev(d1, power, d3, n, device, stream))) return rc;
    return finish_host(out, d3, 48 * n, device, stream);
}
int bn254_pow_batch(const uint64_t* a, const uint64_t* exp, size_t exp_limbs, uint64_t* out, size_t n, int device, void* stream) {
    if (n == 0) return BN254_OK;
    if (!a || (!exp && exp_limbs) || !out) return BN254_ERR_INVALID_ARG;
    if (bn254_device_count() <= 0) return BN254_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return BN254_ERR_INVALID_ARG;
    Staged s; uint64_t *d1, *d3; int rc;
    if ((rc = s.up(a, 48 * n, &d1, (hipStream_t)stream)) || (rc = s.up(nullptr, 48 * n, &d3, (hipStream_t)stream))) return rc;
    if ((rc = bn254_pow_batch_dev(d1, exp, exp_limbs, d3, n, device, stream))) return rc;
    return finish_host(out, d3, 48 * n, device, stream);
}


// ---- host-pointer pipeline, one or several GPUs of this process: contiguous slices of the batch per device,
// no exchange step.  A slice is cut into chunks of PIPE_CHUNK lanes (two full grids); a worker thread owns one private
// stream and device buffers for one chunk and walks every second chunk of its device: it stages its chunk of every limb
// plane (2-D copies straight out of / into the caller's SoA arrays), launches the kernel and copies the result back.  Two
// workers per device alternate, so one worker's copies run under the other's kernel (the kernels fill the chip and
// serialise).  Units are pairings (k = 1) or k-pair groups.

static int run_chunks(int dev, const uint64_t* g1, const uint64_t* g2, uint64_t* out, size_t n_units, size_t k, int do_final_exp, size_t u0,
                      size_t cnt, size_t chunk, size_t first, size_t step) {
    if (cnt == 0 || first * chunk >= cnt) return BN254_OK;
    if (hipSetDevice(dev) != hipSuccess) return BN254_ERR_INVALID_ARG;
    hipStream_t st;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return BN254_ERR_HIP;
    int rc = BN254_OK;
    {
        Staged s; uint64_t *d1, *d2, *d3;
        size_t cap = cnt < chunk ? cnt : chunk, np_all = n_units * k;
        if ((rc = s.up(nullptr, 8 * cap * k, &d1, st)) || (rc = s.up(nullptr, 16 * cap * k, &d2, st)) || (rc = s.up(nullptr, 48 * cap, &d3, st))) goto done;
        for (size_t c0 = first * chunk; c0 < cnt; c0 += step * chunk) {
            size_t m = cnt - c0 < chunk ? cnt - c0 : chunk, np = m * k, base = u0 + c0;
            if (hipMemcpy2DAsync(d1, np * 8, g1 + base * k, np_all * 8, np * 8, 8, hipMemcpyHostToDevice, st) != hipSuccess ||
                hipMemcpy2DAsync(d2, np * 8, g2 + base * k, np_all * 8, np * 8, 16, hipMemcpyHostToDevice, st) != hipSuccess) { rc = BN254_ERR_HIP; goto done; }
            rc = (k == 1 && do_final_exp) ? bn254_pairing_batch_dev(d1, d2, d3, m, dev, st)
                                          : bn254_multi_pairing_batch_dev(d1, d2, d3, m, k, do_final_exp, dev, st);
            if (rc) goto done;
            if (hipMemcpy2DAsync(out + base, n_units * 8, d3, m * 8, m * 8, 48, hipMemcpyDeviceToHost, st) != hipSuccess) { rc = BN254_ERR_HIP; goto done; }
            if ((rc = bn254_last_status(dev, st))) goto done;        // also: the buffers are free for the next chunk
        }
    done:
        (void)hipStreamSynchronize(st);
    }
    (void)bn254_release_stream(dev, st);
    (void)hipStreamDestroy(st);
    return rc;
}

// devices[0..n_dev): the batch is split into n_dev contiguous slices
static int run_pipeline(const int* devices, int n_dev, const uint64_t* g1, const uint64_t* g2, uint64_t* out, size_t n_units, size_t k,
                        int do_final_exp) {
    size_t chunk = PIPE_CHUNK;                        // lanes = units (one unit per lane whatever k is)
    size_t per = (n_units + (size_t)n_dev - 1) / (size_t)n_dev;
    std::vector<int> rcs;
    std::vector<std::thread> th;
    rcs.reserve((size_t)n_dev * 2);
    for (int d = 0; d < n_dev; d++) {
        size_t u0 = per * (size_t)d;
        size_t c = u0 >= n_units ? 0 : (n_units - u0 < per ? n_units - u0 : per);
        size_t workers = c > chunk ? 2 : 1;
        for (size_t w = 0; w < workers; w++) {
            rcs.push_back(BN254_OK);
            int* slot = &rcs.back();
            int dev = devices[d];
            th.emplace_back([=] { *slot = run_chunks(dev, g1, g2, out, n_units, k, do_final_exp, u0, c, chunk, w, workers); });
        }
    }
    for (auto& t : th) t.join();
    for (int rc : rcs) if (rc) return rc;
    return BN254_OK;
}

int bn254_multi_pairing_sharded(const uint64_t* g1, const uint64_t* g2, uint64_t* out, size_t n_groups, size_t k, int do_final_exp, int n_devices) {
    if (n_groups == 0) return BN254_OK;
    if (!g1 || !g2 || !out || k == 0 || k > 64 || n_devices <= 0) return BN254_ERR_INVALID_ARG;
    int cnt = bn254_device_count();
    if (cnt <= 0) return BN254_ERR_NO_DEVICE;
    if (n_devices > cnt) return BN254_ERR_INVALID_ARG;
    std::vector<int> devs((size_t)n_devices);
    for (int d = 0; d < n_devices; d++) devs[(size_t)d] = d;
    return run_pipeline(devs.data(), n_devices, g1, g2, out, n_groups, k, do_final_exp);
}

int bn254_pairing_sharded(const uint64_t* g1, const uint64_t* g2, uint64_t* out, size_t n, int n_devices) {
    return bn254_multi_pairing_sharded(g1, g2, out, n, 1, 1, n_devices);
}

}  // extern "C"
