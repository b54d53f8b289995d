// bn254_pairing.hpp -- C++ host-side mirror of the reference's native API on top of the C ABI
// (include/bn254_pairing.h).  Same names, argument order and error behaviour as the Rust
// functions it replaces; the reference panics where these throw.
//
//   reference                                               here
//   pairing(p, q) -> Fq12                 pairing.rs:20      bn254::pairing(p, q)
//   miller_loop_native(&Q, &P) -> MyFq12  miller_loop_native.rs:320   bn254::miller_loop_native(Q, P)
//   multi_miller_loop_native(pairs)       :324               bn254::multi_miller_loop_native(pairs)
//   final_exp_native(a) -> MyFq12         final_exp_native.rs:209     bn254::final_exp_native(a)
//   frobenius_map_native / pow_native / get_naf / frob_coeffs / conjugate_fp2 / neg_conjugate_fp2
//   SIX_U_PLUS_2_NAF, BN_X
#pragma once
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "bn254_pairing.h"

namespace bn254 {

using Fq = std::array<uint64_t, 4>;                 // Montgomery limbs == ark Fp.0.0
struct Fq2 { Fq c0, c1; };
struct G1Affine { Fq x, y; };                       // (ark's `infinity` flag is outside the contract)
struct G2Affine { Fq2 x, y; };
struct MyFq12 { std::array<Fq, 12> coeffs; bool operator==(const MyFq12& o) const { return coeffs == o.coeffs; } };
struct Fq12 { std::array<Fq, 12> flat; };           // ark order c0.c0.c0, c0.c0.c1, c0.c1.c0, ...

struct Panic : std::runtime_error {                 // the reference panics here
    int status;
    explicit Panic(int s) : std::runtime_error(bn254_strerror(s)), status(s) {}
};
inline void check(int rc) { if (rc != BN254_OK) throw Panic(rc); }

constexpr uint64_t BN_X = 4965661367192848881ull;   // final_exp_native.rs:15
inline const int8_t* SIX_U_PLUS_2_NAF() { return bn254_six_u_plus_2_naf(); }  // miller_loop_native.rs:314-318 (65 digits)

namespace detail {
inline void put(const Fq& f, uint64_t* dst, size_t stride) { for (int l = 0; l < 4; l++) dst[l * stride] = f[l]; }
inline void get(Fq& f, const uint64_t* src, size_t stride) { for (int l = 0; l < 4; l++) f[l] = src[l * stride]; }
// SoA batch of n elements: elem(c, l, i) = buf[(c*4 + l)*n + i]
inline void pack_g1(const G1Affine& p, uint64_t* buf, size_t n, size_t i) { put(p.x, buf + i, n); put(p.y, buf + 4 * n + i, n); }
inline void pack_g2(const G2Affine& q, uint64_t* buf, size_t n, size_t i) {
    put(q.x.c0, buf + i, n); put(q.x.c1, buf + 4 * n + i, n); put(q.y.c0, buf + 8 * n + i, n); put(q.y.c1, buf + 12 * n + i, n);
}
inline void pack_fq12(const MyFq12& a, uint64_t* buf, size_t n, size_t i) { for (int c = 0; c < 12; c++) put(a.coeffs[c], buf + 4 * c * n + i, n); }
inline MyFq12 unpack_fq12(const uint64_t* buf, size_t n, size_t i) { MyFq12 r; for (int c = 0; c < 12; c++) get(r.coeffs[c], buf + 4 * c * n + i, n); return r; }
}  // namespace detail

// Sizes what the library keeps for (device, NULL stream) -- the stream every function of this header uses -- for calls of up to n units x k
// pairs: no later call of that size allocates device memory (bn254_reserve).
inline void reserve(size_t n, size_t k = 1, int device = 0) { check(bn254_reserve(device, nullptr, n, k)); }
// The scalar functions below (one element per call) run on the lane-cooperative kernels by default (bn254_pairing.h: LATENCY PATH);
// batches of at most `n` items do, 0 turns that off.
inline void set_latency_threshold(size_t n) { bn254_set_latency_threshold(n); }
inline size_t latency_threshold() { return bn254_get_latency_threshold(); }
// ... for the NULL stream of `device` alone (the stream every function of this header uses), whatever other users of the library set
// process-wide; BN254_LATENCY_INHERIT / -1 return to the defaults.
inline void set_stream_latency(size_t threshold, int lanes = -1, int device = 0) { check(bn254_set_stream_latency(device, nullptr, threshold, lanes)); }

inline MyFq12 miller_loop_native(const G2Affine& Q, const G1Affine& P, int device = 0) {
    uint64_t g1[8], g2[16], out[48];
    detail::pack_g1(P, g1, 1, 0); detail::pack_g2(Q, g2, 1, 0);
    check(bn254_miller_loop_batch(g1, g2, out, 1, device, nullptr));
    return detail::unpack_fq12(out, 1, 0);
}
inline MyFq12 multi_miller_loop_native(const std::vector<std::pair<const G1Affine*, const G2Affine*>>& pairs, int device = 0) {
    const size_t k = pairs.size();
    if (k == 0) throw Panic(BN254_ERR_INVALID_ARG);          // reference: pairs[0] panics
    std::vector<uint64_t> g1(8 * k), g2(16 * k);
    uint64_t out[48];
    for (size_t j = 0; j < k; j++) { detail::pack_g1(*pairs[j].first, g1.data(), k, j); detail::pack_g2(*pairs[j].second, g2.data(), k, j); }
    check(bn254_multi_pairing_batch(g1.data(), g2.data(), out, 1, k, 0, device, nullptr));
    return detail::unpack_fq12(out, 1, 0);
}
inline MyFq12 final_exp_native(const MyFq12& a, int device = 0) {
    uint64_t in[48], out[48];
    detail::pack_fq12(a, in, 1, 0);
    check(bn254_final_exp_batch(in, out, 1, device, nullptr));
    return detail::unpack_fq12(out, 1, 0);
}
inline Fq12 pairing(const G1Affine& p, const G2Affine& q, int device = 0) {
    uint64_t g1[8], g2[16], out[48];
    detail::pack_g1(p, g1, 1, 0); detail::pack_g2(q, g2, 1, 0);
    check(bn254_pairing_batch(g1, g2, out, 1, device, nullptr));
    MyFq12 m = detail::unpack_fq12(out, 1, 0);
    Fq12 r;
    for (int j = 0; j < 12; j++) r.flat[j] = m.coeffs[bn254_myfq12_to_ark_index(j)];   // `.into()` at pairing.rs:21
    return r;
}
inline MyFq12 frobenius_map_native(const MyFq12& a, size_t power, int device = 0) {
    uint64_t in[48], out[48];
    detail::pack_fq12(a, in, 1, 0);
    check(bn254_frobenius_map_batch(in, power, out, 1, device, nullptr));
    return detail::unpack_fq12(out, 1, 0);
}
inline MyFq12 pow_native(const MyFq12& a, const std::vector<uint64_t>& exp, int device = 0) {
    uint64_t in[48], out[48];
    detail::pack_fq12(a, in, 1, 0);
    check(bn254_pow_batch(in, exp.data(), exp.size(), out, 1, device, nullptr));
    return detail::unpack_fq12(out, 1, 0);
}
inline std::vector<int8_t> get_naf(const std::vector<uint64_t>& exp) {
    std::vector<int8_t> naf(64 * exp.size() + 1);
    long n = bn254_get_naf(exp.data(), exp.size(), naf.data());
    if (n < 0) throw Panic((int)n);
    naf.resize((size_t)n);
    return naf;
}
inline Fq2 frob_coeffs(size_t index) {
    uint64_t o[8];
    check(bn254_frob_coeffs(index % 12, o));     // frobenius_map_native reduces the power mod 12 (:22)
    Fq2 r;
    for (int l = 0; l < 4; l++) { r.c0[l] = o[l]; r.c1[l] = o[4 + l]; }
    return r;
}
// conjugate_fp2 / neg_conjugate_fp2 (miller_loop_native.rs:284-296): p - x on one component
inline Fq fq_neg(const Fq& a) {
    static const Fq P = {0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
    if ((a[0] | a[1] | a[2] | a[3]) == 0) return a;
    Fq r; unsigned __int128 br = 0;
    for (int i = 0; i < 4; i++) { unsigned __int128 d = (unsigned __int128)P[i] - a[i] - (uint64_t)br; r[i] = (uint64_t)d; br = (d >> 64) & 1; }
    return r;
}
inline Fq2 conjugate_fp2(const Fq2& x) { return Fq2{x.c0, fq_neg(x.c1)}; }
inline Fq2 neg_conjugate_fp2(const Fq2& x) { return Fq2{fq_neg(x.c0), x.c1}; }

// ---- batch forms (the reason the engine exists).  std::vector<G1Affine> etc. ARE the element-major layout of
// bn254_pairing.h (8 / 16 / 48 words per element), so the vectors go to the engine as they are: no packing on the host.
static_assert(sizeof(G1Affine) == 64 && sizeof(G2Affine) == 128 && sizeof(MyFq12) == 384 && sizeof(Fq12) == 384, "element-major layout");
namespace detail {
inline const uint64_t* words(const void* p) { return static_cast<const uint64_t*>(p); }
inline uint64_t* words(void* p) { return static_cast<uint64_t*>(p); }
}  // namespace detail

// PAGE-LOCKED HOST MEMORY (bn254_pairing.h: the host-pointer entry points copy by DMA under the kernels when all three arrays are
// page-locked, and a large batch then runs at the resident-data rate).  `pinned_vector<T>` allocates from the runtime; `HostRegistration`
// page-locks memory the caller already owns for its lifetime; the pointer forms below (`*_into`) take either.
template <class T> struct PinnedAllocator {
    using value_type = T;
    PinnedAllocator() = default;
    template <class U> PinnedAllocator(const PinnedAllocator<U>&) {}
    T* allocate(size_t n) { void* p = nullptr; check(bn254_alloc_pinned(n * sizeof(T), &p)); return static_cast<T*>(p); }
    void deallocate(T* p, size_t) noexcept { (void)bn254_free_pinned(p); }
    template <class U> bool operator==(const PinnedAllocator<U>&) const { return true; }
    template <class U> bool operator!=(const PinnedAllocator<U>&) const { return false; }
};
template <class T> using pinned_vector = std::vector<T, PinnedAllocator<T>>;
class HostRegistration {
    void* p_;
public:
    HostRegistration(void* p, size_t bytes) : p_(p) { check(bn254_host_register(p, bytes)); }
    template <class V> explicit HostRegistration(V& v) : HostRegistration(v.data(), v.size() * sizeof(*v.data())) {}
    ~HostRegistration() { (void)bn254_host_unregister(p_); }
    HostRegistration(const HostRegistration&) = delete;
    HostRegistration& operator=(const HostRegistration&) = delete;
};
// n x pairing(p, q) from / into the caller's arrays (page-locked or not): MyFq12 order, or ark's Fq12 order exactly as src/pairing.rs:20-22 returns it
inline void pairing_batch_into(const G1Affine* ps, const G2Affine* qs, MyFq12* out, size_t n, int device = 0) {
    check(bn254_pairing_batch_elems(detail::words(ps), detail::words(qs), detail::words(out), n, BN254_FQ12_MYFQ12, device, nullptr));
}
inline void pairing_batch_fq12_into(const G1Affine* ps, const G2Affine* qs, Fq12* out, size_t n, int device = 0) {
    check(bn254_pairing_batch_elems(detail::words(ps), detail::words(qs), detail::words(out), n, BN254_FQ12_ARK, device, nullptr));
}
inline void multi_pairing_batch_into(const G1Affine* ps, const G2Affine* qs, MyFq12* out, size_t n_groups, size_t k, bool do_final_exp = true, int device = 0) {
    if (k == 0) throw Panic(BN254_ERR_INVALID_ARG);
    check(bn254_multi_pairing_batch_elems(detail::words(ps), detail::words(qs), detail::words(out), n_groups, k, do_final_exp ? 1 : 0, BN254_FQ12_MYFQ12,
                                          device, nullptr));
}

inline std::vector<MyFq12> pairing_batch(const std::vector<G1Affine>& ps, const std::vector<G2Affine>& qs, int device = 0) {
    const size_t n = ps.size();
    if (qs.size() != n) throw Panic(BN254_ERR_INVALID_ARG);
    std::vector<MyFq12> r(n);
    check(bn254_pairing_batch_elems(detail::words(ps.data()), detail::words(qs.data()), detail::words(r.data()), n, BN254_FQ12_MYFQ12, device, nullptr));
    return r;
}
// n x pairing(p, q) -> Fq12 exactly as src/pairing.rs:20-22 returns it (ark coefficient order)
inline std::vector<Fq12> pairing_batch_fq12(const std::vector<G1Affine>& ps, const std::vector<G2Affine>& qs, int device = 0) {
    const size_t n = ps.size();
    if (qs.size() != n) throw Panic(BN254_ERR_INVALID_ARG);
    std::vector<Fq12> r(n);
    check(bn254_pairing_batch_elems(detail::words(ps.data()), detail::words(qs.data()), detail::words(r.data()), n, BN254_FQ12_ARK, device, nullptr));
    return r;
}
inline std::vector<MyFq12> miller_loop_batch(const std::vector<G2Affine>& qs, const std::vector<G1Affine>& ps, int device = 0) {
    const size_t n = ps.size();
    if (qs.size() != n) throw Panic(BN254_ERR_INVALID_ARG);
    std::vector<MyFq12> r(n);
    check(bn254_miller_loop_batch_elems(detail::words(ps.data()), detail::words(qs.data()), detail::words(r.data()), n, device, nullptr));
    return r;
}
inline std::vector<MyFq12> final_exp_batch(const std::vector<MyFq12>& fs, int device = 0) {
    std::vector<MyFq12> r(fs.size());
    check(bn254_final_exp_batch_elems(detail::words(fs.data()), detail::words(r.data()), fs.size(), BN254_FQ12_MYFQ12, BN254_FQ12_MYFQ12, device, nullptr));
    return r;
}
// groups of k pairs: multi_miller_loop_native per group (+ final_exp_native when do_final_exp)
inline std::vector<MyFq12> multi_pairing_batch(const std::vector<G1Affine>& ps, const std::vector<G2Affine>& qs, size_t k, bool do_final_exp = true,
                                               int device = 0) {
    const size_t np = ps.size();
    if (qs.size() != np || k == 0 || np % k != 0) throw Panic(BN254_ERR_INVALID_ARG);
    std::vector<MyFq12> r(np / k);
    check(bn254_multi_pairing_batch_elems(detail::words(ps.data()), detail::words(qs.data()), detail::words(r.data()), np / k, k, do_final_exp ? 1 : 0,
                                          BN254_FQ12_MYFQ12, device, nullptr));
    return r;
}

// Groups whose LAST k_fixed G2 points are the same for the whole batch (a Groth16 verifier's beta, gamma, delta): ps = n x (1 + k_fixed) G1 points,
// group-major (the group's own P first), qs = the n groups' own G2 points, fixed = the shared ones.  final_exp_native(multi_miller_loop_native(...))
// per group, the same limbs as multi_pairing_batch on the expanded pairs, at 0.74 of its cost for 1 + 3 pairs.
inline std::vector<MyFq12> pairing_fixed_g2_batch(const std::vector<G1Affine>& ps, const std::vector<G2Affine>& qs, const std::vector<G2Affine>& fixed, int device = 0) {
    const size_t n = qs.size(), kf = fixed.size();
    if (kf == 0 || ps.size() != n * (kf + 1)) throw Panic(BN254_ERR_INVALID_ARG);
    std::vector<MyFq12> r(n);
    check(bn254_pairing_fixed_g2_batch_elems(detail::words(ps.data()), detail::words(qs.data()), detail::words(fixed.data()), kf, detail::words(r.data()), n,
                                             BN254_FQ12_MYFQ12, device, nullptr));
    return r;
}

// A Groth16 verifier's pairing check for a batch of proofs: verdict[g] = (product of group g's 1 + fixed.size() pairings == *target), target == nullptr:
// MyFq12::one.  With gamma, delta as `fixed` and target = pairing(alpha, beta) a proof costs 1 + 2 pairs.
inline std::vector<uint8_t> pairing_fixed_g2_check_batch(const std::vector<G1Affine>& ps, const std::vector<G2Affine>& qs, const std::vector<G2Affine>& fixed,
                                                         const MyFq12* target = nullptr, int device = 0) {
    // qs empty: the groups have no pair of their own -- ps holds fixed.size() points per group (a KZG / PLONK opening check e(P_1, [tau] G2) e(P_2, G2) == 1)
    const size_t kf = fixed.size(), own = qs.empty() ? 0 : 1;
    if (kf == 0) throw Panic(BN254_ERR_INVALID_ARG);
    const size_t n = own ? qs.size() : ps.size() / kf;
    if (ps.size() != n * (kf + own)) throw Panic(BN254_ERR_INVALID_ARG);
    std::vector<uint8_t> verdict(n);
    check(bn254_pairing_fixed_g2_check_batch_elems(detail::words(ps.data()), own ? detail::words(qs.data()) : nullptr, detail::words(fixed.data()), kf,
                                                   target ? detail::words(target) : nullptr, verdict.data(), n, device, nullptr));
    return verdict;
}

// ark's implicit input contract for untrusted points (`G1Affine::new` / `G2Affine::new`: on the curve, G2 in the r-torsion -- the reference
// calls the latter itself, miller_loop_native.rs:303,311, and panics there): throws Panic(BN254_ERR_INFINITY / _NOT_ON_CURVE /
// _NOT_IN_SUBGROUP) like the reference would; per_point (optional) receives one flag byte per pair (2 | 4 | 8).
inline void check_points_full(const std::vector<G1Affine>& ps, const std::vector<G2Affine>& qs, std::vector<uint8_t>* per_point = nullptr,
                              int device = 0) {
    const size_t n = ps.size();
    if (qs.size() != n) throw Panic(BN254_ERR_INVALID_ARG);
    std::vector<uint64_t> g1(8 * n), g2(16 * n);
    for (size_t i = 0; i < n; i++) { detail::pack_g1(ps[i], g1.data(), n, i); detail::pack_g2(qs[i], g2.data(), n, i); }
    if (per_point) per_point->assign(n, 0);
    check(bn254_check_points_ex(g1.data(), g2.data(), n, BN254_CHECK_INFINITY | BN254_CHECK_ON_CURVE | BN254_CHECK_SUBGROUP,
                                per_point ? per_point->data() : nullptr, device, nullptr));
}

// One process, several GPUs: contiguous slices per device, no exchange step (bn254_pairing_sharded_elems).
inline std::vector<MyFq12> pairing_sharded(const std::vector<G1Affine>& ps, const std::vector<G2Affine>& qs, int n_devices) {
    const size_t n = ps.size();
    if (qs.size() != n) throw Panic(BN254_ERR_INVALID_ARG);
    std::vector<MyFq12> r(n);
    check(bn254_pairing_sharded_elems(detail::words(ps.data()), detail::words(qs.data()), detail::words(r.data()), n, BN254_FQ12_MYFQ12, n_devices));
    return r;
}

// final_exp_native(multi_miller_loop_native(group)) == MyFq12::one for every group of k pairs
// (the product check of final_exp_native.rs:245-263; a Groth16 verifier's shape): one verdict per group.
inline std::vector<uint8_t> multi_pairing_check_batch(const std::vector<G1Affine>& ps, const std::vector<G2Affine>& qs, size_t k, int device = 0) {
    const size_t np = ps.size();
    if (qs.size() != np || k == 0 || np % k != 0) throw Panic(BN254_ERR_INVALID_ARG);
    std::vector<uint8_t> verdict(np / k);
    check(bn254_multi_pairing_check_batch_elems(detail::words(ps.data()), detail::words(qs.data()), verdict.data(), np / k, k, device, nullptr));
    return verdict;
}

}  // namespace bn254
