// A program built on the host binding include/bn254_pairing.hpp (the C++ mirror of the reference's pub fns) and linked
// against the C-ABI library: reads canonical-Montgomery u64 limbs from stdin, calls the scalar entry points the way the
// reference's callers do (pairing.rs:20, miller_loop_native.rs:320,324, final_exp_native.rs:17,56,183,209) and prints the
// result limbs as hex for tests/test_hostbind.py to compare with tests/golden/.
//
//   input:  "<op> <words...>" per line, words = hex u64
//     pairing  g1[8] g2[16]          miller g1[8] g2[16]        multi k (g1[8] g2[16])*k
//     fexp a[48]                     frob power a[48]           pow n_limbs exp[n] a[48]
//     frobc index                    naf n_limbs exp[n]
//     batch n (g1[8] g2[16])*n       -> pairing_batch_fq12 (ark order) and pairing_batch (MyFq12 order), n lines each
//     check k n_groups (g1[8] g2[16])*(k*n_groups)  -> multi_pairing_check_batch verdicts
//     reserve n k                    -> bn254_reserve for the NULL stream (hex words)
#include <cinttypes>
#include <cstdio>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "bn254_pairing.hpp"

using namespace bn254;

static uint64_t rd(std::istringstream& in) { std::string w; in >> w; return std::stoull(w, nullptr, 16); }
static Fq rd_fq(std::istringstream& in) { Fq f; for (int l = 0; l < 4; l++) f[l] = rd(in); return f; }
static G1Affine rd_g1(std::istringstream& in) { G1Affine p; p.x = rd_fq(in); p.y = rd_fq(in); return p; }
static G2Affine rd_g2(std::istringstream& in) { G2Affine q; q.x.c0 = rd_fq(in); q.x.c1 = rd_fq(in); q.y.c0 = rd_fq(in); q.y.c1 = rd_fq(in); return q; }
static MyFq12 rd_fq12(std::istringstream& in) { MyFq12 a; for (auto& c : a.coeffs) c = rd_fq(in); return a; }
static void pr_fq(const Fq& f) { for (int l = 0; l < 4; l++) std::printf(" %016" PRIx64, f[l]); }
static void pr_fq12(const char* tag, const MyFq12& a) { std::printf("%s", tag); for (auto& c : a.coeffs) pr_fq(c); std::printf("\n"); }

int main() {
    std::string line;
    while (std::getline(std::cin, line)) {
        std::istringstream in(line);
        std::string op;
        if (!(in >> op)) continue;
        try {
            if (op == "reserve") {                                        // reserve n k: no later call of that size allocates device memory
                size_t n = rd(in), k = rd(in);
                reserve(n, k);
                std::printf("reserved\n");
            } else if (op == "pairing") {
                G1Affine p = rd_g1(in); G2Affine q = rd_g2(in);
                Fq12 e = pairing(p, q);                                   // ark flat order (`.into()` at pairing.rs:21)
                std::printf("pairing"); for (auto& c : e.flat) pr_fq(c); std::printf("\n");
            } else if (op == "miller") {
                G1Affine p = rd_g1(in); G2Affine q = rd_g2(in);
                pr_fq12("miller", miller_loop_native(q, p));
            } else if (op == "multi") {
                size_t k = (size_t)rd(in);
                std::vector<G1Affine> ps(k); std::vector<G2Affine> qs(k);
                for (size_t j = 0; j < k; j++) { ps[j] = rd_g1(in); qs[j] = rd_g2(in); }
                std::vector<std::pair<const G1Affine*, const G2Affine*>> pairs;
                for (size_t j = 0; j < k; j++) pairs.push_back({&ps[j], &qs[j]});
                pr_fq12("multi", multi_miller_loop_native(pairs));
            } else if (op == "batch") {
                size_t n = (size_t)rd(in);
                std::vector<G1Affine> ps(n); std::vector<G2Affine> qs(n);
                for (size_t j = 0; j < n; j++) { ps[j] = rd_g1(in); qs[j] = rd_g2(in); }
                std::vector<Fq12> ark = pairing_batch_fq12(ps, qs);           // Vec<Fq12>, as n calls of pairing() return them
                std::vector<MyFq12> my = pairing_batch(ps, qs);
                for (auto& e : ark) { std::printf("bark"); for (auto& c : e.flat) pr_fq(c); std::printf("\n"); }
                for (auto& e : my) pr_fq12("bmy", e);
                {   // the same batch from / into page-locked memory: pinned_vector (bn254_alloc_pinned) and a registered std::vector
                    pinned_vector<G1Affine> pp(ps.begin(), ps.end()); pinned_vector<G2Affine> pq(qs.begin(), qs.end()); pinned_vector<Fq12> po(n);
                    pairing_batch_fq12_into(pp.data(), pq.data(), po.data(), n);
                    std::vector<MyFq12> ro(n);
                    int reg_pinned;
                    { HostRegistration r1(ps), r2(qs), r3(ro); reg_pinned = bn254_host_is_pinned(ro.data(), n * sizeof(MyFq12)); pairing_batch_into(ps.data(), qs.data(), ro.data(), n); }
                    bool same = true;
                    for (size_t j = 0; j < n; j++) same = same && po[j].flat == ark[j].flat && ro[j] == my[j];
                    std::printf("bpin %d %d %d %d\n", same ? 1 : 0, bn254_host_is_pinned(po.data(), n * sizeof(Fq12)), reg_pinned,
                                bn254_host_is_pinned(ro.data(), n * sizeof(MyFq12)));
                }
            } else if (op == "check") {
                size_t k = (size_t)rd(in), n = (size_t)rd(in);
                std::vector<G1Affine> ps(k * n); std::vector<G2Affine> qs(k * n);
                for (size_t j = 0; j < k * n; j++) { ps[j] = rd_g1(in); qs[j] = rd_g2(in); }
                std::vector<uint8_t> v = multi_pairing_check_batch(ps, qs, k);
                std::printf("check"); for (uint8_t b : v) std::printf(" %d", (int)b); std::printf("\n");
            } else if (op == "fexp") {
                pr_fq12("fexp", final_exp_native(rd_fq12(in)));
            } else if (op == "frob") {
                size_t power = (size_t)rd(in);
                pr_fq12("frob", frobenius_map_native(rd_fq12(in), power));
            } else if (op == "pow") {
                size_t n = (size_t)rd(in);
                std::vector<uint64_t> e(n); for (auto& w : e) w = rd(in);
                pr_fq12("pow", pow_native(rd_fq12(in), e));
            } else if (op == "frobc") {
                Fq2 c = frob_coeffs((size_t)rd(in));
                std::printf("frobc"); pr_fq(c.c0); pr_fq(c.c1); std::printf("\n");
            } else if (op == "naf") {
                size_t n = (size_t)rd(in);
                std::vector<uint64_t> e(n); for (auto& w : e) w = rd(in);
                std::vector<int8_t> naf = get_naf(e);
                std::printf("naf"); for (int8_t d : naf) std::printf(" %d", (int)d); std::printf("\n");
            } else {
                std::printf("unknown %s\n", op.c_str());
            }
        } catch (const Panic& p) {
            std::printf("panic %d\n", p.status);
        }
    }
    return 0;
}
