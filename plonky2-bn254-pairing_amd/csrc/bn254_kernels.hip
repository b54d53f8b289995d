// bn254_kernels.hip -- kernels + extern "C" ABI (include/bn254_pairing.h) of the MI355X-native batched BN254 pairing engine.
//
// One pairing (or one k-pair group) per lane; 256-thread workgroups (4 waves, one per SIMD), one workgroup per CU (the
// per-lane Fq12 accumulator, two coordinates of the G2 point and two temporaries fill 144 KiB of the CU's 160 KiB LDS; the
// rest of the per-lane state lives in all 512 VGPRs / AGPRs).  Kernels are persistent over 256-lane work items:
// grid = min(#items, #CUs), each workgroup walks items blockIdx.x, blockIdx.x + gridDim.x, ...  so the global scratch is
// sized by the grid, not by the batch.  The kernel bodies are generated gfx950 assembly (tools/kgen4.py, tools/kgen4_prog.py
// -> pairing_asm_gen.h): hipcc contributes the kernel descriptor and the argument SGPRs, the body is one asm statement.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <zlib.h>
#include <string.h>
#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>
#include "bn254_consts_gen.h"
#include "gen_table_gen.h"
#include "pairing_asm_gen.h"
#include "cvm_asm_gen.h"
#include "bn254_point_checks.h"
#include "../../include/bn254_pairing.h"

namespace {
constexpr int BLOCK = 256;                              // 4 waves, one per SIMD; one workgroup per CU (LDS-limited)
constexpr size_t LDS_BYTES = BN254_LDS_BYTES;           // 8 slots x 72 B x 256 lanes
constexpr size_t SLOT_BYTES = BN254_SLOT_BYTES;         // one Fq2: 2 x 9 balanced 29-bit limbs
constexpr size_t MAX_K = 64;                            // pairs per group of the multi-pairing kernels
#ifndef BN254_LATENCY_THRESHOLD_DEFAULT
// per mille of the threshold each program takes batches up to: its own measured crossover against the throughput kernel, re-measured in
// round 5 on the bank-aware programs (profiles/r05_latency.json; the sixteen-lane programs run in passes of 4 096 items, four waves per CU):
// pairing (seven launches above 4 096 items) 5.54 ms at 24 576 against 6.16, 6.33 at 28 672 against 6.16; miller_loop_native 3.20 at 24 576 against 3.69; final_exp_native
// (six launches) 2.72 at 24 576 against 2.95, 3.11 at 28 672 against 2.97; 2-pair product (Miller half + six pieces above 4 096 groups, as the others) 7.25 at 24 576
// against 8.48; 4-pair product 12.31 at 28 672 against 12.85; exact 2-pair value 5.88 at 28 672 against 6.28; exact 4-pair value 11.19 at 24 576 against 11.43
#define BN254_CVM_PM_PAIRING 1500
#define BN254_CVM_PM_MILLER 1500
#define BN254_CVM_PM_FEXP 1500
#define BN254_CVM_PM_MMILLER 1000
// (BN254_CVM_SPLIT_MIN, when defined at build time, replaces cvm_split_min()'s 16 items per CU: 4 096 on the 256 CUs of an MI355X)
#define BN254_LATENCY_THRESHOLD_DEFAULT 16384
#endif

// ------------------------------------------------------------------ kernels
// Where a kernel's code starts matters: the same kernel text ran up to 3 % slower from one library build to the next (the
// k-pair kernels most: their step loop sits at the edge of the 64 KB instruction cache), depending on where the other kernels
// had pushed it.  Every kernel therefore starts on a 64 KB boundary, its body BN254_KERNEL_PAD bytes behind it -- the best of 24
// offsets measured (profiles/r02_ab.txt: +1.0 % on the Groth16 shape, +0.5 % on 2^20 pairings against the best unaligned build,
// worst offset -3.2 %).  What is left is run-to-run: the physical placement of the code object.  (Re-scanned on round 3's kernels:
// ten offsets within +-0.8 %, inside the noise -- profiles/r03_ab.txt.)
#ifndef BN254_KERNEL_ALIGN
#define BN254_KERNEL_ALIGN 65536
#endif
#ifndef BN254_KERNEL_PAD
#define BN254_KERNEL_PAD 13824
#endif
#define BN254_STR2(x) #x
#define BN254_STR(x) BN254_STR2(x)
#define BN254_ASM_KERNEL(NAME, BLOB)                                                                                       \
    __global__ void __launch_bounds__(BLOCK, 1) __attribute__((aligned(BN254_KERNEL_ALIGN)))                               \
    NAME(const uint64_t* g1, const uint64_t* g2, const uint64_t* f_in, uint64_t* out, uint32_t n, uint32_t k, uint4* scratch, \
         uint32_t gslot_stride, int* status) {                                                                             \
        uint32_t tid = threadIdx.x, bid = blockIdx.x, grid = gridDim.x;                                                    \
        asm volatile("s_branch BN254_PAD_%=\n .fill " BN254_STR(BN254_KERNEL_PAD) ", 1, 0\n BN254_PAD_%=:\n" BLOB                 \
                     :                                                                                                     \
                     : "s"(g1), "s"(g2), "s"(f_in), "s"(out), "s"(n), "s"(k), "s"(scratch), "s"(gslot_stride), "s"(status), \
                       "v"(tid), "s"(bid), "s"(grid)                                                                       \
                     : BN254_ASM_CLOBBERS);                                                                                \
    }
BN254_ASM_KERNEL(k_pairing, BN254_ASM_PAIRING)      // pairing() = final_exp_native(miller_loop_native)        src/pairing.rs:20-22
BN254_ASM_KERNEL(k_miller, BN254_ASM_MILLER)        // the exact miller_loop_native value                       miller_loop_native.rs:320
BN254_ASM_KERNEL(k_fexp, BN254_ASM_FEXP)            // final_exp_native on arbitrary Fq12                        final_exp_native.rs:209
BN254_ASM_KERNEL(k_mpairing, BN254_ASM_MPAIRING)    // k pairs per lane, shared f (multi_miller_loop_native, :324) + final exp
BN254_ASM_KERNEL(k_mmiller, BN254_ASM_MMILLER)      // k pairs per lane, exact multi_miller_loop_native value
BN254_ASM_KERNEL(k_mmiller_u, BN254_ASM_MMILLER_U)  // ... without the line scale, on the short chain: a value only a final exponentiation may follow (the spread route's chunks)
BN254_ASM_KERNEL(k_op, BN254_ASM_OP)                // MyFq12 Mul / frobenius_map_native / pow_native (k = op | power << 8 | naf_len << 16)
BN254_ASM_KERNEL(k_generate, BN254_ASM_GENERATE)    // synthetic subgroup points: g1 / g2 = outputs, f_in = table, out = seed
BN254_ASM_KERNEL(k_g2lines, BN254_ASM_G2LINES)      // line table of FIXED G2 points (one point per lane): g2 = points, out = table
BN254_ASM_KERNEL(k_fpairing, BN254_ASM_FPAIRING)    // pairing(P0, Q0) x prod_j pairing(P_j, Qfix_j): g1 = 1 + k points per group, g2 = Q0, f_in = the table of the k fixed points
BN254_ASM_KERNEL(k_subcheck, BN254_ASM_SUBCHECK)    // G2 in the r-torsion? (ark's G2Affine::new, miller_loop_native.rs:303,311): g2 = points, out = one verdict word per point

// The LATENCY path of the scalar signatures (pairing, miller_loop_native, multi_miller_loop_native, final_exp_native): one item on
// sixteen lanes, four items per wave, one wave per workgroup.  The kernel is an interpreter of the round programs in cvm_asm_gen.h
// (tools/cvm.py: the same Miller loop and final exponentiation, scheduled over sixteen lanes -- pairing: 0.49 M instructions deep
// instead of 3.6 M).  `scratch` = the device copy of the program's blob; k = pairs per item.
#define BN254_CVM_KERNEL(NAME, BLOB)                                                                                       \
    __global__ void __launch_bounds__(64) __attribute__((aligned(BN254_KERNEL_ALIGN)))                                     \
    NAME(const uint64_t* g1, const uint64_t* g2, const uint64_t* f_in, uint64_t* out, uint32_t n, uint32_t k, uint4* scratch, \
         uint32_t gslot_stride, int* status) {                                                                             \
        uint32_t tid = threadIdx.x, bid = blockIdx.x, grid = gridDim.x;                                                    \
        asm volatile(BLOB                                                                                                  \
                     :                                                                                                     \
                     : "s"(g1), "s"(g2), "s"(f_in), "s"(out), "s"(n), "s"(k), "s"(scratch), "s"(gslot_stride), "s"(status), \
                       "v"(tid), "s"(bid), "s"(grid)                                                                       \
                     : BN254_CVM_CLOBBERS);                                                                                \
    }
BN254_CVM_KERNEL(k_cvm, BN254_ASM_CVM)               // LDS slots of 48 contiguous bytes: the fastest round (launches of up to three waves per CU)
BN254_CVM_KERNEL(k_cvm_split, BN254_ASM_CVM_SPLIT)   // 36 bytes per slot: four waves of the pairing program per CU (larger launches)
BN254_CVM_KERNEL(k_cvm_wide, BN254_ASM_CVM_WIDE)     // thirty-two lanes per item, two items per wave (launches of at most one wave per SIMD)
BN254_CVM_KERNEL(k_cvm_full, BN254_ASM_CVM_FULL)     // sixty-four lanes per item: products of three and four pairings, at most one wave per SIMD

// verdict[i] = 1 iff Fq12 element i equals the target (MyFq12 coefficient order, ark's Montgomery limbs; by value in the kernel arguments: nothing
// to upload, capturable).  The default target is MyFq12::one (coeffs[0] = R mod p, the rest 0): the check pattern of final_exp_native.rs:245-263
// (a Groth16-style product of pairings == 1), one byte per group; a verifier that holds e(alpha, beta) compares with THAT and saves the pair.
struct Fq12Words { uint64_t w[48]; };
__global__ void __launch_bounds__(256) k_is_one(const uint64_t* __restrict__ f, Fq12Words t, uint8_t* __restrict__ verdict, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t diff = 0;
        for (int w = 0; w < 48; w++) diff |= f[(size_t)w * n + i] ^ t.w[w];
        verdict[i] = diff == 0 ? 1 : 0;
    }
}
// One level of the multiplication tree over the Miller values of a group's chunks (launch_pairing, few groups of many pairs): src holds `cur` Fq12
// per group ([48][G cur] planes); a[g][i] = src[g][i], b[g][i] = src[g][i + h] for i < h = ceil(cur / 2) -- or MyFq12::one where the group has no
// such element (cur odd: its middle element is carried through the product unchanged).
__global__ void __launch_bounds__(256) k_tree_split(const uint64_t* __restrict__ src, uint64_t* __restrict__ a, uint64_t* __restrict__ b, size_t G, size_t cur, size_t h) {
    const uint64_t one[4] = BN254_FQ_ONE_LIMBS;
    size_t m = G * h, n = G * cur;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < m * 48; i += (size_t)gridDim.x * blockDim.x) {
        size_t w = i / m, r = i - w * m, g = r / h, j = r - g * h;
        a[i] = src[w * n + g * cur + j];
        b[i] = (j + h < cur) ? src[w * n + g * cur + j + h] : (w < 4 ? one[w] : 0ull);
    }
}
// Pairs [j0, j0 + ks) of every k-pair group, as a contiguous ks-pair batch (groups of more than MAX_K pairs are walked in
// sub-groups, launch_pairing): plane w of the source has n*k entries, pair j of group g at g*k + j.  HBM-bound, coalesced
// on the destination side.
__global__ void __launch_bounds__(256) k_subgroup(const uint64_t* __restrict__ src, uint64_t* __restrict__ dst, size_t n, size_t k, size_t j0,
                                                  size_t ks, int planes) {
    size_t per = n * ks;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < per * (size_t)planes; i += (size_t)gridDim.x * blockDim.x) {
        size_t w = i / per, r = i - w * per, g = r / ks, j = r - g * ks;
        dst[i] = src[w * n * k + g * k + j0 + j];
    }
}

// status[1] |= 2 when a point of the batch is the point at infinity in ark's affine form (x = y = 0, all limbs zero): the reference reads
// raw x / y and never looks at the `infinity` flag, so such an input is outside its contract; callers that want
// the distinct status SURVEY 8(b) asks for run this check (HBM-bound: every input word is read once).
__global__ void __launch_bounds__(256) k_check_points(const uint64_t* __restrict__ g1, const uint64_t* __restrict__ g2, size_t n, int* status) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t a = 0, b = 0;
        for (int w = 0; w < 8; w++) a |= g1[(size_t)w * n + i];
        for (int w = 0; w < 16; w++) b |= g2[(size_t)w * n + i];
        if (a == 0 || b == 0) atomicOr(status, 2);      // `status` = the stream's point-check word (not the kernels' zero-divisor word)
    }
}

enum { OP_MUL = 0, OP_FROB = 1, OP_POW = 2 };

// Element-major <-> limb-major.  The reference's callers hold `&[G1Affine]`, `Vec<(&G1Affine, &G2Affine)>`, `Vec<MyFq12>`,
// `Vec<Fq12>`: one element after the other, W = 8 / 16 / 48 words each; the pairing kernels want plane (c, l) of all
// elements contiguous.  HBM-bound: every word is read once and written once, both sides in runs of >= 512 B per wave
// (a tile of T elements goes through LDS, rows padded by one word so that the transposed reads are conflict-free).
// W = 48 with order = BN254_FQ12_ARK also applies the MyFq12 <-> ark Fq12 coefficient permutation (`.into()`, pairing.rs:21).
__device__ __forceinline__ int fq12_plane(int w, int order) {      // SoA plane of word w of an element
    if (order == 0) return w;
    int j = w >> 2, h = j / 6, k = (j % 6) >> 1, e = j & 1;        // ark flat coefficient j = ((h*3 + k)*2 + e)
    return (((2 * k + h) + 6 * e) << 2) | (w & 3);
}
template <int W, int T, bool TO_SOA>
__global__ void __launch_bounds__(256) k_layout(const uint64_t* __restrict__ src, uint64_t* __restrict__ dst, size_t n, int order) {
    __shared__ uint64_t tile[T][W + 1];
    size_t i0 = (size_t)blockIdx.x * T;
    int cnt = (int)(n - i0 < (size_t)T ? n - i0 : (size_t)T);
    if (TO_SOA) {
        const uint64_t* in = src + i0 * W;
        for (int idx = threadIdx.x; idx < cnt * W; idx += 256) tile[idx / W][idx % W] = in[idx];
        __syncthreads();
        for (int idx = threadIdx.x; idx < T * W; idx += 256) {
            int w = idx / T, e = idx % T;
            if (e < cnt) dst[(size_t)(W == 48 ? fq12_plane(w, order) : w) * n + i0 + e] = tile[e][w];
        }
    } else {
        for (int idx = threadIdx.x; idx < T * W; idx += 256) {
            int w = idx / T, e = idx % T;
            if (e < cnt) tile[e][w] = src[(size_t)(W == 48 ? fq12_plane(w, order) : w) * n + i0 + e];
        }
        __syncthreads();
        uint64_t* out = dst + i0 * W;
        for (int idx = threadIdx.x; idx < cnt * W; idx += 256) out[idx] = tile[idx / W][idx % W];
    }
}

// ------------------------------------------------------------------ host side
// Scratch, the status word and the staging buffers of the host-pointer entry points are kept per (device, stream): calls on
// different streams of one device are independent ("library is re-entrant, one HIP stream per call").  Calls on
// ONE stream are stream-ordered on the device; on the host they are serialised by the stream context's mutex: a `_dev` call
// holds it from the look-up of its buffers to the launch, a host-pointer call from staging to the read-back of its results
// (two host threads may therefore share a stream -- e.g. the NULL stream behind the scalar Rust / C++ signatures).
struct Buf {
    void* p = nullptr;
    size_t bytes = 0;
};
struct NafSlot {               // pinned host staging of one pow_native call's NAF digits (no stream synchronisation in the call)
    int8_t* host = nullptr;
    size_t bytes = 0;
    hipEvent_t done = nullptr; // recorded behind the copy that reads `host`
    bool used = false;         // `done` has been recorded at least once
};
constexpr size_t NAF_RING_MAX = 64;
struct StreamCtx {
    std::recursive_mutex mu;
    Buf scratch, naf, tmp;
    Buf mid;                   // pairing() in several launches (mid-size batches on the lane-cooperative kernel): the Miller values in between
    Buf fx[5];                 // ... and final_exp_native's: m, m^x, m^(x^2), m^(x^3), the y-chain's first part
    Buf sub[4];                // groups of more than MAX_K pairs: sub-group inputs (G1, G2) and the two Miller values in flight
    Buf stage[8];              // device staging of the host-pointer entry points (inputs / outputs), grown on demand
    // host-pointer fixed-G2 calls: the line tables of the last FIXED_TABLES distinct sets of fixed points (+ room to stage them) and those points (and their
    // layout) -- a verifier's keys do not change between its calls, a table is then made once; the least recently used entry makes room
    static constexpr int FIXED_TABLES = 4;
    Buf fixed_tab[FIXED_TABLES];
    std::vector<uint64_t> fixed_key[FIXED_TABLES];
    uint64_t fixed_used[FIXED_TABLES] = {0, 0, 0, 0}, fixed_clock = 0;
    int fixed_last = -1;       // the entry the running call uses (dropped if the call raises the status: its making may have been what raised it)
    std::vector<void*> retired;   // buffers that were outgrown while work on them may still be queued: freed once the stream has been
                                  // synchronised (bn254_last_status, bn254_release_stream) -- growing never waits for the stream
    std::vector<NafSlot> naf_ring;
    int* status = nullptr;        // TWO words: [0] the zero-divisor flag (the generated kernels store a plain 1), [1] the point-check
                                  // flags (atomicOr: 2 infinity, 4 not on the curve, 8 G2 not in the subgroup) -- separate words, so a
                                  // pairing kernel that trips over the all-zero point cannot overwrite the more specific verdict
    int* status_host = nullptr;   // pinned: the status words are read back on the caller's stream
    std::atomic<size_t> lat_threshold{(size_t)-1};   // bn254_set_stream_latency: this stream's own kernel selection; (size_t)-1 / -1 = the process-wide
    std::atomic<int> lat_lanes{-1};                  // defaults (bn254_set_latency_threshold / _lanes)
    std::atomic<int> last_kernel{0};     // bn254_last_kernel: 1 = throughput kernel (one item per lane), 16 / 32 / 64 = lanes per item (written under the
                                         // context's mutex, read by the diagnostic without it)
    size_t last_pitch = 0;        // scratch geometry of the most recent launch (diagnostic builds read their clock stamps back from it)
    uint32_t last_grid = 0;
    ~StreamCtx() {                // the last holder (bn254_release_stream, after the stream has been synchronised) frees everything
        for (Buf* b : {&scratch, &naf, &tmp, &mid}) if (b->p) (void)hipFree(b->p);
        for (Buf& b : fixed_tab) if (b.p) (void)hipFree(b.p);
        for (Buf& b : fx) if (b.p) (void)hipFree(b.p);
        for (Buf& b : stage) if (b.p) (void)hipFree(b.p);
        for (Buf& b : sub) if (b.p) (void)hipFree(b.p);
        for (void* q : retired) (void)hipFree(q);
        for (NafSlot& n : naf_ring) { if (n.host) (void)hipHostFree(n.host); if (n.done) (void)hipEventDestroy(n.done); }
        if (status) (void)hipFree(status);
        if (status_host) (void)hipHostFree(status_host);
    }
};
struct DeviceCtx {
    std::mutex mu;             // guards `streams`, `shard_streams` and the one-time initialisation; not held across kernel work
    bool init = false;
    int n_cu = 0;
    std::mutex table_mu;       // the generator table upload (138 KB, blocking) has its own lock
    int32_t* gen_table = nullptr;
    uint32_t* cvm_blob[40] = {};    // the latency path's round programs (0.3 - 1.4 MB each, uploaded on first use, same lock)
    bool cvm_init = false;
    std::map<hipStream_t, std::shared_ptr<StreamCtx>> streams;   // shared: a call keeps its context alive across a concurrent release
    std::mutex pipe_mu;        // one host-pointer pipeline at a time per device (its two workers fill the chip anyway)
    hipStream_t pipe_stream[2] = {nullptr, nullptr};   // the workers' private streams: created once, their scratch and staging kept
    hipEvent_t pipe_ev[2] = {nullptr, nullptr};        // ... and the event each worker waits for its chunk on
    void* pipe_table = nullptr;                         // line table of a fixed-G2 pipeline call (+ room to stage its points): made per call, under pipe_mu
    hipEvent_t pipe_table_ev = nullptr;
    std::vector<uint64_t> pipe_table_key;               // the fixed points (and their layout) the table in pipe_table was made from: a verifier calls again with the same key
    std::mutex shard_mu;       // one device-pointer sharded call at a time per device
    std::vector<hipStream_t> shard_streams;            // private streams of bn254_*_sharded_dev (one per shard on this device)
};
DeviceCtx g_ctx[64];

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { return BN254_ERR_HIP; } } while (0)

// target: 48 words in HOST memory (canonical Montgomery limbs, what the pairing entry points return), or null = MyFq12::one
static int launch_is_equal(const uint64_t* f, const uint64_t* target, uint8_t* verdict, size_t n, void* stream) {
    Fq12Words t;
    const uint64_t one[4] = BN254_FQ_ONE_LIMBS;
    for (int w = 0; w < 48; w++) t.w[w] = target ? target[w] : (w < 4 ? one[w] : 0ull);
    size_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_is_one, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)stream, f, t, verdict, n);
    HIPCHK(hipGetLastError());
    return BN254_OK;
}

int check_device(int device) {
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) return BN254_ERR_NO_DEVICE;
    if (device < 0 || device >= cnt || device >= 64) return BN254_ERR_INVALID_ARG;
    return hipSetDevice(device) == hipSuccess ? BN254_OK : BN254_ERR_HIP;
}

// Grows a per-stream buffer WITHOUT waiting for the stream: work queued on the stream may still use the old buffer, so it is
// retired (kept until the next synchronisation point of the context) and a new one is allocated.  bn254_reserve sizes every
// buffer up front, after which no `_dev` call allocates at all.
int ensure(StreamCtx* sc, Buf& b, size_t bytes) {
    if (bytes <= b.bytes) return BN254_OK;
    void* q = nullptr;
    if (hipMalloc(&q, bytes) != hipSuccess) { (void)hipGetLastError(); return BN254_ERR_ALLOC; }
    if (b.p) sc->retired.push_back(b.p);
    b.p = q;
    b.bytes = bytes;
    return BN254_OK;
}
// the stream has just been synchronised (context mutex held): nothing can still be using the outgrown buffers
void free_retired(StreamCtx* sc) {
    for (void* q : sc->retired) (void)hipFree(q);
    sc->retired.clear();
}

std::shared_ptr<StreamCtx> stream_ctx(int device, void* stream) {
    DeviceCtx& c = g_ctx[device];
    std::lock_guard<std::mutex> lk(c.mu);
    std::shared_ptr<StreamCtx>& p = c.streams[(hipStream_t)stream];
    if (!p) p = std::make_shared<StreamCtx>();
    return p;
}

// Scratch layout (pairing_asm_gen.h: BN254_SCRATCH_WG_CONTIGUOUS).  Contiguous per workgroup: [workgroup][slot][wave][...], the
// workgroup pitch rounded up to 2 MiB so that a CU's 80+ slots share one or two pages (the slot-major layout put every slot of
// a workgroup on a different 2 MiB page: 4.7 MB apart at a full grid); the kernels take the pitch as their stride argument.
// + the line area of the split Miller loop (one slot triple per point step and pair; groups of up to BN254_FIS_MAX_K pairs)
size_t scratch_slots(size_t k) {
    size_t lines = (k >= 1 && k <= (size_t)BN254_FIS_MAX_K) ? (size_t)BN254_FIS_LINE_SLOTS * k : 0;
    return (size_t)BN254_GSLOTS + BN254_GSLOTS_PER_PAIR * (k > 1 ? k : 0) + lines;
}
size_t scratch_pitch(size_t k, size_t grid) {
    if (!BN254_SCRATCH_WG_CONTIGUOUS) return grid * BLOCK * SLOT_BYTES;                 // bytes between slots
    size_t wg = scratch_slots(k) * BLOCK * SLOT_BYTES, page = (size_t)2 << 20;
    return (wg + page - 1) / page * page;                                              // bytes between workgroups
}

struct LaunchCtx {
    std::shared_ptr<StreamCtx> s;                    // (declared first: the lock below is released before the context can go)
    std::unique_lock<std::recursive_mutex> lock;     // the stream context stays ours until the launch has been issued
    int n_cu;
    uint4* scratch;
    int* status;
    const int32_t* gen_table;
    uint32_t grid, stride;
};

// n_items 256-lane work items, k pairs per lane (scratch: 80 slots + 7 per pair when k > 1)
int ctx_get(int device, void* stream, size_t k, size_t n_items, LaunchCtx* out, bool want_table = false) {
    int rc = check_device(device);
    if (rc) return rc;
    DeviceCtx& c = g_ctx[device];
    {
        std::lock_guard<std::mutex> lk(c.mu);
        if (!c.init) {
            hipDeviceProp_t prop;
            HIPCHK(hipGetDeviceProperties(&prop, device));
            c.n_cu = prop.multiProcessorCount;
            const void* kernels[] = {(const void*)k_pairing, (const void*)k_miller, (const void*)k_fexp, (const void*)k_mpairing,
                                     (const void*)k_mmiller, (const void*)k_op, (const void*)k_generate, (const void*)k_subcheck,
                                     (const void*)k_g2lines, (const void*)k_fpairing, (const void*)k_mmiller_u};
            for (const void* f : kernels) HIPCHK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
            c.init = true;
        }
    }
    if (want_table) {                              // 138 KB, once per device
        std::lock_guard<std::mutex> lk(c.table_mu);
        if (!c.gen_table) {
            int32_t* t = nullptr;
            if (hipMalloc(&t, sizeof(BN254_GEN_TABLE)) != hipSuccess) return BN254_ERR_ALLOC;
            if (hipMemcpy(t, BN254_GEN_TABLE, sizeof(BN254_GEN_TABLE), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(t); return BN254_ERR_HIP; }
            c.gen_table = t;
        }
    }
    std::shared_ptr<StreamCtx> sc = stream_ctx(device, stream);
    out->lock = std::unique_lock<std::recursive_mutex>(sc->mu);
    out->s = sc;            // before anything can fail: the lock above must never outlive the last reference to its mutex
    if (!sc->status) {
        if (hipMalloc(&sc->status, 2 * sizeof(int)) != hipSuccess) { sc->status = nullptr; return BN254_ERR_ALLOC; }
        HIPCHK(hipMemsetAsync(sc->status, 0, 2 * sizeof(int), (hipStream_t)stream));
    }
    uint32_t grid = (uint32_t)(n_items < (size_t)c.n_cu ? n_items : (size_t)c.n_cu);
    if (grid == 0) grid = 1;
    size_t pitch = scratch_pitch(k, grid);
    // the kernels take the pitch as a 32-bit argument (the workgroup's base blockIdx * pitch is formed in 64 bits) and address
    // slots inside a block with 32-bit byte offsets
    if (pitch >= (1ull << 32)) return BN254_ERR_INVALID_ARG;
    if ((rc = ensure(sc.get(), sc->scratch, BN254_SCRATCH_WG_CONTIGUOUS ? pitch * grid : pitch * scratch_slots(k)))) return rc;
    out->n_cu = c.n_cu;
    out->scratch = (uint4*)sc->scratch.p;
    out->status = sc->status;
    out->gen_table = c.gen_table;
    out->grid = grid;
    out->stride = (uint32_t)pitch;
    sc->last_pitch = pitch;
    sc->last_grid = grid;
    return BN254_OK;
}

int launch_op(int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, size_t power, const int8_t* naf_host, int naf_len,
              int device, void* stream);

// Batches of at most this many items take the lane-cooperative kernel where a program exists (bn254_set_latency_threshold; 0: never).
std::atomic<size_t> g_latency_threshold{BN254_LATENCY_THRESHOLD_DEFAULT};
std::atomic<int> g_latency_lanes{0};       // bn254_set_latency_lanes: 0 = by launch size, 16 / 32 = that program family whatever the size

// the kernel selection of a call: the stream's own setting where it has one (bn254_set_stream_latency), else the process-wide default
void latency_cfg(int device, void* stream, size_t* threshold, int* lanes) {
    size_t t = (size_t)-1;
    int l = -1;
    if (device >= 0 && device < 64) {
        DeviceCtx& c = g_ctx[device];
        std::lock_guard<std::mutex> lk(c.mu);
        auto it = c.streams.find((hipStream_t)stream);
        if (it != c.streams.end()) { t = it->second->lat_threshold.load(); l = it->second->lat_lanes.load(); }
    }
    *threshold = t == (size_t)-1 ? g_latency_threshold.load() : t;
    *lanes = l < 0 ? g_latency_lanes.load() : l;
}

struct CvmProgram {
    const char* b64;          // the program blob, deflated and base64-encoded (cvm_asm_gen.h)
    size_t z_bytes;           // deflated
    size_t bytes;             // inflated
    uint32_t slots;
    uint32_t per_mille;       // share of the threshold this program takes batches up to (its own crossover against the throughput kernel)
    int wide;                 // index of the same function's thirty-two-lane program, or -1
    int full;                 // ... of its sixty-four-lane program, or -1
};
#define CVM_PROGRAM(NAME, PM, WIDE, FULL) {BN254_CVM_##NAME##_B64, BN254_CVM_##NAME##_Z_BYTES, BN254_CVM_##NAME##_BYTES, BN254_CVM_##NAME##_SLOTS, PM, WIDE, FULL}
constexpr int CVM_N_PROGRAMS = 35;
static_assert(CVM_N_PROGRAMS <= 40, "DeviceCtx::cvm_blob");
constexpr int CVM_MILLER_U = 26;      // the Miller half of pairing() (separate launch for mid-size batches)
constexpr int CVM_EASY = 27, CVM_POWX = 28, CVM_YCH1 = 29, CVM_YCH2 = 30;      // final_exp_native in six launches (mid-size batches)
constexpr int CVM_MMILLER_U = 29;    // + k (2, 3, 4): the Miller halves of the k-pair products
const CvmProgram CVM_PROGRAMS[CVM_N_PROGRAMS] = {
    CVM_PROGRAM(PAIRING, BN254_CVM_PM_PAIRING, 9, 18), CVM_PROGRAM(MILLER, BN254_CVM_PM_MILLER, 10, 22), CVM_PROGRAM(FEXP, BN254_CVM_PM_FEXP, 11, 34),
    CVM_PROGRAM(MULTI2, 1500, 12, 19), CVM_PROGRAM(MULTI3, 1500, 13, 20), CVM_PROGRAM(MULTI4, 1750, 14, 21),
    CVM_PROGRAM(MMILLER2, 1750, 15, 23), CVM_PROGRAM(MMILLER3, BN254_CVM_PM_MMILLER, 16, 24), CVM_PROGRAM(MMILLER4, 1500, 17, 25),
    CVM_PROGRAM(PAIRING_W, 0, -1, -1), CVM_PROGRAM(MILLER_W, 0, -1, -1), CVM_PROGRAM(FEXP_W, 0, -1, -1),
    CVM_PROGRAM(MULTI2_W, 0, -1, -1), CVM_PROGRAM(MULTI3_W, 0, -1, -1), CVM_PROGRAM(MULTI4_W, 0, -1, -1),
    CVM_PROGRAM(MMILLER2_W, 0, -1, -1), CVM_PROGRAM(MMILLER3_W, 0, -1, -1), CVM_PROGRAM(MMILLER4_W, 0, -1, -1),
    CVM_PROGRAM(PAIRING_X, 0, -1, -1), CVM_PROGRAM(MULTI2_X, 0, -1, -1), CVM_PROGRAM(MULTI3_X, 0, -1, -1), CVM_PROGRAM(MULTI4_X, 0, -1, -1),
    CVM_PROGRAM(MILLER_X, 0, -1, -1), CVM_PROGRAM(MMILLER2_X, 0, -1, -1), CVM_PROGRAM(MMILLER3_X, 0, -1, -1), CVM_PROGRAM(MMILLER4_X, 0, -1, -1),
    CVM_PROGRAM(MILLER_U, 0, -1, -1), CVM_PROGRAM(EASY, 0, -1, -1), CVM_PROGRAM(POWX, 0, -1, -1), CVM_PROGRAM(YCH1, 0, -1, -1), CVM_PROGRAM(YCH2, 0, -1, -1),
    CVM_PROGRAM(MMILLER2_U, 0, -1, -1), CVM_PROGRAM(MMILLER3_U, 0, -1, -1), CVM_PROGRAM(MMILLER4_U, 0, -1, -1),
    CVM_PROGRAM(FEXP_X, 0, -1, -1)};

// which program serves (Miller loop?, final exponentiation?, k pairs); -1: none
template <bool M, bool F>
int cvm_program(size_t k) {
    if (M && F) return k == 1 ? 0 : (k <= 4 ? 1 + (int)k : -1);
    if (M) return k == 1 ? 1 : (k <= 4 ? 4 + (int)k : -1);
    return k == 1 ? 2 : -1;
}

// The program's blob on the device: inflated (zlib) and uploaded once -- a blocking copy of 0.3 - 1.4 MB on the first call that needs it;
// bn254_reserve does it for every program up front.
int cvm_upload(int device, int prog) {
    DeviceCtx& d = g_ctx[device];
    const CvmProgram& p = CVM_PROGRAMS[prog];
    std::lock_guard<std::mutex> lk(d.table_mu);
    if (!d.cvm_init) {
        HIPCHK(hipFuncSetAttribute((const void*)k_cvm, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHK(hipFuncSetAttribute((const void*)k_cvm_split, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHK(hipFuncSetAttribute((const void*)k_cvm_wide, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHK(hipFuncSetAttribute((const void*)k_cvm_full, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        d.cvm_init = true;
    }
    if (!d.cvm_blob[prog]) {
        std::vector<unsigned char> z(p.z_bytes + 3), raw(p.bytes);
        {   // base64 -> deflate stream
            auto val = [](char ch) -> uint32_t {
                return ch >= 'A' && ch <= 'Z' ? ch - 'A' : ch >= 'a' && ch <= 'z' ? ch - 'a' + 26 : ch >= '0' && ch <= '9' ? ch - '0' + 52 : ch == '+' ? 62 : ch == '/' ? 63 : 0;
            };
            size_t o = 0;
            for (const char* q = p.b64; q[0] && q[1] && q[2] && q[3] && o + 3 <= z.size(); q += 4) {
                uint32_t v = val(q[0]) << 18 | val(q[1]) << 12 | val(q[2]) << 6 | val(q[3]);
                z[o++] = (unsigned char)(v >> 16); z[o++] = (unsigned char)(v >> 8); z[o++] = (unsigned char)v;
            }
            if (o < p.z_bytes) return BN254_ERR_HIP;
        }
        uLongf got = (uLongf)p.bytes;
        if (uncompress(raw.data(), &got, z.data(), (uLong)p.z_bytes) != Z_OK || got != p.bytes) return BN254_ERR_HIP;
        uint32_t* t = nullptr;
        if (hipMalloc(&t, p.bytes) != hipSuccess) { (void)hipGetLastError(); return BN254_ERR_ALLOC; }
        if (hipMemcpy(t, raw.data(), p.bytes, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(t); return BN254_ERR_HIP; }
        d.cvm_blob[prog] = t;
    }
    return BN254_OK;
}

// waves of the interpreter that a CU holds at `lds` bytes each (160 KB of LDS; 248 registers: two waves per SIMD)
size_t resident_waves(size_t lds) {
    size_t w = lds ? (160 * 1024) / lds : 8;
    if (w > 4 && w < 8) w = 4;       // five to seven waves leave one or two SIMDs with two waves and the others with one: the launch then takes as long as
                                     // the two-wave SIMDs need (1.6 .. 1.8 single-wave times for 1.25 .. 1.75 x the waves) -- one per SIMD is faster
    return w > 8 ? 8 : (w ? w : 1);
}

// pairing() / final_exp_native on the lane-cooperative kernel go in several launches above this many items: one wave per SIMD of the fused
// program (4 waves x 4 items per CU)
int n_cu_of(int device) { return device >= 0 && device < 64 ? g_ctx[device].n_cu : 0; }      // 0 before the device's first launch
size_t cvm_split_min(int n_cu) {
#ifdef BN254_CVM_SPLIT_MIN
    (void)n_cu;
    return (size_t)BN254_CVM_SPLIT_MIN;
#else
    return (size_t)16 * (size_t)(n_cu > 0 ? n_cu : 256);
#endif
}
// ... and the largest item count that path can take under threshold `thr`: thr x the largest per-mille share of a program that has the
// several-launch form (pairing, final_exp_native, the 2 / 3 / 4-pair products), below the kernels' 2^22-lane limit
size_t cvm_split_max(size_t thr) {
    uint32_t pm = 0;
    for (int prog : {0, 2, 3, 4, 5}) if (CVM_PROGRAMS[prog].per_mille > pm) pm = CVM_PROGRAMS[prog].per_mille;
    unsigned __int128 cap = (unsigned __int128)thr * pm / 1000u;
    return cap >= ((size_t)1 << 22) ? ((size_t)1 << 22) - 1 : (size_t)cap;
}

int launch_cvm(int prog, int lanes, const uint64_t* g1, const uint64_t* g2, const uint64_t* f_in, uint64_t* out, size_t n, size_t k, int device, void* stream) {
    LaunchCtx c;
    int rc = ctx_get(device, stream, 1, 1, &c);          // the status word and the stream context; the kernel needs no scratch
    if (rc) return rc;
    DeviceCtx& d = g_ctx[device];
    // The smallest launches -- at most one wave per SIMD with two items per wave -- take the function's thirty-two-lane program where
    // there is one: fewer, fuller rounds (pairing: 979 instead of 1 326).
    // ... and products of three and four pairings, while the launch is at most one wave per SIMD with ONE item per wave, the sixty-four-lane
    // program: the lines of a step multiplied with each other off f's chain (four-pair product: 1 008 rounds instead of 1 297)
    if (CVM_PROGRAMS[prog].full >= 0) {
        int x = CVM_PROGRAMS[prog].full;
        const CvmProgram& px = CVM_PROGRAMS[x];
        size_t lds_x = (size_t)px.slots * BN254_CVM_SLOT_BYTES, cap = resident_waves(lds_x) * (size_t)c.n_cu;
        size_t once = cap < (size_t)4 * (size_t)c.n_cu ? cap : (size_t)4 * (size_t)c.n_cu;       // one pass, one wave per SIMD at most
        if (lanes == 64 || (lanes == 0 && n <= once)) {
            if ((rc = cvm_upload(device, x))) return rc;
            hipLaunchKernelGGL(k_cvm_full, dim3((uint32_t)(n < cap ? n : cap)), dim3(64), lds_x, (hipStream_t)stream, g1, g2, f_in, out, (uint32_t)n,
                               (uint32_t)k, (uint4*)d.cvm_blob[x], 0u, c.status);
            HIPCHK(hipGetLastError());
            c.s->last_kernel = 64;
            return BN254_OK;
        }
    }
    if (CVM_PROGRAMS[prog].wide >= 0 && (lanes == 32 || lanes == 64 || (lanes == 0 && n <= (size_t)2 * 4 * (size_t)c.n_cu))) {
        int w = CVM_PROGRAMS[prog].wide;
        if ((rc = cvm_upload(device, w))) return rc;
        const CvmProgram& pw = CVM_PROGRAMS[w];
        size_t lds_w = (size_t)2 * pw.slots * BN254_CVM_SLOT_BYTES, cap = resident_waves(lds_w) * (size_t)c.n_cu, need = (n + 1) / 2;
        hipLaunchKernelGGL(k_cvm_wide, dim3((uint32_t)(need < cap ? need : cap)), dim3(64), lds_w, (hipStream_t)stream, g1, g2, f_in,
                           out, (uint32_t)n, (uint32_t)k, (uint4*)d.cvm_blob[w], 0u, c.status);
        HIPCHK(hipGetLastError());
        c.s->last_kernel = 32;
        return BN254_OK;
    }
    const CvmProgram& p = CVM_PROGRAMS[prog];
    if ((rc = cvm_upload(device, prog))) return rc;
    // One wave per workgroup; the grid is what is RESIDENT (LDS-limited, at most two waves per SIMD), every wave walks its items: the
    // hardware's own distribution of more workgroups than fit left CUs a whole pass behind the others.  The contiguous slot layout
    // is the faster one (one address computation less per operand); the split layout needs 3/4 of the LDS, so more waves fit a CU:
    // taken when the launch has more waves than the contiguous layout holds resident AND the split layout holds more.
    size_t need = (n + BN254_CVM_GROUPS - 1) / BN254_CVM_GROUPS;
    size_t lds = (size_t)BN254_CVM_GROUPS * p.slots * BN254_CVM_SLOT_BYTES, lds_split = (size_t)BN254_CVM_GROUPS * p.slots * BN254_CVM_SLOT_BYTES_SPLIT;
    size_t per_cu = resident_waves(lds), per_cu_split = resident_waves(lds_split);
    bool split = per_cu_split > per_cu && need > per_cu * (size_t)c.n_cu;
    size_t cap = (split ? per_cu_split : per_cu) * (size_t)c.n_cu;
    hipLaunchKernelGGL(split ? k_cvm_split : k_cvm, dim3((uint32_t)(need < cap ? need : cap)), dim3(64), split ? lds_split : lds, (hipStream_t)stream,
                       g1, g2, f_in, out, (uint32_t)n, (uint32_t)k, (uint4*)d.cvm_blob[prog], 0u, c.status);
    HIPCHK(hipGetLastError());
    c.s->last_kernel = 16;
    return BN254_OK;
}

// final_exp_native on a mid-size batch as SIX launches of the lane-cooperative kernel -- easy part, the three x-powers (one program,
// three times), the y-chain in two parts (the first takes three Fq12 batches through the g1 / g2 / f_in arguments, the second two) --
// through per-stream buffers.  The whole program holds 274 slots per item (BN254_CVM_FEXP_SLOTS: four waves per CU); its pieces 63 / 141 /
// 141 / 135 (BN254_CVM_EASY / POWX / YCH1 / YCH2_SLOTS): eight waves per CU -- two per SIMD, one's operand fetch under the other's
// arithmetic -- the easy part in either LDS layout, the other three through the 36-byte split layout only (the contiguous layout would hold
// six or seven, which resident_waves() clamps to four).  Same values: the same operations in the same order on the same limbs
// (tests/test_cvm.py composes the pieces on integers; tests/test_gpu_latency.py on the GPU).
int launch_fexp_pieces(const uint64_t* f_in, uint64_t* out, size_t n, int device, void* stream) {
    LaunchCtx hold;
    int rc = ctx_get(device, stream, 1, 1, &hold);
    if (rc) return rc;
    StreamCtx* sc = hold.s.get();
    for (Buf& b : sc->fx)
        if ((rc = ensure(sc, b, 384 * n))) return rc;
    uint64_t *m = (uint64_t*)sc->fx[0].p, *mx = (uint64_t*)sc->fx[1].p, *mx2 = (uint64_t*)sc->fx[2].p, *mx3 = (uint64_t*)sc->fx[3].p, *t1 = (uint64_t*)sc->fx[4].p;
    if ((rc = launch_cvm(CVM_EASY, 16, nullptr, nullptr, f_in, m, n, 1, device, stream))) return rc;
    if ((rc = launch_cvm(CVM_POWX, 16, nullptr, nullptr, m, mx, n, 1, device, stream))) return rc;
    if ((rc = launch_cvm(CVM_POWX, 16, nullptr, nullptr, mx, mx2, n, 1, device, stream))) return rc;
    if ((rc = launch_cvm(CVM_POWX, 16, nullptr, nullptr, mx2, mx3, n, 1, device, stream))) return rc;
    if ((rc = launch_cvm(CVM_YCH1, 16, mx, mx2, mx3, t1, n, 1, device, stream))) return rc;
    return launch_cvm(CVM_YCH2, 16, m, nullptr, t1, out, n, 1, device, stream);
}

// I/O layout of a launch of the generated kernels (bits 28..30 of their k argument, tools/kgen4_prog.py: S_MODE): which of the arrays are
// ELEMENT-major -- one G1Affine / G2Affine / Fq12 after the other, as the reference's callers hold them -- instead of limb-major planes
enum { IO_IN_ELEMS = 1, IO_OUT_ELEMS = 2, IO_OUT_ARK = 4, IO_NO_OWN = 8 };      // (IO_NO_OWN: k_fpairing only -- groups without a pair of their own, MODE_NO_OWN)

constexpr size_t FULL_GRID_LANES = 65536;          // one 256-lane work item on each of 256 CUs
std::atomic<size_t> g_wide_groups{FULL_GRID_LANES};   // bn254_set_wide_groups: batches of fewer groups (of more than MAX_K pairs) spread a group over several lanes
constexpr int BN254_ERR_NOT_WIDE = -1000;          // (internal: launch_wide declines, the caller goes on with its own route)
// FEW groups of MANY pairs (one aggregated check over thousands of pairs): a lane per group would leave the chip empty and walk its group for seconds.
// More than MAX_K pairs: whenever the batch has less than a grid's worth of groups.  5 .. MAX_K pairs (no lane-cooperative program): when the spread route's
// estimated time beats one launch of the k-pair kernel, which on a partial grid is latency-bound (one group of 64 pairs: 145 ms on one lane).
// Pairs per lane of the spread route, C | k, C < k: the divisor with the smallest estimated time -- passes over the grid x what a lane of C pairs costs
// (the C-pair Miller kernel: 2.2 ms per pair with the shared f, + 1.5), + the tree and the final exponentiation.  Measured anchors: one lane of 8 / 64 pairs
// 21.6 / 145 ms, the exact one-pair Miller kernel 3.7 ms per grid.
inline size_t wide_chunk(size_t n_groups, size_t k, double* est_ms) {
    size_t best = 1;
    double best_ms = 1e300;
    for (size_t d = k < MAX_K ? k - 1 : MAX_K; d >= 1; d--) {
        if (d >= k || k % d) continue;
        size_t lanes = n_groups * (k / d);
        if (lanes >= ((size_t)1 << 22)) continue;
        double ms = (double)((lanes + FULL_GRID_LANES - 1) / FULL_GRID_LANES) * (2.2 * (double)d + 1.5) + 4.0;
        if (ms < best_ms) { best_ms = ms; best = d; }
    }
    if (est_ms) *est_ms = best_ms;
    return best;
}
thread_local bool tl_in_wide = false;               // (the spread route's own Miller launch never spreads again: its buffers are in use -- by the choice of C it would not anyway)
inline bool takes_wide_route(size_t n_groups, size_t k) {
    if (tl_in_wide || k <= 4 || n_groups >= g_wide_groups.load()) return false;
    double wide_ms;
    (void)wide_chunk(n_groups, k, &wide_ms);
    if (wide_ms > 1e299) return false;                       // (no divisor keeps the lane count below the kernels' limit)
    if (k > MAX_K) return true;                              // (the alternative walks every group on one lane, sub-group after sub-group)
    return wide_ms * 1.15 < 2.2 * (double)k + 4.0;           // 5 .. MAX_K pairs: against one launch of the k-pair kernel, latency-bound on a partial grid
}
// would launch_pairing<M, F> serve this batch on the lane-cooperative kernel (whose programs read limb-major planes only)?
template <bool M, bool F>
bool takes_latency_kernel(size_t n_groups, size_t k, int device, void* stream) {
    int prog = cvm_program<M, F>(k);
    if (prog < 0 || n_groups * k >= (1u << 22)) return false;
    size_t thr; int lanes;
    latency_cfg(device, stream, &thr, &lanes);
    return (unsigned __int128)n_groups * 1000u <= (unsigned __int128)thr * CVM_PROGRAMS[prog].per_mille;
}
// can the throughput kernel take this batch straight from / into element-major arrays?  The lane offsets are 32-bit: 128 bytes per G2 element, 384
// per Fq12 -- 2^23 units keep the largest (the result's: 3.2 GB) below 4 GB; larger batches take the transposition route (64-bit plane walks)
template <bool M, bool F>
bool direct_elems_ok(size_t n_groups, size_t k, int device, void* stream) {
    return k >= 1 && k <= MAX_K && n_groups * k <= ((size_t)1 << 23) && !takes_latency_kernel<M, F>(n_groups, k, device, stream) && !(M && takes_wide_route(n_groups, k));
}

template <bool M, bool F>
int launch_pairing(const uint64_t* g1, const uint64_t* g2, const uint64_t* f_in, uint64_t* out, size_t n_groups, size_t k, int device, void* stream, int io_mode = 0);
// The Miller value of n groups of k pairs as ONE FACTOR of a value that a final exponentiation follows (the chunks of a spread group, the sub-groups of
// a long one): it need not be the exact multi_miller_loop_native value -- no line scale, the short chain (k_mmiller_u: 2^20 pairs in chunks of 16
// 41 ms instead of 46).  One pair per lane, or a batch the lane-cooperative programs serve: the exact kernels.
int launch_miller_part(const uint64_t* g1, const uint64_t* g2, uint64_t* out, size_t n, size_t k, int device, void* stream) {
    if (k < 2 || k > MAX_K || takes_latency_kernel<true, false>(n, k, device, stream)) return launch_pairing<true, false>(g1, g2, nullptr, out, n, k, device, stream, 0);
    LaunchCtx c;
    int rc = ctx_get(device, stream, k, (n + BLOCK - 1) / BLOCK, &c);
    if (rc) return rc;
    hipLaunchKernelGGL(k_mmiller_u, dim3(c.grid), dim3(BLOCK), LDS_BYTES, (hipStream_t)stream, g1, g2, (const uint64_t*)nullptr, out, (uint32_t)n, (uint32_t)k, c.scratch, c.stride,
                       c.status);
    HIPCHK(hipGetLastError());
    c.s->last_kernel = 1;
    return BN254_OK;
}
// A group's pairs are contiguous, so the batch IS also n_groups k / C groups of C pairs for any divisor C of k: one launch of the C-pair Miller kernel
// over all those lanes, then a multiplication tree over each group's k / C values (MyFq12 Mul; the odd one out is carried through a level by a
// multiplication by one), then the final exponentiation of n_groups values.  C: wide_chunk's choice (1: every pair its own lane).  The same field element as the shared-f loop over the whole group, hence the same limbs.
template <bool F>
int launch_wide(StreamCtx* sc, const uint64_t* g1, const uint64_t* g2, uint64_t* out, size_t n_groups, size_t k, int device, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const size_t C = wide_chunk(n_groups, k, nullptr);
    const size_t S = k / C, lanes = n_groups * S, half = n_groups * ((S + 1) / 2);
    if (lanes >= ((size_t)1 << 22)) return BN254_ERR_NOT_WIDE;
    int rc;
    if ((rc = ensure(sc, sc->sub[2], 384 * lanes)) || (rc = ensure(sc, sc->sub[0], 384 * half)) || (rc = ensure(sc, sc->sub[1], 384 * half))) return rc;
    uint64_t *V = (uint64_t*)sc->sub[2].p, *A = (uint64_t*)sc->sub[0].p, *B = (uint64_t*)sc->sub[1].p;
    tl_in_wide = true;
    rc = F ? launch_miller_part(g1, g2, V, lanes, C, device, stream) : launch_pairing<true, false>(g1, g2, nullptr, V, lanes, C, device, stream, 0);
    tl_in_wide = false;
    if (rc) return rc;
    for (size_t cur = S; cur > 1;) {
        const size_t h = (cur + 1) / 2, m = n_groups * h;
        size_t blocks = (m * 48 + 255) / 256;
        hipLaunchKernelGGL(k_tree_split, dim3((uint32_t)(blocks > 16384 ? 16384 : blocks)), dim3(256), 0, st, (const uint64_t*)V, A, B, n_groups, cur, h);
        HIPCHK(hipGetLastError());
        if ((rc = launch_op(OP_MUL, A, B, (h == 1 && !F) ? out : V, m, 0, nullptr, 0, device, stream))) return rc;
        cur = h;
    }
    if (F) return launch_pairing<false, true>(nullptr, nullptr, V, out, n_groups, 1, device, stream, 0);
    return BN254_OK;
}
template <bool M, bool F>
int launch_pairing(const uint64_t* g1, const uint64_t* g2, const uint64_t* f_in, uint64_t* out, size_t n_groups, size_t k, int device, void* stream, int io_mode) {
    if (n_groups == 0) return BN254_OK;
    if (io_mode && !direct_elems_ok<M, F>(n_groups, k, device, stream)) return BN254_ERR_INVALID_ARG;      // (callers ask first)
    if (!out || (M && (!g1 || !g2)) || (!M && !f_in) || k == 0 || (!M && k != 1)) return BN254_ERR_INVALID_ARG;
    if (n_groups * k >= (1ull << 29) || n_groups * k / k != n_groups) return BN254_ERR_INVALID_ARG;   // 32-bit byte offsets of the SoA planes in the kernels
    hipStream_t st = (hipStream_t)stream;
    if (M && !io_mode && k <= MAX_K && takes_wide_route(n_groups, k)) {      // 5 .. MAX_K pairs, a handful of groups: spread over the lanes (launch_wide)
        LaunchCtx hold;
        int rc = ctx_get(device, stream, 1, 1, &hold);
        if (rc) return rc;
        rc = launch_wide<F>(hold.s.get(), g1, g2, out, n_groups, k, device, stream);
        if (rc != BN254_ERR_NOT_WIDE) return rc;
    }
    if (k > MAX_K) {
        // multi_miller_loop_native takes any Vec (miller_loop_native.rs:192-282, :324-326); the k-pair kernels hold at most MAX_K
        // pairs' state.  The shared-f Miller value of a group IS the product of the Miller values of any partition of its pairs
        // -- the same field element, hence the same limbs (the reference's own T1, :336-348, asserts it for singletons): walk
        // the group in sub-groups of <= MAX_K pairs, multiply the values (MyFq12 Mul), one final exponentiation at the end.
        LaunchCtx hold;                                 // keeps the context and its (recursive) lock for the whole walk
        int rc = ctx_get(device, stream, 1, (n_groups + BLOCK - 1) / BLOCK, &hold);
        if (rc) return rc;
        StreamCtx* sc = hold.s.get();
        if (takes_wide_route(n_groups, k)) {
            rc = launch_wide<F>(sc, g1, g2, out, n_groups, k, device, stream);
            if (rc != BN254_ERR_NOT_WIDE) return rc;
        }
        if ((rc = ensure(sc, sc->sub[0], 64 * n_groups * MAX_K)) || (rc = ensure(sc, sc->sub[1], 128 * n_groups * MAX_K)) ||
            (rc = ensure(sc, sc->sub[2], 384 * n_groups)) || (rc = ensure(sc, sc->sub[3], 384 * n_groups)))
            return rc;
        uint64_t *s1 = (uint64_t*)sc->sub[0].p, *s2 = (uint64_t*)sc->sub[1].p, *acc = (uint64_t*)sc->sub[2].p, *val = (uint64_t*)sc->sub[3].p;
        size_t n_sub = (k + MAX_K - 1) / MAX_K;
        for (size_t s = 0; s < n_sub; s++) {
            size_t j0 = s * MAX_K, ks = k - j0 < MAX_K ? k - j0 : MAX_K;
            size_t blocks = (n_groups * ks * 16 + 255) / 256;
            if (blocks > 16384) blocks = 16384;
            hipLaunchKernelGGL(k_subgroup, dim3((uint32_t)blocks), dim3(256), 0, st, g1, s1, n_groups, k, j0, ks, 8);
            hipLaunchKernelGGL(k_subgroup, dim3((uint32_t)blocks), dim3(256), 0, st, g2, s2, n_groups, k, j0, ks, 16);
            HIPCHK(hipGetLastError());
            bool last = s + 1 == n_sub;
            uint64_t* dst = s == 0 ? acc : val;
            tl_in_wide = true;                          // (the walk's sub-group launches never spread: sub[] is in use here)
            rc = F ? launch_miller_part(s1, s2, dst, n_groups, ks, device, stream) : launch_pairing<true, false>(s1, s2, nullptr, dst, n_groups, ks, device, stream);
            tl_in_wide = false;
            if (rc) return rc;
            // acc <- acc * val; the last product lands in `out` when no final exponentiation follows
            if (s > 0 && (rc = launch_op(OP_MUL, acc, val, (last && !F) ? out : acc, n_groups, 0, nullptr, 0, device, stream))) return rc;
        }
        if (F) return launch_pairing<false, true>(nullptr, nullptr, acc, out, n_groups, 1, device, stream);
        return BN254_OK;
    }
    if (!io_mode) {
        int prog = cvm_program<M, F>(k);
        if (prog >= 0 && n_groups * k < (1u << 22)) {
            size_t thr; int lanes;
            latency_cfg(device, stream, &thr, &lanes);
            // n_groups * 1000 <= thr * per_mille without the overflow of a huge threshold ("always": SIZE_MAX)
            unsigned __int128 lhs = (unsigned __int128)n_groups * 1000u, rhs = (unsigned __int128)thr * CVM_PROGRAMS[prog].per_mille;
            if (lhs <= rhs) {
                // pairing(), mid-size batch: SEVEN launches -- the Miller loop without the line scale (117 slots per item: eight waves per
                // CU, two per SIMD, one wave's operand fetch under the other's arithmetic), then final_exp_native on its values in six
                // pieces (launch_fexp_pieces) -- while the launch has more waves than one per SIMD (below that nothing overlaps).
                // (the k-pair products likewise: their Miller halves keep four waves per CU -- 155 .. 231 slots -- but the final exponentiation's pieces
                // run with eight)
                if (M && F && k <= 4 && lanes == 0 && n_groups > cvm_split_min(n_cu_of(device))) {
                    LaunchCtx hold;
                    int rc = ctx_get(device, stream, 1, 1, &hold);
                    if (rc) return rc;
                    if ((rc = ensure(hold.s.get(), hold.s->mid, 384 * n_groups))) return rc;
                    uint64_t* mid = (uint64_t*)hold.s->mid.p;
                    if ((rc = launch_cvm(k == 1 ? CVM_MILLER_U : CVM_MMILLER_U + (int)k, 16, g1, g2, nullptr, mid, n_groups, k, device, stream))) return rc;
                    return launch_fexp_pieces(mid, out, n_groups, device, stream);
                }
                if (!M && F && lanes == 0 && n_groups > cvm_split_min(n_cu_of(device)))          // final_exp_native alone, mid-size batch: the same six launches
                    return launch_fexp_pieces(f_in, out, n_groups, device, stream);
                return launch_cvm(prog, lanes, g1, g2, f_in, out, n_groups, k, device, stream);
            }
        }
    }
    LaunchCtx c;
    int rc = ctx_get(device, stream, k, (n_groups + BLOCK - 1) / BLOCK, &c);
    if (rc) return rc;
    auto go = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3(c.grid), dim3(BLOCK), LDS_BYTES, st, g1, g2, f_in, out, (uint32_t)n_groups, (uint32_t)k | ((uint32_t)io_mode << 28),
                           c.scratch, c.stride, c.status);
    };
    if (k == 1) {
        if (M && F) go(k_pairing);
        else if (M) go(k_miller);
        else go(k_fexp);
    } else {
        if (F) go(k_mpairing);
        else go(k_mmiller);
    }
    HIPCHK(hipGetLastError());
    c.s->last_kernel = 1;
    return BN254_OK;
}

int launch_op(int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, size_t power, const int8_t* naf_host, int naf_len,
              int device, void* stream) {
    if (n == 0) return BN254_OK;
    if (!a || !out || (op == OP_MUL && !b) || n >= (1ull << 29)) return BN254_ERR_INVALID_ARG;
    LaunchCtx c;
    int rc = ctx_get(device, stream, 1, (n + BLOCK - 1) / BLOCK, &c);
    if (rc) return rc;
    const int8_t* naf_dev = nullptr;
    uint32_t kk = op == OP_MUL ? 0u : (1u | ((uint32_t)(power % 12) << 8));
    if (op == OP_POW) {
        while (naf_len > 0 && naf_host[naf_len - 1] == 0) naf_len--;        // the top digit of a NAF is +1
        if (naf_len < 1 || naf_len >= 65536) return BN254_ERR_INVALID_ARG;   // 16-bit length field of the kernel's k argument
        if ((rc = ensure(c.s.get(), c.s->naf, (size_t)naf_len + 64))) return rc;
        // naf_host is a caller temporary and the call must not wait for the stream: the digits go through a pinned staging
        // slot, which is reused once the copy that read it has completed (its event).  The ring takes a new slot while all
        // are in flight (bn254_reserve creates four): only with NAF_RING_MAX pow calls queued on one stream does a call wait.
        // The device-side buffer is shared: copies and kernels are ordered by the stream.
        NafSlot* free_slot = nullptr;
        for (NafSlot& ns : c.s->naf_ring)
            if (ns.done && (!ns.used || hipEventQuery(ns.done) == hipSuccess)) { free_slot = &ns; break; }
        (void)hipGetLastError();                                                   // hipErrorNotReady of the queries
        if (!free_slot && c.s->naf_ring.size() < NAF_RING_MAX) {
            c.s->naf_ring.emplace_back();
            free_slot = &c.s->naf_ring.back();
            if (hipEventCreateWithFlags(&free_slot->done, hipEventDisableTiming) != hipSuccess) { c.s->naf_ring.pop_back(); return BN254_ERR_HIP; }
        }
        if (!free_slot) { free_slot = &c.s->naf_ring[0]; HIPCHK(hipEventSynchronize(free_slot->done)); }
        NafSlot& slot = *free_slot;
        if (slot.bytes < (size_t)naf_len) {
            if (slot.host) HIPCHK(hipHostFree(slot.host));
            slot.host = nullptr; slot.bytes = 0;
            size_t cap = ((size_t)naf_len + 4095) & ~(size_t)4095;
            if (hipHostMalloc((void**)&slot.host, cap, hipHostMallocDefault) != hipSuccess) return BN254_ERR_ALLOC;
            slot.bytes = cap;
        }
        memcpy(slot.host, naf_host, (size_t)naf_len);
        HIPCHK(hipMemcpyAsync(c.s->naf.p, slot.host, (size_t)naf_len, hipMemcpyHostToDevice, (hipStream_t)stream));
        HIPCHK(hipEventRecord(slot.done, (hipStream_t)stream));
        slot.used = true;
        naf_dev = (const int8_t*)c.s->naf.p;
        kk = 2u | ((uint32_t)naf_len << 16);
        for (int t = 0; t < naf_len; t++) if (naf_host[t] < 0) { kk |= 1u << 8; break; }   // 1/a is needed
    }
    hipLaunchKernelGGL(k_op, dim3(c.grid), dim3(BLOCK), LDS_BYTES, (hipStream_t)stream, b, (const uint64_t*)naf_dev, a, out, (uint32_t)n, kk,
                       c.scratch, c.stride, c.status);
    HIPCHK(hipGetLastError());
    return BN254_OK;
}

int launch_layout(bool to_soa, const uint64_t* src, uint64_t* dst, size_t words, size_t n, int order, int device, void* stream) {
    if (n == 0) return BN254_OK;
    if (!src || !dst || src == dst || (words != 8 && words != 16 && words != 48) || (order != BN254_FQ12_MYFQ12 && order != BN254_FQ12_ARK) ||
        n >= (1ull << 29))
        return BN254_ERR_INVALID_ARG;
    int rc = check_device(device);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    auto go = [&](auto kern, size_t tile) { hipLaunchKernelGGL(kern, dim3((uint32_t)((n + tile - 1) / tile)), dim3(256), 0, st, src, dst, n, order); };
    if (words == 8) { if (to_soa) go(k_layout<8, 256, true>, 256); else go(k_layout<8, 256, false>, 256); }
    else if (words == 16) { if (to_soa) go(k_layout<16, 128, true>, 128); else go(k_layout<16, 128, false>, 128); }
    else { if (to_soa) go(k_layout<48, 32, true>, 32); else go(k_layout<48, 32, false>, 32); }      // measured: T = 16 / 32 / 64 / 128 -> 5.1 / 5.6 / 5.1 / 3.4 TB/s
    HIPCHK(hipGetLastError());
    return BN254_OK;
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

// host-pointer entry points: inputs are staged through per-stream device buffers that are kept (and grown) across calls.
// A Stage owns the stream context from the first staged byte to the end of the call (the read-back in finish_host): a second
// host thread on the same (device, stream) waits instead of overwriting staged inputs or freeing a buffer under the kernel.
struct Stage {
    std::shared_ptr<StreamCtx> sc;
    std::unique_lock<std::recursive_mutex> lock;
    hipStream_t st;
    int used = 0;
    int init(int device, void* stream) {
        int rc = check_device(device);
        if (rc) return rc;
        sc = stream_ctx(device, stream);
        lock = std::unique_lock<std::recursive_mutex>(sc->mu);
        st = (hipStream_t)stream;
        return BN254_OK;
    }
    int up(const void* h, size_t bytes, uint64_t** d) {
        if (used >= 8) return BN254_ERR_INVALID_ARG;
        Buf& b = sc->stage[used++];
        int rc = ensure(sc.get(), b, bytes ? bytes : 8);
        if (rc) return rc;
        *d = (uint64_t*)b.p;
        if (h && bytes && hipMemcpyAsync(b.p, h, bytes, hipMemcpyHostToDevice, st) != hipSuccess) return BN254_ERR_HIP;
        return BN254_OK;
    }
};

}  // namespace

// ------------------------------------------------------------------ extern "C"
extern "C" {

constexpr size_t PIPE_CHUNK = 65536;          // lanes per chunk of the host-pointer pipeline (one full grid: one work item per CU)
struct HostFmt {               // how the caller's host arrays are laid out
    bool elems = false;        // element-major (one G1 / G2 / Fq12 after the other) instead of limb-major planes
    int out_order = BN254_FQ12_MYFQ12;
};
struct FixedJob {              // a fixed-G2 batch through the pipeline: units of 1 + k_fixed G1 points and ONE G2 point, the fixed points in host memory
    const uint64_t* g2_fixed = nullptr;   // (limb-major planes of k_fixed points, or element-major with fmt.elems)
    size_t k_fixed = 0;
    uint8_t* verdict = nullptr;           // != null: one byte per unit (product == target) instead of the Fq12
    const uint64_t* target = nullptr;     // 48 host words or null = MyFq12::one
    const uint64_t* table = nullptr;      // (device: filled in by run_pipeline, with the event recorded behind its making)
    hipEvent_t table_ready = nullptr;
};
static int run_pipeline(const int* devices, int n_dev, const uint64_t* g1, const uint64_t* g2, uint64_t* out, size_t n_units, size_t k,
                        int do_final_exp, HostFmt fmt = HostFmt(), const FixedJob* fixed = nullptr);
static bool host_pinned(const void* p, size_t bytes);

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
        case BN254_ERR_INFINITY: return "a point at infinity in the batch (outside the reference's contract: it reads raw x / y)";
        case BN254_ERR_NOT_ON_CURVE: return "a point of the batch is not on its curve, or a coordinate is not below p (ark's G1Affine::new / G2Affine::new panic here)";
        case BN254_ERR_NOT_IN_SUBGROUP: return "a G2 point of the batch is not in the r-torsion (ark's G2Affine::new panics here: miller_loop_native.rs:303,311)";
        default: return "unknown status";
    }
}

int bn254_last_status(int device, void* stream) {
    int rc = check_device(device);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    std::shared_ptr<StreamCtx> sc;
    {
        DeviceCtx& c = g_ctx[device];
        std::lock_guard<std::mutex> lk(c.mu);
        auto it = c.streams.find(st);
        if (it != c.streams.end()) sc = it->second;
    }
    if (!sc) { HIPCHK(hipStreamSynchronize(st)); return BN254_OK; }
    std::lock_guard<std::recursive_mutex> lk(sc->mu);
    if (!sc->status) { HIPCHK(hipStreamSynchronize(st)); free_retired(sc.get()); return BN254_OK; }
    // the read-back (and the clearing store) are enqueued on the caller's stream: other streams are not touched
    if (!sc->status_host && hipHostMalloc((void**)&sc->status_host, 2 * sizeof(int), hipHostMallocDefault) != hipSuccess) return BN254_ERR_ALLOC;
    HIPCHK(hipMemcpyAsync(sc->status_host, sc->status, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    free_retired(sc.get());
    if (sc->status_host[0] | sc->status_host[1]) {
        int zero_div = sc->status_host[0], pts = sc->status_host[1];
        HIPCHK(hipMemsetAsync(sc->status, 0, 2 * sizeof(int), st));
        HIPCHK(hipStreamSynchronize(st));
        // the most specific verdict first: an invalid input makes everything computed from it meaningless (an all-zero point also
        // trips the zero-divisor flag of the pairing kernels behind the check)
        if (pts & bn254_chk::PT_INFINITY) return BN254_ERR_INFINITY;
        if (pts & bn254_chk::PT_NOT_ON_CURVE) return BN254_ERR_NOT_ON_CURVE;
        if (pts & bn254_chk::PT_NOT_IN_SUBGROUP) return BN254_ERR_NOT_IN_SUBGROUP;
        return zero_div ? BN254_ERR_ZERO_DIVISOR : BN254_ERR_HIP;
    }
    return BN254_OK;
}

static int finish_host(void* h_out, const void* d_out, size_t bytes, int device, void* stream) {
    if (hipMemcpyAsync(h_out, d_out, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return BN254_ERR_HIP;
    return bn254_last_status(device, stream);
}

size_t bn254_scratch_bytes(size_t n, size_t k) {
    size_t items = (n + BLOCK - 1) / BLOCK, cus = 256;              // MI355X: 256 CUs unless a device this library already uses says otherwise
    for (DeviceCtx& c : g_ctx) {                                     // informational: never initialises the HIP runtime itself
        std::lock_guard<std::mutex> lk(c.mu);
        if (c.init && c.n_cu > 0) { cus = (size_t)c.n_cu; break; }
    }
    size_t grid = items < cus ? (items ? items : 1) : cus;           // persistent kernels: the grid never exceeds the CU count
    size_t kk = k > MAX_K ? MAX_K : k;                                // larger groups are walked in sub-groups of MAX_K pairs
    return BN254_SCRATCH_WG_CONTIGUOUS ? scratch_pitch(kk, grid) * grid : scratch_pitch(kk, grid) * scratch_slots(kk);
}

void bn254_set_latency_lanes(int lanes) { g_latency_lanes.store(lanes == 16 || lanes == 32 || lanes == 64 ? lanes : 0); }
int bn254_get_latency_lanes(void) { return g_latency_lanes.load(); }
void bn254_set_latency_threshold(size_t n) { g_latency_threshold.store(n); }
int bn254_last_kernel(int device, void* stream) {
    if (device < 0 || device >= 64) return BN254_ERR_INVALID_ARG;
    DeviceCtx& c = g_ctx[device];
    std::lock_guard<std::mutex> lk(c.mu);
    auto it = c.streams.find((hipStream_t)stream);
    return it == c.streams.end() ? 0 : it->second->last_kernel.load();
}
int bn254_set_stream_latency(int device, void* stream, size_t threshold, int lanes) {
    if (device < 0 || device >= 64 || !(lanes == -1 || lanes == 0 || lanes == 16 || lanes == 32 || lanes == 64)) return BN254_ERR_INVALID_ARG;
    std::shared_ptr<StreamCtx> sc = stream_ctx(device, stream);      // (no HIP call: the setting may precede the first launch)
    DeviceCtx& c = g_ctx[device];
    std::lock_guard<std::mutex> lk(c.mu);
    sc->lat_threshold.store(threshold);
    sc->lat_lanes.store(lanes);
    return BN254_OK;
}
size_t bn254_get_latency_threshold(void) { return g_latency_threshold.load(); }

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
        int rc = check_device(device);
        if (rc) return rc;
        if (out != a) HIPCHK(hipMemcpyAsync(out, a, 48 * n * sizeof(uint64_t), hipMemcpyDeviceToDevice, (hipStream_t)stream));
        return BN254_OK;
    }
    return launch_op(OP_POW, a, nullptr, out, n, 0, naf.data(), (int)len, device, stream);
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
const int8_t* bn254_six_u_plus_2_naf(void) { return BN254_SIX_U_PLUS_2_NAF; }
uint64_t bn254_bn_x(void) { return BN254_BN_X; }
int bn254_myfq12_to_ark_index(int j) {
    if (j < 0 || j > 11) return -1;
    int h = j / 6, k = (j % 6) / 2, e = j % 2;   // ark flat index j = (h*3 + k)*2 + e
    return (2 * k + h) + 6 * e;
}

int bn254_generate_pairs_dev(uint64_t seed, uint64_t* g1_out, uint64_t* g2_out, size_t n, int device, void* stream) {
    if (n == 0) return BN254_OK;
    if (!g1_out || !g2_out || n >= (1ull << 29)) return BN254_ERR_INVALID_ARG;
    LaunchCtx c;
    int rc = ctx_get(device, stream, 1, (n + BLOCK - 1) / BLOCK, &c, true);
    if (rc) return rc;
    hipLaunchKernelGGL(k_generate, dim3(c.grid), dim3(BLOCK), LDS_BYTES, (hipStream_t)stream, (const uint64_t*)g1_out, (const uint64_t*)g2_out,
                       (const uint64_t*)c.gen_table, (uint64_t*)(uintptr_t)seed, (uint32_t)n, 1u, c.scratch, c.stride, c.status);
    HIPCHK(hipGetLastError());
    return BN254_OK;
}

int bn254_check_points_dev(const uint64_t* g1, const uint64_t* g2, size_t n, int device, void* stream) {
    if (n == 0) return BN254_OK;
    if (!g1 || !g2 || n >= (1ull << 29)) return BN254_ERR_INVALID_ARG;
    LaunchCtx c;
    int rc = ctx_get(device, stream, 1, 1, &c);
    if (rc) return rc;
    size_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_check_points, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)stream, g1, g2, n, c.status + 1);
    HIPCHK(hipGetLastError());
    return BN254_OK;
}
// The subgroup criterion runs on the generated kernel k_subcheck (one G2 point per lane, 0.50 M instructions: 62 doublings + 24 mixed and 2 general
// additions in Jacobian coordinates on the 29-bit limbs); the plain HIP C++ kernel of bn254_point_checks.h does everything else (infinity, on the
// curve) and decides which points are ELIGIBLE for the subgroup verdict (finite and on the twist -- for anything else the criterion's value means
// nothing); a third, tiny kernel merges the two.  BN254_CHECK_SUBGROUP_PORTABLE keeps the whole check on the C++ kernel: the cross-check of the
// generated one (tests/test_point_checks.py).
__global__ void __launch_bounds__(256) k_subcheck_merge(const uint32_t* __restrict__ fails, uint8_t* __restrict__ flags_io, uint8_t* __restrict__ per_point,
                                                        size_t n, int* __restrict__ status) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        int bad = flags_io[i];
        if (!(bad & bn254_chk::PT_SKIP_SUBGROUP) && fails[i]) { bad |= bn254_chk::PT_NOT_IN_SUBGROUP; atomicOr(status, bn254_chk::PT_NOT_IN_SUBGROUP); }
        if (per_point) per_point[i] = (uint8_t)(bad & ~bn254_chk::PT_SKIP_SUBGROUP);
    }
}

int bn254_check_points_ex_dev(const uint64_t* g1, const uint64_t* g2, size_t n, int flags, uint8_t* per_point, int device, void* stream) {
    if (n == 0) return BN254_OK;
    if (!g1 || !g2 || n >= (1ull << 29) || (flags & ~15) || !(flags & 15)) return BN254_ERR_INVALID_ARG;
    const bool portable = (flags & BN254_CHECK_SUBGROUP_PORTABLE) != 0;
    if (portable) flags = (flags & 7) | BN254_CHECK_SUBGROUP;
    const bool generated = (flags & BN254_CHECK_SUBGROUP) && !portable;
    LaunchCtx c;
    int rc = ctx_get(device, stream, 1, generated ? (n + BLOCK - 1) / BLOCK : 1, &c);
    if (rc) return rc;
    static const bn254_chk::Consts K = BN254_CHK_CONSTS;
    size_t blocks = (n + 63) / 64, cap = (size_t)c.n_cu * 32;       // one point per thread; wave-sized workgroups spread small batches over the CUs
    if (blocks > cap) blocks = cap;
    hipStream_t st = (hipStream_t)stream;
    if (!generated) {
        hipLaunchKernelGGL(bn254_chk::k_check_points_ex, dim3((uint32_t)blocks), dim3(64), 0, st, g1, g2, n, flags, K, per_point, c.status + 1);
        HIPCHK(hipGetLastError());
        return BN254_OK;
    }
    // the stream's verdict buffer (bn254_reserve sizes it: 384 n bytes) takes the verdict words (4 n) and the eligibility / flag bytes (n)
    if ((rc = ensure(c.s.get(), c.s->tmp, 8 * n))) return rc;
    uint32_t* fails = (uint32_t*)c.s->tmp.p;
    uint8_t* fl = (uint8_t*)c.s->tmp.p + 4 * n;
    hipLaunchKernelGGL(bn254_chk::k_check_points_ex, dim3((uint32_t)blocks), dim3(64), 0, st, g1, g2, n, flags | bn254_chk::CHECK_DEFER_SUBGROUP, K, fl, c.status + 1);
    hipLaunchKernelGGL(k_subcheck, dim3(c.grid), dim3(BLOCK), LDS_BYTES, st, g1, g2, (const uint64_t*)nullptr, (uint64_t*)fails, (uint32_t)n, 1u, c.scratch,
                       c.stride, c.status);
    size_t mb = (n + 255) / 256;
    if (mb > 4096) mb = 4096;
    hipLaunchKernelGGL(k_subcheck_merge, dim3((uint32_t)mb), dim3(256), 0, st, (const uint32_t*)fails, fl, per_point, n, c.status + 1);
    HIPCHK(hipGetLastError());
    return BN254_OK;
}
int bn254_check_points_ex(const uint64_t* g1, const uint64_t* g2, size_t n, int flags, uint8_t* per_point, int device, void* stream) {
    if (n == 0) return BN254_OK;
    if (!g1 || !g2) return BN254_ERR_INVALID_ARG;
    Stage s; uint64_t *d1, *d2, *d3 = nullptr; int rc;
    if ((rc = s.init(device, stream)) || (rc = s.up(g1, 64 * n, &d1)) || (rc = s.up(g2, 128 * n, &d2)) || (per_point && (rc = s.up(nullptr, n, &d3)))) return rc;
    if ((rc = bn254_check_points_ex_dev(d1, d2, n, flags, (uint8_t*)d3, device, stream))) return rc;
    if (per_point && hipMemcpyAsync(per_point, d3, n, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return BN254_ERR_HIP;
    return bn254_last_status(device, stream);
}
int bn254_check_points(const uint64_t* g1, const uint64_t* g2, size_t n, int device, void* stream) {
    if (n == 0) return BN254_OK;
    if (!g1 || !g2) return BN254_ERR_INVALID_ARG;
    Stage s; uint64_t *d1, *d2; int rc;
    if ((rc = s.init(device, stream)) || (rc = s.up(g1, 64 * n, &d1)) || (rc = s.up(g2, 128 * n, &d2))) return rc;
    if ((rc = bn254_check_points_dev(d1, d2, n, device, stream))) return rc;
    return bn254_last_status(device, stream);
}

int bn254_multi_pairing_check_target_batch_dev(const uint64_t* g1, const uint64_t* g2, const uint64_t* target, uint8_t* verdict, size_t n_groups, size_t k,
                                               int device, void* stream) {
    if (n_groups == 0) return BN254_OK;
    if (!g1 || !g2 || !verdict || k == 0) return BN254_ERR_INVALID_ARG;
    LaunchCtx c;
    int rc = ctx_get(device, stream, k > MAX_K ? 1 : k, (n_groups + BLOCK - 1) / BLOCK, &c);
    if (rc) return rc;
    if ((rc = ensure(c.s.get(), c.s->tmp, 384 * n_groups))) return rc;
    if ((rc = launch_pairing<true, true>(g1, g2, nullptr, (uint64_t*)c.s->tmp.p, n_groups, k, device, stream))) return rc;
    return launch_is_equal((const uint64_t*)c.s->tmp.p, target, verdict, n_groups, stream);
}
int bn254_multi_pairing_check_batch_dev(const uint64_t* g1, const uint64_t* g2, uint8_t* verdict, size_t n_groups, size_t k, int device,
                                        void* stream) {
    return bn254_multi_pairing_check_target_batch_dev(g1, g2, nullptr, verdict, n_groups, k, device, stream);
}

// The G2 points of n groups of 1 + kf pairs whose last kf points are the table's fixed ones, as limb-major planes of n (1 + kf) points (what the k-pair
// programs read): pair 0 of group g = the group's own point (limb-major planes of n points, or element-major structs), pair j = fixed point j - 1
// (planes of kf points, the tail of the line table).
__global__ void __launch_bounds__(256) k_expand_fixed(const uint64_t* __restrict__ var, const uint64_t* __restrict__ fix, uint64_t* __restrict__ dst, size_t n, size_t kf,
                                                      int var_elems) {
    size_t own = var ? 1 : 0, k = kf + own, per = n * k;          // (var == null: groups without a point of their own)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < per * 16; i += (size_t)gridDim.x * blockDim.x) {
        size_t w = i / per, r = i - w * per, g = r / k, j = r - g * k;
        dst[i] = (j >= own) ? fix[w * kf + (j - own)] : (var_elems ? var[g * 16 + w] : var[w * n + g]);
    }
}
// ---- fixed G2 points (a Groth16 verifier's beta, gamma, delta: the same for every proof).  bn254_g2_lines_dev walks the point steps of each
// fixed point ONCE and leaves every step's line coefficients in a table; bn254_pairing_fixed_g2_batch_dev then computes, per group,
// final_exp_native(multi_miller_loop_native([(P0, Q0), (P1, Qfix_1), ..., (Pk, Qfix_k)])) with the fixed pairs reduced to one line scaling and one
// sparse multiplication per step (no point step, no per-lane point state): miller_loop_native.rs:192-282 with k of the b's shared by the batch.
static size_t g2_lines_only_bytes(size_t k_fixed) { return k_fixed * (size_t)BN254_FIXED_LINES * BN254_FIXED_LINE_SLOTS * SLOT_BYTES; }
// the lines, then the points themselves (limb-major planes of k_fixed points): small batches are served by the lane-cooperative k-pair programs on the expanded pairs
size_t bn254_g2_lines_bytes(size_t k_fixed) { return g2_lines_only_bytes(k_fixed) + 128 * k_fixed; }

int bn254_g2_lines_dev(const uint64_t* g2_fixed, size_t k_fixed, uint64_t* table, int device, void* stream) {
    if (!g2_fixed || !table || k_fixed == 0 || k_fixed > (size_t)BN254_FIXED_MAX) return BN254_ERR_INVALID_ARG;
    LaunchCtx c;
    int rc = ctx_get(device, stream, 1, 1, &c);
    if (rc) return rc;
    hipLaunchKernelGGL(k_g2lines, dim3(1), dim3(BLOCK), LDS_BYTES, (hipStream_t)stream, (const uint64_t*)nullptr, g2_fixed, (const uint64_t*)nullptr, table, (uint32_t)k_fixed, 1u,
                       c.scratch, c.stride, c.status);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync((char*)table + g2_lines_only_bytes(k_fixed), g2_fixed, 128 * k_fixed, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return BN254_OK;
}

static int launch_fixed(const uint64_t* g1, const uint64_t* g2, const uint64_t* table, size_t k_fixed, uint64_t* out, size_t n, int io_mode, int device, void* stream) {
    if (n == 0) return BN254_OK;
    // g2 == null: groups WITHOUT a pair of their own (every G2 point is one of the table's: a KZG-style check e(P_1, Qfix_1) e(P_2, Qfix_2)): g1 holds k_fixed points per group
    const size_t own = g2 ? 1 : 0;
    if (!g2) io_mode |= IO_NO_OWN;
    if (!g1 || !table || !out || k_fixed == 0 || k_fixed > (size_t)BN254_FIXED_MAX || n * (k_fixed + 1) > ((size_t)1 << 23)) return BN254_ERR_INVALID_ARG;      // (32-bit lane offsets of the element-major form: 384 n < 4 GB)
    LaunchCtx c;
    int rc = ctx_get(device, stream, 1, (n + BLOCK - 1) / BLOCK, &c);
    if (rc) return rc;
    if (takes_latency_kernel<true, true>(n, k_fixed + own, device, stream)) {
        // a batch this small is a fraction of one grid of the throughput kernel (8 ms whatever n is): expand the pairs and let the lane-cooperative k-pair
        // program take them (a single group of 1 + 3 pairs: 0.74 ms) -- the same value, hence the same limbs (the final exponentiation does not see how the
        // Miller value was reached)
        StreamCtx* sc = c.s.get();
        const size_t k = k_fixed + own, np = n * k;
        const bool in_e = io_mode & IO_IN_ELEMS, out_e = io_mode & IO_OUT_ELEMS;
        if ((rc = ensure(sc, sc->sub[1], 128 * np)) || (in_e && (rc = ensure(sc, sc->sub[0], 64 * np))) || (out_e && (rc = ensure(sc, sc->sub[2], 384 * n)))) return rc;
        uint64_t *p1 = (uint64_t*)sc->sub[0].p, *p2 = (uint64_t*)sc->sub[1].p, *p3 = (uint64_t*)sc->sub[2].p;
        size_t blocks = (np * 16 + 255) / 256;
        hipLaunchKernelGGL(k_expand_fixed, dim3((uint32_t)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, (hipStream_t)stream, g2,
                           (const uint64_t*)((const char*)table + g2_lines_only_bytes(k_fixed)), p2, n, k_fixed, in_e ? 1 : 0);
        HIPCHK(hipGetLastError());
        if (in_e && (rc = launch_layout(true, g1, p1, 8, np, 0, device, stream))) return rc;
        if ((rc = launch_pairing<true, true>(in_e ? p1 : g1, p2, nullptr, out_e ? p3 : out, n, k, device, stream))) return rc;
        return out_e ? launch_layout(false, p3, out, 48, n, (io_mode & IO_OUT_ARK) ? BN254_FQ12_ARK : BN254_FQ12_MYFQ12, device, stream) : BN254_OK;
    }
    hipLaunchKernelGGL(k_fpairing, dim3(c.grid), dim3(BLOCK), LDS_BYTES, (hipStream_t)stream, g1, g2, table, out, (uint32_t)n, (uint32_t)k_fixed | ((uint32_t)io_mode << 28),
                       c.scratch, c.stride, c.status);
    HIPCHK(hipGetLastError());
    c.s->last_kernel = 1;
    return BN254_OK;
}
int bn254_pairing_fixed_g2_batch_dev(const uint64_t* g1, const uint64_t* g2_var, const uint64_t* table, size_t k_fixed, uint64_t* out, size_t n, int device, void* stream) {
    return launch_fixed(g1, g2_var, table, k_fixed, out, n, 0, device, stream);
}
int bn254_pairing_fixed_g2_batch_elems_dev(const uint64_t* g1, const uint64_t* g2_var, const uint64_t* table, size_t k_fixed, uint64_t* out, size_t n, int out_order,
                                           int device, void* stream) {
    if (out_order != BN254_FQ12_MYFQ12 && out_order != BN254_FQ12_ARK) return BN254_ERR_INVALID_ARG;
    return launch_fixed(g1, g2_var, table, k_fixed, out, n, IO_IN_ELEMS | IO_OUT_ELEMS | (out_order == BN254_FQ12_ARK ? IO_OUT_ARK : 0), device, stream);
}
int bn254_pairing_fixed_g2_check_target_batch_dev(const uint64_t* g1, const uint64_t* g2_var, const uint64_t* table, size_t k_fixed, const uint64_t* target,
                                                  uint8_t* verdict, size_t n, int device, void* stream) {
    if (n == 0) return BN254_OK;
    if (!verdict) return BN254_ERR_INVALID_ARG;
    LaunchCtx c;
    int rc = ctx_get(device, stream, 1, (n + BLOCK - 1) / BLOCK, &c);
    if (rc) return rc;
    if ((rc = ensure(c.s.get(), c.s->tmp, 384 * n))) return rc;
    if ((rc = launch_fixed(g1, g2_var, table, k_fixed, (uint64_t*)c.s->tmp.p, n, 0, device, stream))) return rc;
    return launch_is_equal((const uint64_t*)c.s->tmp.p, target, verdict, n, stream);
}
int bn254_pairing_fixed_g2_check_batch_dev(const uint64_t* g1, const uint64_t* g2_var, const uint64_t* table, size_t k_fixed, uint8_t* verdict, size_t n, int device,
                                           void* stream) {
    return bn254_pairing_fixed_g2_check_target_batch_dev(g1, g2_var, table, k_fixed, nullptr, verdict, n, device, stream);
}

// host-pointer forms: stage, make the table (2.1 ms), launch, copy back.  `elems`: every array element-major (the fixed points too), result in out_order.
// the line table of `g2_fixed` (host memory; element-major if `elems`) on this stream: made (2 ms) unless the stream's last host-pointer call had the same points
static int host_table(Stage& s, const uint64_t* g2_fixed, size_t k_fixed, bool elems, int device, void* stream, const uint64_t** table) {
    StreamCtx* sc = s.sc.get();
    const size_t tab_max = bn254_g2_lines_bytes(BN254_FIXED_MAX);
    std::vector<uint64_t> key(g2_fixed, g2_fixed + 16 * k_fixed);
    key.push_back(elems ? 1 : 0);
    int slot = -1, lru = 0;
    for (int i = 0; i < StreamCtx::FIXED_TABLES; i++) {
        if (!sc->fixed_key[i].empty() && sc->fixed_key[i] == key) { slot = i; break; }
        if (sc->fixed_used[i] < sc->fixed_used[lru]) lru = i;
    }
    const bool hit = slot >= 0;
    if (!hit) slot = lru;
    sc->fixed_used[slot] = ++sc->fixed_clock;
    sc->fixed_last = slot;
    int rc = ensure(sc, sc->fixed_tab[slot], tab_max + 2 * 128 * BN254_FIXED_MAX);
    if (rc) return rc;
    uint64_t *tab = (uint64_t*)sc->fixed_tab[slot].p, *in = (uint64_t*)((char*)tab + tab_max), *planes = in + 16 * BN254_FIXED_MAX;
    *table = tab;
    if (hit) return BN254_OK;
    sc->fixed_key[slot].clear();
    if (hipMemcpyAsync(in, g2_fixed, 128 * k_fixed, hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess) return BN254_ERR_HIP;
    if (elems && (rc = launch_layout(true, in, planes, 16, k_fixed, 0, device, stream))) return rc;      // (the table kernel reads limb-major planes)
    if ((rc = bn254_g2_lines_dev(elems ? planes : in, k_fixed, tab, device, stream))) return rc;
    sc->fixed_key[slot] = std::move(key);
    return BN254_OK;
}
// ... which must not outlive a call that raised the status (its making may have been what raised it)
static int finish_fixed_host(Stage& s, void* h_out, const void* d_out, size_t bytes, int device, void* stream) {
    int rc = finish_host(h_out, d_out, bytes, device, stream);
    if (rc && s.sc->fixed_last >= 0) s.sc->fixed_key[s.sc->fixed_last].clear();
    return rc;
}

static int fixed_host(const uint64_t* g1, const uint64_t* g2_var, const uint64_t* g2_fixed, size_t k_fixed, uint64_t* out, size_t n, bool elems, int out_order,
                      int device, void* stream) {
    if (n == 0) return BN254_OK;
    if (!g1 || !g2_fixed || !out || k_fixed == 0 || k_fixed > (size_t)BN254_FIXED_MAX || n * (k_fixed + 1) >= ((size_t)1 << 29) ||
        (out_order != BN254_FQ12_MYFQ12 && out_order != BN254_FQ12_ARK))
        return BN254_ERR_INVALID_ARG;
    const size_t own = g2_var ? 1 : 0;    // (g2_var == null: groups without a pair of their own, k_fixed G1 points each)
    if (n > PIPE_CHUNK) {                 // large batch: chunked, copies overlapped with compute (private streams), the table made once
        int rc0 = check_device(device);
        if (rc0) return rc0;
        HIPCHK(hipStreamSynchronize((hipStream_t)stream));
        HostFmt fmt; fmt.elems = elems; fmt.out_order = out_order;
        FixedJob job; job.g2_fixed = g2_fixed; job.k_fixed = k_fixed;
        return run_pipeline(&device, 1, g1, g2_var, out, n, k_fixed + own, 1, fmt, &job);
    }
    Stage s; uint64_t *d1, *d2 = nullptr, *d3; const uint64_t* dt; int rc;
    if ((rc = s.init(device, stream)) || (rc = s.up(g1, 64 * n * (k_fixed + own), &d1)) || (own && (rc = s.up(g2_var, 128 * n, &d2))) || (rc = s.up(nullptr, 384 * n, &d3)) ||
        (rc = host_table(s, g2_fixed, k_fixed, elems, device, stream, &dt)))
        return rc;
    rc = elems ? bn254_pairing_fixed_g2_batch_elems_dev(d1, d2, dt, k_fixed, d3, n, out_order, device, stream)
               : bn254_pairing_fixed_g2_batch_dev(d1, d2, dt, k_fixed, d3, n, device, stream);
    if (rc) return rc;
    return finish_fixed_host(s, out, d3, 384 * n, device, stream);
}
int bn254_pairing_fixed_g2_batch(const uint64_t* g1, const uint64_t* g2_var, const uint64_t* g2_fixed, size_t k_fixed, uint64_t* out, size_t n, int device, void* stream) {
    return fixed_host(g1, g2_var, g2_fixed, k_fixed, out, n, false, BN254_FQ12_MYFQ12, device, stream);
}
int bn254_pairing_fixed_g2_batch_elems(const uint64_t* g1, const uint64_t* g2_var, const uint64_t* g2_fixed, size_t k_fixed, uint64_t* out, size_t n, int out_order,
                                       int device, void* stream) {
    return fixed_host(g1, g2_var, g2_fixed, k_fixed, out, n, true, out_order, device, stream);
}
// host pointers, element-major structs in, one verdict byte per group out (target: 48 host words or null = one): a Groth16 verifier's whole pairing check
int bn254_pairing_fixed_g2_check_batch_elems(const uint64_t* g1, const uint64_t* g2_var, const uint64_t* g2_fixed, size_t k_fixed, const uint64_t* target,
                                             uint8_t* verdict, size_t n, int device, void* stream) {
    if (n == 0) return BN254_OK;
    if (!g1 || !g2_fixed || !verdict || k_fixed == 0 || k_fixed > (size_t)BN254_FIXED_MAX || n * (k_fixed + 1) >= ((size_t)1 << 29)) return BN254_ERR_INVALID_ARG;
    const size_t own = g2_var ? 1 : 0;
    if (n > PIPE_CHUNK) {
        int rc0 = check_device(device);
        if (rc0) return rc0;
        HIPCHK(hipStreamSynchronize((hipStream_t)stream));
        HostFmt fmt; fmt.elems = true;
        FixedJob job; job.g2_fixed = g2_fixed; job.k_fixed = k_fixed; job.verdict = verdict; job.target = target;
        return run_pipeline(&device, 1, g1, g2_var, nullptr, n, k_fixed + own, 1, fmt, &job);
    }
    Stage s; uint64_t *d1, *d2 = nullptr, *d3, *dv; const uint64_t* dt; int rc;
    if ((rc = s.init(device, stream)) || (rc = s.up(g1, 64 * n * (k_fixed + own), &d1)) || (own && (rc = s.up(g2_var, 128 * n, &d2))) || (rc = s.up(nullptr, 384 * n, &d3)) ||
        (rc = s.up(nullptr, (n + 7) & ~(size_t)7, &dv)) || (rc = host_table(s, g2_fixed, k_fixed, true, device, stream, &dt)))
        return rc;
    if ((rc = launch_fixed(d1, d2, dt, k_fixed, d3, n, IO_IN_ELEMS, device, stream)) || (rc = launch_is_equal(d3, target, (uint8_t*)dv, n, stream))) return rc;
    return finish_fixed_host(s, verdict, dv, n, device, stream);
}

void bn254_set_wide_groups(size_t max_groups) { g_wide_groups.store(max_groups); }
size_t bn254_get_wide_groups(void) { return g_wide_groups.load(); }

int bn254_release_stream(int device, void* stream) {
    int rc = check_device(device);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    std::shared_ptr<StreamCtx> sc;
    {
        DeviceCtx& c = g_ctx[device];
        std::lock_guard<std::mutex> lk(c.mu);
        auto it = c.streams.find((hipStream_t)stream);
        if (it == c.streams.end()) return BN254_OK;
        sc = std::move(it->second);
        c.streams.erase(it);
    }
    // A call that looked the context up before the erase still holds a reference: the context (and its buffers) go when the
    // last holder lets go.  Such a call may have launched after the synchronisation above: wait for that work too.
    { std::lock_guard<std::recursive_mutex> lk(sc->mu); }
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    return BN254_OK;
}

// ---- page-locked host memory for the host-pointer entry points.  The reference's callers hold their G1Affine / G2Affine / Fq12 values in
// host memory (src/pairing.rs:20, miller_loop_native.rs:324); from pageable memory every byte goes through the runtime's staging buffers
// (a host memcpy on the issuing thread: ~5 GB/s), from page-locked memory the copy engines move it at the link rate under the kernels.
int bn254_host_register(void* ptr, size_t bytes) {
    if (!ptr || !bytes) return BN254_ERR_INVALID_ARG;
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) return BN254_ERR_NO_DEVICE;
    hipError_t e = hipHostRegister(ptr, bytes, hipHostRegisterPortable);
    if (e == hipErrorHostMemoryAlreadyRegistered) { (void)hipGetLastError(); return BN254_OK; }
    if (e != hipSuccess) { (void)hipGetLastError(); return e == hipErrorOutOfMemory ? BN254_ERR_ALLOC : BN254_ERR_HIP; }
    return BN254_OK;
}
int bn254_host_unregister(void* ptr) {
    if (!ptr) return BN254_ERR_INVALID_ARG;
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) return BN254_ERR_NO_DEVICE;
    if (hipHostUnregister(ptr) != hipSuccess) { (void)hipGetLastError(); return BN254_ERR_INVALID_ARG; }
    return BN254_OK;
}
int bn254_alloc_pinned(size_t bytes, void** out) {
    if (!out) return BN254_ERR_INVALID_ARG;
    *out = nullptr;
    if (!bytes) return BN254_ERR_INVALID_ARG;
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) return BN254_ERR_NO_DEVICE;
    if (hipHostMalloc(out, bytes, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); *out = nullptr; return BN254_ERR_ALLOC; }
    return BN254_OK;
}
int bn254_free_pinned(void* ptr) {
    if (!ptr) return BN254_OK;
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) return BN254_ERR_NO_DEVICE;
    if (hipHostFree(ptr) != hipSuccess) { (void)hipGetLastError(); return BN254_ERR_INVALID_ARG; }
    return BN254_OK;
}
int bn254_host_is_pinned(const void* ptr, size_t bytes) {
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) return 0;
    return host_pinned(ptr, bytes) ? 1 : 0;
}

#ifdef BN254_DEBUG_STAMPS
// DIAGNOSTIC builds only (tools/exp/build_variant.sh with KGEN_CLOCK_STAMP=1 and -DBN254_DEBUG_STAMPS; not part of the ABI): the
// kernels of such a build leave, per wave, (d s_memtime, d s_memrealtime) around their item loop in the slack at the end of their
// workgroup's scratch block.  out: 4 x u64 per workgroup-wave pair = [wg][wave]{d_memtime, d_memrealtime}; returns the grid size.
int bn254_debug_stamps(int device, void* stream, uint64_t* out, size_t max_wg) {
    int rc = check_device(device);
    if (rc) return rc;
    std::shared_ptr<StreamCtx> sc = stream_ctx(device, stream);
    std::lock_guard<std::recursive_mutex> lk(sc->mu);
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    if (!sc->scratch.p || !sc->last_grid) return BN254_ERR_INVALID_ARG;
    size_t n = sc->last_grid < max_wg ? sc->last_grid : max_wg;
    for (size_t g = 0; g < n; g++)
        HIPCHK(hipMemcpy(out + g * 8, (const char*)sc->scratch.p + (g + 1) * sc->last_pitch - BN254_STAMP_OFFSET_FROM_END, 64, hipMemcpyDeviceToHost));
    return (int)n;
}
#ifdef BN254_PROFILE_IDS
// per-routine cycle / call counters of a KGEN_PROFILE_L2 build: out = [wg][wave][3][64] dwords (cycles lo, cycles hi, calls; lane = routine id)
int bn254_debug_profile(int device, void* stream, uint32_t* out, size_t max_wg) {
    int rc = check_device(device);
    if (rc) return rc;
    std::shared_ptr<StreamCtx> sc = stream_ctx(device, stream);
    std::lock_guard<std::recursive_mutex> lk(sc->mu);
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    if (!sc->scratch.p || !sc->last_grid) return BN254_ERR_INVALID_ARG;
    size_t n = sc->last_grid < max_wg ? sc->last_grid : max_wg;
    for (size_t g = 0; g < n; g++)
        HIPCHK(hipMemcpy(out + g * 768, (const char*)sc->scratch.p + (g + 1) * sc->last_pitch - BN254_PROFILE_OFFSET_FROM_END, 3072, hipMemcpyDeviceToHost));
    return (int)n;
}
const char* bn254_debug_profile_ids(void) { return BN254_PROFILE_IDS; }
#endif
#endif

int bn254_reserve(int device, void* stream, size_t n, size_t k) {
    if (k == 0 || n * k >= (1ull << 29) || (n && n * k / n != k)) return BN254_ERR_INVALID_ARG;
    LaunchCtx c;                                       // scratch for k pairs per lane (sub-groups of MAX_K above that) + the status word
    int rc = ctx_get(device, stream, k > MAX_K ? MAX_K : k, (n + BLOCK - 1) / BLOCK, &c);
    if (rc) return rc;
    StreamCtx* sc = c.s.get();
    {   // "up to k pairs": the scratch of a smaller group may be LARGER (groups of up to BN254_FIS_MAX_K pairs keep a line area)
        size_t need = 0, kmax = k > MAX_K ? MAX_K : k;
        for (size_t kk = 1; kk <= kmax; kk++) {
            size_t pitch = scratch_pitch(kk, c.grid);
            if (pitch >= (1ull << 32)) return BN254_ERR_INVALID_ARG;
            size_t bytes = BN254_SCRATCH_WG_CONTIGUOUS ? pitch * c.grid : pitch * scratch_slots(kk);
            if (bytes > need) need = bytes;
        }
        if ((rc = ensure(sc, sc->scratch, need))) return rc;
    }
    if ((rc = ensure(sc, sc->tmp, 384 * (n ? n : 1)))) return rc;                                   // the `== one` verdict's Fq12 values
    {   // pairing() / final_exp_native in several launches (mid-size batches on the lane-cooperative kernel): the values in between.  Sized for the
        // LARGEST item count that path can take under this stream's (else the process') threshold -- whatever n is: after reserve(2^20) a call of
        // 8 192 items must not allocate either (and a hipGraph capture of it must not meet a hipMalloc).  Nothing when the path is switched off.
        size_t thr = sc->lat_threshold.load() == (size_t)-1 ? g_latency_threshold.load() : sc->lat_threshold.load();
        size_t split_max = cvm_split_max(thr);
        if (split_max > n) split_max = n;
        if (split_max > cvm_split_min(c.n_cu)) {
            if ((rc = ensure(sc, sc->mid, 384 * split_max))) return rc;
            for (Buf& b : sc->fx)
                if ((rc = ensure(sc, b, 384 * split_max))) return rc;
        }
    }
    if (k >= 2 && k <= (size_t)BN254_FIXED_MAX) {      // a small fixed-G2 batch of 1 + (k - 1) pairs expands its pairs for the lane-cooperative k-pair program (launch_fixed)
        size_t thr = sc->lat_threshold.load() == (size_t)-1 ? g_latency_threshold.load() : sc->lat_threshold.load();
        size_t m = cvm_split_max(thr);
        if (m > n) m = n;
        if (m && ((rc = ensure(sc, sc->sub[0], 64 * m * k)) || (rc = ensure(sc, sc->sub[1], 128 * m * k)) || (rc = ensure(sc, sc->sub[2], 384 * m)))) return rc;
    }
    if ((rc = ensure(sc, sc->naf, 65536 + 64))) return rc;                                          // pow_native digits (16-bit length field)
    if (k > MAX_K && ((rc = ensure(sc, sc->sub[0], 64 * n * MAX_K)) || (rc = ensure(sc, sc->sub[1], 128 * n * MAX_K)) ||
                      (rc = ensure(sc, sc->sub[2], 384 * n)) || (rc = ensure(sc, sc->sub[3], 384 * n))))
        return rc;
    if (n && takes_wide_route(n, k)) {      // few groups of many pairs: the chunk values and the two operand buffers of the multiplication tree (launch_wide)
        const size_t C = wide_chunk(n, k, nullptr);
        const size_t S = k / C, lanes = n * S, half = n * ((S + 1) / 2);
        if (lanes < ((size_t)1 << 22)) {
            if ((rc = ensure(sc, sc->sub[2], 384 * lanes)) || (rc = ensure(sc, sc->sub[0], 384 * half)) || (rc = ensure(sc, sc->sub[1], 384 * half))) return rc;
            LaunchCtx cw;                               // ... and the scratch of the Miller launch over all the chunks
            if ((rc = ctx_get(device, stream, C, (lanes + BLOCK - 1) / BLOCK, &cw))) return rc;
        }
    }
    while (sc->naf_ring.size() < 4) {
        NafSlot ns;
        if (hipHostMalloc((void**)&ns.host, 65536, hipHostMallocDefault) != hipSuccess) return BN254_ERR_ALLOC;
        ns.bytes = 65536;
        if (hipEventCreateWithFlags(&ns.done, hipEventDisableTiming) != hipSuccess) { (void)hipHostFree(ns.host); return BN254_ERR_HIP; }
        sc->naf_ring.push_back(ns);
    }
    if (!sc->status_host && hipHostMalloc((void**)&sc->status_host, 2 * sizeof(int), hipHostMallocDefault) != hipSuccess) return BN254_ERR_ALLOC;
    if ((sc->lat_threshold.load() == (size_t)-1 ? g_latency_threshold.load() : sc->lat_threshold.load()) != 0)                    // the latency path's round programs (28 MB in all): small calls upload nothing later
        for (int prog = 0; prog < CVM_N_PROGRAMS; prog++)
            if ((rc = cvm_upload(device, prog))) return rc;
    return BN254_OK;
}

// ---- host-pointer variants: stage through the device, synchronise, report the status word.  The scalar Rust / C++
// signatures of the reference land here with n = 1: ONE lane of one launch (drop-in correctness, not a fast path).
int bn254_multi_pairing_check_batch(const uint64_t* g1, const uint64_t* g2, uint8_t* verdict, size_t n_groups, size_t k, int device, void* stream) {
    if (n_groups == 0) return BN254_OK;
    if (!g1 || !g2 || !verdict || k == 0) return BN254_ERR_INVALID_ARG;
    Stage s; uint64_t *d1, *d2, *d3; int rc; size_t np = n_groups * k;
    if ((rc = s.init(device, stream)) || (rc = s.up(g1, 64 * np, &d1)) || (rc = s.up(g2, 128 * np, &d2)) || (rc = s.up(nullptr, n_groups, &d3))) return rc;
    if ((rc = bn254_multi_pairing_check_batch_dev(d1, d2, (uint8_t*)d3, n_groups, k, device, stream))) return rc;
    return finish_host(verdict, d3, n_groups, device, stream);
}

int bn254_pairing_batch(const uint64_t* g1, const uint64_t* g2, uint64_t* out, size_t n, int device, void* stream) {
    if (n == 0) return BN254_OK;
    if (!g1 || !g2 || !out) return BN254_ERR_INVALID_ARG;
    int rc = check_device(device);
    if (rc) return rc;
    if (n > PIPE_CHUNK) {                 // large batch: chunked, copies overlapped with compute (private streams)
        if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return BN254_ERR_HIP;
        return run_pipeline(&device, 1, g1, g2, out, n, 1, 1);
    }
    Stage s; uint64_t *d1, *d2, *d3;
    if ((rc = s.init(device, stream)) || (rc = s.up(g1, 64 * n, &d1)) || (rc = s.up(g2, 128 * n, &d2)) || (rc = s.up(nullptr, 384 * n, &d3))) return rc;
    if ((rc = bn254_pairing_batch_dev(d1, d2, d3, n, device, stream))) return rc;
    return finish_host(out, d3, 384 * n, device, stream);
}
int bn254_miller_loop_batch(const uint64_t* g1, const uint64_t* g2, uint64_t* f_out, size_t n, int device, void* stream) {
    if (n == 0) return BN254_OK;
    if (!g1 || !g2 || !f_out) return BN254_ERR_INVALID_ARG;
    Stage s; uint64_t *d1, *d2, *d3; int rc;
    if ((rc = s.init(device, stream)) || (rc = s.up(g1, 64 * n, &d1)) || (rc = s.up(g2, 128 * n, &d2)) || (rc = s.up(nullptr, 384 * n, &d3))) return rc;
    if ((rc = bn254_miller_loop_batch_dev(d1, d2, d3, n, device, stream))) return rc;
    return finish_host(f_out, d3, 384 * n, device, stream);
}
int bn254_final_exp_batch(const uint64_t* f_in, uint64_t* out, size_t n, int device, void* stream) {
    if (n == 0) return BN254_OK;
    if (!f_in || !out) return BN254_ERR_INVALID_ARG;
    Stage s; uint64_t *d1, *d3; int rc;
    if ((rc = s.init(device, stream)) || (rc = s.up(f_in, 384 * n, &d1)) || (rc = s.up(nullptr, 384 * n, &d3))) return rc;
    if ((rc = bn254_final_exp_batch_dev(d1, d3, n, device, stream))) return rc;
    return finish_host(out, d3, 384 * n, device, stream);
}
int bn254_multi_pairing_batch(const uint64_t* g1, const uint64_t* g2, uint64_t* out, size_t n_groups, size_t k, int do_final_exp,
                              int device, void* stream) {
    if (n_groups == 0) return BN254_OK;
    if (!g1 || !g2 || !out || k == 0) return BN254_ERR_INVALID_ARG;
    int rc = check_device(device);
    if (rc) return rc;
    if (n_groups > PIPE_CHUNK) {
        if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return BN254_ERR_HIP;
        return run_pipeline(&device, 1, g1, g2, out, n_groups, k, do_final_exp);
    }
    Stage s; uint64_t *d1, *d2, *d3; size_t np = n_groups * k;
    if ((rc = s.init(device, stream)) || (rc = s.up(g1, 64 * np, &d1)) || (rc = s.up(g2, 128 * np, &d2)) || (rc = s.up(nullptr, 384 * n_groups, &d3))) return rc;
    if ((rc = bn254_multi_pairing_batch_dev(d1, d2, d3, n_groups, k, do_final_exp, device, stream))) return rc;
    return finish_host(out, d3, 384 * n_groups, device, stream);
}
int bn254_fq12_mul_batch(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, int device, void* stream) {
    if (n == 0) return BN254_OK;
    if (!a || !b || !out) return BN254_ERR_INVALID_ARG;
    Stage s; uint64_t *d1, *d2, *d3; int rc;
    if ((rc = s.init(device, stream)) || (rc = s.up(a, 384 * n, &d1)) || (rc = s.up(b, 384 * n, &d2)) || (rc = s.up(nullptr, 384 * n, &d3))) return rc;
    if ((rc = bn254_fq12_mul_batch_dev(d1, d2, d3, n, device, stream))) return rc;
    return finish_host(out, d3, 384 * n, device, stream);
}
int bn254_frobenius_map_batch(const uint64_t* a, size_t power, uint64_t* out, size_t n, int device, void* stream) {
    if (n == 0) return BN254_OK;
    if (!a || !out) return BN254_ERR_INVALID_ARG;
    Stage s; uint64_t *d1, *d3; int rc;
    if ((rc = s.init(device, stream)) || (rc = s.up(a, 384 * n, &d1)) || (rc = s.up(nullptr, 384 * n, &d3))) return rc;
    if ((rc = bn254_frobenius_map_batch_dev(d1, power, d3, n, device, stream))) return rc;
    return finish_host(out, d3, 384 * n, device, stream);
}
int bn254_pow_batch(const uint64_t* a, const uint64_t* exp, size_t exp_limbs, uint64_t* out, size_t n, int device, void* stream) {
    if (n == 0) return BN254_OK;
    if (!a || (!exp && exp_limbs) || !out) return BN254_ERR_INVALID_ARG;
    Stage s; uint64_t *d1, *d3; int rc;
    if ((rc = s.init(device, stream)) || (rc = s.up(a, 384 * n, &d1)) || (rc = s.up(nullptr, 384 * n, &d3))) return rc;
    if ((rc = bn254_pow_batch_dev(d1, exp, exp_limbs, d3, n, device, stream))) return rc;
    return finish_host(out, d3, 384 * n, device, stream);
}

// ---- element-major data ("elems"): the order the reference's callers hold their values in (&[G1Affine], Vec<MyFq12>, Vec<Fq12>)
int bn254_soa_from_elems_dev(const uint64_t* elems, uint64_t* soa, size_t words, size_t n, int fq12_order, int device, void* stream) {
    return launch_layout(true, elems, soa, words, n, words == 48 ? fq12_order : 0, device, stream);
}
int bn254_soa_to_elems_dev(const uint64_t* soa, uint64_t* elems, size_t words, size_t n, int fq12_order, int device, void* stream) {
    return launch_layout(false, soa, elems, words, n, words == 48 ? fq12_order : 0, device, stream);
}
int bn254_multi_pairing_batch_elems(const uint64_t* g1, const uint64_t* g2, uint64_t* out, size_t n_groups, size_t k, int do_final_exp,
                                    int out_order, int device, void* stream) {
    if (n_groups == 0) return BN254_OK;
    if (!g1 || !g2 || !out || k == 0 || (out_order != BN254_FQ12_MYFQ12 && out_order != BN254_FQ12_ARK)) return BN254_ERR_INVALID_ARG;
    int rc = check_device(device);
    if (rc) return rc;
    if (n_groups > PIPE_CHUNK) {
        if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return BN254_ERR_HIP;
        HostFmt fmt; fmt.elems = true; fmt.out_order = out_order;
        return run_pipeline(&device, 1, g1, g2, out, n_groups, k, do_final_exp, fmt);
    }
    Stage s; uint64_t *e1, *e2, *e3, *d1, *d2, *d3; size_t np = n_groups * k;
    if ((rc = s.init(device, stream)) || (rc = s.up(g1, 64 * np, &e1)) || (rc = s.up(g2, 128 * np, &e2)) || (rc = s.up(nullptr, 384 * n_groups, &e3))) return rc;
    {   // the throughput kernels read and write element-major arrays themselves: no transposition pass on either side
        const int mode = IO_IN_ELEMS | IO_OUT_ELEMS | (out_order == BN254_FQ12_ARK ? IO_OUT_ARK : 0);
        if (do_final_exp ? direct_elems_ok<true, true>(n_groups, k, device, stream) : direct_elems_ok<true, false>(n_groups, k, device, stream)) {
            rc = do_final_exp ? launch_pairing<true, true>(e1, e2, nullptr, e3, n_groups, k, device, stream, mode)
                              : launch_pairing<true, false>(e1, e2, nullptr, e3, n_groups, k, device, stream, mode);
            if (rc) return rc;
            return finish_host(out, e3, 384 * n_groups, device, stream);
        }
    }
    if ((rc = s.up(nullptr, 64 * np, &d1)) || (rc = s.up(nullptr, 128 * np, &d2)) || (rc = s.up(nullptr, 384 * n_groups, &d3))) return rc;
    if ((rc = launch_layout(true, e1, d1, 8, np, 0, device, stream)) || (rc = launch_layout(true, e2, d2, 16, np, 0, device, stream))) return rc;
    rc = (k == 1 && do_final_exp) ? bn254_pairing_batch_dev(d1, d2, d3, n_groups, device, stream)
                                  : bn254_multi_pairing_batch_dev(d1, d2, d3, n_groups, k, do_final_exp, device, stream);
    if (rc || (rc = launch_layout(false, d3, e3, 48, n_groups, out_order, device, stream))) return rc;
    return finish_host(out, e3, 384 * n_groups, device, stream);
}
// Device-resident element-major batches: the throughput kernels take them as they are; batches that the lane-cooperative programs serve
// (which read planes) go through the transposition kernels and per-stream staging buffers.
int bn254_multi_pairing_batch_elems_dev(const uint64_t* g1, const uint64_t* g2, uint64_t* out, size_t n_groups, size_t k, int do_final_exp,
                                        int out_order, int device, void* stream) {
    if (n_groups == 0) return BN254_OK;
    if (!g1 || !g2 || !out || k == 0 || (out_order != BN254_FQ12_MYFQ12 && out_order != BN254_FQ12_ARK) || n_groups * k >= (1ull << 29)) return BN254_ERR_INVALID_ARG;
    int rc = check_device(device);
    if (rc) return rc;
    const int mode = IO_IN_ELEMS | IO_OUT_ELEMS | (out_order == BN254_FQ12_ARK ? IO_OUT_ARK : 0);
    if (do_final_exp ? direct_elems_ok<true, true>(n_groups, k, device, stream) : direct_elems_ok<true, false>(n_groups, k, device, stream))
        return do_final_exp ? launch_pairing<true, true>(g1, g2, nullptr, out, n_groups, k, device, stream, mode)
                            : launch_pairing<true, false>(g1, g2, nullptr, out, n_groups, k, device, stream, mode);
    Stage s; uint64_t *d1, *d2, *d3; size_t np = n_groups * k;
    if ((rc = s.init(device, stream)) || (rc = s.up(nullptr, 64 * np, &d1)) || (rc = s.up(nullptr, 128 * np, &d2)) || (rc = s.up(nullptr, 384 * n_groups, &d3))) return rc;
    if ((rc = launch_layout(true, g1, d1, 8, np, 0, device, stream)) || (rc = launch_layout(true, g2, d2, 16, np, 0, device, stream))) return rc;
    rc = (k == 1 && do_final_exp) ? bn254_pairing_batch_dev(d1, d2, d3, n_groups, device, stream)
                                  : bn254_multi_pairing_batch_dev(d1, d2, d3, n_groups, k, do_final_exp, device, stream);
    if (rc) return rc;
    return launch_layout(false, d3, out, 48, n_groups, out_order, device, stream);
}
int bn254_pairing_batch_elems_dev(const uint64_t* g1, const uint64_t* g2, uint64_t* out, size_t n, int out_order, int device, void* stream) {
    return bn254_multi_pairing_batch_elems_dev(g1, g2, out, n, 1, 1, out_order, device, stream);
}
int bn254_multi_pairing_check_batch_elems(const uint64_t* g1, const uint64_t* g2, uint8_t* verdict, size_t n_groups, size_t k, int device, void* stream) {
    if (n_groups == 0) return BN254_OK;
    if (!g1 || !g2 || !verdict || k == 0) return BN254_ERR_INVALID_ARG;
    Stage s; uint64_t *e1, *e2, *d1, *d2, *d3; int rc; size_t np = n_groups * k;
    if ((rc = s.init(device, stream)) || (rc = s.up(g1, 64 * np, &e1)) || (rc = s.up(g2, 128 * np, &e2)) || (rc = s.up(nullptr, 64 * np, &d1)) ||
        (rc = s.up(nullptr, 128 * np, &d2)) || (rc = s.up(nullptr, n_groups, &d3))) return rc;
    if ((rc = launch_layout(true, e1, d1, 8, np, 0, device, stream)) || (rc = launch_layout(true, e2, d2, 16, np, 0, device, stream)) ||
        (rc = bn254_multi_pairing_check_batch_dev(d1, d2, (uint8_t*)d3, n_groups, k, device, stream))) return rc;
    return finish_host(verdict, d3, n_groups, device, stream);
}
int bn254_pairing_batch_elems(const uint64_t* g1, const uint64_t* g2, uint64_t* out, size_t n, int out_order, int device, void* stream) {
    return bn254_multi_pairing_batch_elems(g1, g2, out, n, 1, 1, out_order, device, stream);
}
int bn254_miller_loop_batch_elems(const uint64_t* g1, const uint64_t* g2, uint64_t* f_out, size_t n, int device, void* stream) {
    return bn254_multi_pairing_batch_elems(g1, g2, f_out, n, 1, 0, BN254_FQ12_MYFQ12, device, stream);
}
int bn254_final_exp_batch_elems(const uint64_t* f_in, uint64_t* out, size_t n, int in_order, int out_order, int device, void* stream) {
    if (n == 0) return BN254_OK;
    if (!f_in || !out) return BN254_ERR_INVALID_ARG;
    if ((in_order != BN254_FQ12_MYFQ12 && in_order != BN254_FQ12_ARK) || (out_order != BN254_FQ12_MYFQ12 && out_order != BN254_FQ12_ARK)) return BN254_ERR_INVALID_ARG;
    Stage s; uint64_t *e1, *e3, *d1, *d3; int rc;
    if ((rc = s.init(device, stream)) || (rc = s.up(f_in, 384 * n, &e1)) || (rc = s.up(nullptr, 384 * n, &e3))) return rc;
    if (in_order == BN254_FQ12_MYFQ12 && direct_elems_ok<false, true>(n, 1, device, stream)) {      // (an ark-ordered INPUT still takes the transposition pass)
        if ((rc = launch_pairing<false, true>(nullptr, nullptr, e1, e3, n, 1, device, stream, IO_IN_ELEMS | IO_OUT_ELEMS | (out_order == BN254_FQ12_ARK ? IO_OUT_ARK : 0))))
            return rc;
        return finish_host(out, e3, 384 * n, device, stream);
    }
    if ((rc = s.up(nullptr, 384 * n, &d1)) || (rc = s.up(nullptr, 384 * n, &d3))) return rc;
    if ((rc = launch_layout(true, e1, d1, 48, n, in_order, device, stream)) || (rc = bn254_final_exp_batch_dev(d1, d3, n, device, stream)) ||
        (rc = launch_layout(false, d3, e3, 48, n, out_order, device, stream))) return rc;
    return finish_host(out, e3, 384 * n, device, stream);
}

// ---- host-pointer pipeline, one or several GPUs of this process: contiguous slices of the batch per device,
// no exchange step.  A slice is cut into chunks of PIPE_CHUNK lanes (one full grid); a worker thread owns one private
// stream and device buffers for one chunk and walks every second chunk of its device: it stages its chunk of every limb
// plane (2-D copies straight out of / into the caller's SoA arrays), launches the kernel and copies the result back.  Two
// workers per device alternate, so one worker's copies run under the other's kernel (the kernels fill the chip and
// serialise).  Units are pairings (k = 1) or k-pair groups.
// contiguous copy between pageable host memory and the device, issued as a 2-D copy of 1 MiB rows: measured 2-3x faster
// than hipMemcpyAsync on the same (pageable) buffers, whose staging path blocks the issuing thread
static hipError_t copy_rows(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t st) {
    const size_t row = 1u << 20;
    size_t rows = bytes / row, rest = bytes - rows * row;
    hipError_t e = hipSuccess;
    if (rows) e = hipMemcpy2DAsync(dst, row, src, row, row, rows, kind, st);
    if (e == hipSuccess && rest) e = hipMemcpyAsync((char*)dst + rows * row, (const char*)src + rows * row, rest, kind, st);
    return e;
}

// Is [p, p + bytes) page-locked memory the HIP runtime knows (hipHostMalloc / hipHostRegister, also through bn254_alloc_pinned /
// bn254_host_register)?  Copies from / to such memory are true asynchronous DMA at the link rate; pageable memory goes through the
// runtime's staging buffers (a host memcpy on the issuing thread).
static bool host_pinned(const void* p, size_t bytes) {
    if (!p || !bytes) return false;
    for (const char* q : {(const char*)p, (const char*)p + bytes - 1}) {
        hipPointerAttribute_t a;
        if (hipPointerGetAttributes(&a, q) != hipSuccess) { (void)hipGetLastError(); return false; }
        if (a.type != hipMemoryTypeHost) return false;
    }
    return true;
}
struct PipeWorker {            // one worker of the host-pointer pipeline: its private stream and the event it waits for its chunks on
    hipStream_t st;
    hipEvent_t out_done;
};

static int run_chunks(int dev, PipeWorker pw, const uint64_t* g1, const uint64_t* g2, uint64_t* out, size_t n_units, size_t k, int do_final_exp,
                      size_t u0, size_t cnt, size_t chunk, size_t first, size_t step, HostFmt fmt, const FixedJob* fx) {
    if (cnt == 0 || first * chunk >= cnt) return BN254_OK;
    if (hipSetDevice(dev) != hipSuccess) return BN254_ERR_INVALID_ARG;
    int rc = BN254_OK;
    hipStream_t st = pw.st;
    {
        Stage s; uint64_t *d1, *d2, *d3, *e1 = nullptr, *e2 = nullptr, *e3 = nullptr;
        std::vector<uint64_t> hres;                             // (verdict calls: a chunk's Fq12 values on their way to the comparison)
        const size_t k2 = fx ? (g2 ? 1 : 0) : k;               // G2 points per unit (a fixed-G2 unit brings its own point only -- or none)
        size_t cap = cnt < chunk ? cnt : chunk, np_all = n_units * k, np2_all = n_units * k2;
        if ((rc = s.init(dev, st)) || (rc = s.up(nullptr, 64 * cap * k, &d1)) || (rc = s.up(nullptr, 128 * cap * (k2 ? k2 : 1), &d2)) || (rc = s.up(nullptr, 384 * cap, &d3))) goto done;
        if (fmt.elems && ((rc = s.up(nullptr, 64 * cap * k, &e1)) || (rc = s.up(nullptr, 128 * cap * (k2 ? k2 : 1), &e2)) || (rc = s.up(nullptr, 384 * cap, &e3)))) goto done;
        for (size_t c0 = first * chunk; c0 < cnt; c0 += step * chunk) {
            size_t m = cnt - c0 < chunk ? cnt - c0 : chunk, np = m * k, np2 = m * k2, base = u0 + c0;
            // a chunk of an element-major array is one contiguous run, and the throughput kernels take it as it is (no transposition pass, which --
            // a kernel -- would wait for the other worker's launch to leave the CUs); the lane-cooperative programs of a small last chunk read
            // planes: those are then made on the device
            const bool direct = fmt.elems && !fx && (do_final_exp ? direct_elems_ok<true, true>(m, k, dev, st) : direct_elems_ok<true, false>(m, k, dev, st));
            if (fmt.elems) {
                if (copy_rows(e1, g1 + base * k * 8, np * 64, hipMemcpyHostToDevice, st) != hipSuccess ||
                    (np2 && copy_rows(e2, g2 + base * k2 * 16, np2 * 128, hipMemcpyHostToDevice, st) != hipSuccess)) { rc = BN254_ERR_HIP; goto done; }
                if (!direct && !fx && ((rc = launch_layout(true, e1, d1, 8, np, 0, dev, st)) || (rc = launch_layout(true, e2, d2, 16, np, 0, dev, st)))) goto done;
            } else if (hipMemcpy2DAsync(d1, np * 8, g1 + base * k, np_all * 8, np * 8, 8, hipMemcpyHostToDevice, st) != hipSuccess ||
                       (np2 && hipMemcpy2DAsync(d2, np2 * 8, g2 + base * k2, np2_all * 8, np2 * 8, 16, hipMemcpyHostToDevice, st) != hipSuccess)) { rc = BN254_ERR_HIP; goto done; }
            if (fx) {
                // the fixed-G2 kernel reads either layout itself (a small last chunk: launch_fixed expands the pairs for the lane-cooperative program)
                const bool vd = fx->verdict != nullptr;                      // (verdicts: element-major input only)
                if (c0 == first * chunk && fx->table_ready && hipStreamWaitEvent(st, fx->table_ready, 0) != hipSuccess) { rc = BN254_ERR_HIP; goto done; }
                const bool last = c0 + chunk >= cnt;                           // nothing is queued behind the job's last launch
                const bool planes_out = !fmt.elems || (vd && last);
                const int mode = fmt.elems ? (IO_IN_ELEMS | (planes_out ? 0 : IO_OUT_ELEMS | (!vd && fmt.out_order == BN254_FQ12_ARK ? IO_OUT_ARK : 0))) : 0;
                if ((rc = launch_fixed(fmt.elems ? e1 : d1, !k2 ? nullptr : (fmt.elems ? e2 : d2), fx->table, fx->k_fixed, planes_out ? d3 : e3, m, mode, dev, st))) goto done;
                if (vd && last) {
                    if ((rc = launch_is_equal(d3, fx->target, (uint8_t*)e1, m, st))) goto done;              // (e1: this chunk's inputs, dead behind its launch)
                    if (hipMemcpy2DAsync(fx->verdict + base, m, e1, m, m, 1, hipMemcpyDeviceToHost, st) != hipSuccess) { rc = BN254_ERR_HIP; goto done; }
                } else if (vd) {
                    // The comparison is the worker THREAD's: a compare kernel behind this launch would have to wait for the other worker's launch to leave
                    // the CUs (8 ms), and this worker's next upload with it.  The Fq12 values come back under the other worker's kernel like any
                    // result; 384 bytes per unit are compared while that kernel runs.  (The job's LAST chunk has no launch behind it: compared on the device.)
                    hres.resize(48 * cap);
                    if (copy_rows(hres.data(), e3, m * 384, hipMemcpyDeviceToHost, st) != hipSuccess) { rc = BN254_ERR_HIP; goto done; }
                } else if (fmt.elems) {
                    if (copy_rows(out + base * 48, e3, m * 384, hipMemcpyDeviceToHost, st) != hipSuccess) { rc = BN254_ERR_HIP; goto done; }
                } else if (hipMemcpy2DAsync(out + base, n_units * 8, d3, m * 8, m * 8, 48, hipMemcpyDeviceToHost, st) != hipSuccess) { rc = BN254_ERR_HIP; goto done; }
                if (hipEventRecord(pw.out_done, st) != hipSuccess || hipEventSynchronize(pw.out_done) != hipSuccess) { rc = BN254_ERR_HIP; goto done; }
                if (vd && !last) {
                    uint64_t t[48];
                    const uint64_t one[4] = BN254_FQ_ONE_LIMBS;
                    for (int w = 0; w < 48; w++) t[w] = fx->target ? fx->target[w] : (w < 4 ? one[w] : 0ull);
                    for (size_t i = 0; i < m; i++) fx->verdict[base + i] = memcmp(hres.data() + 48 * i, t, 384) == 0 ? 1 : 0;
                }
                continue;
            }
            if (direct) {
                const int mode = IO_IN_ELEMS | IO_OUT_ELEMS | (fmt.out_order == BN254_FQ12_ARK ? IO_OUT_ARK : 0);
                rc = do_final_exp ? launch_pairing<true, true>(e1, e2, nullptr, e3, m, k, dev, st, mode) : launch_pairing<true, false>(e1, e2, nullptr, e3, m, k, dev, st, mode);
            } else
                rc = (k == 1 && do_final_exp) ? bn254_pairing_batch_dev(d1, d2, d3, m, dev, st)
                                              : bn254_multi_pairing_batch_dev(d1, d2, d3, m, k, do_final_exp, dev, st);
            if (rc) goto done;
            if (fmt.elems) {
                if (!direct && (rc = launch_layout(false, d3, e3, 48, m, fmt.out_order, dev, st))) goto done;
                if (copy_rows(out + base * 48, e3, m * 384, hipMemcpyDeviceToHost, st) != hipSuccess) { rc = BN254_ERR_HIP; goto done; }
            } else if (hipMemcpy2DAsync(out + base, n_units * 8, d3, m * 8, m * 8, 48, hipMemcpyDeviceToHost, st) != hipSuccess) { rc = BN254_ERR_HIP; goto done; }
            // the buffers are free for the next chunk once the stream has drained -- waited for on an EVENT: the status read-back that used to
            // stand here is an 8-byte copy, which the runtime performs with a blit kernel, and that kernel waited for the other worker's launch
            // to leave the CUs (0.2 ms per chunk between the kernels).  The sticky status words are read once, behind the last chunk.
            if (hipEventRecord(pw.out_done, st) != hipSuccess || hipEventSynchronize(pw.out_done) != hipSuccess) { rc = BN254_ERR_HIP; goto done; }
        }
        rc = bn254_last_status(dev, st);
    done:
        (void)hipStreamSynchronize(st);
    }
    return rc;
}

// devices[0..n_dev): the batch is split into n_dev contiguous slices
static int run_pipeline(const int* devices, int n_dev, const uint64_t* g1, const uint64_t* g2, uint64_t* out, size_t n_units, size_t k,
                        int do_final_exp, HostFmt fmt, const FixedJob* fixed) {
    size_t chunk = PIPE_CHUNK;                        // lanes = units (one unit per lane whatever k is)
    // (element-major data passes two small kernels on the way in and one on the way out (k_layout); a pairing launch holds every register of every
    // CU, so they run at the launch boundaries: 0.5 ms per chunk, 8 %.  Launches of one CU less do not help -- the other worker's launch takes the
    // free CU: measured 116 ms against 109.5 for 2^20 -- nor does a third stream per worker: copies behind a cross-stream event run as blit kernels.)
    size_t per = (n_units + (size_t)n_dev - 1) / (size_t)n_dev;
    // the workers' streams (and with them the scratch and the staging buffers, which are kept per stream) live as long as
    // the library: a pipeline call costs no allocation after the first.  Devices are locked in ascending order.
    std::vector<std::unique_lock<std::mutex>> locks;
    for (int d = 0; d < n_dev; d++) {
        int rc = check_device(devices[d]);
        if (rc) return rc;
        if (d && devices[d] <= devices[d - 1]) return BN254_ERR_INVALID_ARG;
        DeviceCtx& c = g_ctx[devices[d]];
        locks.emplace_back(c.pipe_mu);
        for (hipStream_t& st : c.pipe_stream)
            if (!st && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return BN254_ERR_HIP;
        for (hipEvent_t& ev : c.pipe_ev)
            if (!ev && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return BN254_ERR_HIP;
    }
    std::vector<FixedJob> jobs((size_t)n_dev);
    if (fixed) {
        // the line table of this call's fixed points, once per device (2 ms) on the first worker's stream, before the workers start
        const size_t kf = fixed->k_fixed, tab_max = bn254_g2_lines_bytes(BN254_FIXED_MAX);
        for (int d = 0; d < n_dev; d++) {
            DeviceCtx& c = g_ctx[devices[d]];
            hipStream_t st = c.pipe_stream[0];
            if (hipSetDevice(devices[d]) != hipSuccess) return BN254_ERR_HIP;
            if (!c.pipe_table && hipMalloc(&c.pipe_table, tab_max + 2 * 128 * BN254_FIXED_MAX) != hipSuccess) { c.pipe_table = nullptr; return BN254_ERR_ALLOC; }
            uint64_t* tab = (uint64_t*)c.pipe_table;
            uint64_t *in = (uint64_t*)((char*)c.pipe_table + tab_max), *planes = in + 16 * BN254_FIXED_MAX;
            jobs[(size_t)d] = *fixed;
            jobs[(size_t)d].table = tab;
            // the table of the previous call stands when the fixed points are the same (a verifier's key does not change between its batches: 2 ms saved)
            std::vector<uint64_t> key(fixed->g2_fixed, fixed->g2_fixed + 16 * kf);
            key.push_back(fmt.elems ? 1 : 0);
            if (key == c.pipe_table_key) continue;
            c.pipe_table_key.clear();
            if (hipMemcpyAsync(in, fixed->g2_fixed, 128 * kf, hipMemcpyHostToDevice, st) != hipSuccess) return BN254_ERR_HIP;
            int rc = BN254_OK;
            if (fmt.elems && (rc = launch_layout(true, in, planes, 16, kf, 0, devices[d], st))) return rc;
            if ((rc = bn254_g2_lines_dev(fmt.elems ? planes : in, kf, tab, devices[d], st))) return rc;
            // (no wait here: worker 0 works on this very stream; worker 1 waits for the event in front of its first LAUNCH, its uploads run meanwhile)
            if (!c.pipe_table_ev && hipEventCreateWithFlags(&c.pipe_table_ev, hipEventDisableTiming) != hipSuccess) return BN254_ERR_HIP;
            if (hipEventRecord(c.pipe_table_ev, st) != hipSuccess) return BN254_ERR_HIP;
            jobs[(size_t)d].table_ready = c.pipe_table_ev;
            c.pipe_table_key = std::move(key);
        }
    }
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
            DeviceCtx& dc = g_ctx[dev];
            PipeWorker pw{dc.pipe_stream[w], dc.pipe_ev[w]};
            const FixedJob* fj = fixed ? &jobs[(size_t)d] : nullptr;
            th.emplace_back([=] { *slot = run_chunks(dev, pw, g1, g2, out, n_units, k, do_final_exp, u0, c, chunk, w, workers, fmt, fj); });
        }
    }
    for (auto& t : th) t.join();
    for (int rc : rcs)
        if (rc) {
            if (fixed)              // (a table whose making raised the status must not be taken for good by the next call)
                for (int d = 0; d < n_dev; d++) g_ctx[devices[d]].pipe_table_key.clear();
            return rc;
        }
    return BN254_OK;
}

static int sharded(const uint64_t* g1, const uint64_t* g2, uint64_t* out, size_t n_groups, size_t k, int do_final_exp, int n_devices, HostFmt fmt) {
    if (n_groups == 0) return BN254_OK;
    if (!g1 || !g2 || !out || k == 0 || n_devices <= 0 || (fmt.out_order != BN254_FQ12_MYFQ12 && fmt.out_order != BN254_FQ12_ARK))
        return BN254_ERR_INVALID_ARG;
    int cnt = bn254_device_count();
    if (cnt <= 0) return BN254_ERR_NO_DEVICE;
    if (n_devices > cnt) return BN254_ERR_INVALID_ARG;
    std::vector<int> devs((size_t)n_devices);
    for (int d = 0; d < n_devices; d++) devs[(size_t)d] = d;
    return run_pipeline(devs.data(), n_devices, g1, g2, out, n_groups, k, do_final_exp, fmt);
}
int bn254_multi_pairing_sharded(const uint64_t* g1, const uint64_t* g2, uint64_t* out, size_t n_groups, size_t k, int do_final_exp, int n_devices) {
    return sharded(g1, g2, out, n_groups, k, do_final_exp, n_devices, HostFmt());
}
int bn254_multi_pairing_sharded_elems(const uint64_t* g1, const uint64_t* g2, uint64_t* out, size_t n_groups, size_t k, int do_final_exp,
                                      int out_order, int n_devices) {
    HostFmt fmt; fmt.elems = true; fmt.out_order = out_order;
    return sharded(g1, g2, out, n_groups, k, do_final_exp, n_devices, fmt);
}
int bn254_pairing_sharded_elems(const uint64_t* g1, const uint64_t* g2, uint64_t* out, size_t n, int out_order, int n_devices) {
    return bn254_multi_pairing_sharded_elems(g1, g2, out, n, 1, 1, out_order, n_devices);
}

// the fixed-G2 verdicts over the first n_devices GPUs of this process: contiguous slices of the groups per device, each device makes its own line table
int bn254_pairing_fixed_g2_check_sharded_elems(const uint64_t* g1, const uint64_t* g2_var, const uint64_t* g2_fixed, size_t k_fixed, const uint64_t* target,
                                               uint8_t* verdict, size_t n, int n_devices) {
    if (n == 0) return BN254_OK;
    if (!g1 || !g2_fixed || !verdict || k_fixed == 0 || k_fixed > (size_t)BN254_FIXED_MAX || n * (k_fixed + 1) >= ((size_t)1 << 29) || n_devices <= 0) return BN254_ERR_INVALID_ARG;
    int cnt = bn254_device_count();
    if (cnt <= 0) return BN254_ERR_NO_DEVICE;
    if (n_devices > cnt) return BN254_ERR_INVALID_ARG;
    std::vector<int> devs((size_t)n_devices);
    for (int d = 0; d < n_devices; d++) devs[(size_t)d] = d;
    HostFmt fmt; fmt.elems = true;
    FixedJob job; job.g2_fixed = g2_fixed; job.k_fixed = k_fixed; job.verdict = verdict; job.target = target;
    return run_pipeline(devs.data(), n_devices, g1, g2_var, nullptr, n, k_fixed + (g2_var ? 1 : 0), 1, fmt, &job);
}

int bn254_pairing_sharded(const uint64_t* g1, const uint64_t* g2, uint64_t* out, size_t n, int n_devices) {
    return bn254_multi_pairing_sharded(g1, g2, out, n, 1, 1, n_devices);
}

// ---- device-pointer multi-GPU entry (one process): the batch lives in HBM of devices[0] (limb-major planes),
// shard i = contiguous slice i of the units runs on devices[i] on a private stream of that device.  The slices travel as
// plane-by-plane 1-D copies -- hipMemcpyPeerAsync between devices: copy engines over xGMI, no CU on either side, so they run
// under the other shards' kernels (which fill their chips: profiles/r03_coresidency.txt) -- ordered behind `stream` by an
// event.  A device may appear several times (two shards on one GPU take turns on it): that is how the path is tested on a
// one-GPU box.  Synchronous: returns when every shard's results are in `out` and its status word has been read.
static int sharded_dev(const uint64_t* g1, const uint64_t* g2, uint64_t* out, size_t n_units, size_t k, int do_final_exp, const int* devices,
                       int n_dev, void* stream) {
    if (n_units == 0) return BN254_OK;
    if (!g1 || !g2 || !out || k == 0 || n_dev <= 0 || n_dev > 64 || n_units * k >= (1ull << 29)) return BN254_ERR_INVALID_ARG;
    int cnt = bn254_device_count();
    if (cnt <= 0) return BN254_ERR_NO_DEVICE;
    std::vector<int> devs((size_t)n_dev);
    for (int i = 0; i < n_dev; i++) {
        devs[(size_t)i] = devices ? devices[i] : i;
        if (devs[(size_t)i] < 0 || devs[(size_t)i] >= cnt || devs[(size_t)i] >= 64) return BN254_ERR_INVALID_ARG;
    }
    const int root = devs[0];
    std::vector<std::unique_lock<std::mutex>> locks;          // distinct devices, ascending: no lock-order inversion between calls
    for (int d = 0; d < 64; d++)
        for (int x : devs) if (x == d) { locks.emplace_back(g_ctx[d].shard_mu); break; }
    std::vector<hipStream_t> sts((size_t)n_dev);
    for (int i = 0; i < n_dev; i++) {
        int d = devs[(size_t)i];
        size_t j = 0;
        for (int e = 0; e < i; e++) j += devs[(size_t)e] == d;
        if (hipSetDevice(d) != hipSuccess) return BN254_ERR_HIP;
        if (d != root) {                                       // direct xGMI access both ways (already enabled: fine)
            (void)hipDeviceEnablePeerAccess(root, 0);
            (void)hipGetLastError();
        }
        DeviceCtx& c = g_ctx[d];
        std::lock_guard<std::mutex> lk(c.mu);
        while (c.shard_streams.size() <= j) {
            hipStream_t st = nullptr;
            if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return BN254_ERR_HIP;
            c.shard_streams.push_back(st);
        }
        sts[(size_t)i] = c.shard_streams[j];
    }
    if (hipSetDevice(root) != hipSuccess) return BN254_ERR_HIP;
    for (int d : devs) if (d != root) { (void)hipDeviceEnablePeerAccess(d, 0); (void)hipGetLastError(); }
    hipEvent_t ready = nullptr;
    HIPCHK(hipEventCreateWithFlags(&ready, hipEventDisableTiming));
    int rc = BN254_OK;
    if (hipEventRecord(ready, (hipStream_t)stream) != hipSuccess) rc = BN254_ERR_HIP;
    const size_t np_all = n_units * k;
    std::vector<Stage> stages((size_t)n_dev);
    for (int i = 0; i < n_dev && !rc; i++) {
        size_t lo = n_units * (size_t)i / (size_t)n_dev, hi = n_units * (size_t)(i + 1) / (size_t)n_dev, m = hi - lo, np = m * k;
        if (m == 0) continue;
        int d = devs[(size_t)i];
        hipStream_t st = sts[(size_t)i];
        Stage& s = stages[(size_t)i];
        uint64_t *d1, *d2, *d3;
        if ((rc = s.init(d, st)) || (rc = s.up(nullptr, 64 * np, &d1)) || (rc = s.up(nullptr, 128 * np, &d2)) || (rc = s.up(nullptr, 384 * m, &d3))) break;
        if (hipStreamWaitEvent(st, ready, 0) != hipSuccess) { rc = BN254_ERR_HIP; break; }
        auto plane = [&](void* dst, int dst_dev, const void* src, int src_dev, size_t bytes) {
            return dst_dev == src_dev ? hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st)
                                      : hipMemcpyPeerAsync(dst, dst_dev, src, src_dev, bytes, st);
        };
        for (size_t w = 0; w < 8 && !rc; w++) if (plane(d1 + w * np, d, g1 + w * np_all + lo * k, root, np * 8) != hipSuccess) rc = BN254_ERR_HIP;
        for (size_t w = 0; w < 16 && !rc; w++) if (plane(d2 + w * np, d, g2 + w * np_all + lo * k, root, np * 8) != hipSuccess) rc = BN254_ERR_HIP;
        if (rc) break;
        rc = (k == 1 && do_final_exp) ? bn254_pairing_batch_dev(d1, d2, d3, m, d, st) : bn254_multi_pairing_batch_dev(d1, d2, d3, m, k, do_final_exp, d, st);
        for (size_t w = 0; w < 48 && !rc; w++) if (plane(out + w * n_units + lo, root, d3 + w * m, d, m * 8) != hipSuccess) rc = BN254_ERR_HIP;
    }
    for (int i = 0; i < n_dev; i++) {                          // always drain what was issued, then report the first failure
        if (!stages[(size_t)i].sc) continue;
        int r = bn254_last_status(devs[(size_t)i], sts[(size_t)i]);
        if (!rc) rc = r;
    }
    (void)hipSetDevice(root);
    (void)hipEventDestroy(ready);
    return rc;
}
int bn254_multi_pairing_sharded_dev(const uint64_t* g1, const uint64_t* g2, uint64_t* out, size_t n_groups, size_t k, int do_final_exp,
                                    const int* devices, int n_devices, void* stream) {
    return sharded_dev(g1, g2, out, n_groups, k, do_final_exp, devices, n_devices, stream);
}
int bn254_pairing_sharded_dev(const uint64_t* g1, const uint64_t* g2, uint64_t* out, size_t n, const int* devices, int n_devices, void* stream) {
    return sharded_dev(g1, g2, out, n, 1, 1, devices, n_devices, stream);
}

}  // extern "C"
