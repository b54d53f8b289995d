// valu_calib.hip -- gfx950 VALU issue-rate calibration for the integer ops a 254-bit
// Montgomery multiply is built from.  Prints cycles per wave-instruction (s_memtime
// deltas inside the kernel) at 1, 2 and 4 waves per SIMD, plus chip-wide G instr/s.
// Build: hipcc --offload-arch=gfx950 -O2 tools/valu_calib.hip -o tools/valu_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <string>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

// Each kernel: 8 independent chains (or 1 dependent chain), ITER iterations of 64 instrs.
template <int KIND>
__global__ void __launch_bounds__(1024) k_calib(uint64_t* out, int iters) {
    extern __shared__ uint32_t lds_pad[];
    if (iters < 0) lds_pad[threadIdx.x] = 1;  // keep the allocation (occupancy limiter)
    uint32_t a = threadIdx.x * 2654435761u + 12345u, b = blockIdx.x * 40503u + 77u + threadIdx.x;
    uint64_t c0 = a, c1 = b, c2 = a ^ b, c3 = a + b, c4 = 5, c5 = 6, c6 = 7, c7 = 8;
    uint32_t h0 = 1, h1 = 2, h2 = 3, h3 = 4, h4 = 5, h5 = 6, h6 = 7, h7 = 8;
    double d0 = a, d1 = b, d2 = 1.5, d3 = 2.5, d4 = 3.5, d5 = 4.5, d6 = 5.5, d7 = 6.5, da = 1.0000001, db = 0.5;
    uint64_t r0 = __builtin_amdgcn_s_memrealtime();
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {  // v_mad_u64_u32, 8 independent chains
            asm volatile(REP8(
                "v_mad_u64_u32 %0, s[10:11], %8, %9, %0\n\t" "v_mad_u64_u32 %1, s[10:11], %8, %9, %1\n\t"
                "v_mad_u64_u32 %2, s[10:11], %8, %9, %2\n\t" "v_mad_u64_u32 %3, s[10:11], %8, %9, %3\n\t"
                "v_mad_u64_u32 %4, s[10:11], %8, %9, %4\n\t" "v_mad_u64_u32 %5, s[10:11], %8, %9, %5\n\t"
                "v_mad_u64_u32 %6, s[10:11], %8, %9, %6\n\t" "v_mad_u64_u32 %7, s[10:11], %8, %9, %7\n\t")
                : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b) : "s10", "s11");
        } else if (KIND == 1) {  // v_mad_u64_u32 single dependent chain
            asm volatile(REP64("v_mad_u64_u32 %0, s[10:11], %1, %2, %0\n\t") : "+v"(c0) : "v"(a), "v"(b) : "s10", "s11");
        } else if (KIND == 2) {  // mad(carry->vcc) + addc pattern, 4 independent column accumulators
            asm volatile(REP8(
                "v_mad_u64_u32 %0, vcc, %8, %9, %0\n\t v_addc_co_u32_e32 %4, vcc, 0, %4, vcc\n\t"
                "v_mad_u64_u32 %1, vcc, %8, %9, %1\n\t v_addc_co_u32_e32 %5, vcc, 0, %5, vcc\n\t"
                "v_mad_u64_u32 %2, vcc, %8, %9, %2\n\t v_addc_co_u32_e32 %6, vcc, 0, %6, vcc\n\t"
                "v_mad_u64_u32 %3, vcc, %8, %9, %3\n\t v_addc_co_u32_e32 %7, vcc, 0, %7, vcc\n\t")
                : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3) : "v"(a), "v"(b) : "vcc");
        } else if (KIND == 3) {  // v_mul_lo_u32, 8 chains
            asm volatile(REP8(
                "v_mul_lo_u32 %0, %0, %8\n\t v_mul_lo_u32 %1, %1, %8\n\t v_mul_lo_u32 %2, %2, %8\n\t v_mul_lo_u32 %3, %3, %8\n\t"
                "v_mul_lo_u32 %4, %4, %8\n\t v_mul_lo_u32 %5, %5, %8\n\t v_mul_lo_u32 %6, %6, %8\n\t v_mul_lo_u32 %7, %7, %8\n\t")
                : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3), "+v"(h4), "+v"(h5), "+v"(h6), "+v"(h7) : "v"(a));
        } else if (KIND == 4) {  // v_mul_hi_u32
            asm volatile(REP8(
                "v_mul_hi_u32 %0, %0, %8\n\t v_mul_hi_u32 %1, %1, %8\n\t v_mul_hi_u32 %2, %2, %8\n\t v_mul_hi_u32 %3, %3, %8\n\t"
                "v_mul_hi_u32 %4, %4, %8\n\t v_mul_hi_u32 %5, %5, %8\n\t v_mul_hi_u32 %6, %6, %8\n\t v_mul_hi_u32 %7, %7, %8\n\t")
                : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3), "+v"(h4), "+v"(h5), "+v"(h6), "+v"(h7) : "v"(a));
        } else if (KIND == 5) {  // add_co/addc pairs (64-bit add the classic way), 4 chains
            asm volatile(REP8(
                "v_add_co_u32_e32 %0, vcc, %8, %0\n\t v_addc_co_u32_e32 %4, vcc, 0, %4, vcc\n\t"
                "v_add_co_u32_e32 %1, vcc, %8, %1\n\t v_addc_co_u32_e32 %5, vcc, 0, %5, vcc\n\t"
                "v_add_co_u32_e32 %2, vcc, %8, %2\n\t v_addc_co_u32_e32 %6, vcc, 0, %6, vcc\n\t"
                "v_add_co_u32_e32 %3, vcc, %8, %3\n\t v_addc_co_u32_e32 %7, vcc, 0, %7, vcc\n\t")
                : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3), "+v"(h4), "+v"(h5), "+v"(h6), "+v"(h7) : "v"(a) : "vcc");
        } else if (KIND == 6) {  // v_lshl_add_u64, 8 chains
            asm volatile(REP8(
                "v_lshl_add_u64 %0, %0, 0, %8\n\t v_lshl_add_u64 %1, %1, 0, %8\n\t v_lshl_add_u64 %2, %2, 0, %8\n\t v_lshl_add_u64 %3, %3, 0, %8\n\t"
                "v_lshl_add_u64 %4, %4, 0, %8\n\t v_lshl_add_u64 %5, %5, 0, %8\n\t v_lshl_add_u64 %6, %6, 0, %8\n\t v_lshl_add_u64 %7, %7, 0, %8\n\t")
                : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(c0 ^ 0x1234567ull));
        } else if (KIND == 7) {  // v_fma_f64, 8 chains
            asm volatile(REP8(
                "v_fma_f64 %0, %0, %8, %9\n\t v_fma_f64 %1, %1, %8, %9\n\t v_fma_f64 %2, %2, %8, %9\n\t v_fma_f64 %3, %3, %8, %9\n\t"
                "v_fma_f64 %4, %4, %8, %9\n\t v_fma_f64 %5, %5, %8, %9\n\t v_fma_f64 %6, %6, %8, %9\n\t v_fma_f64 %7, %7, %8, %9\n\t")
                : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(da), "v"(db));
        } else if (KIND == 8) {  // v_mad_u32_u24
            asm volatile(REP8(
                "v_mad_u32_u24 %0, %0, %8, %9\n\t v_mad_u32_u24 %1, %1, %8, %9\n\t v_mad_u32_u24 %2, %2, %8, %9\n\t v_mad_u32_u24 %3, %3, %8, %9\n\t"
                "v_mad_u32_u24 %4, %4, %8, %9\n\t v_mad_u32_u24 %5, %5, %8, %9\n\t v_mad_u32_u24 %6, %6, %8, %9\n\t v_mad_u32_u24 %7, %7, %8, %9\n\t")
                : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3), "+v"(h4), "+v"(h5), "+v"(h6), "+v"(h7) : "v"(a), "v"(b));
        } else if (KIND == 9) {  // v_mul_hi_u32_u24
            asm volatile(REP8(
                "v_mul_hi_u32_u24 %0, %0, %8\n\t v_mul_hi_u32_u24 %1, %1, %8\n\t v_mul_hi_u32_u24 %2, %2, %8\n\t v_mul_hi_u32_u24 %3, %3, %8\n\t"
                "v_mul_hi_u32_u24 %4, %4, %8\n\t v_mul_hi_u32_u24 %5, %5, %8\n\t v_mul_hi_u32_u24 %6, %6, %8\n\t v_mul_hi_u32_u24 %7, %7, %8\n\t")
                : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3), "+v"(h4), "+v"(h5), "+v"(h6), "+v"(h7) : "v"(a));
        } else if (KIND == 10) {  // v_dot2_u32_u16
            asm volatile(REP8(
                "v_dot2_u32_u16 %0, %8, %9, %0\n\t v_dot2_u32_u16 %1, %8, %9, %1\n\t v_dot2_u32_u16 %2, %8, %9, %2\n\t v_dot2_u32_u16 %3, %8, %9, %3\n\t"
                "v_dot2_u32_u16 %4, %8, %9, %4\n\t v_dot2_u32_u16 %5, %8, %9, %5\n\t v_dot2_u32_u16 %6, %8, %9, %6\n\t v_dot2_u32_u16 %7, %8, %9, %7\n\t")
                : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3), "+v"(h4), "+v"(h5), "+v"(h6), "+v"(h7) : "v"(a), "v"(b));
        } else if (KIND == 11) {  // v_add3_u32
            asm volatile(REP8(
                "v_add3_u32 %0, %0, %8, %9\n\t v_add3_u32 %1, %1, %8, %9\n\t v_add3_u32 %2, %2, %8, %9\n\t v_add3_u32 %3, %3, %8, %9\n\t"
                "v_add3_u32 %4, %4, %8, %9\n\t v_add3_u32 %5, %5, %8, %9\n\t v_add3_u32 %6, %6, %8, %9\n\t v_add3_u32 %7, %7, %8, %9\n\t")
                : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3), "+v"(h4), "+v"(h5), "+v"(h6), "+v"(h7) : "v"(a), "v"(b));
        } else if (KIND == 12) {  // v_add_co_u32 dependent carry chain (8-limb add: add_co + 7 addc), the Fq add pattern
            asm volatile(REP8(
                "v_add_co_u32_e32 %0, vcc, %8, %0\n\t v_addc_co_u32_e32 %1, vcc, %8, %1, vcc\n\t v_addc_co_u32_e32 %2, vcc, %8, %2, vcc\n\t v_addc_co_u32_e32 %3, vcc, %8, %3, vcc\n\t"
                "v_addc_co_u32_e32 %4, vcc, %8, %4, vcc\n\t v_addc_co_u32_e32 %5, vcc, %8, %5, vcc\n\t v_addc_co_u32_e32 %6, vcc, %8, %6, vcc\n\t v_addc_co_u32_e32 %7, vcc, %8, %7, vcc\n\t")
                : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3), "+v"(h4), "+v"(h5), "+v"(h6), "+v"(h7) : "v"(a) : "vcc");
        } else if (KIND == 13) {  // v_mov_b32
            asm volatile(REP8(
                "v_mov_b32 %0, %8\n\t v_mov_b32 %1, %8\n\t v_mov_b32 %2, %8\n\t v_mov_b32 %3, %8\n\t"
                "v_mov_b32 %4, %8\n\t v_mov_b32 %5, %8\n\t v_mov_b32 %6, %8\n\t v_mov_b32 %7, %8\n\t")
                : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3), "+v"(h4), "+v"(h5), "+v"(h6), "+v"(h7) : "v"(a));
        } else if (KIND == 14) {  // v_dot4_u32_u8
            asm volatile(REP8(
                "v_dot4_u32_u8 %0, %8, %9, %0\n\t v_dot4_u32_u8 %1, %8, %9, %1\n\t v_dot4_u32_u8 %2, %8, %9, %2\n\t v_dot4_u32_u8 %3, %8, %9, %3\n\t"
                "v_dot4_u32_u8 %4, %8, %9, %4\n\t v_dot4_u32_u8 %5, %8, %9, %5\n\t v_dot4_u32_u8 %6, %8, %9, %6\n\t v_dot4_u32_u8 %7, %8, %9, %7\n\t")
                : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3), "+v"(h4), "+v"(h5), "+v"(h6), "+v"(h7) : "v"(a), "v"(b));
        } else if (KIND == 15) {  // mixed: 1 mad_u64_u32 + 2 independent addc-type ops (does full-rate work hide under the multiplier?)
            asm volatile(REP8(
                "v_mad_u64_u32 %0, s[10:11], %8, %9, %0\n\t v_add3_u32 %4, %4, %8, %9\n\t v_add3_u32 %5, %5, %8, %9\n\t"
                "v_mad_u64_u32 %1, s[10:11], %8, %9, %1\n\t v_add3_u32 %6, %6, %8, %9\n\t v_add3_u32 %7, %7, %8, %9\n\t"
                "v_mad_u64_u32 %2, s[10:11], %8, %9, %2\n\t v_add3_u32 %4, %4, %8, %9\n\t v_add3_u32 %5, %5, %8, %9\n\t"
                "v_mad_u64_u32 %3, s[10:11], %8, %9, %3\n\t v_add3_u32 %6, %6, %8, %9\n\t v_add3_u32 %7, %7, %8, %9\n\t")
                : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(h4), "+v"(h5), "+v"(h6), "+v"(h7) : "v"(a), "v"(b) : "s10", "s11");
        }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    uint64_t r1 = __builtin_amdgcn_s_memrealtime();
    uint64_t sink = c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7 ^ h0 ^ h1 ^ h2 ^ h3 ^ h4 ^ h5 ^ h6 ^ h7 ^
                    (uint64_t)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
    size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    out[2 * gid] = t1 - t0;
    out[2 * gid + 1] = (r1 - r0) + (sink == 0x123456789abcdefull ? 1 : 0);
}

struct Kind { int id; const char* name; int per_iter; };

template <int K> static int run(const Kind& kd, uint64_t* dbuf, int iters) {
    // exact occupancy: one (or two) workgroups per CU forced by the LDS allocation
    struct Cfg { int wps, threads, lds; };
    const Cfg cfgs[] = {{1, 256, 100 * 1024}, {2, 512, 100 * 1024}, {4, 1024, 100 * 1024}, {8, 1024, 72 * 1024}};
    for (const Cfg& c : cfgs) {
        int per_cu = (c.wps == 8) ? 2 : 1;
        int rounds = 4;
        int blocks = 256 * per_cu * rounds;
        size_t nthreads = (size_t)blocks * c.threads;
        std::vector<uint64_t> h(nthreads * 2);
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipFuncSetAttribute((const void*)k_calib<K>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipLaunchKernelGGL(k_calib<K>, dim3(blocks), dim3(c.threads), c.lds, 0, dbuf, iters / 4);  // warm
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_calib<K>, dim3(blocks), dim3(c.threads), c.lds, 0, dbuf, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(h.data(), dbuf, h.size() * 8, hipMemcpyDeviceToHost));
        std::vector<double> cyc, clk;
        for (size_t w = 0; w < nthreads / 64; ++w) { cyc.push_back((double)h[2 * (w * 64)]); clk.push_back((double)h[2 * (w * 64)] / ((double)h[2 * (w * 64) + 1] * 10.0)); }
        std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
        double med = cyc[cyc.size() / 2], mclk = clk[clk.size() / 2];
        double n_inst = (double)iters * kd.per_iter;                 // per wave
        double ginst = n_inst * (nthreads / 64) / (ms * 1e-3) / 1e9;  // G wave-instr/s chip-wide (wall)
        printf("%-26s w/SIMD=%d  ms=%8.3f  wall Gwave-inst/s=%8.2f  shader-cyc/inst/wave=%7.3f  => cyc/inst/SIMD=%6.3f  clk(GHz)=%5.3f\n",
               kd.name, c.wps, ms, ginst, med / n_inst, med / n_inst / c.wps, mclk);
    }
    return 0;
}

// --mad-only [--json]: the one figure bench.py's roofline needs, re-taken on THIS lease -- the chip-wide issue rate of a pure
// v_mad_u64_u32 stream (the 32 x 32 + 64 multiply-add every limb product of the pairing kernels is; the signed form v_mad_i64_i32
// issues at the same rate) at eight waves per SIMD, held for about half a second so that the package settles at the clock its power
// limit allows under that load; the mean of the second half of the launches counts.
static int mad_only(bool json) {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int threads = 1024, lds = 72 * 1024, blocks = 256 * 2 * 4, iters = 8192, launches = 16;
    size_t nthreads = (size_t)blocks * threads;
    uint64_t* dbuf; CK(hipMalloc(&dbuf, nthreads * 2 * 8));
    CK(hipFuncSetAttribute((const void*)k_calib<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(k_calib<0>, dim3(blocks), dim3(threads), lds, 0, dbuf, iters / 4);
    CK(hipDeviceSynchronize());
    std::vector<hipEvent_t> ev(launches + 1);
    for (auto& evt : ev) CK(hipEventCreate(&evt));
    CK(hipEventRecord(ev[0]));
    for (int l = 0; l < launches; l++) {
        hipLaunchKernelGGL(k_calib<0>, dim3(blocks), dim3(threads), lds, 0, dbuf, iters);
        CK(hipEventRecord(ev[l + 1]));
    }
    CK(hipDeviceSynchronize());
    std::vector<uint64_t> h(nthreads * 2);
    CK(hipMemcpy(h.data(), dbuf, h.size() * 8, hipMemcpyDeviceToHost));      // stamps of the LAST launch
    std::vector<double> clk;
    for (size_t w = 0; w < nthreads / 64; ++w) clk.push_back((double)h[2 * (w * 64)] / ((double)h[2 * (w * 64) + 1] * 10.0));
    std::sort(clk.begin(), clk.end());
    double n_inst = (double)iters * 64, waves = (double)(nthreads / 64), sum = 0, best = 0, worst = 1e30;
    for (int l = launches / 2; l < launches; l++) {
        float ms = 0; CK(hipEventElapsedTime(&ms, ev[l], ev[l + 1]));
        double g = n_inst * waves / (ms * 1e-3);
        sum += g; best = std::max(best, g); worst = std::min(worst, g);
    }
    double rate = sum / (launches - launches / 2);
    if (json)
        printf("{\"what\": \"v_mad_u64_u32 x8 independent chains, 8 waves/SIMD, %d launches of %d x 64 instructions per wave, mean of the second half (wall, HIP events)\", "
               "\"wave_inst_per_s\": %.6e, \"mul32_per_s\": %.6e, \"best\": %.6e, \"worst\": %.6e, \"in_kernel_clock_ghz_median\": %.4f, "
               "\"device\": \"%s\", \"cus\": %d, \"nominal_clock_khz\": %d}\n",
               launches, iters, rate, rate * 64, best, worst, clk[clk.size() / 2], prop.name, prop.multiProcessorCount, prop.clockRate);
    else
        printf("v_mad_u64_u32 x8 indep  w/SIMD=8  wall Gwave-inst/s=%8.2f (best %8.2f, worst %8.2f)  clk(GHz)=%5.3f\n", rate / 1e9, best / 1e9, worst / 1e9,
               clk[clk.size() / 2]);
    CK(hipFree(dbuf));
    return 0;
}

int main(int argc, char** argv) {
    bool only = false, json = false;
    for (int i = 1; i < argc; i++) { only |= std::string(argv[i]) == "--mad-only"; json |= std::string(argv[i]) == "--json"; }
    if (only) return mad_only(json);
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s  CUs=%d  clock=%d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    uint64_t* dbuf; CK(hipMalloc(&dbuf, (size_t)256 * 2 * 4 * 1024 * 2 * 8));
    int iters = 8192;
    const Kind kinds[] = {
        {0, "v_mad_u64_u32 x8 indep", 64}, {1, "v_mad_u64_u32 dependent", 64}, {2, "mad(vcc)+addc pairs", 64},
        {3, "v_mul_lo_u32", 64}, {4, "v_mul_hi_u32", 64}, {5, "add_co+addc pairs", 64}, {6, "v_lshl_add_u64", 64},
        {7, "v_fma_f64", 64}, {8, "v_mad_u32_u24", 64}, {9, "v_mul_hi_u32_u24", 64}, {10, "v_dot2_u32_u16", 64},
        {11, "v_add3_u32", 64}, {12, "add_co + 7 addc chain", 64}, {13, "v_mov_b32", 64}, {14, "v_dot4_u32_u8", 64},
        {15, "1 mad64 + 2 add3 mixed", 96},
    };
    run<0>(kinds[0], dbuf, iters); run<1>(kinds[1], dbuf, iters); run<2>(kinds[2], dbuf, iters); run<3>(kinds[3], dbuf, iters);
    run<4>(kinds[4], dbuf, iters); run<5>(kinds[5], dbuf, iters); run<6>(kinds[6], dbuf, iters); run<7>(kinds[7], dbuf, iters);
    run<8>(kinds[8], dbuf, iters); run<9>(kinds[9], dbuf, iters); run<10>(kinds[10], dbuf, iters); run<11>(kinds[11], dbuf, iters);
    run<12>(kinds[12], dbuf, iters); run<13>(kinds[13], dbuf, iters); run<14>(kinds[14], dbuf, iters); run<15>(kinds[15], dbuf, iters);
    CK(hipFree(dbuf));
    return 0;
}
