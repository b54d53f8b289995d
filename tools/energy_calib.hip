// energy_calib.hip -- what does an instruction cost in ENERGY on MI355X?  k_pairing runs at the package power limit (1.3 kW of
// 1.4 kW, tools/gpu_power.sh), so its throughput is set by joules per pairing, not by issue slots.  One wave per SIMD (as the
// pairing kernels); each mode runs ~2.5 s of one instruction class on pseudo-random operands while a host thread samples
// `rocm-smi --showpower`; the program prints average power, wall time and instructions, from which
//     energy per wave-instruction = (P_mode - P_sleep) * t / N.
// Build: hipcc --offload-arch=gfx950 -O2 tools/energy_calib.hip -o build/energy_calib -lpthread
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

// 40 instructions per group
#define MAD8 \
    "v_mad_i64_i32 v[10:11], s[10:11], v2, v3, v[10:11]\n\t v_mad_i64_i32 v[12:13], s[10:11], v4, v5, v[12:13]\n\t" \
    "v_mad_i64_i32 v[14:15], s[10:11], v6, v7, v[14:15]\n\t v_mad_i64_i32 v[16:17], s[10:11], v3, v6, v[16:17]\n\t" \
    "v_mad_i64_i32 v[18:19], s[10:11], v2, v5, v[18:19]\n\t v_mad_i64_i32 v[20:21], s[10:11], v4, v7, v[20:21]\n\t" \
    "v_mad_i64_i32 v[22:23], s[10:11], v5, v6, v[22:23]\n\t v_mad_i64_i32 v[24:25], s[10:11], v2, v7, v[24:25]\n\t"
#define ADD8 \
    "v_add_u32_e32 v10, v2, v10\n\t v_sub_u32_e32 v12, v12, v3\n\t v_add_u32_e32 v14, v4, v14\n\t v_sub_u32_e32 v16, v16, v5\n\t" \
    "v_add_u32_e32 v18, v6, v18\n\t v_sub_u32_e32 v20, v20, v7\n\t v_add_u32_e32 v22, v2, v22\n\t v_sub_u32_e32 v24, v24, v3\n\t"
#define ADD64_8 \
    "v_lshl_add_u64 v[10:11], v[2:3], 0, v[10:11]\n\t v_lshl_add_u64 v[12:13], v[4:5], 0, v[12:13]\n\t" \
    "v_lshl_add_u64 v[14:15], v[6:7], 0, v[14:15]\n\t v_lshl_add_u64 v[16:17], v[2:3], 0, v[16:17]\n\t" \
    "v_lshl_add_u64 v[18:19], v[4:5], 0, v[18:19]\n\t v_lshl_add_u64 v[20:21], v[6:7], 0, v[20:21]\n\t" \
    "v_lshl_add_u64 v[22:23], v[2:3], 0, v[22:23]\n\t v_lshl_add_u64 v[24:25], v[4:5], 0, v[24:25]\n\t"
#define ACC8 \
    "v_accvgpr_write_b32 a0, v10\n\t v_accvgpr_read_b32 v12, a1\n\t v_accvgpr_write_b32 a2, v14\n\t v_accvgpr_read_b32 v16, a3\n\t" \
    "v_accvgpr_write_b32 a1, v18\n\t v_accvgpr_read_b32 v20, a0\n\t v_accvgpr_write_b32 a3, v22\n\t v_accvgpr_read_b32 v24, a2\n\t"
#define DIG8 \
    "v_bfe_i32 v26, v10, 0, 29\n\t v_ashrrev_i64 v[10:11], 29, v[12:13]\n\t v_bfe_i32 v27, v14, 0, 29\n\t v_ashrrev_i64 v[12:13], 29, v[16:17]\n\t" \
    "v_bfe_i32 v28, v18, 0, 29\n\t v_ashrrev_i64 v[14:15], 29, v[20:21]\n\t v_bfe_i32 v29, v22, 0, 29\n\t v_ashrrev_i64 v[16:17], 29, v[24:25]\n\t"
#define MULLO8 \
    "v_mul_lo_u32 v10, v2, v3\n\t v_mul_lo_u32 v12, v4, v5\n\t v_mul_lo_u32 v14, v6, v7\n\t v_mul_lo_u32 v16, v3, v6\n\t" \
    "v_mul_lo_u32 v18, v2, v5\n\t v_mul_lo_u32 v20, v4, v7\n\t v_mul_lo_u32 v22, v5, v6\n\t v_mul_lo_u32 v24, v2, v7\n\t"
#define MADU8 \
    "v_mad_u64_u32 v[10:11], s[10:11], v2, v3, v[10:11]\n\t v_mad_u64_u32 v[12:13], s[10:11], v4, v5, v[12:13]\n\t" \
    "v_mad_u64_u32 v[14:15], s[10:11], v6, v7, v[14:15]\n\t v_mad_u64_u32 v[16:17], s[10:11], v3, v6, v[16:17]\n\t" \
    "v_mad_u64_u32 v[18:19], s[10:11], v2, v5, v[18:19]\n\t v_mad_u64_u32 v[20:21], s[10:11], v4, v7, v[20:21]\n\t" \
    "v_mad_u64_u32 v[22:23], s[10:11], v5, v6, v[22:23]\n\t v_mad_u64_u32 v[24:25], s[10:11], v2, v7, v[24:25]\n\t"
#define SLEEP8 "s_sleep 8\n\t s_sleep 8\n\t s_sleep 8\n\t s_sleep 8\n\t s_sleep 8\n\t s_sleep 8\n\t s_sleep 8\n\t s_sleep 8\n\t"

// MODE: 0 sleep  1 mad_i64_i32 (29-bit balanced operands, as the field code)  2 add/sub 32  3 add 64  4 accvgpr  5 bfe/ashr64
//       6 mul_lo  7 mad_i64_i32 with SMALL operands (8 significant bits)  8 mad_u64_u32 (32-bit random operands)
template <int MODE>
__global__ void __launch_bounds__(256) k_energy(uint32_t* out, int iters, uint32_t seed) {
    extern __shared__ uint32_t lds_pad[];
    if (iters < 0) lds_pad[threadIdx.x] = 1;
    uint32_t r;
    uint32_t x = (threadIdx.x + 1) * 2654435761u ^ (blockIdx.x * 40503u) ^ seed;
    asm volatile(
        // six pseudo-random operands: 29-bit balanced (modes 1..6), 8-bit (mode 7), 32-bit (mode 8)
        "v_mov_b32 v2, %1\n\t"
        "s_mov_b32 s10, 0x9E3779B1\n\t v_mul_lo_u32 v3, v2, s10\n\t s_mov_b32 s10, 0x85EBCA77\n\t v_mul_lo_u32 v4, v3, s10\n\t"
        "s_mov_b32 s10, 0xC2B2AE3D\n\t v_mul_lo_u32 v5, v4, s10\n\t s_mov_b32 s10, 0x27D4EB2F\n\t v_mul_lo_u32 v6, v5, s10\n\t"
        "s_mov_b32 s10, 0x165667B1\n\t v_mul_lo_u32 v7, v6, s10\n\t"
        ".if %c3 == 7\n\t v_bfe_i32 v2, v2, 3, 8\n\t v_bfe_i32 v3, v3, 3, 8\n\t v_bfe_i32 v4, v4, 3, 8\n\t v_bfe_i32 v5, v5, 3, 8\n\t v_bfe_i32 v6, v6, 3, 8\n\t v_bfe_i32 v7, v7, 3, 8\n\t"
        ".elseif %c3 != 8\n\t v_bfe_i32 v2, v2, 3, 29\n\t v_bfe_i32 v3, v3, 3, 29\n\t v_bfe_i32 v4, v4, 3, 29\n\t v_bfe_i32 v5, v5, 3, 29\n\t v_bfe_i32 v6, v6, 3, 29\n\t v_bfe_i32 v7, v7, 3, 29\n\t"
        ".endif\n\t"
        "v_mov_b32 v10, v2\n\t v_mov_b32 v11, v3\n\t v_mov_b32 v12, v4\n\t v_mov_b32 v13, v5\n\t v_mov_b32 v14, v6\n\t v_mov_b32 v15, v7\n\t"
        "v_mov_b32 v16, v3\n\t v_mov_b32 v17, v4\n\t v_mov_b32 v18, v5\n\t v_mov_b32 v19, v6\n\t v_mov_b32 v20, v7\n\t v_mov_b32 v21, v2\n\t"
        "v_mov_b32 v22, v4\n\t v_mov_b32 v23, v5\n\t v_mov_b32 v24, v6\n\t v_mov_b32 v25, v7\n\t"
        "v_accvgpr_write_b32 a0, v2\n\t v_accvgpr_write_b32 a1, v3\n\t v_accvgpr_write_b32 a2, v4\n\t v_accvgpr_write_b32 a3, v5\n\t"
        "s_mov_b32 s12, %2\n\t"
        "1:\n\t"
        ".rept 100\n\t"
        ".if %c3 == 0\n\t" SLEEP8 ".endif\n\t"
        ".if %c3 == 1 || %c3 == 7\n\t" MAD8 MAD8 MAD8 MAD8 MAD8 ".endif\n\t"
        ".if %c3 == 2\n\t" ADD8 ADD8 ADD8 ADD8 ADD8 ".endif\n\t"
        ".if %c3 == 3\n\t" ADD64_8 ADD64_8 ADD64_8 ADD64_8 ADD64_8 ".endif\n\t"
        ".if %c3 == 4\n\t" ACC8 ACC8 ACC8 ACC8 ACC8 ".endif\n\t"
        ".if %c3 == 5\n\t" DIG8 DIG8 DIG8 DIG8 DIG8 ".endif\n\t"
        ".if %c3 == 6\n\t" MULLO8 MULLO8 MULLO8 MULLO8 MULLO8 ".endif\n\t"
        ".if %c3 == 8\n\t" MADU8 MADU8 MADU8 MADU8 MADU8 ".endif\n\t"
        ".endr\n\t"
        "s_sub_u32 s12, s12, 1\n\t"
        "s_cmp_lg_u32 s12, 0\n\t"
        "s_cbranch_scc1 1b\n\t"
        "v_add_u32 %0, v10, v12\n\t v_add_u32 %0, %0, v14\n\t v_add_u32 %0, %0, v16\n\t v_add_u32 %0, %0, v18\n\t v_add_u32 %0, %0, v20\n\t"
        "v_add_u32 %0, %0, v22\n\t v_add_u32 %0, %0, v24\n\t v_add_u32 %0, %0, v26\n\t"
        : "=v"(r) : "v"(x), "s"(iters), "i"(MODE)
        : "v2", "v3", "v4", "v5", "v6", "v7", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23",
          "v24", "v25", "v26", "v27", "v28", "v29", "a0", "a1", "a2", "a3", "s10", "s11", "s12", "scc");
    if (r == 0x12345) out[0] = r;
}

static std::atomic<bool> g_sampling{false};
static std::vector<double> g_samples;
static void sampler() {
    while (g_sampling.load()) {
        FILE* f = popen("rocm-smi --showpower 2>/dev/null | grep -o 'Power (W): [0-9.]*' | head -1 | grep -o '[0-9.]*$'", "r");
        char buf[64] = {0};
        if (f) { if (fgets(buf, sizeof buf, f)) g_samples.push_back(atof(buf)); pclose(f); }
    }
}

template <int MODE>
int run(const char* tag, uint32_t* d_out, int n_cu, int iters, double insts_per_iter) {
    CK(hipFuncSetAttribute((const void*)k_energy<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 147456));
    hipLaunchKernelGGL(k_energy<MODE>, dim3(n_cu), dim3(256), 147456, 0, d_out, 50, 1u);      // warm-up
    CK(hipDeviceSynchronize());
    g_samples.clear();
    g_sampling = true;
    std::thread th(sampler);
    auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(k_energy<MODE>, dim3(n_cu), dim3(256), 147456, 0, d_out, iters, 7u);
    CK(hipDeviceSynchronize());
    double t = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    g_sampling = false;
    th.join();
    // drop the first and the last sample (ramps)
    double p = 0; int c = 0;
    for (size_t i = 1; i + 1 < g_samples.size(); i++) { p += g_samples[i]; c++; }
    p = c ? p / c : 0;
    double n = (double)iters * insts_per_iter * n_cu * 4;        // wave-instructions
    printf("%-34s t %6.3f s  power %7.1f W (%2d samples)  wave-instr %.3e  cycles/instr at 2.4 GHz %5.2f\n", tag, t, p, c, n,
           t * 2.4e9 / ((double)iters * insts_per_iter));
    printf("DATA %s %.6f %.2f %.6e\n", tag, t, p, n);
    return 0;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    int n_cu = prop.multiProcessorCount;
    uint32_t* d_out;
    CK(hipMalloc(&d_out, 4096));
    const double per = 100 * 40;                     // instructions per loop iteration (sleep: 100 * 8 s_sleep)
    const int it = 330000;                           // ~2.4 s at 4.2 cycles per instruction
    if (run<0>("s_sleep_(baseline)", d_out, n_cu, 14000, 800)) return 1;
    if (run<1>("v_mad_i64_i32_29bit_operands", d_out, n_cu, it, per)) return 1;
    if (run<7>("v_mad_i64_i32_8bit_operands", d_out, n_cu, it, per)) return 1;
    if (run<8>("v_mad_u64_u32_32bit_operands", d_out, n_cu, it, per)) return 1;
    if (run<6>("v_mul_lo_u32", d_out, n_cu, it, per)) return 1;
    if (run<2>("v_add_u32/v_sub_u32", d_out, n_cu, it, per)) return 1;
    if (run<3>("v_lshl_add_u64", d_out, n_cu, it, per)) return 1;
    if (run<5>("v_bfe_i32/v_ashrrev_i64", d_out, n_cu, it, per)) return 1;
    if (run<4>("v_accvgpr_read/write", d_out, n_cu, it, per)) return 1;
    return 0;
}
