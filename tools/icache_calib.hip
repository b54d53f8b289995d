// icache_calib.hip -- what does straight-line code cost on gfx950 once it no longer fits the 64 KB instruction cache
// (shared by two CUs)?  One wave per SIMD (512 registers requested through the LDS/launch bounds as in the pairing
// kernels), every wave runs a loop whose body is a straight-line block of S bytes of the pairing kernel's instruction mix
// (runs of v_mad_i64_i32 with a one-dword filler every 8th slot); the number of iterations is scaled so that every
// configuration executes the same number of instructions.
// Build: hipcc --offload-arch=gfx950 -O2 tools/icache_calib.hip -o build/icache_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

// one group = 8 mads (8 bytes each) + 1 add (4 bytes) + 1 add (4 bytes) = 72 bytes + 8 = 80 bytes, 10 instructions
#define GROUP \
    "v_mad_i64_i32 v[10:11], s[10:11], v2, v3, v[10:11]\n\t" \
    "v_mad_i64_i32 v[12:13], s[10:11], v2, v4, v[12:13]\n\t" \
    "v_mad_i64_i32 v[14:15], s[10:11], v2, v5, v[14:15]\n\t" \
    "v_mad_i64_i32 v[16:17], s[10:11], v2, v6, v[16:17]\n\t" \
    "v_add_u32_e32 v30, v30, v2\n\t" \
    "v_mad_i64_i32 v[18:19], s[10:11], v3, v3, v[18:19]\n\t" \
    "v_mad_i64_i32 v[20:21], s[10:11], v3, v4, v[20:21]\n\t" \
    "v_mad_i64_i32 v[22:23], s[10:11], v3, v5, v[22:23]\n\t" \
    "v_mad_i64_i32 v[24:25], s[10:11], v3, v6, v[24:25]\n\t" \
    "v_add_u32_e32 v31, v31, v3\n\t"

template <int GROUPS>
__global__ void __launch_bounds__(256) k_code(uint32_t* out, int iters) {
    extern __shared__ uint32_t lds_pad[];
    if (iters < 0) lds_pad[threadIdx.x] = 1;
    uint32_t r;
    asm volatile(
        "v_mov_b32 v2, %1\n\t v_add_u32 v3, 3, v2\n\t v_add_u32 v4, 5, v2\n\t v_add_u32 v5, 7, v2\n\t v_add_u32 v6, 11, v2\n\t"
        "v_mov_b32 v30, 0\n\t v_mov_b32 v31, 0\n\t"
        "s_mov_b32 s12, %2\n\t"
        "s_getpc_b64 s[14:15]\n\t"              // loop head (s_setpc reaches any distance, s_cbranch only 128 KB)
        ".rept %c3\n\t" GROUP ".endr\n\t"
        "s_sub_u32 s12, s12, 1\n\t"
        "s_cmp_lg_u32 s12, 0\n\t"
        "s_cbranch_scc0 2f\n\t"
        "s_setpc_b64 s[14:15]\n\t"
        "2:\n\t"
        "v_add_u32 %0, v30, v31\n\t v_add_u32 %0, %0, v10\n\t v_add_u32 %0, %0, v24\n\t"
        : "=v"(r) : "v"(threadIdx.x), "s"(iters), "i"(GROUPS)
        : "v2", "v3", "v4", "v5", "v6", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23",
          "v24", "v25", "v30", "v31", "s10", "s11", "s12", "s14", "s15", "scc");
    if (r == 0x12345) out[0] = r;
}

template <int GROUPS>
int run(const char* tag, uint32_t* d_out, int n_cu, long total_groups) {
    int iters = (int)(total_groups / GROUPS);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute((const void*)k_code<GROUPS>, hipFuncAttributeMaxDynamicSharedMemorySize, 147456));
    std::vector<float> ms;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_code<GROUPS>, dim3(n_cu), dim3(256), 147456, 0, d_out, iters);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1));
        if (rep) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    double instr = (double)iters * GROUPS * 10;
    printf("%-10s body %7.1f KB  iters %7d  ms %8.3f  ns/instr/wave %7.4f  => cycles/instr at 2.4 GHz %6.3f\n", tag, GROUPS * 80 / 1024.0, iters,
           ms[ms.size() / 2], ms[ms.size() / 2] * 1e6 / instr, ms[ms.size() / 2] * 1e6 / instr * 2.4);
    return 0;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    int n_cu = prop.multiProcessorCount;
    uint32_t* d_out;
    CK(hipMalloc(&d_out, 4096));
    long total = 1L << 21;            // groups per wave: 21 M instructions, ~40 ms
    printf("device %s, %d CUs, one 256-thread workgroup per CU (one wave per SIMD)\n", prop.name, n_cu);
    if (run<100>("8KB", d_out, n_cu, total)) return 1;
    if (run<200>("16KB", d_out, n_cu, total)) return 1;
    if (run<400>("31KB", d_out, n_cu, total)) return 1;
    if (run<600>("47KB", d_out, n_cu, total)) return 1;
    if (run<750>("59KB", d_out, n_cu, total)) return 1;
    if (run<820>("64KB", d_out, n_cu, total)) return 1;
    if (run<1000>("78KB", d_out, n_cu, total)) return 1;
    if (run<1300>("102KB", d_out, n_cu, total)) return 1;
    if (run<1650>("129KB", d_out, n_cu, total)) return 1;
    if (run<3300>("258KB", d_out, n_cu, total)) return 1;
    if (run<6600>("516KB", d_out, n_cu, total)) return 1;
    if (run<13200>("1031KB", d_out, n_cu, total)) return 1;
    return 0;
}
