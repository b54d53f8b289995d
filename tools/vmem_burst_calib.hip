// vmem_burst_calib.hip -- does a burst of global loads stall a lone wave?  One wave per SIMD (as the pairing kernels), a loop
// body of 40 global_load_dwordx4 (each 64 lanes x 16 B = one contiguous KiB, results never read inside the loop) and 4000
// VALU instructions (the 4 mads + 1 add mix), arranged either as ONE burst of 40 loads followed by the VALU work or as one
// load every 100 VALU instructions.  Same instruction count either way.
// Build: hipcc --offload-arch=gfx950 -O2 tools/vmem_burst_calib.hip -o build/vmem_burst_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

#define VALU10 \
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
#define LOAD "global_load_dwordx4 v[40:43], v8, s[16:17]\n\t s_add_u32 s16, s16, 0x100000\n\t s_addc_u32 s17, s17, 0\n\t"
#define STORE "global_store_dwordx4 v8, v[44:47], s[16:17]\n\t s_add_u32 s16, s16, 0x100000\n\t s_addc_u32 s17, s17, 0\n\t"

// MODE 0: burst of 40 loads, then 400 x VALU10.  MODE 1: 40 x (1 load + 10 x VALU10).  MODE 2: no loads.  MODE 3/4: stores.
template <int MODE>
__global__ void __launch_bounds__(256) k_burst(uint32_t* out, const uint4* buf, int iters) {
    extern __shared__ uint32_t lds_pad[];
    if (iters < 0) lds_pad[threadIdx.x] = 1;
    uint32_t r;
    const uint4* base = buf + (size_t)blockIdx.x * 4096 * 16;          // 1 MiB per workgroup and step, 64 KiB per ... see host
    asm volatile(
        "v_mov_b32 v2, %1\n\t v_add_u32 v3, 3, v2\n\t v_add_u32 v4, 5, v2\n\t v_add_u32 v5, 7, v2\n\t v_add_u32 v6, 11, v2\n\t"
        "v_mov_b32 v30, 0\n\t v_mov_b32 v31, 0\n\t v_lshlrev_b32 v8, 4, %1\n\t"
        "v_mov_b32 v44, 1\n\t v_mov_b32 v45, 2\n\t v_mov_b32 v46, 3\n\t v_mov_b32 v47, 4\n\t"
        "s_mov_b32 s12, %2\n\t"
        "1:\n\t"
        "s_mov_b64 s[16:17], %3\n\t"
        ".if %c4 == 0\n\t .rept 40\n\t" LOAD ".endr\n\t .rept 400\n\t" VALU10 ".endr\n\t .endif\n\t"
        ".if %c4 == 1\n\t .rept 40\n\t" LOAD ".rept 10\n\t" VALU10 ".endr\n\t .endr\n\t .endif\n\t"
        ".if %c4 == 2\n\t .rept 400\n\t" VALU10 ".endr\n\t .endif\n\t"
        ".if %c4 == 3\n\t .rept 40\n\t" STORE ".endr\n\t .rept 400\n\t" VALU10 ".endr\n\t .endif\n\t"
        ".if %c4 == 4\n\t .rept 40\n\t" STORE ".rept 10\n\t" VALU10 ".endr\n\t .endr\n\t .endif\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        "s_sub_u32 s12, s12, 1\n\t"
        "s_cmp_lg_u32 s12, 0\n\t"
        "s_cbranch_scc1 1b\n\t"
        "v_add_u32 %0, v30, v31\n\t v_add_u32 %0, %0, v10\n\t v_add_u32 %0, %0, v24\n\t v_add_u32 %0, %0, v40\n\t"
        : "=v"(r) : "v"(threadIdx.x), "s"(iters), "s"(base), "i"(MODE)
        : "v2", "v3", "v4", "v5", "v6", "v8", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23",
          "v24", "v25", "v30", "v31", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "s10", "s11", "s12", "s16", "s17", "scc", "memory");
    if (r == 0x12345) out[0] = r;
}

template <int MODE>
int run(const char* tag, uint32_t* d_out, const uint4* buf, int n_cu) {
    int iters = 400;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute((const void*)k_burst<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 147456));
    std::vector<float> ms;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_burst<MODE>, dim3(n_cu), dim3(256), 147456, 0, d_out, buf, iters);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1));
        if (rep) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    double m = ms[ms.size() / 2];
    printf("%-44s ms %8.3f  => %8.1f cycles per loop body at 2.2 GHz (4000 VALU = 16000 issue cycles)\n", tag, m, m * 1e-3 * 2.2e9 / iters);
    return 0;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    int n_cu = prop.multiProcessorCount;
    uint32_t* d_out;
    CK(hipMalloc(&d_out, 4096));
    uint4* buf;                                                         // 40 steps x 1 MiB stride + 256 workgroups x 64 KiB
    size_t bytes = (size_t)41 * (1 << 20) + (size_t)n_cu * 4096 * 16 * sizeof(uint4);
    CK(hipMalloc(&buf, bytes));
    CK(hipMemset(buf, 1, bytes));
    printf("%d CUs, one 256-thread workgroup per CU (one wave per SIMD, the four waves of a CU in lock step)\n", n_cu);
    if (run<2>("no memory instructions", d_out, buf, n_cu)) return 1;
    if (run<0>("burst of 40 loads, then 4000 VALU", d_out, buf, n_cu)) return 1;
    if (run<1>("one load every 100 VALU", d_out, buf, n_cu)) return 1;
    if (run<3>("burst of 40 stores, then 4000 VALU", d_out, buf, n_cu)) return 1;
    if (run<4>("one store every 100 VALU", d_out, buf, n_cu)) return 1;
    return 0;
}
