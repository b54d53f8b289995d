// lds_bank_calib.hip -- how gfx950's LDS serialises ds_read_b128 / ds_read_b64 / ds_write_b128 of one wave whose lanes name
// arbitrary addresses: which lanes are served together, and what makes two of them collide.  The lane-cooperative kernels
// (tools/cvm_kernel.py) read their operands with per-lane slot addresses; tools/cvm.py assigns the slots.
//
// One wave per workgroup, one workgroup; every lane takes its byte address from a table; the kernel times REPS x 8 independent
// reads with s_memtime.  Patterns (host side): linear, strides, and "only the lanes of one candidate group collide" tables that
// tell the grouping apart.
// Build: hipcc --offload-arch=gfx950 -O2 tools/lds_bank_calib.hip -o tools/lds_bank_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>
#include <functional>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void __launch_bounds__(512) k_lds(const uint32_t* addr_tab, uint64_t* out, int reps) {
    extern __shared__ uint32_t lds[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = i;
    __syncthreads();
    uint32_t a = addr_tab[threadIdx.x & 63];
    uint32_t acc = 0;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; r++) {
        if (KIND == 0) {
            u32x4 x0, x1, x2, x3;
            asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:16384\n ds_read_b128 %2, %4 offset:32768\n ds_read_b128 %3, %4 offset:49152\n"
                         "ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:16384\n ds_read_b128 %2, %4 offset:32768\n ds_read_b128 %3, %4 offset:49152\n s_waitcnt lgkmcnt(0)"
                         : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3) : "v"(a));
            acc += x0.x ^ x1.y ^ x2.z ^ x3.w;
        } else if (KIND == 1) {
            u32x2 x0, x1, x2, x3;
            asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:16384\n ds_read_b64 %2, %4 offset:32768\n ds_read_b64 %3, %4 offset:49152\n"
                         "ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:16384\n ds_read_b64 %2, %4 offset:32768\n ds_read_b64 %3, %4 offset:49152\n s_waitcnt lgkmcnt(0)"
                         : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3) : "v"(a));
            acc += x0.x ^ x1.y ^ x2.x ^ x3.y;
        } else {
            u32x4 x = {acc, (uint32_t)r, a, 7u};
            asm volatile("ds_write_b128 %0, %1\n ds_write_b128 %0, %1 offset:16384\n ds_write_b128 %0, %1 offset:32768\n ds_write_b128 %0, %1 offset:49152\n"
                         "ds_write_b128 %0, %1\n ds_write_b128 %0, %1 offset:16384\n ds_write_b128 %0, %1 offset:32768\n ds_write_b128 %0, %1 offset:49152\n s_waitcnt lgkmcnt(0)"
                         : : "v"(a), "v"(x) : "memory");
        }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = acc; }
}

int main() {
    uint32_t* dtab; uint64_t* dout;
    CK(hipMalloc(&dtab, 64 * 4)); CK(hipMalloc(&dout, 16));
    const int reps = 2000;
    CK(hipFuncSetAttribute((const void*)k_lds<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipFuncSetAttribute((const void*)k_lds<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipFuncSetAttribute((const void*)k_lds<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    struct Pat { std::string name; std::function<uint32_t(int)> f; };
    std::vector<Pat> pats = {
        {"linear 16 B (lane*16)", [](int l) { return (uint32_t)l * 16; }},
        {"all lanes one address", [](int) { return 0u; }},
        {"stride 32", [](int l) { return (uint32_t)l * 32; }},
        {"stride 48 (consecutive slots)", [](int l) { return (uint32_t)l * 48; }},
        {"stride 64", [](int l) { return (uint32_t)l * 64; }},
        {"stride 128", [](int l) { return (uint32_t)l * 128; }},
        {"stride 256 (all lanes one bank group?)", [](int l) { return (uint32_t)(l * 256) % 16384; }},
        {"stride 512", [](int l) { return (uint32_t)(l * 512) % 16384 + (l / 32) * 16; }},
        // only the lanes of an aligned group of G collide (same 16-byte column of different 256-byte rows); groups use different columns
        {"collide within aligned 2", [](int l) { return (uint32_t)((l / 2) % 16) * 16 + (l % 2) * 256 + (l / 32) * 512; }},
        {"collide within aligned 4", [](int l) { return (uint32_t)((l / 4) % 16) * 16 + (l % 4) * 256; }},
        {"collide within aligned 8", [](int l) { return (uint32_t)(l / 8) * 16 + (l % 8) * 256; }},
        {"collide within aligned 16", [](int l) { return (uint32_t)(l / 16) * 16 + (l % 16) * 256; }},
        {"collide lanes l, l+8 (16 apart free)", [](int l) { return (uint32_t)(l % 8) * 16 + ((l / 8) % 2) * 256 + (l / 16) * 128; }},
        {"collide lanes l, l+16", [](int l) { return (uint32_t)(l % 16) * 16 + ((l / 16) % 2) * 256 + (l / 32) * 512; }},
        {"collide lanes l, l+32", [](int l) { return (uint32_t)(l % 32) * 16 + (l / 32) * 512; }},
        {"collide lanes l, l+32 (row 256 apart)", [](int l) { return (uint32_t)(l % 16) * 16 + ((l / 16) % 2) * 1024 * 0 + (l / 32) * 256 + ((l / 16) % 2) * 2048; }},
        {"same address pairs (l, l+1) bcast", [](int l) { return (uint32_t)(l / 2) * 16; }},
        {"same address quads bcast", [](int l) { return (uint32_t)(l / 4) * 16; }},
        {"128-byte period test: l*16 + (l/8)*128", [](int l) { return (uint32_t)(l % 8) * 16 + (l / 8) * 128; }},
        {"random slots x48 (seed 1)", [](int l) { uint32_t x = (uint32_t)l * 2654435761u + 12345u; x ^= x >> 13; return (x % 271) * 48; }},
        {"random slots x48 (seed 2)", [](int l) { uint32_t x = (uint32_t)l * 40503u * 2654435761u + 99u; x ^= x >> 11; return (x % 271) * 48; }},
        {"random slots x64", [](int l) { uint32_t x = (uint32_t)l * 2654435761u + 12345u; x ^= x >> 13; return (x % 200) * 64; }},
        {"16 lanes x 4 groups, slots 0..15 x48 per group (+13008 B per group)", [](int l) { return (uint32_t)((l % 16) * 48 + (l / 16) * 13008) % 16384; }},
    };
    for (int waves : {1, 2, 4, 8})
    for (int kind = 0; kind < 3; kind++) {
        printf("== %s, %d wave(s) in the workgroup (one CU): cycles per instruction of wave 0\n", kind == 0 ? "ds_read_b128" : kind == 1 ? "ds_read_b64" : "ds_write_b128", waves);
        for (auto& p : pats) {
            if (waves > 1 && p.name.find("linear") == std::string::npos && p.name.find("random slots x48 (seed 1)") == std::string::npos && p.name.find("stride 256") == std::string::npos) continue;
            uint32_t tab[64];
            for (int l = 0; l < 64; l++) tab[l] = p.f(l) & ~15u;
            CK(hipMemcpy(dtab, tab, sizeof(tab), hipMemcpyHostToDevice));
            for (int w = 0; w < 2; w++) {
                if (kind == 0) hipLaunchKernelGGL(k_lds<0>, dim3(1), dim3(64 * waves), 65536, 0, dtab, dout, reps);
                else if (kind == 1) hipLaunchKernelGGL(k_lds<1>, dim3(1), dim3(64 * waves), 65536, 0, dtab, dout, reps);
                else hipLaunchKernelGGL(k_lds<2>, dim3(1), dim3(64 * waves), 65536, 0, dtab, dout, reps);
                CK(hipDeviceSynchronize());
            }
            uint64_t h[2]; CK(hipMemcpy(h, dout, 16, hipMemcpyDeviceToHost));
            printf("  %-70s %7.2f cycles per instruction\n", p.name.c_str(), (double)h[0] / (reps * 8.0));
        }
    }
    return 0;
}
