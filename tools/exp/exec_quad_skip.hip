// exec_quad_skip.hip -- does a wave64 VALU instruction cost fewer cycles when only 16 (or 32) of its lanes are active?
// (If the SIMD skipped the 16-lane passes whose lanes are all masked off, a one-pairing launch -- one active lane -- would run up to
// four times faster for free.)  One wave per SIMD, a dependent chain of v_mad_i64_i32; time per instruction for exec = 64 / 32 / 16 / 1 lanes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void __launch_bounds__(64) k(uint64_t mask, int iters, uint64_t* out) {
    uint64_t t0, t1;
    uint32_t a = threadIdx.x + 3, b = threadIdx.x * 7 + 1;
    uint64_t acc = 0;
    asm volatile("s_mov_b64 exec, %0" :: "s"(mask));
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int i = 0; i < iters; i++) {
        asm volatile(
            "v_mad_i64_i32 %0, vcc, %1, %2, %0\n v_mad_i64_i32 %0, vcc, %1, %2, %0\n v_mad_i64_i32 %0, vcc, %1, %2, %0\n v_mad_i64_i32 %0, vcc, %1, %2, %0\n"
            "v_mad_i64_i32 %0, vcc, %1, %2, %0\n v_mad_i64_i32 %0, vcc, %1, %2, %0\n v_mad_i64_i32 %0, vcc, %1, %2, %0\n v_mad_i64_i32 %0, vcc, %1, %2, %0\n"
            "v_mad_i64_i32 %0, vcc, %1, %2, %0\n v_mad_i64_i32 %0, vcc, %1, %2, %0\n v_mad_i64_i32 %0, vcc, %1, %2, %0\n v_mad_i64_i32 %0, vcc, %1, %2, %0\n"
            "v_mad_i64_i32 %0, vcc, %1, %2, %0\n v_mad_i64_i32 %0, vcc, %1, %2, %0\n v_mad_i64_i32 %0, vcc, %1, %2, %0\n v_mad_i64_i32 %0, vcc, %1, %2, %0\n"
            : "+v"(acc) : "v"(a), "v"(b) : "vcc");
    }
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    asm volatile("s_mov_b64 exec, -1");
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = acc; }
}
int main() {
    uint64_t* d; hipMalloc(&d, 16 * 1024);
    const int iters = 20000;
    struct { const char* name; uint64_t m; } cases[] = {{"64 lanes", ~0ull}, {"32 lanes (low half)", 0xFFFFFFFFull}, {"16 lanes", 0xFFFFull}, {"1 lane", 1ull},
                                                        {"16 lanes (one per quarter: 0x000F000F000F000F)", 0x000F000F000F000Full}};
    for (auto& c : cases) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, c.m, iters, d);
        hipDeviceSynchronize();
        uint64_t h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("%-50s %8.3f shader cycles per v_mad_i64_i32 (one wave, dependent chain)\n", c.name, (double)h[0] / (16.0 * iters));
    }
    return 0;
}
