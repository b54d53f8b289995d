#!/usr/bin/env python3
"""l1_bench.py -- time straight-line gfx950 instruction sequences at 1 wave per SIMD (the pairing kernels' occupancy).

    python tools/exp/l1_bench.py variants.json out.hip      (then hipcc --offload-arch=gfx950 -O2 out.hip -o out; run on the GPU)

variants.json: {"name": ["asm line", ...], ...}: physical registers v0..v119, a0..a63, SGPRs s36..s59 (s36..s45 = modulus limbs in
radix 2^27, s46 = -p^-1 mod 2^27 like the v3 kernels), vcc.  Every sequence is timed unrolled x2 and x4 inside a loop; the difference
removes the loop overhead.  Prints cycles per sequence and per instruction."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from asmcore import align_code  # noqa: E402

P_L = [0x7cfd47, 0x1842c36, 0x2e5346f, 0x68ddb52, 0x455f06d, 0x360ab71, 0x7316de1, 0x4a028d7, 0x6131a02, 0x30644e7 >> 0]


def esc(l):
    return l.replace("\\", "\\\\").replace('"', '\\"')


def main():
    variants = json.load(open(sys.argv[1]))
    out = []
    out.append('#include <hip/hip_runtime.h>\n#include <cstdio>\n#include <cstdint>\n#include <vector>\n#include <algorithm>\n')
    out.append('#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)\n')
    clob = ", ".join([f'"v{i}"' for i in range(120)] + [f'"a{i}"' for i in range(64)] + [f'"s{i}"' for i in range(36, 60)] + ['"vcc"', '"scc"', '"memory"'])
    init = [f"v_mov_b32 v{i}, 0x{(0x1234567 * (i + 3)) & 0x7ffffff:x}" for i in range(120)]
    init += [f"v_accvgpr_write_b32 a{i}, v{i}" for i in range(64)]
    init += [f"s_mov_b32 s{36 + i}, 0x{(0x2345671 * (i + 1)) & 0x7ffffff:x}" for i in range(10)] + ["s_mov_b32 s46, 0x5e4c2b9"]
    names = list(variants)
    for vi, name in enumerate(names):
        lines = variants[name]
        if not name.startswith("raw:"):
            lines = [".p2align 3"] + align_code(lines)         # 8-byte instructions 8-byte aligned, as in the kernels
        body = " \\\n".join(f'"{esc(l)}\\n"' for l in lines)
        out.append(f"#define BODY{vi} {body if body else chr(34) + chr(34)}\n")
        for rep, tag in ((2, "a"), (4, "b")):
            out.append(f'''__global__ void __launch_bounds__(256) k{vi}{tag}(uint64_t* out, int iters) {{
    extern __shared__ uint32_t lds_pad[];
    if (iters < 0) lds_pad[threadIdx.x] = 1;
    asm volatile({" ".join('"' + l + chr(92) + 'n"' for l in init)} ::: {clob});
    asm volatile("v_lshlrev_b32 v119, 4, %0\\n v_lshlrev_b32 v118, 2, %0\\n v_lshlrev_b32 v117, 3, %0" :: "v"(threadIdx.x) : "v117", "v118", "v119");
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    if ({1 if name.startswith("solo:") else 0} && threadIdx.x >= 64) iters = 0;      // "solo:" variants: one wave per CU runs, the other three idle
    for (int it = 0; it < iters; ++it) asm volatile({" ".join(["BODY%d" % vi] * rep)} ::: {clob});
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    uint32_t sink;
    asm volatile("v_xor_b32 %0, v0, v1\\n v_xor_b32 %0, %0, v20\\n v_xor_b32 %0, %0, v40" : "=v"(sink) :: {clob});
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = (t1 - t0) + (sink == 0x12345678u ? 1 : 0);
}}
''')
    out.append('''template <typename K> static double timeit(K kern, uint64_t* dbuf, int iters) {
    int threads = 256, blocks = 512;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 100 * 1024, 0, dbuf, iters / 4);
    (void)hipDeviceSynchronize();
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 100 * 1024, 0, dbuf, iters);
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    std::vector<uint64_t> h((size_t)blocks * threads);
    (void)hipMemcpy(h.data(), dbuf, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> cyc;
    for (size_t w = 0; w < h.size() / 64; ++w) cyc.push_back((double)h[w * 64]);
    std::sort(cyc.begin(), cyc.end());
    return cyc[cyc.size() * 7 / 8] / (double)iters;          // upper octile: idle waves of "solo:" variants report ~0
}
int main() {
    uint64_t* dbuf; CK(hipMalloc(&dbuf, (size_t)512 * 256 * 8));
    int iters = 2048;
''')
    for vi, name in enumerate(names):
        n = len([l for l in variants[name] if l.strip() and not l.strip().endswith(":")])
        out.append(f'    {{ double a = timeit(k{vi}a, dbuf, iters), b = timeit(k{vi}b, dbuf, iters); double c = (b - a) / 2.0;\n'
                   f'      printf("%-44s %5d instr  %9.1f cycles  %6.3f cycles/instr\\n", "{name}", {n}, c, c / {max(n, 1)}.0); fflush(stdout); }}\n')
    out.append("    CK(hipFree(dbuf));\n    return 0;\n}\n")
    open(sys.argv[2], "w").write("".join(out))


if __name__ == "__main__":
    main()
