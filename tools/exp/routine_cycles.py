#!/usr/bin/env python3
"""routine_cycles.py out.hip [variant ...] -- shader cycles of the generated LEAF ROUTINES as the kernels run them: one wave per SIMD, all
CUs busy, the routine body (tools/kgen4.py, aligned like the kernels' code) in a loop, s_memtime around it.  Prints cycles per call
and per instruction (4.000 = one instruction per issue slot); anything above says which routine loses cycles to something other
than its instruction count.  Variants: generator switches as NAME=ENV1=V,ENV2=V (evaluated in a fresh import).  Run the built
program on the GPU box (gpurun)."""
import importlib
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))

ROUTINES = ["mul", "sqr", "mulfq", "mul3", "mul6", "dblstep", "addstep", "sqr4c", "redn", "mulxi", "fqsqr"]


def bodies(env):
    for k in list(os.environ):
        if k.startswith("KGEN_"):
            del os.environ[k]
    os.environ.update(env)
    for m in ("kgen4", "asmcore"):
        sys.modules.pop(m, None)
    K4 = importlib.import_module("kgen4")
    AC = importlib.import_module("asmcore")
    out = {}
    for n in ROUTINES:
        e = AC.Emitter()
        K4.routine_body(e, n)
        out[n] = AC.align_code(e.finalize())
    return out, K4


def main():
    out_path = sys.argv[1]
    variants = [("shipped", {})]
    for a in sys.argv[2:]:
        name, rest = a.split("=", 1)
        variants.append((name, dict(kv.split("=") for kv in rest.split(","))))
    src = ['#include <hip/hip_runtime.h>\n#include <cstdio>\n#include <cstdint>\n#include <vector>\n#include <algorithm>\n']
    clob = ", ".join([f'"v{i}"' for i in range(248)] + [f'"a{i}"' for i in range(256)] + [f'"s{i}"' for i in range(36, 64)] + ['"vcc"', '"scc"', '"memory"'])
    kernels = []
    for vname, env in variants:
        bd, K4 = bodies(env)
        init = [f"v_mov_b32 v{i}, 0x{(0x00234567 * (i + 3)) & 0x0fffffff:x}" for i in range(248)]
        init += [f"s_mov_b32 s{K4.S_P + i}, 0x{K4.P_L[i] & 0xffffffff:x}" for i in range(K4.NL)]
        init += [f"s_mov_b32 s{K4.S_N0}, 0x{K4.N0P:x}", f"s_mov_b32 s{K4.S_REDN}, 0x{K4.REDN_C & 0xffffffff:x}", f"s_mov_b32 s{K4.S_HALF}, 0x10000000", f"s_mov_b32 s{K4.S_HALF + 1}, 0", f"s_mov_b32 s{K4.S_M30}, 0x{(-30) & 0xffffffff:x}"]
        for n in ROUTINES:
            kid = f"k_{vname}_{n}"
            body = " \\\n".join('"%s\\n"' % l for l in [".p2align 3"] + bd[n])
            n_ins = len([l for l in bd[n] if not l.endswith(":")])
            kernels.append((kid, vname, n, n_ins))
            src.append(f'''__global__ void __launch_bounds__(256) {kid}(uint64_t* out, int iters) {{
    extern __shared__ uint32_t lds_pad[];
    if (iters < 0) lds_pad[threadIdx.x] = 1;
    asm volatile({" ".join('"' + l + chr(92) + 'n"' for l in init)} ::: {clob});
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) asm volatile({body} ::: {clob});
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    uint32_t sink;
    asm volatile("v_xor_b32 %0, v0, v1\\n v_xor_b32 %0, %0, v20\\n v_xor_b32 %0, %0, v90" : "=v"(sink) :: {clob});
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = (t1 - t0) + (sink == 0x12345678u ? 1 : 0);
}}
''')
    src.append('''template <typename K> static double timeit(K kern, uint64_t* dbuf, int iters) {
    int threads = 256, blocks = 512;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 100 * 1024, 0, dbuf, iters);
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    std::vector<uint64_t> h((size_t)blocks * threads);
    (void)hipMemcpy(h.data(), dbuf, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> cyc;
    for (size_t w = 0; w < h.size() / 64; ++w) cyc.push_back((double)h[w * 64]);
    std::sort(cyc.begin(), cyc.end());
    return cyc[cyc.size() / 2];
}
int main() {
    uint64_t* dbuf; if (hipMalloc(&dbuf, (size_t)512 * 256 * 8) != hipSuccess) return 1;
''')
    for kid, vname, n, n_ins in kernels:
        src.append(f'    {{ double a = timeit({kid}, dbuf, 64), b = timeit({kid}, dbuf, 192); double c = (b - a) / 128.0;\n'
                   f'      printf("%-10s %-9s %5d instr  %9.1f cycles  %6.3f cycles/instr\\n", "{vname}", "{n}", {n_ins}, c, c / {n_ins}.0); fflush(stdout); }}\n')
    src.append("    return 0;\n}\n")
    open(out_path, "w").write("".join(src))


if __name__ == "__main__":
    main()
