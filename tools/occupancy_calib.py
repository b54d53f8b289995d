#!/usr/bin/env python3
"""occupancy_calib.py out.hip -- what do the pairing kernels' instruction classes cost when a SIMD holds MORE THAN ONE wave?

The shipped kernels run one wave per SIMD (512 registers each) and every VALU instruction of a lone wave issues in 4 cycles
(profiles/r04_class_cycles.txt).  A wave64 instruction needs only 2 passes through the SIMD-32 for the classes that are not
multiplier-bound, so a SECOND wave on the SIMD may fill the other two cycles.  This generator emits one HIP program that prices that
operating point (run the built binary on the GPU box):

  (a) every instruction class of profiles/r05_instr_histogram.json alone at 1 / 2 / 4 waves per SIMD;
  (b) HETEROGENEOUS pairs: waves 0..3 of a workgroup (one per SIMD) run a pure v_mad_i64_i32 stream, waves 4..7 (their SIMD partners)
      run another class / the kernels' real non-multiply mix;
  (c) the generated LEAF ROUTINES themselves (tools/kgen4.py: mul, mul3, sqr4c, mul6, redn -- they touch VGPRs only, so two waves
      of 256 registers fit a SIMD) at 1 and 2 waves per SIMD.

Every wave runs its stream until a deadline in shader cycles (s_memtime) and counts its iterations, so streams of different speed
share the SIMD for the whole measurement.  Printed per test: instructions per wave, cycles per instruction per WAVE and per SIMD
(the aggregate issue cost: 4.0 = no gain over the shipped operating point), the in-kernel clock (s_memtime / s_memrealtime), package
power (hwmon, when readable) and the SIMD ids seen (checks the wave -> SIMD assumption).  Reference cost centres:
/root/reference/src/final_exp_native.rs:56-84,130-169, miller_loop_native.rs:46-96."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

N_BODY = 768          # instructions per stream body
INNER = 8             # bodies per deadline check
NV = 48               # v0..v31 destinations / accumulators, v32..v47 operands (a 1024-thread workgroup gets 64 VGPRs + 64 AGPRs)


def rot(n, f):
    return [f(k) for k in range(n)]


def stream_mad(n=N_BODY):
    return rot(n, lambda k: f"v_mad_i64_i32 v[{2 * (k % 16)}:{2 * (k % 16) + 1}], s[10:11], v{32 + k % 8}, v{40 + (k * 3) % 8}, v[{2 * (k % 16)}:{2 * (k % 16) + 1}]")


CLASSES = {
    "v_mad_i64_i32": lambda k: f"v_mad_i64_i32 v[{2 * (k % 16)}:{2 * (k % 16) + 1}], s[10:11], v{32 + k % 8}, v{40 + (k * 3) % 8}, v[{2 * (k % 16)}:{2 * (k % 16) + 1}]",
    "v_bfe_i32": lambda k: f"v_bfe_i32 v{k % 32}, v{32 + k % 16}, 0, 29",
    "v_ashrrev_i32": lambda k: f"v_ashrrev_i32_e32 v{k % 32}, 29, v{32 + k % 16}",
    "v_ashrrev_i64": lambda k: f"v_ashrrev_i64 v[{2 * (k % 16)}:{2 * (k % 16) + 1}], 29, v[{32 + 2 * (k % 8)}:{33 + 2 * (k % 8)}]",
    "v_and_b32 (literal mask)": lambda k: f"v_and_b32_e32 v{k % 32}, 0x1fffffff, v{32 + k % 16}",
    "v_and_b32 (registers)": lambda k: f"v_and_b32_e32 v{k % 32}, v{32 + k % 8}, v{40 + (k * 3) % 8}",
    "v_add_u32": lambda k: f"v_add_u32_e32 v{k % 32}, v{32 + k % 8}, v{40 + (k * 3) % 8}",
    "v_sub_u32": lambda k: f"v_sub_u32_e32 v{k % 32}, v{32 + k % 8}, v{40 + (k * 3) % 8}",
    "v_add_u32 (accumulating)": lambda k: f"v_add_u32_e32 v{k % 32}, v{32 + k % 8}, v{k % 32}",
    "v_lshl_add_u32": lambda k: f"v_lshl_add_u32 v{k % 32}, v{32 + k % 8}, 3, v{40 + (k * 3) % 8}",
    "v_add3_u32": lambda k: f"v_add3_u32 v{k % 32}, v{32 + k % 8}, v{40 + (k * 3) % 8}, v{k % 32}",
    "v_lshl_add_u64": lambda k: f"v_lshl_add_u64 v[{2 * (k % 16)}:{2 * (k % 16) + 1}], v[{32 + 2 * (k % 8)}:{33 + 2 * (k % 8)}], 0, v[{2 * (k % 16)}:{2 * (k % 16) + 1}]",
    "v_sub_co + v_subb (64-bit)": lambda k: (f"v_sub_co_u32_e64 v{2 * ((k // 2) % 16)}, s[10:11], v{2 * ((k // 2) % 16)}, v{32 + (k // 2) % 8}" if k % 2 == 0 else
                                            f"v_subb_co_u32_e64 v{2 * ((k // 2) % 16) + 1}, s[10:11], v{2 * ((k // 2) % 16) + 1}, v{40 + (k // 2) % 8}, s[10:11]"),
    "v_accvgpr_read_b32": lambda k: f"v_accvgpr_read_b32 v{k % 32}, a{k % 16}",
    "v_accvgpr_write_b32": lambda k: f"v_accvgpr_write_b32 a{k % 16}, v{32 + k % 16}",
    "v_mul_lo_u32": lambda k: f"v_mul_lo_u32 v{k % 32}, v{32 + k % 8}, v{40 + (k * 3) % 8}",
    "v_mul_hi_i32": lambda k: f"v_mul_hi_i32 v{k % 32}, v{32 + k % 8}, v{40 + (k * 3) % 8}",
    "v_mov_b32 (e32)": lambda k: f"v_mov_b32_e32 v{k % 32}, v{32 + k % 16}",
    "v_mov_b32 (e64)": lambda k: f"v_mov_b32_e64 v{k % 32}, v{32 + k % 16}",
}

# the non-multiply instructions of one pairing by opcode (profiles/r05_instr_histogram.json by_class, split by the leaf routines' opcode
# counts): digit handling 370 k = bfe + ashr64, 64-bit combinations 219 k = lshl_add_u64 + sub_co/subb pairs, AGPR moves 143 k,
# limb-wise add / sub 142 k, mul_lo 64 k, mov 43 k  -> a pattern of 97
NONMUL_PATTERN = (["v_bfe_i32"] * 18 + ["v_ashrrev_i64"] * 19 + ["v_lshl_add_u64"] * 16 + ["v_sub_co + v_subb (64-bit)"] * 6 +
                  ["v_accvgpr_read_b32"] * 7 + ["v_accvgpr_write_b32"] * 7 + ["v_sub_u32"] * 14 + ["v_mul_lo_u32"] * 6 + ["v_mov_b32 (e32)"] * 4)


def stream_nonmul(n=N_BODY):
    # spread the classes evenly (largest remainder), keep each sub_co directly in front of its subb
    order = []
    counts = {}
    for c in NONMUL_PATTERN:
        counts[c] = counts.get(c, 0) + 1
    acc = {c: 0.0 for c in counts}
    total = len(NONMUL_PATTERN)
    pair = "v_sub_co + v_subb (64-bit)"
    while len(order) < n:
        for c in acc:
            acc[c] += counts[c] / total
        c = max(acc, key=lambda x: acc[x])
        acc[c] -= 1.0
        if c == pair:
            if len(order) + 2 > n:
                continue
            acc[c] -= 1.0
            order += [(c, 0), (c, 1)]
        else:
            order.append((c, None))
    out, idx = [], {}
    for c, half in order[:n]:
        k = idx.get(c, 0)
        if c == pair:
            out.append(CLASSES[c](2 * k + half))
            if half == 1:
                idx[c] = k + 1
        else:
            out.append(CLASSES[c](k))
            idx[c] = k + 1
    return out


def stream_kernel_mix(n=N_BODY):
    """71 % multiply-adds, 29 % the non-multiply mix, interleaved as the column sweeps do (runs of 5 - 9 mads, then the digit work)"""
    mads, non = stream_mad(n), stream_nonmul(n)
    out, im, inn = [], 0, 0
    while len(out) < n:
        for _ in range(7):
            out.append(mads[im % n]); im += 1
        for _ in range(3):
            out.append(non[inn % n]); inn += 1
    return out[:n]


def c_str(lines):
    return " ".join('"%s\\n"' % l for l in lines)


def main():
    out_path = sys.argv[1]
    import asmcore as AC
    import kgen4 as K4
    tests = []       # (name, body A, body B or None, waves per SIMD list, kind)
    bodies = {}

    def body(name, lines):
        if name not in bodies:
            bodies[name] = (len(bodies), [".p2align 3"] + AC.align_code(list(lines)), len(lines))
        return name

    for cname, f in CLASSES.items():
        body(cname, rot(N_BODY, f))
        tests.append((cname, cname, None, (1, 2, 4)))
    body("non-multiply mix", stream_nonmul())
    body("kernel mix (71 % mad)", stream_kernel_mix())
    tests.append(("non-multiply mix of the kernels", "non-multiply mix", None, (1, 2, 4)))
    tests.append(("kernel mix: 7 mads + 3 of the mix", "kernel mix (71 % mad)", None, (1, 2, 4)))
    for partner in ["v_mad_i64_i32", "v_bfe_i32", "v_ashrrev_i64", "v_sub_u32", "v_lshl_add_u64", "v_sub_co + v_subb (64-bit)", "v_accvgpr_read_b32",
                    "v_accvgpr_write_b32", "v_mul_lo_u32", "v_mov_b32 (e32)", "non-multiply mix"]:
        tests.append((f"PAIR mad | {partner}", "v_mad_i64_i32", partner, (2, 4)))

    # the generated leaf routines (VGPRs only)
    routines = ["mul", "mul3", "sqr4c", "mul6", "redn"]
    rbody = {}
    for n in routines:
        e = AC.Emitter()
        K4.routine_body(e, n)
        lines = AC.align_code(e.finalize())
        rbody[n] = ([".p2align 3"] + lines, len([l for l in lines if not l.endswith(":")]))

    src = ['// generated by tools/occupancy_calib.py -- do not edit\n#include <hip/hip_runtime.h>\n#include <cstdio>\n#include <cstdint>\n#include <cstdlib>\n#include <vector>\n'
           '#include <algorithm>\n#include <atomic>\n#include <thread>\n#include <chrono>\n#include <string>\n#include <dirent.h>\n#include <map>\n']
    clob_small = ", ".join([f'"v{i}"' for i in range(NV)] + [f'"a{i}"' for i in range(16)] + ['"s10"', '"s11"', '"vcc"', '"scc"', '"memory"'])
    init_small = [f"v_mov_b32 v{i}, 0x{(0x00234567 * (i + 3)) & 0x0fffffff:x}" for i in range(NV)] + [f"v_accvgpr_write_b32 a{i}, v{i}" for i in range(16)]
    for name, (bi, lines, n) in bodies.items():
        src.append(f"#define BODY{bi} {c_str(lines)}\n")
    src.append(f'''
struct Rec {{ uint32_t iters, hw_id, role, pad; uint64_t cycles, real; }};
#define RUN_STREAM(BODY) do {{ \\
    uint32_t n = 0; uint64_t t; \\
    do {{ _Pragma("unroll 1") for (int k = 0; k < {INNER}; ++k) asm volatile(BODY ::: {clob_small}); ++n; t = __builtin_amdgcn_s_memtime(); }} while (t - t0 < deadline); \\
    iters = n; t1 = t; }} while (0)
template <int A, int B> __global__ void __launch_bounds__(1024) k_class(Rec* out, uint64_t deadline) {{
    extern __shared__ uint32_t lds_pad[];
    if (deadline == 0) lds_pad[threadIdx.x] = 1;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t role = (B >= 0) ? ((wave >> 2) & 1u) : 0u;
    asm volatile({c_str(init_small)} ::: {clob_small});
    uint32_t hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    uint32_t iters = 0; uint64_t t1 = 0;
    const uint64_t r0 = __builtin_amdgcn_s_memrealtime();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
''')
    src.append("    if (role == 0) { switch (A) {\n")
    for name, (bi, lines, n) in bodies.items():
        src.append(f"        case {bi}: RUN_STREAM(BODY{bi}); break;\n")
    src.append("    } } else { switch (B) {\n")
    for name, (bi, lines, n) in bodies.items():
        src.append(f"        case {bi}: RUN_STREAM(BODY{bi}); break;\n")
    src.append(f'''    }} }}
    const uint64_t r1 = __builtin_amdgcn_s_memrealtime();
    uint32_t sink; asm volatile("v_xor_b32 %0, v0, v1\\n v_xor_b32 %0, %0, v20\\n v_xor_b32 %0, %0, v40" : "=v"(sink) :: {clob_small});
    if ((threadIdx.x & 63) == 0) {{
        Rec r; r.iters = iters + (sink == 0x12345678u ? 1 : 0); r.hw_id = hw; r.role = role; r.pad = 0; r.cycles = t1 - t0; r.real = r1 - r0;
        out[(size_t)blockIdx.x * (blockDim.x >> 6) + wave] = r;
    }}
}}
''')
    # leaf routines: full register file of a 256-register wave
    clob_big = ", ".join([f'"v{i}"' for i in range(248)] + [f'"s{i}"' for i in range(36, 64)] + ['"vcc"', '"scc"', '"memory"'])
    init_big = [f"v_mov_b32 v{i}, 0x{(0x00234567 * (i + 3)) & 0x0fffffff:x}" for i in range(248)]
    init_big += [f"s_mov_b32 s{K4.S_P + i}, 0x{K4.P_L[i] & 0xffffffff:x}" for i in range(K4.NL)]
    init_big += [f"s_mov_b32 s{K4.S_N0}, 0x{K4.N0P:x}", f"s_mov_b32 s{K4.S_REDN}, 0x{K4.REDN_C & 0xffffffff:x}", f"s_mov_b32 s{K4.S_HALF}, 0x10000000",
                 f"s_mov_b32 s{K4.S_HALF + 1}, 0", f"s_mov_b32 s{K4.S_M30}, 0x{(-30) & 0xffffffff:x}"]
    for n in routines:
        lines, n_ins = rbody[n]
        if any("accvgpr" in l for l in lines):
            continue
        src.append(f'''__global__ void __launch_bounds__(512) k_rt_{n}(Rec* out, uint64_t deadline) {{
    extern __shared__ uint32_t lds_pad[];
    if (deadline == 0) lds_pad[threadIdx.x] = 1;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    asm volatile({c_str(init_big)} ::: {clob_big});
    uint32_t hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    uint32_t n = 0; uint64_t t;
    const uint64_t r0 = __builtin_amdgcn_s_memrealtime();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    do {{ _Pragma("unroll 1") for (int k = 0; k < 4; ++k) asm volatile({c_str(lines)} ::: {clob_big}); ++n; t = __builtin_amdgcn_s_memtime(); }} while (t - t0 < deadline);
    const uint64_t r1 = __builtin_amdgcn_s_memrealtime();
    uint32_t sink; asm volatile("v_xor_b32 %0, v0, v1\\n v_xor_b32 %0, %0, v20\\n v_xor_b32 %0, %0, v90" : "=v"(sink) :: {clob_big});
    if ((threadIdx.x & 63) == 0) {{
        Rec r; r.iters = n + (sink == 0x12345678u ? 1 : 0); r.hw_id = hw; r.role = 0; r.pad = 0; r.cycles = t - t0; r.real = r1 - r0;
        out[(size_t)blockIdx.x * (blockDim.x >> 6) + wave] = r;
    }}
}}
''')
    src.append(r'''
static std::atomic<bool> g_sampling{false};
static std::vector<double> g_samples;
static std::string g_hwmon;
static void find_hwmon() {
    const char* env = getenv("CALIB_HWMON");
    if (env) { g_hwmon = env; return; }
    for (int card = 0; card < 16 && g_hwmon.empty(); ++card) {
        std::string base = "/sys/class/drm/card" + std::to_string(card) + "/device/hwmon";
        DIR* d = opendir(base.c_str());
        if (!d) continue;
        while (dirent* e = readdir(d)) {
            if (e->d_name[0] == '.') continue;
            for (const char* leaf : {"power1_average", "power1_input"}) {
                std::string p = base + "/" + e->d_name + "/" + leaf;
                FILE* f = fopen(p.c_str(), "r");
                if (f) { double v = 0; if (fscanf(f, "%lf", &v) == 1 && v > 0) g_hwmon = p; fclose(f); }
                if (!g_hwmon.empty()) break;
            }
        }
        closedir(d);
    }
}
static void sampler() {
    while (g_sampling) {
        if (!g_hwmon.empty()) { FILE* f = fopen(g_hwmon.c_str(), "r"); if (f) { double v = 0; if (fscanf(f, "%lf", &v) == 1) g_samples.push_back(v * 1e-6); fclose(f); } }
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
    }
}
struct Res { double inst_per_wave[2], cyc_per_inst_wave[2], cyc_per_inst_simd, clk, power, ms; int simd_ok; };
template <typename K> static int measure(K kern, int threads, Rec* dbuf, double seconds, const int n_inst[2], int inner, Res* res) {
    int n_cu = 256; hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, 0) == hipSuccess) n_cu = prop.multiProcessorCount;
    const int wpb = threads / 64;
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return 1;
    const uint64_t deadline = (uint64_t)(seconds * 2.4e9);
    hipLaunchKernelGGL(kern, dim3(n_cu), dim3(threads), 100 * 1024, 0, dbuf, deadline / 8 + 1);     // warm-up (code, clocks)
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    g_samples.clear(); g_sampling = true; std::thread th(sampler);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(n_cu), dim3(threads), 100 * 1024, 0, dbuf, deadline);
    (void)hipEventRecord(e1);
    if (hipDeviceSynchronize() != hipSuccess) { g_sampling = false; th.join(); return 1; }
    g_sampling = false; th.join();
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<Rec> h((size_t)n_cu * wpb);
    if (hipMemcpy(h.data(), dbuf, h.size() * sizeof(Rec), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    double inst[2] = {0, 0}, cyc[2] = {0, 0}; int cnt[2] = {0, 0}; double clk = 0; int simd_bad = 0;
    double rate_sum = 0;           // instructions per cycle, summed over all waves
    for (int b = 0; b < n_cu; ++b) for (int w = 0; w < wpb; ++w) {
        const Rec& r = h[(size_t)b * wpb + w];
        int role = r.role & 1;
        double ni = (double)r.iters * inner * n_inst[role];
        inst[role] += ni; cyc[role] += (double)r.cycles; cnt[role]++;
        rate_sum += ni / (double)r.cycles;
        clk += (double)r.cycles / ((double)r.real * 10.0);
        int simd = (r.hw_id >> 4) & 3;
        if (simd != (w & 3)) simd_bad++;
    }
    for (int r = 0; r < 2; ++r) { res->inst_per_wave[r] = cnt[r] ? inst[r] / cnt[r] : 0; res->cyc_per_inst_wave[r] = inst[r] > 0 ? cyc[r] / inst[r] : 0; }
    res->cyc_per_inst_simd = (double)n_cu * 4.0 / rate_sum;
    res->clk = clk / h.size(); res->ms = ms; res->simd_ok = simd_bad == 0;
    double p = 0; int c = 0; for (size_t i = g_samples.size() / 4; i < g_samples.size(); ++i) { p += g_samples[i]; c++; }
    res->power = c ? p / c : 0;
    return 0;
}
''')
    src.append("int main(int argc, char** argv) {\n    double seconds = argc > 1 ? atof(argv[1]) : 0.35;\n    find_hwmon();\n"
               "    printf(\"# tools/occupancy_calib.py: issue cost per instruction class at 1 / 2 / 4 waves per SIMD, heterogeneous pairs, leaf routines (hwmon: %s)\\n\", g_hwmon.empty() ? \"none\" : g_hwmon.c_str());\n"
               "    printf(\"# cyc/inst/SIMD = aggregate issue cost (4.0 = the shipped one-wave operating point); A = waves 0-3 (+8-11), B = their SIMD partners 4-7 (+12-15)\\n\");\n"
               "    Rec* dbuf; if (hipMalloc(&dbuf, (size_t)1024 * 16 * sizeof(Rec)) != hipSuccess) return 1;\n    Res r; int ni[2];\n")
    for name, a, b, wlist in tests:
        ai, _, an = bodies[a]
        bi, bn = (bodies[b][0], bodies[b][2]) if b else (-1, an)
        for w in wlist:
            src.append(f'    ni[0] = {an}; ni[1] = {bn};\n'
                       f'    if (measure(k_class<{ai}, {bi}>, {256 * w}, dbuf, seconds, ni, {INNER}, &r)) {{ printf("FAILED {name}\\n"); return 1; }}\n'
                       f'    printf("%-44s w/SIMD=%d  cyc/inst/SIMD=%6.3f  A: %6.3f cyc/inst/wave  B: %6.3f  inst A:B = %.3e : %.3e  clk=%5.3f GHz  P=%6.1f W  simd_map_ok=%d\\n", "{name}", {w}, '
                       f'r.cyc_per_inst_simd, r.cyc_per_inst_wave[0], r.cyc_per_inst_wave[1], r.inst_per_wave[0], r.inst_per_wave[1], r.clk, r.power, r.simd_ok); fflush(stdout);\n')
    for n in routines:
        lines, n_ins = rbody[n]
        if any("accvgpr" in l for l in lines):
            continue
        for w in (1, 2):
            src.append(f'    ni[0] = {n_ins}; ni[1] = {n_ins};\n'
                       f'    if (measure(k_rt_{n}, {256 * w}, dbuf, seconds, ni, 4, &r)) {{ printf("FAILED routine {n}\\n"); return 1; }}\n'
                       f'    printf("%-44s w/SIMD=%d  cyc/inst/SIMD=%6.3f  A: %6.3f cyc/inst/wave  ({n_ins} instructions per call)  clk=%5.3f GHz  P=%6.1f W  simd_map_ok=%d\\n", "ROUTINE {n}", {w}, '
                       f'r.cyc_per_inst_simd, r.cyc_per_inst_wave[0], r.clk, r.power, r.simd_ok); fflush(stdout);\n')
    src.append("    (void)hipFree(dbuf);\n    return 0;\n}\n")
    open(out_path, "w").write("".join(src))
    print("wrote", out_path, "tests:", sum(len(t[3]) for t in tests) + 2 * len(routines))


if __name__ == "__main__":
    main()
