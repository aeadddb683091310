// Micro-benchmarks of the VALU ceilings that bound the field arithmetic: issue cost of the integer / f64 instructions
// the field arithmetic is made of, and whole-chip throughput of the Fr and Fq Montgomery products.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I zkvm_amd/csrc tools/ubench.hip -o tools/ubench && tools/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
#include <vector>
#include "ff.hpp"
#include "fq28.hpp"

using namespace zk;

#define CHECK(x)                                                                       \
    do {                                                                               \
        hipError_t e = (x);                                                            \
        if (e != hipSuccess) {                                                         \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            return 1;                                                                  \
        }                                                                              \
    } while (0)

// Each kernel runs 8 independent copies of one instruction per iteration (so issue rate, not latency, is measured),
// `iters` iterations, and stamps the shader clock (s_memtime) and the constant 100 MHz clock (s_memrealtime) around the
// loop: their ratio is the clock the chip held while the loop ran.
#define RATE_KERNEL(NAME, BODY)                                                                    \
    __global__ void __launch_bounds__(256) NAME(uint64_t *cycles, uint32_t *sink, uint32_t seed, uint32_t iters) { \
        uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3;        \
        uint32_t a4 = a0 * 11 + 4, a5 = a0 * 13 + 5, a6 = a0 * 17 + 6, a7 = a0 * 19 + 7;            \
        uint64_t q0 = a0, q1 = a1, q2 = a2, q3 = a3, q4 = a4, q5 = a5, q6 = a6, q7 = a7;            \
        double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;              \
        uint32_t m = seed | 1;                                                                      \
        uint32_t x0 = a0 + 100, x1 = a1 + 101, x2 = a2 + 102, x3 = a3 + 103, x4 = a4 + 104, x5 = a5 + 105, x6 = a6 + 106, x7 = a7 + 107; \
        uint32_t y0 = a0 + 200, y1 = a1 + 201, y2 = a2 + 202, y3 = a3 + 203, y4 = a4 + 204, y5 = a5 + 205, y6 = a6 + 206, y7 = a7 + 207; \
        uint64_t p0 = q1 + 9, p1 = q2 + 9, p2 = q3 + 9, p3 = q4 + 9, p4 = q5 + 9, p5 = q6 + 9, p6 = q7 + 9, p7 = q0 + 9; \
        uint64_t u0 = q1 + 5, u1 = q2 + 5, u2 = q3 + 5, u3 = q4 + 5, u4 = q5 + 5, u5 = q6 + 5, u6 = q7 + 5, u7 = q0 + 5; \
        asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));        \
        asm volatile("" : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "+v"(y4), "+v"(y5), "+v"(y6), "+v"(y7));        \
        asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7));        \
        asm volatile("" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7));        \
        double dm = 1.0000001;                                                                      \
        (void)q0; (void)d0; (void)dm; (void)m;                                                      \
        uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();          \
        for (uint32_t i = 0; i < iters; i++) { BODY BODY BODY BODY }                                \
        uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();          \
        uint32_t s = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                                         \
        s ^= (uint32_t)(q0 ^ q1 ^ q2 ^ q3 ^ q4 ^ q5 ^ q6 ^ q7);                                     \
        s ^= (uint32_t)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);                                     \
        sink[blockIdx.x * blockDim.x + threadIdx.x] = s;                                            \
        if (threadIdx.x == 0) {                                                                     \
            cycles[2 * blockIdx.x] = t1 - t0;                                                       \
            cycles[2 * blockIdx.x + 1] = r1 - r0;                                                   \
        }                                                                                           \
    }

#define X8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)

#define OP_MAD64(k) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q##k) : "v"(a##k), "v"(m) : "vcc");
#define OP_MULLO(k) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a##k) : "v"(m));
#define OP_MULHI(k) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a##k) : "v"(m));
#define OP_ADD(k) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a##k) : "v"(m) : "vcc");
#define OP_ADDC(k) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a##k) : "v"(m) : "vcc");
#define OP_ADD64(k) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q##k) : "v"(q7));
#define OP_MAD24(k) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a##k) : "v"(m));
#define OP_FMA64(k) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d##k) : "v"(dm));
#define OP_FMA32(k) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a##k) : "v"(m));
#define OP_MAD32(k) asm volatile("v_mad_u32_u24 %0, %0, %1, %0\n\tv_mul_hi_u32_u24 %0, %0, %1" : "+v"(a##k) : "v"(m));

// Variants with a register of its own for every operand of every copy (x, y: per-copy multiplicands; the plain forms
// above share the multiplier `m` among the eight copies and read the accumulator twice), the two-source VOP2 encoding
// of the f32 multiply-add, and the 64-bit multiply-add with its carry-out in eight different SGPR pairs instead of vcc.
#define OP_FMA32D(k) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a##k) : "v"(x##k), "v"(y##k));
#define OP_FMAC32(k) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a##k) : "v"(x##k), "v"(y##k));
#define OP_PKFMA(k) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(q##k) : "v"(p##k), "v"(u##k));
#define OP_MAD64D(k) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q##k) : "v"(x##k), "v"(y##k) : "vcc");
#define OP_MAD64S(k) asm volatile("v_mad_u64_u32 %0, s[" MAD64S_LO(k) ":" MAD64S_HI(k) "], %1, %2, %0" : "+v"(q##k) : "v"(x##k), "v"(y##k) : "s" MAD64S_LO(k), "s" MAD64S_HI(k));
#define MAD64S_LO(k) MAD64S_LO_##k
#define MAD64S_HI(k) MAD64S_HI_##k
#define MAD64S_LO_0 "20"
#define MAD64S_HI_0 "21"
#define MAD64S_LO_1 "22"
#define MAD64S_HI_1 "23"
#define MAD64S_LO_2 "24"
#define MAD64S_HI_2 "25"
#define MAD64S_LO_3 "26"
#define MAD64S_HI_3 "27"
#define MAD64S_LO_4 "28"
#define MAD64S_HI_4 "29"
#define MAD64S_LO_5 "30"
#define MAD64S_HI_5 "31"
#define MAD64S_LO_6 "32"
#define MAD64S_HI_6 "33"
#define MAD64S_LO_7 "34"
#define MAD64S_HI_7 "35"
#define OP_MULLOD(k) asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(a##k) : "v"(x##k), "v"(y##k));

#define OP_SHR64(k) asm volatile("v_lshrrev_b64 %0, 28, %0" : "+v"(q##k));
#define OP_AND(k) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a##k) : "v"(m));
#define OP_ALIGN(k) asm volatile("v_alignbit_b32 %0, %0, %1, 28" : "+v"(a##k) : "v"(m));
#define OP_SHR32(k) asm volatile("v_lshrrev_b32 %0, 28, %0" : "+v"(a##k));
#define OP_ADD3(k) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a##k) : "v"(m));

RATE_KERNEL(k_shr64, X8(OP_SHR64))
RATE_KERNEL(k_and, X8(OP_AND))
RATE_KERNEL(k_align, X8(OP_ALIGN))
RATE_KERNEL(k_shr32, X8(OP_SHR32))
RATE_KERNEL(k_add3, X8(OP_ADD3))
RATE_KERNEL(k_mad64, X8(OP_MAD64))
RATE_KERNEL(k_mullo, X8(OP_MULLO))
RATE_KERNEL(k_mulhi, X8(OP_MULHI))
RATE_KERNEL(k_add, X8(OP_ADD))
RATE_KERNEL(k_addc, X8(OP_ADDC))
RATE_KERNEL(k_add64, X8(OP_ADD64))
RATE_KERNEL(k_mad24, X8(OP_MAD24))
RATE_KERNEL(k_fma64, X8(OP_FMA64))
RATE_KERNEL(k_fma32, X8(OP_FMA32))
RATE_KERNEL(k_fma32d, X8(OP_FMA32D))
RATE_KERNEL(k_fmac32, X8(OP_FMAC32))
RATE_KERNEL(k_pkfma, X8(OP_PKFMA))
RATE_KERNEL(k_mad64d, X8(OP_MAD64D))
RATE_KERNEL(k_mad64s, X8(OP_MAD64S))
RATE_KERNEL(k_mullod, X8(OP_MULLOD))

template <class F>
__global__ void k_field_mul(F *out, uint32_t iters) {
    F a, b;
    for (int i = 0; i < F::N; i++) {
        a.v[i] = threadIdx.x * 7 + i + 1 + blockIdx.x;
        b.v[i] = threadIdx.x * 13 + i * 3 + 2;
    }
    a.v[F::N - 1] &= 0x0fffffff;
    b.v[F::N - 1] &= 0x0fffffff;
    for (uint32_t i = 0; i < iters; i++) {
        a = F::mul(a, b);
        b = F::mul(b, a);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = F::add(a, b);
}

// Workgroups of 256 threads (one wave per SIMD of a CU), 8 per CU in the grid; `waves_per_simd` of them are resident on
// a CU at a time (the dynamic LDS request caps the occupancy), so every SIMD holds that many waves for the whole launch
// and the launch lasts several milliseconds.  Reports the chip-wide issue rate, the clock held, and from the two the
// cycles one SIMD spends per wave-instruction.
static int g_cus = 256;
template <class K>
static int run_rate(const char *name, K kernel, int waves_per_simd, uint32_t iters) {
    const int blocks = g_cus * 8, threads = 256;
    const size_t lds = (size_t)(160 * 1024) / waves_per_simd - (waves_per_simd > 1 ? 512 : 0);
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    uint64_t *cyc;
    uint32_t *sink;
    CHECK(hipMalloc(&cyc, (size_t)blocks * 16));
    CHECK(hipMalloc(&sink, (size_t)blocks * threads * 4));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), lds, 0, cyc, sink, 12345u, iters / 8);
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), lds, 0, cyc, sink, 777u, iters);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<uint64_t> h((size_t)blocks * 2);
    CHECK(hipMemcpy(h.data(), cyc, (size_t)blocks * 16, hipMemcpyDeviceToHost));
    std::vector<double> clk(blocks);
    for (int i = 0; i < blocks; i++) clk[i] = h[2 * i + 1] ? (double)h[2 * i] / (double)h[2 * i + 1] * 0.1 : 0.0;   // GHz
    std::sort(clk.begin(), clk.end());
    const double ghz = clk[blocks / 2];
    const double insts_per_wave = (double)iters * 32;
    const double total_wave_insts = insts_per_wave * blocks * (threads / 64.0);
    const double ginst = total_wave_insts / (ms * 1e-3) / 1e9;
    const double simds = g_cus * 4.0;
    printf("%-10s waves/SIMD=%d  launch %8.3f ms  clock %.2f GHz  %8.1f G wave-inst/s chip-wide  %.2f cycles per wave-inst per SIMD\n",
           name, waves_per_simd, ms, ghz, ginst, simds * ghz / ginst);
    CHECK(hipFree(cyc));
    CHECK(hipFree(sink));
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
    return 0;
}

template <class F>
static int run_field(const char *name, int threads_per_block, int blocks) {
    F *out;
    const uint32_t iters = 2000;
    CHECK(hipMalloc(&out, (size_t)blocks * threads_per_block * sizeof(F)));
    hipLaunchKernelGGL(k_field_mul<F>, dim3(blocks), dim3(threads_per_block), 0, 0, out, 10u);
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_field_mul<F>, dim3(blocks), dim3(threads_per_block), 0, 0, out, iters);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    double muls = 2.0 * iters * blocks * threads_per_block;
    printf("%-10s %d x %d threads: %.3f ms -> %.2f G mul/s\n", name, blocks, threads_per_block, ms, muls / (ms * 1e-3) / 1e9);
    CHECK(hipFree(out));
    return 0;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    g_cus = prop.multiProcessorCount;
    printf("device: %s, %d CUs, clock %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    const uint32_t iters = 16384;   // x 32 instructions per wave: several milliseconds per launch
    for (int w = 1; w <= 8; w *= 2) {
        if (run_rate("fma_f32", k_fma32, w, iters)) return 1;
        if (run_rate("fma_f32_d", k_fma32d, w, iters)) return 1;    // distinct registers for every operand
        if (run_rate("fmac_f32", k_fmac32, w, iters)) return 1;     // VOP2 encoding (4-byte instruction)
        if (run_rate("pk_fma_f32", k_pkfma, w, iters)) return 1;
        if (run_rate("mad_u64", k_mad64, w, iters)) return 1;
        if (run_rate("mad_u64_d", k_mad64d, w, iters)) return 1;    // distinct multiplicands, carry-out in vcc
        if (run_rate("mad_u64_s", k_mad64s, w, iters)) return 1;    // distinct multiplicands, carry-out in 8 SGPR pairs
        if (run_rate("mul_lo_d", k_mullod, w, iters)) return 1;
        if (run_rate("mul_lo", k_mullo, w, iters)) return 1;
        if (run_rate("mul_hi", k_mulhi, w, iters)) return 1;
        if (run_rate("add_co", k_add, w, iters)) return 1;
        if (run_rate("addc_co", k_addc, w, iters)) return 1;
        if (run_rate("add_u64", k_add64, w, iters)) return 1;
        if (run_rate("mad_u24", k_mad24, w, iters)) return 1;
        if (run_rate("lshr_b64", k_shr64, w, iters)) return 1;
        if (run_rate("and_b32", k_and, w, iters)) return 1;
        if (run_rate("alignbit", k_align, w, iters)) return 1;
        if (run_rate("lshr_b32", k_shr32, w, iters)) return 1;
        if (run_rate("add3_u32", k_add3, w, iters)) return 1;
        if (run_rate("fma_f64", k_fma64, w, iters)) return 1;
    }
    for (int tpb : {256, 512, 1024}) {
        if (run_field<Fr>("Fr::mul", tpb, 8192)) return 1;
        if (run_field<Fq>("Fq::mul", tpb, 8192)) return 1;
        if (run_field<Fq28>("Fq28::mul", tpb, 8192)) return 1;
    }
    return 0;
}
