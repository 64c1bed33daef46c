// Microbenchmark: cost of the requantisation epilogue (per packed dword = 4 output bytes) on gfx950, for the
// packing variants tried in k_common.hpp.  No memory traffic in the loop: this is the VALU floor of every
// fast kernel.  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off epi_rate.hip -o epi_rate
// Also built as scripts/ubench/libepi_rate.so (microflow_rs_amd/build.py): bench.py calls mf_ubench_requant_ns() outside its
// timed region and prices the VALU-bound kernels against the rate measured in the same run.
// Variants 4 / 5 are the library's own requant_pack4<1> / <2> (k_common.hpp), i.e. exactly what the kernels execute.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "../../microflow_rs_amd/csrc/k_common.hpp"

#ifndef ITERS
#define ITERS 4096 // (the library build uses 1024: a variant costs bench.py ~17 ms of GPU time)
#endif
#define NG 8 // independent dword groups per iteration

using mf::k::pack4;
__device__ __forceinline__ float rq(int acc, float A, float S, float lo, float hi) {
    const float f = __fsub_rn(__int_as_float(acc), 12582912.0f);
    const float x = __fadd_rn(A, __fmul_rn(S, f));
    const float r = __fadd_rn(x, __builtin_copysignf(0x1.fffffep-2f, x));
    return __builtin_amdgcn_fmed3f(r, lo, hi);
}

template <int V>
__global__ __launch_bounds__(256) void bench(uint32_t *out, int seed, float A, float S, float lo, float hi) {
    int acc[NG][4];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[g][k] = 0x4B400000 + (int)(threadIdx.x * 37 + g * 11 + k * 3 + seed) % 4000;
    uint32_t sum = 0;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            uint32_t d = 0;
            if (V == 0) { // requant + v_cvt + 3 v_perm (r01)
                d = pack4((int)rq(acc[g][0], A, S, lo, hi), (int)rq(acc[g][1], A, S, lo, hi), (int)rq(acc[g][2], A, S, lo, hi),
                          (int)rq(acc[g][3], A, S, lo, hi));
            } else if (V == 1) { // SDWA cvt into byte k, dependent chain, s_nop between (round 2's form, retired from the library in round 5)
                const float r0 = rq(acc[g][0], A, S, lo, hi), r1 = rq(acc[g][1], A, S, lo, hi), r2 = rq(acc[g][2], A, S, lo, hi),
                            r3 = rq(acc[g][3], A, S, lo, hi);
                asm("v_cvt_i32_f32_sdwa %0, %1 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD\n\ts_nop 0\n\t"
                    "v_cvt_i32_f32_sdwa %0, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\ts_nop 0\n\t"
                    "v_cvt_i32_f32_sdwa %0, %3 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\ts_nop 0\n\t"
                    "v_cvt_i32_f32_sdwa %0, %4 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\ts_nop 0"
                    : "=&v"(d) : "v"(r0), "v"(r1), "v"(r2), "v"(r3));
            } else if (V == 8) { // epilogue mode 3: v_fma_f32 + v_cvt_pk_u8_f32 per byte (the library's requant_pack4<3>)
                d = mf::k::requant_pack4<3, 0u>(acc[g][0], acc[g][1], acc[g][2], acc[g][3], make_float4(A, A, A, A), make_float4(S, S, S, S), lo, hi);
            } else if (V == 9) { // mode 3, two dwords per call
                if (g & 1) {
                    uint32_t da, db;
                    const mf::k::v4i a0 = {acc[g - 1][0], acc[g - 1][1], acc[g - 1][2], acc[g - 1][3]}, a1 = {acc[g][0], acc[g][1], acc[g][2], acc[g][3]};
                    mf::k::requant_pack4x2<3, 0u>(a0, make_float4(A, A, A, A), make_float4(S, S, S, S), a1, make_float4(A, A, A, A), make_float4(S, S, S, S), lo, hi, da, db);
                    d = da + db;
                }
            } else if (V == 2) { // requant only (no conversion, no packing): lower bound of the float part
                d = __float_as_uint(rq(acc[g][0], A, S, lo, hi)) ^ __float_as_uint(rq(acc[g][1], A, S, lo, hi)) ^
                    __float_as_uint(rq(acc[g][2], A, S, lo, hi)) ^ __float_as_uint(rq(acc[g][3], A, S, lo, hi));
            } else if (V == 3) { // v_cvt_pk_u8_f32: truncating conversion + [0,255] clamp + byte insert (u8 domain only)
                const float r0 = rq(acc[g][0], A, S, lo, hi), r1 = rq(acc[g][1], A, S, lo, hi), r2 = rq(acc[g][2], A, S, lo, hi),
                            r3 = rq(acc[g][3], A, S, lo, hi);
                asm("v_cvt_pk_u8_f32 %0, %1, 0, 0\n\t"
                    "v_cvt_pk_u8_f32 %0, %2, 1, %0\n\t"
                    "v_cvt_pk_u8_f32 %0, %3, 2, %0\n\t"
                    "v_cvt_pk_u8_f32 %0, %4, 3, %0"
                    : "=&v"(d) : "v"(r0), "v"(r1), "v"(r2), "v"(r3));
            }
            else if (V == 4) { // sticky-bit RNE + v_med3 (epilogue mode 1)
                d = mf::k::requant_pack4<1, 0u>(acc[g][0], acc[g][1], acc[g][2], acc[g][3], make_float4(A, A, A, A), make_float4(S, S, S, S), lo, hi);
            } else if (V == 5) { // sticky-bit RNE + saturating pack (epilogue mode 2)
                d = mf::k::requant_pack4<2, 0u>(acc[g][0], acc[g][1], acc[g][2], acc[g][3], make_float4(A, A, A, A), make_float4(S, S, S, S), lo, hi);
            } else if (V == 6) { // two dwords at a time, mode 1
                if (g & 1) {
                    uint32_t da, db;
                    const mf::k::v4i a0 = {acc[g - 1][0], acc[g - 1][1], acc[g - 1][2], acc[g - 1][3]}, a1 = {acc[g][0], acc[g][1], acc[g][2], acc[g][3]};
                    mf::k::requant_pack4x2<1, 0u>(a0, make_float4(A, A, A, A), make_float4(S, S, S, S), a1, make_float4(A, A, A, A), make_float4(S, S, S, S), lo, hi, da, db);
                    d = da + db;
                }
            } else if (V == 7) { // two dwords at a time, mode 2
                if (g & 1) {
                    uint32_t da, db;
                    const mf::k::v4i a0 = {acc[g - 1][0], acc[g - 1][1], acc[g - 1][2], acc[g - 1][3]}, a1 = {acc[g][0], acc[g][1], acc[g][2], acc[g][3]};
                    mf::k::requant_pack4x2<2, 0u>(a0, make_float4(A, A, A, A), make_float4(S, S, S, S), a1, make_float4(A, A, A, A), make_float4(S, S, S, S), lo, hi, da, db);
                    d = da + db;
                }
            }
            sum += d;
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[g][k] += 1; // one full-rate op per value keeps the loop body live
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = sum;
}

static bool g_quiet = false;
template <int V> static double run(uint32_t *d, const char *name, double base_extra) {
    const int grid = 256 * 8; // 8 blocks of 4 waves per CU = 8 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipLaunchKernelGGL(bench<V>, dim3(grid), dim3(256), 0, 0, d, 1, 0.5f, 0.01f, -128.0f, 127.0f);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(bench<V>, dim3(grid), dim3(256), 0, 0, d, r, 0.5f, 0.01f, -128.0f, 127.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double groups = 5.0 * grid * 4 /*waves*/ * ITERS * NG;       // wave-level dword groups
    const double ns_per_group_simd = ms * 1e6 / (groups / 1024.0);      // per SIMD
    if (!g_quiet) std::printf("%-34s %8.3f ms  %6.2f ns per dword-group per SIMD (%.1f per byte)  [loop overhead included]\n", name, ms,
                ns_per_group_simd, ns_per_group_simd / 4);
    (void)base_extra;
    return ns_per_group_simd;
}

// ns per packed dword (4 output bytes) per SIMD of variant v (4: epilogue mode 1, 5: mode 2, 8: mode 3, 1: round 2's form), loop
// overhead included (one v_add per value); < 0 on failure.  The chip-wide rate is 1024 SIMDs * 4 bytes / that.
extern "C" double mf_ubench_requant_ns(int variant) {
    uint32_t *d = nullptr;
    if (hipMalloc(&d, 256 * 8 * 256 * 4) != hipSuccess) return -1.0;
    g_quiet = true;
    double ns = -1.0;
    switch (variant) {
    case 1: ns = run<1>(d, "", 0); break;
    case 4: ns = run<4>(d, "", 0); break;
    case 5: ns = run<5>(d, "", 0); break;
    case 6: ns = run<6>(d, "", 0); break;
    case 7: ns = run<7>(d, "", 0); break;
    case 8: ns = run<8>(d, "", 0); break;
    case 9: ns = run<9>(d, "", 0); break;
    default: break;
    }
    g_quiet = false;
    if (hipDeviceSynchronize() != hipSuccess) ns = -1.0;
    (void)hipFree(d);
    return ns;
}
#ifndef MF_UBENCH_LIB
int main() {
    uint32_t *d;
    hipMalloc(&d, 256 * 8 * 256 * 4);
    run<2>(d, "requant only (no cvt/pack)", 0);
    run<0>(d, "requant + cvt + 3 v_perm (r01)", 0);
    run<1>(d, "requant + SDWA cvt (r02)", 0);
    run<3>(d, "requant + v_cvt_pk_u8_f32", 0);
    run<4>(d, "mode 1: med3 + sticky RNE pack", 0);
    run<5>(d, "mode 2: sticky RNE + sat pack", 0);
    run<6>(d, "mode 1, two dwords per block", 0);
    run<7>(d, "mode 2, two dwords per block", 0);
    run<8>(d, "mode 3: fma + cvt_pk_u8", 0);
    run<9>(d, "mode 3, two dwords per call", 0);
    return 0;
}
#endif
