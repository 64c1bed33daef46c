// What v_cvt_pknorm_i16_f32 does on gfx950 -- in the default rounding mode and with MODE.FP_ROUND (single precision) = toward zero --
// and what it costs.  Why: its result's HIGH bytes are two int8 values in two's complement (round(x * 32767) >> 8 for |x| <= 1), so
// "v_fma_f32 x 4, v_cvt_pknorm_i16_f32 x 2, v_perm_b32" would requantise four accumulators into a packed i8 dword in 7 instructions
// where epilogue mode 3 (k_common.hpp: v_fma_f32 + v_cvt_pk_u8_f32 per byte, one v_xor per dword) takes 9.
// Part 1: all 2^32 float bit patterns against candidate models of the conversion (integer arithmetic, exact).
// Part 2: issue rates of the instructions and of the two sequences, 8 independent chains per wave, 8 waves per SIMD.
//     hipcc --offload-arch=gfx950 -O2 pknorm_probe.hip -o pknorm_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>

typedef short s16x2 __attribute__((ext_vector_type(2)));

// n = round(clamp(x, -1, 1) * 32767) under model m:
//   0 exact product, round half to even      1 exact product, half away from zero      2 exact product, truncate
//   3 exact product, floor(p + 1/2)           4 product rounded to f32 (nearest even), then round half to even
//   5 product truncated to f32, then truncate 6 product rounded to f32 (nearest even), then half away from zero
__device__ int model(float x, int m) {
    const uint32_t b = __float_as_uint(x);
    const int neg = b >> 31;
    uint32_t e = (b >> 23) & 255u;
    uint64_t man = b & 0x7fffffu;
    if (e == 255u) {
        if (man) return 0;           // NaN
        return neg ? -32767 : 32767; // infinities clamp
    }
    if (e == 0) return 0; // denormals and zeros: |p| < 2^-111
    man |= 0x800000u;
    if (e >= 127u) return neg ? -32767 : 32767; // |x| >= 1
    uint64_t P = man * 32767ull;                // value = P * 2^-(150 - e), 150 - e >= 24
    int sh = 150 - (int)e;
    if (m >= 4) { // first round the product to 24 significant bits
        const int bits = 64 - __clzll((long long)P);
        const int drop = bits - 24;
        if (drop > 0) {
            const uint64_t q = P >> drop, fr = P & ((1ull << drop) - 1ull), hf = 1ull << (drop - 1);
            uint64_t r = q;
            if (m == 4 || m == 6) r += (fr > hf || (fr == hf && (q & 1ull))) ? 1ull : 0ull;
            P = r << drop;
        }
    }
    uint64_t ip, fr, hf;
    if (sh >= 63) ip = 0, fr = 1, hf = 2; // far below one half
    else ip = P >> sh, fr = P & ((1ull << sh) - 1ull), hf = 1ull << (sh - 1);
    uint64_t r = ip;
    if (m == 0 || m == 4) r += (fr > hf || (fr == hf && (ip & 1ull))) ? 1ull : 0ull;
    else if (m == 1 || m == 6) r += fr >= hf ? 1ull : 0ull;
    else if (m == 3) { // floor(p + 1/2) on the signed value
        if (!neg) r += fr >= hf ? 1ull : 0ull;
        else r += fr > hf ? 1ull : 0ull;
    }
    const int n = (int)(r > 32767ull ? 32767ull : r);
    return neg ? -n : n;
}
#define NM 7
template <int RZ> __global__ void probe(unsigned long long *bad, unsigned *first) {
    if (RZ) __builtin_amdgcn_s_setreg(0x801, 3); // hwreg(HW_REG_MODE, 0, 2) = FP_ROUND for f32: 3 = toward zero
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned c[NM] = {};
    for (unsigned long long b = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; b < (1ull << 32); b += stride) {
        const float x = __uint_as_float((uint32_t)b);
        const s16x2 r = __builtin_amdgcn_cvt_pknorm_i16(x, 0.25f);
        const int got = (int)r[0];
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const bool ne = got != model(x, m);
            c[m] += ne;
            if (ne) atomicMin(first + m, (unsigned)b & 0x7fffffffu); // (smallest magnitude that differs)
        }
    }
    for (int m = 0; m < NM; ++m) atomicAdd(bad + m, (unsigned long long)c[m]);
}
template <int RZ> __global__ void show(const float *x, int *y, int n) {
    if (RZ) __builtin_amdgcn_s_setreg(0x801, 3);
    const int i = threadIdx.x;
    if (i < n) {
        const s16x2 r = __builtin_amdgcn_cvt_pknorm_i16(x[i], -x[i]);
        y[2 * i] = r[0], y[2 * i + 1] = r[1];
    }
}

// ---- rates ----
#define ITERS 2048
#define NG 8
template <int V> __global__ __launch_bounds__(256) void rate(uint32_t *out, float S, float C) {
    __builtin_amdgcn_s_setreg(0x801, 3);
    float f[NG][4];
    uint32_t d[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        d[g] = threadIdx.x * 7 + g;
#pragma unroll
        for (int k = 0; k < 4; ++k) f[g][k] = __uint_as_float(0x4B400000u + (threadIdx.x * 37 + g * 11 + k * 3) % 4000);
    }
    uint32_t sum = 0;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (V == 0) { // 4 x v_fma_f32
#pragma unroll
                for (int k = 0; k < 4; ++k) f[g][k] = __fmaf_rn(S, f[g][k], C);
            } else if (V == 1) { // 4 x v_cvt_pk_u8_f32
#pragma unroll
                for (int k = 0; k < 4; ++k) d[g] = __builtin_amdgcn_cvt_pk_u8_f32(f[g][k], (uint32_t)k, d[g]);
                f[g][0] = __uint_as_float(d[g] | 0x3f000000u);
            } else if (V == 2) { // 4 x v_cvt_pknorm_i16_f32
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const s16x2 r = __builtin_amdgcn_cvt_pknorm_i16(f[g][k], f[g][(k + 1) & 3]);
                    f[g][k] = __uint_as_float(((uint32_t)(uint16_t)r[0] | ((uint32_t)(uint16_t)r[1] << 16)) | 0x3e000000u);
                }
            } else if (V == 3) { // 4 x v_perm_b32
#pragma unroll
                for (int k = 0; k < 4; ++k) d[g] = __builtin_amdgcn_perm(d[g], __float_as_uint(f[g][k]), 0x07050301u);
            } else if (V == 4) { // candidate: 4 fma + 2 pknorm + 1 perm  (per packed dword)
                const float a = __fmaf_rn(S, f[g][0], C), b = __fmaf_rn(S, f[g][1], C), c = __fmaf_rn(S, f[g][2], C), e = __fmaf_rn(S, f[g][3], C);
                const s16x2 p = __builtin_amdgcn_cvt_pknorm_i16(a, b), q = __builtin_amdgcn_cvt_pknorm_i16(c, e);
                uint32_t pu, qu;
                __builtin_memcpy(&pu, &p, 4), __builtin_memcpy(&qu, &q, 4);
                d[g] = __builtin_amdgcn_perm(qu, pu, 0x07050301u);
                sum += d[g];
#pragma unroll
                for (int k = 0; k < 4; ++k) f[g][k] = __uint_as_float(__float_as_uint(f[g][k]) + 1u);
            } else if (V == 5) { // mode 3: 4 fma + 4 cvt_pk_u8 + xor
                const float a = __fmaf_rn(S, f[g][0], C), b = __fmaf_rn(S, f[g][1], C), c = __fmaf_rn(S, f[g][2], C), e = __fmaf_rn(S, f[g][3], C);
                uint32_t w = __builtin_amdgcn_cvt_pk_u8_f32(a, 0u, 0u);
                w = __builtin_amdgcn_cvt_pk_u8_f32(b, 1u, w), w = __builtin_amdgcn_cvt_pk_u8_f32(c, 2u, w), w = __builtin_amdgcn_cvt_pk_u8_f32(e, 3u, w);
                d[g] = w ^ 0x80808080u;
                sum += d[g];
#pragma unroll
                for (int k = 0; k < 4; ++k) f[g][k] = __uint_as_float(__float_as_uint(f[g][k]) + 1u);
            } else if (V == 6) { // the shared overhead of 4 and 5: sum + 4 adds
                sum += __float_as_uint(f[g][0]);
#pragma unroll
                for (int k = 0; k < 4; ++k) f[g][k] = __uint_as_float(__float_as_uint(f[g][k]) + 1u);
            }
        }
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) sum += d[g] + __float_as_uint(f[g][0]) + __float_as_uint(f[g][1]) + __float_as_uint(f[g][2]) + __float_as_uint(f[g][3]);
    out[blockIdx.x * 256 + threadIdx.x] = sum;
}
template <int V> static void run_rate(uint32_t *d, const char *name, int per_group) {
    const int grid = 256 * 8;
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipLaunchKernelGGL(rate<V>, dim3(grid), dim3(256), 0, 0, d, 0.999f, 0.001f);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(rate<V>, dim3(grid), dim3(256), 0, 0, d, 0.999f, 0.001f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double groups = 5.0 * grid * 4 * (double)ITERS * NG;
    const double ns = ms * 1e6 / (groups / 1024.0);
    printf("%-44s %8.3f ms  %6.2f ns per group per SIMD  (%d instructions per group: %.2f ns each)\n", name, ms, ns, per_group, ns / per_group);
}

int main() {
    unsigned long long *d, h[NM];
    unsigned *df, hf[NM];
    hipMalloc(&d, 8 * NM), hipMalloc(&df, 4 * NM);
    for (int rz = 0; rz < 2; ++rz) {
        hipMemset(d, 0, 8 * NM), hipMemset(df, 0xff, 4 * NM);
        if (rz) hipLaunchKernelGGL(probe<1>, dim3(4096), dim3(256), 0, 0, d, df);
        else hipLaunchKernelGGL(probe<0>, dim3(4096), dim3(256), 0, 0, d, df);
        hipMemcpy(h, d, 8 * NM, hipMemcpyDeviceToHost), hipMemcpy(hf, df, 4 * NM, hipMemcpyDeviceToHost);
        printf("FP_ROUND %s: mismatches over 2^32 inputs:", rz ? "toward zero" : "default");
        const char *nm[NM] = {"exact/RNE", "exact/half-away", "exact/trunc", "exact/floor(p+.5)", "f32(RNE)/RNE", "f32(RZ)/trunc", "f32(RNE)/half-away"};
        for (int m = 0; m < NM; ++m) printf("  %s %llu (first |x| bits 0x%08x)", nm[m], h[m], hf[m]);
        printf("\n");
    }
    float xs[] = {0.0f, 1.0f, -1.0f, 0.5f, 0.25f, 1.5f / 32767.0f, 2.5f / 32767.0f, 0.5f / 32767.0f, 0.49999f / 32767.0f, 0.50001f / 32767.0f, 0.9999847f, 0.99998474f, 0.999985f,
                  2.0f, -2.0f, NAN, INFINITY, 127.5f / 128.0f, 3.0517578e-05f, 100.5f / 32767.0f, 101.5f / 32767.0f, 1e-40f};
    const int n = sizeof(xs) / 4;
    float *dx;
    int *dy, hy[2 * 64];
    hipMalloc(&dx, sizeof(xs)), hipMalloc(&dy, 8 * 64), hipMemcpy(dx, xs, sizeof(xs), hipMemcpyHostToDevice);
    for (int rz = 0; rz < 2; ++rz) {
        if (rz) hipLaunchKernelGGL(show<1>, dim3(1), dim3(64), 0, 0, dx, dy, n);
        else hipLaunchKernelGGL(show<0>, dim3(1), dim3(64), 0, 0, dx, dy, n);
        hipMemcpy(hy, dy, 8 * n, hipMemcpyDeviceToHost);
        printf("FP_ROUND %s:\n", rz ? "toward zero" : "default");
        for (int i = 0; i < n; ++i) printf("  x = %-16.9g  x * 32767 = %-14.8f -> %6d   (-x -> %6d)\n", xs[i], (double)xs[i] * 32767.0, hy[2 * i], hy[2 * i + 1]);
    }
    uint32_t *o;
    hipMalloc(&o, 256 * 8 * 256 * 4);
    run_rate<0>(o, "v_fma_f32 x 4", 4);
    run_rate<1>(o, "v_cvt_pk_u8_f32 x 4 (+ 1 v_or)", 5);
    run_rate<2>(o, "v_cvt_pknorm_i16_f32 x 4 (+ 4 v_or)", 8);
    run_rate<3>(o, "v_perm_b32 x 4", 4);
    run_rate<6>(o, "loop overhead of the two sequences (5 v_add)", 5);
    run_rate<4>(o, "candidate: 4 fma + 2 pknorm + 1 perm (+ 5)", 12);
    run_rate<5>(o, "mode 3:    4 fma + 4 cvt_pk_u8 + 1 xor (+ 5)", 14);
    return 0;
}
