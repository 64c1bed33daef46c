// Do the matrix pipe and the VALU of one SIMD overlap when the work has the shape of the fused kernels' units?
// A "unit" here is what a wave of quad_rr does per 16 pixels x 16 channels: three chained v_mfma_i32_16x16x64_i8 (depthwise taps)
// -> requantise the four accumulators to one packed dword (4 x (v_fma_f32 + v_cvt_pk_u8_f32) + v_xor) -> two v_mfma_i32_16x16x32_i8
// that read that dword -> requantise their eight accumulators (18 VALU) -> fold into a checksum.  Everything inside a unit is one
// dependency chain; units are independent.  Variants: the full unit, its matrix instructions alone, its VALU instructions alone;
// U = 1, 2, 3 units written side by side in the loop body (the compiler may interleave them); 1, 2, 3 waves per SIMD; no barriers.
// Output: ns per unit per SIMD.  If full ~ max(mfma, valu) the pipes overlap; if full ~ mfma + valu they do not.
//     hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip -o mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
#define ITERS 1024

__device__ __forceinline__ uint32_t pack4(v4i a, float s, float c) {
    uint32_t d = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) d = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(s, __int_as_float(a[j]), c), j, d);
    return d ^ 0x80808080u;
}
// MODE 0 full, 1 matrix instructions only, 2 VALU only
template <int MODE, int U> __global__ __launch_bounds__(768) void k(uint32_t *out, int seed, float s, float c) {
    v4i wa[3], kc = {seed, seed + 1, seed + 2, seed + 3};
#pragma unroll
    for (int t = 0; t < 3; ++t) wa[t] = v4i{seed + t, seed, seed, seed};
    long wp = seed;
    uint32_t sum[U];
    v4i bsrc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) sum[u] = 0, bsrc[u] = v4i{seed + u, seed, seed + 2 * u, seed};
    v4i av[U], p0v[U], p1v[U]; // MODE 2: stand-ins for the accumulators, made opaque in place every iteration
#pragma unroll
    for (int u = 0; u < U; ++u) av[u] = kc, p0v[u] = kc, p1v[u] = kc;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            v4i acc = kc;
            if (MODE != 2) {
#pragma unroll
                for (int t = 0; t < 3; ++t) acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa[t], bsrc[u], acc, 0, 0, 0);
            } else { // (an empty asm makes the values opaque: no instruction, and nothing of the epilogue is loop-invariant)
                asm volatile("" : "+v"(av[u][0]), "+v"(av[u][1]), "+v"(av[u][2]), "+v"(av[u][3]));
                acc = av[u];
            }
            uint32_t d;
            if (MODE != 1) d = pack4(acc, s, c);
            else d = (uint32_t)acc[0];
            long b2 = ((long)d << 32) | d;
            v4i p0 = kc, p1 = kc;
            if (MODE != 2) {
                p0 = __builtin_amdgcn_mfma_i32_16x16x32_i8(wp, b2, p0, 0, 0, 0);
                p1 = __builtin_amdgcn_mfma_i32_16x16x32_i8(wp + 1, b2, p1, 0, 0, 0);
            } else {
                asm volatile("" : "+v"(p0v[u][0]), "+v"(p0v[u][1]), "+v"(p0v[u][2]), "+v"(p0v[u][3]), "+v"(p1v[u][0]), "+v"(p1v[u][1]), "+v"(p1v[u][2]),
                             "+v"(p1v[u][3])
                             : "v"(d));
                p0 = p0v[u], p1 = p1v[u];
            }
            if (MODE != 1) sum[u] += pack4(p0, s, c) + pack4(p1, c, s);
            else sum[u] += (uint32_t)(p0[0] + p1[3]);
            bsrc[u][1] = (int)sum[u]; // the next unit of this slot depends on nothing expensive: a cheap loop-carried link keeps it live
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) r += sum[u];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int MODE, int U> static double run(uint32_t *d, int threads) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, U>), dim3(256), dim3(threads), 0, 0, d, 1, 0.001f, 3.5f);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<MODE, U>), dim3(256), dim3(threads), 0, 0, d, 1, 0.001f, 3.5f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6 / (5.0 * (threads / 256) * ITERS * U); // ns per unit per SIMD
}
template <int U> static void row(uint32_t *d) {
    for (int threads : {256, 512, 768}) {
        const double f = run<0, U>(d, threads), m = run<1, U>(d, threads), v = run<2, U>(d, threads);
        printf("U = %d units side by side, %d wave(s)/SIMD: full %6.1f ns   matrix only %6.1f   VALU only %6.1f   (sum %6.1f, max %6.1f)  per unit per SIMD\n",
               U, threads / 256, f, m, v, m + v, m > v ? m : v);
    }
}
int main() {
    uint32_t *d;
    (void)hipMalloc(&d, 256 * 768 * 4);
    row<1>(d), row<2>(d), row<3>(d);
    return 0;
}
