// Issue interval and dependent-chain latency of the int8 matrix instructions the fused kernels use (and two they might):
//   v_mfma_i32_16x16x64_i8 (depthwise taps), v_mfma_i32_16x16x32_i8 (1x1 convolutions with K <= 32), v_mfma_i32_32x32x32_i8 (the GEMM),
//   v_smfmac_i32_16x16x128_i8 / 16x16x64 (2:4 structured-sparse A: a depthwise tap matrix is 1-in-16 dense).
// Per instruction: NCH independent accumulator chains issued round-robin, one wave per SIMD (and three), zero operands (no
// power throttling from the data) -- ns per instruction per SIMD.  NCH = 1 is the dependent latency, NCH = 8 the issue interval.
//     hipcc --offload-arch=gfx950 -O2 mfma_rates.hip -o mfma_rates
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define ITERS 2048

template <int KIND, int NCH> __global__ __launch_bounds__(768) void k(int *out, int seed) {
    v4i a4 = {seed, seed, seed, seed}, b4 = a4;
    v8i b8 = {seed, seed, seed, seed, seed, seed, seed, seed};
    long a2 = seed, b2 = seed;
    v4i c4[8];
    v16i c16[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) c4[i] = v4i{0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) c16[i][r] = 0;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = j % NCH;
            if (KIND == 0) c4[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a4, b4, c4[i], 0, 0, 0);
            if (KIND == 1) c4[i] = __builtin_amdgcn_mfma_i32_16x16x32_i8(a2, b2, c4[i], 0, 0, 0);
            if (KIND == 2) c16[i & 3] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a4, b4, c16[i & 3], 0, 0, 0);
            if (KIND == 3) c4[i] = __builtin_amdgcn_smfmac_i32_16x16x128_i8(a4, b8, c4[i], seed, 0, 0);
            if (KIND == 4) c4[i] = __builtin_amdgcn_smfmac_i32_16x16x64_i8(v2i{seed, seed}, b4, c4[i], seed, 0, 0);
            if (KIND == 5) c16[i & 3] = __builtin_amdgcn_mfma_i32_32x32x16_i8(a2, b2, c16[i & 3], 0, 0, 0);
        }
    }
    int s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += c4[i][0] + c4[i][3];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += c16[i][0] + c16[i][15];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND, int NCH> static void run(int *d, const char *name, int threads) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND, NCH>), dim3(256), dim3(threads), 0, 0, d, 0);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<KIND, NCH>), dim3(256), dim3(threads), 0, 0, d, 0);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = 5.0 * (threads / 256) * ITERS * 8; // instructions per SIMD
    printf("%-28s %d chain(s), %d wave(s)/SIMD: %7.2f ns per instruction per SIMD  (%.1f cycles at 2.4 GHz)\n", name, NCH, threads / 256, ms * 1e6 / per_simd,
           ms * 1e6 / per_simd * 2.4);
}
#define ALL(KIND, NAME)                                                                                   \
    run<KIND, 1>(d, NAME, 256), run<KIND, 2>(d, NAME, 256), run<KIND, 4>(d, NAME, 256), run<KIND, 8>(d, NAME, 256); \
    run<KIND, 1>(d, NAME, 768), run<KIND, 2>(d, NAME, 768), run<KIND, 8>(d, NAME, 768);
int main() {
    int *d;
    (void)hipMalloc(&d, 256 * 768 * 4);
    ALL(0, "v_mfma_i32_16x16x64_i8")
    ALL(1, "v_mfma_i32_16x16x32_i8")
    ALL(2, "v_mfma_i32_32x32x32_i8")
    ALL(5, "v_mfma_i32_32x32x16_i8")
    ALL(3, "v_smfmac_i32_16x16x128_i8")
    ALL(4, "v_smfmac_i32_16x16x64_i8")
    return 0;
}
