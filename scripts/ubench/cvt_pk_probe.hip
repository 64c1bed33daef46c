// What v_cvt_pk_u8_f32 does on gfx950, in the default rounding mode and with MODE.FP_ROUND (single precision) = toward zero:
// all 2^32 float bit patterns against three candidate models (truncate / round half to even / round half up), a few printed values,
// and whether v_fma_f32 follows the mode.  hipcc --offload-arch=gfx950 -O2 cvt_pk_probe.hip -o cvt_pk_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
__device__ uint32_t model(float x, int m) {
    if (!(x > 0.0f)) return 0u;
    if (x >= 256.0f) return 255u;
    // (integer arithmetic: independent of the rounding mode the kernel runs in)
    const uint32_t b = __float_as_uint(x), e = b >> 23, man = (b & 0x7fffffu) | 0x800000u;
    if (e < 126) return 0u;                      // x < 0.5
    const int sh = 150 - (int)e;                 // value = man * 2^-sh, sh in 16 .. 24 here (0.5 <= x < 256)
    const uint32_t ip = man >> sh, frac = man & ((1u << sh) - 1u), half = 1u << (sh - 1);
    uint32_t r = ip;
    if (m == 1) r += (frac > half || (frac == half && (ip & 1u))) ? 1u : 0u;
    if (m == 2) r += frac >= half ? 1u : 0u;
    return r > 255u ? 255u : r;
}
template <int RZ> __global__ void probe(unsigned long long *bad) {
    if (RZ) __builtin_amdgcn_s_setreg(0x801, 3); // hwreg(HW_REG_MODE, 0, 2) = FP_ROUND for f32: 3 = toward zero
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned c0 = 0, c1 = 0, c2 = 0;
    for (unsigned long long b = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; b < (1ull << 32); b += stride) {
        const float x = __uint_as_float((uint32_t)b);
        const uint32_t got = __builtin_amdgcn_cvt_pk_u8_f32(x, 0u, 0u);
        c0 += got != model(x, 0), c1 += got != model(x, 1), c2 += got != model(x, 2);
    }
    atomicAdd(bad + 0, (unsigned long long)c0), atomicAdd(bad + 1, (unsigned long long)c1), atomicAdd(bad + 2, (unsigned long long)c2);
}
template <int RZ> __global__ void show(const float *x, uint32_t *y, float *f, int n) {
    if (RZ) __builtin_amdgcn_s_setreg(0x801, 3);
    const int i = threadIdx.x;
    if (i < n) y[i] = __builtin_amdgcn_cvt_pk_u8_f32(x[i], 1u, 0xAABBCCDDu);
    // fma whose exact result is 1 + 2^-24 + 2^-48 (rounds to 1 + 2^-23 to nearest, to 1 toward zero) and its negative
    if (i == 0) f[0] = __fmaf_rn(x[n], x[n], x[n + 1]), f[1] = __fmaf_rn(-x[n], x[n], -x[n + 1]);
}
int main() {
    unsigned long long *d, h[3];
    hipMalloc(&d, 24);
    for (int rz = 0; rz < 2; ++rz) {
        hipMemset(d, 0, 24);
        if (rz) hipLaunchKernelGGL(probe<1>, dim3(4096), dim3(256), 0, 0, d);
        else hipLaunchKernelGGL(probe<0>, dim3(4096), dim3(256), 0, 0, d);
        hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        printf("FP_ROUND %s: mismatches over 2^32 inputs: truncate %llu, round-half-even %llu, round-half-up %llu\n", rz ? "toward zero" : "default", h[0], h[1], h[2]);
    }
    float xs[] = {0.49999997f, 0.5f, 0.50000006f, 1.5f, 2.5f, 3.5f, 1.4999999f, 254.5f, 254.49998f, 255.4f, 255.5f, 256.0f, 1e9f, -0.5f, -0.4f, -1.0f, NAN, INFINITY, -INFINITY, 127.5f, 128.5f, 0.99999994f, 254.99998f,
                  1.0f + 0x1p-24f, 0x1p-23f - 0x1p-47f + 0.0f};
    const int n = sizeof(xs) / 4 - 2;
    // (1 + 2^-24)^2 = 1 + 2^-23 + 2^-48; minus ... keep it simple: a = 1 + 2^-12 -> a*a = 1 + 2^-11 + 2^-24, + c = 2^-25: exact 1 + 2^-11 + 2^-24 + 2^-25 -> nearest: round up (3/4 ulp), toward zero: down
    xs[n] = 1.0f + 0x1p-12f, xs[n + 1] = 0x1p-25f;
    float *dx, *df, hf[2]; uint32_t *dy, hy[32];
    hipMalloc(&dx, sizeof(xs)), hipMalloc(&dy, 128), hipMalloc(&df, 8), hipMemcpy(dx, xs, sizeof(xs), hipMemcpyHostToDevice);
    for (int rz = 0; rz < 2; ++rz) {
        if (rz) hipLaunchKernelGGL(show<1>, dim3(1), dim3(64), 0, 0, dx, dy, df, n);
        else hipLaunchKernelGGL(show<0>, dim3(1), dim3(64), 0, 0, dx, dy, df, n);
        hipMemcpy(hy, dy, n * 4, hipMemcpyDeviceToHost), hipMemcpy(hf, df, 8, hipMemcpyDeviceToHost);
        printf("FP_ROUND %s: fma(1+2^-12, 1+2^-12, 2^-25) = 1 + 2^-11 + %g ulp; negated: -(1 + 2^-11 + %g ulp)\n", rz ? "toward zero" : "default",
               (hf[0] - (1.0f + 0x1p-11f)) / 0x1p-23f, (-hf[1] - (1.0f + 0x1p-11f)) / 0x1p-23f);
        for (int i = 0; i < n; ++i) printf("  %-14.9g -> byte %3u\n", xs[i], (hy[i] >> 8) & 255);
    }
    return 0;
}
