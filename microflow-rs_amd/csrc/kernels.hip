// kernels.hip -- hand-written HIP kernels for gfx950 (MI355X / CDNA4).
//
// One kernel family per reference function in src/ops/ (upstream file:line in each
// header comment).  All of them share one arithmetic contract:
//
//   acc (i32)  = sum over ALL window taps of (v' - izp) * (w - wzp)
//                with v' = izp at out-of-range taps          [== x0 - x1 - k2 + k3]
//   y          = sat_i8(roundf((f32(ozp) + c0[c]) + c1[c] * f32(acc)))   then activation
//
// The device never evaluates `izp`-dependent terms per pixel: the host folds
//   Kc[c] = -izp * sum_all_taps w[c] + T * izp * wzp[c]   (T = taps contributing to c)
// so that acc = dot(v', w) - wzp[c] * sum(v') + Kc[c]; padding is realised by
// filling the halo with izp, which makes every pixel -- border or interior -- run
// the same code.  A[c] = fl32(f32(ozp) + c0[c]) and S[c] = c1[c or 0] are folded on
// the host too, and the activation is a clamp [lo, hi] (relu: lo = ozp; relu6:
// hi = quantize(6.0)).
//
// f32 rules: no contraction (this file is built with -ffp-contract=off AND uses the
// explicit __fmul_rn/__fadd_rn/__fdiv_rn forms), int->float is v_cvt_f32_i32 (RNE),
// roundf is trunc(x + copysign(pred(0.5), x)) which is exact for every finite x.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <atomic>
#include <type_traits>

#include "kernels.hpp"

namespace mf {
namespace k {

typedef int v4i __attribute__((ext_vector_type(4)));
// native vector type for register-resident staging arrays: arrays of HIP's struct-based
// uint4 that are conditionally re-assigned are NOT promoted to registers by hipcc (they
// end up in scratch memory); ext_vector_type arrays are.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------
// shared device helpers
// ------------------------------------------------------------------------
__device__ __forceinline__ int requant(int acc, float A, float S, float lo_f, float hi_f) {
    // (f32(ozp) + c0) + c1 * f32(acc): two roundings, like the reference (conv_2d.rs:93-98)
    const float x = __fadd_rn(A, __fmul_rn(S, (float)acc));
    // libm::roundf = half away from zero = trunc(x + copysign(0x1.fffffep-2, x))
    float r = __fadd_rn(x, __builtin_copysignf(0x1.fffffep-2f, x));
    // saturating cast + activation clamp, done in f32 so that the int conversion below
    // is always in range (it truncates toward zero)
    r = __builtin_amdgcn_fmed3f(r, lo_f, hi_f);
    return (int)r;
}

// "Magic" accumulators (MG = true): the accumulator is started at Kc + 0x4B400000, so that its
// bit pattern read as f32 is 12582912 + acc, exactly, whenever |acc| < 2^22; f32(acc) is then
// one v_add_f32 (full rate) instead of v_cvt_f32_i32 (half rate).  Both are exact, so the
// epilogue's value is unchanged.  The host enables it per operator only when the worst-case
// |acc| over ALL inputs -- max|v - izp| * sum|w - wzp| per channel -- is below 2^22 (ops.hip).
constexpr int MF_MAGIC_I = 0x4B400000;
template <bool MG>
__device__ __forceinline__ int requant_t(int acc, float A, float S, float lo_f, float hi_f) {
    if constexpr (!MG) {
        return requant(acc, A, S, lo_f, hi_f);
    } else {
        const float f = __fsub_rn(__int_as_float(acc), 12582912.0f);
        const float x = __fadd_rn(A, __fmul_rn(S, f));
        float r = __fadd_rn(x, __builtin_copysignf(0x1.fffffep-2f, x));
        r = __builtin_amdgcn_fmed3f(r, lo_f, hi_f);
        return (int)r;
    }
}
template <bool MG> __device__ __forceinline__ int4 magic4(int4 k) {
    if constexpr (MG) k.x += MF_MAGIC_I, k.y += MF_MAGIC_I, k.z += MF_MAGIC_I, k.w += MF_MAGIC_I;
    return k;
}

// The shape-generic kernels also honour Rust's `NaN as i8 == 0` (a NaN can only come from
// non-finite constants, i.e. a degenerate model); operators with non-finite constants are never
// routed to the shape-specialised kernels (ops.hip), whose requant() skips this test.
__device__ __forceinline__ int requant_any(int acc, float A, float S, float lo_f, float hi_f) {
    const float x = __fadd_rn(A, __fmul_rn(S, (float)acc));
    float r = __fadd_rn(x, __builtin_copysignf(0x1.fffffep-2f, x));
    r = (r != r) ? 0.0f : r; // NaN -> 0, then the activation clamp like any other value
    r = __builtin_amdgcn_fmed3f(r, lo_f, hi_f);
    return (int)r;
}

// 4 ints in [-128,127] -> one dword of int8 (byte 0 = a)
__device__ __forceinline__ uint32_t pack4(int a, int b, int c, int d) {
    const uint32_t lo = __builtin_amdgcn_perm((uint32_t)b, (uint32_t)a, 0x0c0c0400u);
    const uint32_t hi = __builtin_amdgcn_perm((uint32_t)d, (uint32_t)c, 0x0c0c0400u);
    return __builtin_amdgcn_perm(hi, lo, 0x05040100u);
}

// pack4 for either element type: XR4 = 0x80808080 moves u8-domain epilogue results (0..255) back
// to the stored i8 domain (kernels.hpp), XR4 = 0 is plain i8 (the XOR disappears at compile time)
template <uint32_t XR4>
__device__ __forceinline__ uint32_t pack4x(int a, int b, int c, int d) {
    return pack4(a, b, c, d) ^ XR4;
}

__device__ __forceinline__ int sdot4(uint32_t a, uint32_t b, int c) {
    return __builtin_amdgcn_sdot4((int)a, (int)b, c, false);
}
// First tap of an accumulator: d = dot4(a, b) + c with c a value that stays live (the folded
// constant Kc).  hipcc otherwise copies c into d and uses the destructive v_dot4c (one v_mov per
// accumulator per task); the three-address form needs no copy.
__device__ __forceinline__ int sdot4_first(uint32_t a, uint32_t b, int c) {
    int d;
    asm("v_dot4_i32_i8 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// ------------------------------------------------------------------------
// Generic (any shape) kernels: one thread per output element.  They exist for
// API completeness (the reference's unit KATs, odd filter shapes, non-zero weight
// zero points) and as the in-product cross-check of the fast paths.
// ------------------------------------------------------------------------

// microflow::ops::conv_2d  (src/ops/conv_2d.rs:28-108)
__global__ __launch_bounds__(256) void conv2d_generic(const int8_t *__restrict__ in,
                                                      int8_t *__restrict__ out, ConvArgs p,
                                                      size_t total) {
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * 256) {
        const int n = (int)(idx % p.N);
        size_t t = idx / p.N;
        const int ox = (int)(t % p.OW);
        t /= p.OW;
        const int oy = (int)(t % p.OH);
        const size_t img = t / p.OH;
        const int8_t *ip = in + img * (size_t)p.H * p.W * p.C;
        const int8_t *wp = p.w + (size_t)n * p.KH * p.KW * p.C;
        const int shy = p.pad_same ? (p.KH - 1) / 2 : 0, shx = p.pad_same ? (p.KW - 1) / 2 : 0;
        int dot = 0, vs = 0;
        for (int ky = 0; ky < p.KH; ++ky) {
            const int iy = oy * p.sh + ky - shy;
            for (int kx = 0; kx < p.KW; ++kx) {
                const int ix = ox * p.sw + kx - shx;
                const bool ok = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
                const int8_t *vp = ip + ((size_t)iy * p.W + ix) * p.C;
                const int8_t *fp = wp + ((size_t)ky * p.KW + kx) * p.C;
                for (int c = 0; c < p.C; ++c) {
                    const int v = ok ? (int)vp[c] : p.izp;
                    dot += v * (int)fp[c];
                    vs += v;
                }
            }
        }
        const int acc = dot - p.wzp[n] * vs + p.Kc[n];
        out[idx] = (int8_t)(requant_any(acc, p.A[n], p.S[n], p.lo_f, p.hi_f) ^ p.xr);
    }
}

// microflow::ops::depthwise_conv_2d  (src/ops/depthwise_conv_2d.rs:28-105)
__global__ __launch_bounds__(256) void dwconv_generic(const int8_t *__restrict__ in,
                                                      int8_t *__restrict__ out, ConvArgs p,
                                                      size_t total) {
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * 256) {
        const int c = (int)(idx % p.N);
        size_t t = idx / p.N;
        const int ox = (int)(t % p.OW);
        t /= p.OW;
        const int oy = (int)(t % p.OH);
        const size_t img = t / p.OH;
        const int ci = c < p.C ? c : 0; // v.get(c).copied().unwrap_or(v[0])  (depthwise_conv_2d.rs:67)
        const int8_t *ip = in + img * (size_t)p.H * p.W * p.C;
        const int shy = p.pad_same ? (p.KH - 1) / 2 : 0, shx = p.pad_same ? (p.KW - 1) / 2 : 0;
        int dot = 0, vs = 0;
        for (int ky = 0; ky < p.KH; ++ky) {
            const int iy = oy * p.sh + ky - shy;
            for (int kx = 0; kx < p.KW; ++kx) {
                const int ix = ox * p.sw + kx - shx;
                const bool ok = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
                const int v = ok ? (int)ip[((size_t)iy * p.W + ix) * p.C + ci] : p.izp;
                dot += v * (int)p.w[((size_t)ky * p.KW + kx) * p.N + c];
                vs += v;
            }
        }
        const int acc = dot - p.wzp[c] * vs + p.Kc[c];
        out[idx] = (int8_t)(requant_any(acc, p.A[c], p.S[c], p.lo_f, p.hi_f) ^ p.xr);
    }
}

// microflow::ops::average_pool_2d  (src/ops/average_pool_2d.rs:29-66)
//   x = (1 / f32(len)) * f32(sum over the zero-filled window);  y = roundf(c0 * x + c1)
__global__ __launch_bounds__(256) void avgpool_generic(const int8_t *__restrict__ in,
                                                       int8_t *__restrict__ out, PoolArgs p,
                                                       size_t total) {
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * 256) {
        const int c = (int)(idx % p.C);
        size_t t = idx / p.C;
        const int ox = (int)(t % p.OW);
        t /= p.OW;
        const int oy = (int)(t % p.OH);
        const size_t img = t / p.OH;
        const int8_t *ip = in + img * (size_t)p.H * p.W * p.C;
        const int shy = p.pad_same ? (p.KH - 1) / 2 : 0, shx = p.pad_same ? (p.KW - 1) / 2 : 0;
        int sum = 0, len = 0;
        for (int ky = 0; ky < p.KH; ++ky) {
            const int iy = oy * p.sh + ky - shy;
            for (int kx = 0; kx < p.KW; ++kx) {
                const int ix = ox * p.sw + kx - shx;
                if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) {
                    sum += (int)ip[((size_t)iy * p.W + ix) * p.C + c] + p.bias;
                    ++len;
                }
            }
        }
        const float inv = __fdiv_rn(1.0f, (float)len);
        const float x = __fmul_rn(inv, (float)sum);
        const float y = __fadd_rn(__fmul_rn(p.c0, x), p.c1);
        float r = __fadd_rn(y, __builtin_copysignf(0x1.fffffep-2f, y));
        // NaN (len == 0) converts to 0 like Rust's `as`; fmed3 is skipped for it
        int q = (r != r) ? 0 : (int)__builtin_amdgcn_fmed3f(r, p.sat_lo, p.sat_hi);
        q = max(q, p.lo);
        q = min(q, p.hi);
        out[idx] = (int8_t)(q ^ p.xr);
    }
}

// microflow::ops::fully_connected  (src/ops/fully_connected.rs:24-82), any M/K/N.
// in [batch*M][K], w [N][K], out [batch*M][N].
__global__ __launch_bounds__(256) void fc_generic(const int8_t *__restrict__ in,
                                                  int8_t *__restrict__ out, FcArgs p,
                                                  size_t total) {
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * 256) {
        const int j = (int)(idx % p.N);
        const size_t row = idx / p.N;
        const int8_t *x = in + row * (size_t)p.K;
        const int8_t *w = p.w + (size_t)j * p.K;
        int dot = 0, rs = 0;
        if ((p.K & 3) == 0) {
            const uint32_t *x4 = (const uint32_t *)x, *w4 = (const uint32_t *)w;
            for (int k = 0; k < p.K / 4; ++k) {
                const uint32_t v = x4[k];
                dot = sdot4(v, w4[k], dot);
                rs = sdot4(v, 0x01010101u, rs);
            }
        } else {
            for (int k = 0; k < p.K; ++k) {
                dot += (int)x[k] * (int)w[k];
                rs += (int)x[k];
            }
        }
        const int acc = dot - p.wzp * rs + p.Kc[j];
        out[idx] = (int8_t)(requant_any(acc, p.A[j], p.S, p.lo_f, p.hi_f) ^ p.xr);
    }
}

// FullyConnected with few outputs and a long reduction (speech: K=4000, N=4):
// one wavefront per input row, 16-byte coalesced loads, DPP/shuffle reduction.
// Memory-bound: every input byte is read exactly once.
template <int N>
__global__ __launch_bounds__(256) void fc_rowwave(const int8_t *__restrict__ in,
                                                  int8_t *__restrict__ out, FcArgs p, size_t rows) {
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * 256) >> 6;
    const int K16 = p.K >> 4; // K % 16 == 0 is a routing precondition
    for (size_t row = wave; row < rows; row += nwaves) {
        const uint4 *x = (const uint4 *)(in + row * (size_t)p.K);
        int dot[N], rs = 0;
#pragma unroll
        for (int j = 0; j < N; ++j) dot[j] = 0;
        for (int k = lane; k < K16; k += 64) {
            const uint4 v = x[k];
            rs = sdot4(v.x, 0x01010101u, rs);
            rs = sdot4(v.y, 0x01010101u, rs);
            rs = sdot4(v.z, 0x01010101u, rs);
            rs = sdot4(v.w, 0x01010101u, rs);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const uint4 w = ((const uint4 *)(p.w + (size_t)j * p.K))[k];
                dot[j] = sdot4(v.x, w.x, dot[j]);
                dot[j] = sdot4(v.y, w.y, dot[j]);
                dot[j] = sdot4(v.z, w.z, dot[j]);
                dot[j] = sdot4(v.w, w.w, dot[j]);
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            rs += __shfl_xor(rs, off, 64);
#pragma unroll
            for (int j = 0; j < N; ++j) dot[j] += __shfl_xor(dot[j], off, 64);
        }
        if (lane < N) {
            int d = 0;
#pragma unroll
            for (int j = 0; j < N; ++j) d = (lane == j) ? dot[j] : d;
            const int acc = d - p.wzp * rs + p.Kc[lane];
            out[row * N + lane] = (int8_t)(requant(acc, p.A[lane], p.S, p.lo_f, p.hi_f) ^ p.xr);
        }
    }
}

// fc_rowwave followed by the softmax over its N outputs (speech.tflite: FullyConnected 4000 -> 4,
// Softmax) in one launch: the FC result of a row IS the whole [1][N] softmax tensor, so lanes
// 0..N-1 exchange their exp-table entries by shuffles and every one of them forms the sum in the
// reference's order (src/ops/softmax.rs:20-21) before quantising its own probability.
template <int N>
__global__ __launch_bounds__(256) void fc_rowwave_softmax(const int8_t *__restrict__ in, int8_t *__restrict__ out,
                                                          FcArgs p, SoftmaxArgs sm, size_t rows) {
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * 256) >> 6;
    const int K16 = p.K >> 4;
    for (size_t row = wave; row < rows; row += nwaves) {
        const uint4 *x = (const uint4 *)(in + row * (size_t)p.K);
        int dot[N], rs = 0;
#pragma unroll
        for (int j = 0; j < N; ++j) dot[j] = 0;
        for (int k = lane; k < K16; k += 64) {
            const uint4 v = x[k];
            rs = sdot4(v.x, 0x01010101u, rs);
            rs = sdot4(v.y, 0x01010101u, rs);
            rs = sdot4(v.z, 0x01010101u, rs);
            rs = sdot4(v.w, 0x01010101u, rs);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const uint4 w = ((const uint4 *)(p.w + (size_t)j * p.K))[k];
                dot[j] = sdot4(v.x, w.x, dot[j]);
                dot[j] = sdot4(v.y, w.y, dot[j]);
                dot[j] = sdot4(v.z, w.z, dot[j]);
                dot[j] = sdot4(v.w, w.w, dot[j]);
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            rs += __shfl_xor(rs, off, 64);
#pragma unroll
            for (int j = 0; j < N; ++j) dot[j] += __shfl_xor(dot[j], off, 64);
        }
        int d = 0;
#pragma unroll
        for (int j = 0; j < N; ++j) d = (lane == j) ? dot[j] : d;
        const int ch = lane < N ? lane : 0;
        const int acc = d - p.wzp * rs + p.Kc[ch];
        // the FullyConnected output byte as it would be stored (i8 domain), then softmax's table index
        const int y = (int)(int8_t)(requant(acc, p.A[ch], p.S, p.lo_f, p.hi_f) ^ p.xr);
        const float e = sm.exp_table[y + 128];
        float sum = 0.0f;
#pragma unroll
        for (int j = 0; j < N; ++j) sum = __fadd_rn(sum, __shfl(e, j, 64));
        const float prob = __fdiv_rn(e, sum);
        const float q = __fadd_rn(__fdiv_rn(prob, sm.oscale), sm.ozp_f);
        const float r = __fadd_rn(q, __builtin_copysignf(0x1.fffffep-2f, q));
        const int qi = (r != r) ? 0 : (int)__builtin_amdgcn_fmed3f(r, sm.sat_lo, sm.sat_hi);
        if (lane < N) out[row * N + lane] = (int8_t)(qi ^ sm.xr);
    }
}

// microflow::ops::softmax  (src/ops/softmax.rs:15-27).  One thread per inference.
// e_k = f32(q_k) * input_scale has only 256 possible values, so expf comes from a
// host-built table (libm's algorithm runs on the host, never the device's expf).
// The sum runs over the whole rows x cols tensor in column-major order.
__global__ __launch_bounds__(256) void softmax_table(const int8_t *__restrict__ in,
                                                     int8_t *__restrict__ out, SoftmaxArgs p,
                                                     size_t batch) {
    for (size_t b = (size_t)blockIdx.x * 256 + threadIdx.x; b < batch;
         b += (size_t)gridDim.x * 256) {
        const int8_t *x = in + b * (size_t)p.rows * p.cols;
        int8_t *y = out + b * (size_t)p.rows * p.cols;
        float sum = 0.0f;
        for (int j = 0; j < p.cols; ++j)
            for (int i = 0; i < p.rows; ++i) sum = __fadd_rn(sum, p.exp_table[(int)x[i * p.cols + j] + 128]);
        for (int i = 0; i < p.rows * p.cols; ++i) {
            const float e = p.exp_table[(int)x[i] + 128];
            const float prob = __fdiv_rn(e, sum);
            const float q = __fadd_rn(__fdiv_rn(prob, p.oscale), p.ozp_f); // quantize (quantize.rs:17)
            const float r = __fadd_rn(q, __builtin_copysignf(0x1.fffffep-2f, q));
            const int qi = (r != r) ? 0 : (int)__builtin_amdgcn_fmed3f(r, p.sat_lo, p.sat_hi);
            y[i] = (int8_t)(qi ^ p.xr);
        }
    }
}

// src/quantize.rs:16-18 over a buffer: q = sat(roundf(x / scale + f32(zp)))
__global__ __launch_bounds__(256) void quantize_f32(const float *__restrict__ in,
                                                    int8_t *__restrict__ out, size_t n, float scale,
                                                    float zp_f, float sat_lo, float sat_hi, int xr) {
    // 4 values per thread: one 16-byte load, one 4-byte store (scalar when a pointer is not aligned for it)
    const size_t n4 = ((((uintptr_t)in & 15) | ((uintptr_t)out & 3)) == 0) ? n >> 2 : 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = ((const float4 *)in)[i];
        int q[4];
        const float xs[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float t = __fadd_rn(__fdiv_rn(xs[k], scale), zp_f);
            const float r = __fadd_rn(t, __builtin_copysignf(0x1.fffffep-2f, t));
            q[k] = (r != r) ? 0 : (int)__builtin_amdgcn_fmed3f(r, sat_lo, sat_hi);
        }
        ((uint32_t *)out)[i] = pack4(q[0], q[1], q[2], q[3]) ^ (0x01010101u * (uint32_t)xr);
    }
    for (size_t i = (n4 << 2) + (size_t)blockIdx.x * 256 + threadIdx.x; i < n;
         i += (size_t)gridDim.x * 256) {
        const float t = __fadd_rn(__fdiv_rn(in[i], scale), zp_f);
        const float r = __fadd_rn(t, __builtin_copysignf(0x1.fffffep-2f, t));
        const int qi = (r != r) ? 0 : (int)__builtin_amdgcn_fmed3f(r, sat_lo, sat_hi);
        out[i] = (int8_t)(qi ^ xr);
    }
}

// u8 <-> internal i8 domain at the quantized boundary of a u8 model: byte ^ 0x80
__global__ __launch_bounds__(256) void xor80_bytes(const int8_t *in, int8_t *out,  // may alias (in place)
                                                   size_t n) {
    const size_t n16 = n >> 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        u32x4 v = ((const u32x4 *)in)[i];
        v ^= 0x80808080u;
        ((u32x4 *)out)[i] = v;
    }
    for (size_t i = (n16 << 4) + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        out[i] = (int8_t)(in[i] ^ 0x80);
}

// src/quantize.rs:27-29: x = scale * (f32(q) - f32(zp))
__global__ __launch_bounds__(256) void dequantize_i8(const int8_t *__restrict__ in,
                                                     float *__restrict__ out, size_t n, float scale,
                                                     float zp_f, int raw_u8) {
    // raw_u8: the bytes are real u8 values (ABI-level mf_dequantize_u8), not the internal domain
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int q = raw_u8 ? (int)(uint8_t)in[i] : (int)in[i];
        out[i] = __fmul_rn(scale, __fsub_rn((float)q, zp_f));
    }
}

// counter-based synthetic input: 8 bytes per thread
__global__ __launch_bounds__(256) void synth_i8(int8_t *__restrict__ out, size_t n, uint64_t seed,
                                                uint64_t first) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const uint64_t g = first + i;
        const uint64_t w = splitmix64(seed + (g >> 3));
        out[i] = (int8_t)(uint8_t)(w >> ((g & 7) * 8));
    }
}

// position-sensitive checksum: sum (u8 + 1) * splitmix64(i); one atomic per block
__global__ __launch_bounds__(256) void checksum_i8(const int8_t *__restrict__ in, size_t n,
                                                   unsigned long long *__restrict__ result) {
    __shared__ unsigned long long part[256];
    unsigned long long s = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        s += ((unsigned long long)(uint8_t)in[i] + 1ull) * splitmix64(i);
    part[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) part[threadIdx.x] += part[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(result, part[0]);
}

// ------------------------------------------------------------------------
// HBM -> LDS staging for the depthwise kernels: LDS-DMA (`global_load_lds_dwordx4`,
// gfx950), 16 bytes per lane straight into LDS with no VGPR round trip.
//
// Why not registers: (1) a predicated register prefetch makes hipcc branch around every
// element and wait vmcnt(0) after each one; (2) an array of HIP's struct-based uint4 that
// is conditionally re-assigned is not promoted to registers (it went to scratch); (3) even
// with both fixed, hipcc flushed vmcnt(0) in the pre-header of the compute loop, i.e. the
// prefetch never overlapped the compute.  A DMA has no destination register, so nothing
// waits on it except the one explicit `s_waitcnt vmcnt(0)` + barrier per step below.
// The LDS destination of a DMA instruction is wave-uniform base + lane * 16.
// ------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

__device__ __forceinline__ void dma16(const int8_t *src_lane, uint8_t *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gbl_void_t *)src_lane, (lds_void_t *)lds_wave_base, 16, 0, 0);
}

// ------------------------------------------------------------------------
// Stride-1 depthwise 3x3 task: R output rows x 2 adjacent pixels x 4 channels.
// Per INPUT row the 4 pixels ox0-1 .. ox0+2 (4 channel dwords) are byte-transposed (8 v_perm)
// into 4 per-channel windows [v(-1), v(0), v(+1), v(+2)]; each window feeds up to three
// output rows (as filter row ky = input row - output row): pixel ox0 = window . (w0,w1,w2,0),
// pixel ox0+1 = window . (0,w0,w1,w2).  R = 2 shares the transposes of the two middle
// input rows: 5 VALU ops per output byte instead of 6 (and 9 for byte-masked taps).
// `base` = LDS address of (input row oy0-1, pixel ox0-1, this lane's channel group).
// ------------------------------------------------------------------------
// Output rows per depthwise task.  More rows share more input-row transposes (stride 1: 6, 5,
// 4.67, 4.5 VALU ops per output byte for R = 1..4; stride 2: 7.5, 6.75, 6.5) but give fewer,
// bigger tasks, which the fixed workgroup sizes then fill less evenly.  Measured per compiled
// shape (r01, fused and layer-wise kernels alike): R = 3 wins wherever OH % 3 == 0 except on
// the 48-row stride-1 layer and the 12-row stride-2 one; R = 4 never wins.
constexpr int dw_rows_per_task(int OH, int S) {
    if (OH % 3 == 0 && !(S == 1 && OH == 48) && !(S == 2 && OH == 12)) return 3;
    return (OH % 2 == 0) ? 2 : 1;
}
template <int R, int ROW, int C>
__device__ __forceinline__ void dw_s1_task(const uint8_t *base, const uint32_t (&wA)[3][4],
                                           const uint32_t (&wB)[3][4], const int4 Kc,
                                           int (&o0)[R][4], int (&o1)[R][4]) {
#pragma unroll
    for (int j = 0; j < R; ++j) {
        o0[j][0] = o1[j][0] = Kc.x, o0[j][1] = o1[j][1] = Kc.y;
        o0[j][2] = o1[j][2] = Kc.z, o0[j][3] = o1[j][3] = Kc.w;
    }
#pragma unroll
    for (int r = 0; r < R + 2; ++r) {
        const uint32_t s0 = *(const uint32_t *)(base + r * ROW);
        const uint32_t s1 = *(const uint32_t *)(base + r * ROW + C);
        const uint32_t s2 = *(const uint32_t *)(base + r * ROW + 2 * C);
        const uint32_t s3 = *(const uint32_t *)(base + r * ROW + 3 * C);
        const uint32_t ab_lo = __builtin_amdgcn_perm(s1, s0, 0x05010400u);
        const uint32_t ab_hi = __builtin_amdgcn_perm(s1, s0, 0x07030602u);
        const uint32_t cd_lo = __builtin_amdgcn_perm(s3, s2, 0x05010400u);
        const uint32_t cd_hi = __builtin_amdgcn_perm(s3, s2, 0x07030602u);
        uint32_t win[4];
        win[0] = __builtin_amdgcn_perm(cd_lo, ab_lo, 0x05040100u);
        win[1] = __builtin_amdgcn_perm(cd_lo, ab_lo, 0x07060302u);
        win[2] = __builtin_amdgcn_perm(cd_hi, ab_hi, 0x05040100u);
        win[3] = __builtin_amdgcn_perm(cd_hi, ab_hi, 0x07060302u);
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int ky = r - j; // filter row this input row plays for output row j
            if (ky >= 0 && ky <= 2) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (ky == 0) { // this accumulator's first tap (compile-time)
                        o0[j][k] = sdot4_first(win[k], wA[0][k], o0[j][k]);
                        o1[j][k] = sdot4_first(win[k], wB[0][k], o1[j][k]);
                    } else {
                        o0[j][k] = sdot4(win[k], wA[ky][k], o0[j][k]);
                        o1[j][k] = sdot4(win[k], wB[ky][k], o1[j][k]);
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------
// Stride-2 depthwise 3x3 task: R output rows x 2 adjacent output pixels x 4 channels.
// Output pixel ox0 reads input pixels 2ox0-1 .. 2ox0+1, pixel ox0+1 reads 2ox0+1 .. 2ox0+3: per
// INPUT row five channel dwords s0..s4.  s0..s3 are byte-transposed into per-channel windows
// [v0,v1,v2,v3] as in the stride-1 task (8 v_perm); one more v_perm per channel builds
// [v2,v3,v4,0] from the window and s4, so both pixels use the same weight dword (w0,w1,w2,0)
// in one real 3-MAC dot4 each.  With R = 2 the middle input row (2oy0+1) serves both output
// rows: 6.75 VALU ops per output byte instead of 9 for byte-masked taps.
// `base` = LDS address of (input row 2oy0-1, pixel 2ox0-1, this lane's channel group).
// ------------------------------------------------------------------------
template <int R, int ROW, int C>
__device__ __forceinline__ void dw_s2_task(const uint8_t *base, const uint32_t (&wA)[3][4], const int4 Kc,
                                           int (&o0)[R][4], int (&o1)[R][4]) {
#pragma unroll
    for (int j = 0; j < R; ++j) {
        o0[j][0] = o1[j][0] = Kc.x, o0[j][1] = o1[j][1] = Kc.y;
        o0[j][2] = o1[j][2] = Kc.z, o0[j][3] = o1[j][3] = Kc.w;
    }
#pragma unroll
    for (int r = 0; r < 2 * R + 1; ++r) {
        const uint32_t s0 = *(const uint32_t *)(base + r * ROW);
        const uint32_t s1 = *(const uint32_t *)(base + r * ROW + C);
        const uint32_t s2 = *(const uint32_t *)(base + r * ROW + 2 * C);
        const uint32_t s3 = *(const uint32_t *)(base + r * ROW + 3 * C);
        const uint32_t s4 = *(const uint32_t *)(base + r * ROW + 4 * C);
        const uint32_t ab_lo = __builtin_amdgcn_perm(s1, s0, 0x05010400u);
        const uint32_t ab_hi = __builtin_amdgcn_perm(s1, s0, 0x07030602u);
        const uint32_t cd_lo = __builtin_amdgcn_perm(s3, s2, 0x05010400u);
        const uint32_t cd_hi = __builtin_amdgcn_perm(s3, s2, 0x07030602u);
        uint32_t win[4], winb[4];
        win[0] = __builtin_amdgcn_perm(cd_lo, ab_lo, 0x05040100u);
        win[1] = __builtin_amdgcn_perm(cd_lo, ab_lo, 0x07060302u);
        win[2] = __builtin_amdgcn_perm(cd_hi, ab_hi, 0x05040100u);
        win[3] = __builtin_amdgcn_perm(cd_hi, ab_hi, 0x07060302u);
        // [v2, v3, v4 (= byte k of s4), 0]
        winb[0] = __builtin_amdgcn_perm(s4, win[0], 0x0c040302u);
        winb[1] = __builtin_amdgcn_perm(s4, win[1], 0x0c050302u);
        winb[2] = __builtin_amdgcn_perm(s4, win[2], 0x0c060302u);
        winb[3] = __builtin_amdgcn_perm(s4, win[3], 0x0c070302u);
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int ky = r - 2 * j; // filter row this input row plays for output row j
            if (ky >= 0 && ky <= 2) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (ky == 0) {
                        o0[j][k] = sdot4_first(win[k], wA[0][k], o0[j][k]);
                        o1[j][k] = sdot4_first(winb[k], wA[0][k], o1[j][k]);
                    } else {
                        o0[j][k] = sdot4(win[k], wA[ky][k], o0[j][k]);
                        o1[j][k] = sdot4(winb[k], wA[ky][k], o1[j][k]);
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------
// FAST PATH 1 -- DepthwiseConv2D 3x3, SAME, NHWC, C % 4 == 0, weight zp == 0.
// (src/ops/depthwise_conv_2d.rs:28-105; person_detect ops 1,3,5,...,25)
//
// HBM-bound by construction: every input byte is read from HBM once, every output
// byte written once.  One workgroup owns G whole images per step and double-buffers
// them in LDS:
//   stage  : one DMA instruction per image row (W*C <= 1 KiB) into an LDS tile whose
//            1-pixel halo ring was pre-filled with izp once (padding == izp makes border
//            pixels identical to interior ones).  The DMAs of step i+1 are issued right
//            after the barrier of step i and fly during its compute.
//   compute: lane = (pixel, 4-channel group).  The 4-channel group of a lane never
//            changes, so its 9 tap-weight dwords live in VGPRs as 36 byte-masked
//            copies: acc[k] += sdot4(v, w & (0xff << 8k)) is one VALU op per MAC with
//            no unpacking of either operand.
//   store  : one dword (4 channels) per lane, consecutive lanes = consecutive addresses.
// LDS row layout: [LP pad][W*C bytes][LP pad], LP = max(C,16): rows start 16-byte
// aligned and tap (ky,kx) of a lane is the constant offset ky*ROW + kx*C from its base.
// One barrier per step: after it, every wave has finished reading the other buffer
// (safe to overwrite) and every wave's DMAs into this buffer have landed.
// ------------------------------------------------------------------------
template <int H, int W, int C, int S, int G, int NTHR, bool MG, uint32_t XR4>
__global__ __launch_bounds__(NTHR) void dw3x3_nhwc(const int8_t *__restrict__ in,
                                                  int8_t *__restrict__ out, DwFastArgs p,
                                                  int batch) {
    constexpr int C4 = C / 4;
    constexpr int OH = (H + S - 1) / S, OW = (W + S - 1) / S;
    constexpr int LP = C < 16 ? 16 : C;
    constexpr int ROWB = W * C;                   // payload bytes per image row
    constexpr int ROW = LP + ROWB + LP;           // bytes per LDS row
    constexpr int TILE = (H + 2) * ROW;           // bytes per image tile (1 halo row above/below)
    constexpr int BUF = G * TILE;                 // one staging buffer (two are allocated)
    constexpr int IMG = H * ROWB;                 // bytes per input image
    constexpr int ROWCH = ROWB / 16;              // 16-byte chunks (= DMA lanes) per row
    constexpr int NROWS = G * H;                  // DMA instructions per step
    constexpr int NWAVE = NTHR / 64;
    static_assert(NTHR % C4 == 0, "channel group of a lane must be loop-invariant");
    static_assert(ROWB % 16 == 0 && ROWCH <= 64, "one DMA instruction per row");

    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // both buffers := izp, once; the DMAs only ever rewrite the interiors
    for (int i = tid; i < 2 * BUF / 16; i += NTHR)
        ((uint4 *)lds)[i] = make_uint4(p.izp4, p.izp4, p.izp4, p.izp4);

    // per-lane constants of this lane's channel group
    const int cg = tid & (C4 - 1);
    // S == 2: 9 tap dwords as 36 byte-masked copies (one sdot4 per MAC, no unpacking).
    // S == 1: per filter row and channel the 3 taps as ONE dword (w0,w1,w2,0) and its
    //         shifted twin (0,w0,w1,w2): a 4-pixel window transposed to per-channel dwords
    //         then yields TWO adjacent outputs with two real 3-MAC sdot4s.
    uint32_t wA[3][4], wB[3][4]; // (w0,w1,w2,0) and (0,w0,w1,w2) per filter row and channel; wB: stride 1 only
    {
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const uint32_t w0 = ((const uint32_t *)p.w)[(ky * 3 + 0) * C4 + cg];
        const uint32_t w1 = ((const uint32_t *)p.w)[(ky * 3 + 1) * C4 + cg];
        const uint32_t w2 = ((const uint32_t *)p.w)[(ky * 3 + 2) * C4 + cg];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            wA[ky][k] = ((w0 >> (8 * k)) & 0xffu) | (((w1 >> (8 * k)) & 0xffu) << 8) |
                        (((w2 >> (8 * k)) & 0xffu) << 16);
            wB[ky][k] = wA[ky][k] << 8;
        }
    }
    }
    const float4 A = ((const float4 *)p.A)[cg], Sc = ((const float4 *)p.S)[cg];
    const int4 Kc = magic4<MG>(((const int4 *)p.Kc)[cg]);
    __syncthreads(); // halo fill complete before any DMA lands

    auto stage = [&](int st, int buf) {
#pragma unroll
        for (int k = 0; k < (NROWS + NWAVE - 1) / NWAVE; ++k) {
            const int r = k * NWAVE + wave;           // wave-uniform row of the step
            const int g = r / H, y = r % H;
            if (r < NROWS && st * G + g < batch && lane < ROWCH)
                dma16(in + ((size_t)(st * G + g) * IMG + y * ROWB + lane * 16),
                      lds + buf * BUF + g * TILE + (y + 1) * ROW + LP);
        }
    };

    const int nsteps = (batch + G - 1) / G;
    int step = blockIdx.x, cur = 0;
    if (step < nsteps) stage(step, 0);

    for (; step < nsteps; step += gridDim.x, cur ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's DMAs (and old stores) done
        __syncthreads();                                  // ... and everyone else's
        const int next = step + gridDim.x;
        if (next < nsteps) stage(next, cur ^ 1);          // flies during the compute below

        const uint8_t *tile = lds + cur * BUF;
        uint32_t *dst = (uint32_t *)out + (size_t)step * G * OH * OW * C4;
        {
            // task = R output rows x 2 adjacent pixels x 4 channels (see dw_s1_task)
            constexpr int R = dw_rows_per_task(OH, S);
            constexpr int OWP = (OW + 1) / 2, OHR = OH / R;
            constexpr int TASKS = G * OHR * OWP * C4;
            constexpr int NTASK = (TASKS + NTHR - 1) / NTHR;
            const int gvalid = min(G, batch - step * G);
#pragma unroll 1
            for (int i = 0; i < NTASK; ++i) {
                const int t = tid + NTHR * i;
                const int pp = t / C4;
                const int g = pp / (OHR * OWP), rem = pp % (OHR * OWP);
                const int oy0 = R * (rem / OWP), ox0 = 2 * (rem % OWP);
                if (t < TASKS && g < gvalid) {
                    int o0[R][4], o1[R][4];
                    const uint8_t *base = tile + g * TILE + (oy0 * S) * ROW + LP + (ox0 * S - 1) * C + cg * 4;
                    if constexpr (S == 2) dw_s2_task<R, ROW, C>(base, wA, Kc, o0, o1);
                    else dw_s1_task<R, ROW, C>(base, wA, wB, Kc, o0, o1);
#pragma unroll
                    for (int j = 0; j < R; ++j) {
                        uint32_t *dp = dst + ((size_t)(g * OH + oy0 + j) * OW + ox0) * C4 + cg;
                        dp[0] = pack4x<XR4>(requant_t<MG>(o0[j][0], A.x, Sc.x, p.lo_f, p.hi_f), requant_t<MG>(o0[j][1], A.y, Sc.y, p.lo_f, p.hi_f),
                                      requant_t<MG>(o0[j][2], A.z, Sc.z, p.lo_f, p.hi_f), requant_t<MG>(o0[j][3], A.w, Sc.w, p.lo_f, p.hi_f));
                        if (ox0 + 1 < OW)
                            dp[C4] = pack4x<XR4>(requant_t<MG>(o1[j][0], A.x, Sc.x, p.lo_f, p.hi_f), requant_t<MG>(o1[j][1], A.y, Sc.y, p.lo_f, p.hi_f),
                                           requant_t<MG>(o1[j][2], A.z, Sc.z, p.lo_f, p.hi_f), requant_t<MG>(o1[j][3], A.w, Sc.w, p.lo_f, p.hi_f));
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------
// FAST PATH 2 -- DepthwiseConv2D 3x3 stride 2 SAME with ONE input channel and DM
// output channels (the network stem: person_detect op 0, 96x96x1 -> 48x48x8).
// (src/ops/depthwise_conv_2d.rs:67: every output channel reads input channel 0.)
//
// A lane produces two horizontally adjacent output pixels x DM=8 channels = one
// 16-byte store.  Per filter row it reads two LDS dwords, builds the two 3-tap
// windows with one v_perm and one shift, and issues one sdot4 per (pixel, channel,
// row) against wave-uniform weight dwords [w(ky,0,c), w(ky,1,c), w(ky,2,c), 0] that
// live in SGPRs.  Staging: the image is copied verbatim (contiguous 1 KiB DMAs) between
// two izp rows; the only tap that is not covered by those rows, column -1 of the first
// pixel pair, is patched with a select.
// F32IN = true fuses the model-boundary quantisation (M::predict: Tensor::quantize, lib.rs:189,
// src/quantize.rs:16-18) into the staging: `in` then points to f32 pixels, each thread loads the
// next step's float4s into registers before the compute of this step, quantises them afterwards
// (true division, roundf, saturating cast -- the arithmetic of quantize_f32) and writes the int8
// tile itself, so the 4x larger f32 image crosses HBM once and no int8 copy of it ever does.
// ------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int H, int W, int G, bool MG, uint32_t XR4, bool F32IN>
__global__ __launch_bounds__(256) void dw3x3_stem8(const int8_t *__restrict__ in,
                                                   int8_t *__restrict__ out, DwStemArgs p,
                                                   int batch) {
    constexpr int DM = 8, S = 2;
    constexpr int OH = (H + S - 1) / S, OW = (W + S - 1) / S;
    constexpr int GUARD = 16;                     // the j == 0 lanes read 4 bytes before a row
    constexpr int TILE = GUARD + (H + 2) * W;     // [guard][izp row][H rows][izp row]
    constexpr int BUF = G * TILE;
    constexpr int IMG = H * W;
    constexpr int NI = IMG / 1024;                // 1 KiB DMA instructions per image
    constexpr int PAIRS = OW / 2;                 // lane tasks per output row
    constexpr int TASKS = G * OH * PAIRS;
    constexpr int NTASK = (TASKS + 255) / 256;
    static_assert(IMG % 1024 == 0 && W % 16 == 0 && OW % 2 == 0 && TILE % 16 == 0, "stem geometry");

    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 2 * BUF / 16; i += 256)
        ((uint4 *)lds)[i] = make_uint4(p.izp4, p.izp4, p.izp4, p.izp4);
    __syncthreads();

    auto stage = [&](int st, int buf) {
#pragma unroll
        for (int k = 0; k < (G * NI + 3) / 4; ++k) {
            const int r = k * 4 + wave;           // wave-uniform 1 KiB piece of the step
            const int g = r / NI, c = r % NI;
            if (r < G * NI && st * G + g < batch)
                dma16(in + ((size_t)(st * G + g) * IMG + c * 1024 + lane * 16),
                      lds + buf * BUF + g * TILE + GUARD + W + c * 1024);
        }
    };

    // f32 staging: float4 k of this thread is pixels 4*(k*256 + tid) .. +3 of the step's G images
    constexpr int NF = F32IN ? G * IMG / 4 / 256 : 1;
    static_assert(!F32IN || (G * IMG) % 1024 == 0, "f32 staging geometry");
    f32x4 pre[NF];
    auto load_f32 = [&](int st) {
        const f32x4 *src = (const f32x4 *)in;
#pragma unroll
        for (int k = 0; k < NF; ++k) {
            const int idx = k * 256 + tid, g = idx / (IMG / 4);
            const size_t img = (size_t)st * G + g;
            const size_t at = (img < (size_t)batch ? img : (size_t)batch - 1) * (IMG / 4) + idx % (IMG / 4);
            pre[k] = src[at]; // clamped, unconditional: a ragged last step re-reads the last image
        }
    };
    auto store_f32 = [&](int buf) {
#pragma unroll
        for (int k = 0; k < NF; ++k) {
            const int idx = k * 256 + tid, g = idx / (IMG / 4), c = idx % (IMG / 4);
            int q[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t = __fadd_rn(__fdiv_rn(pre[k][e], p.in_scale), p.in_zp_f);
                const float r = __fadd_rn(t, __builtin_copysignf(0x1.fffffep-2f, t));
                q[e] = (r != r) ? 0 : (int)__builtin_amdgcn_fmed3f(r, p.in_sat_lo, p.in_sat_hi);
            }
            *(uint32_t *)(lds + buf * BUF + g * TILE + GUARD + W + c * 4) = pack4(q[0], q[1], q[2], q[3]) ^ p.in_xr4;
        }
    };

    const int nsteps = (batch + G - 1) / G;
    int step = blockIdx.x, cur = 0;
    if constexpr (F32IN) {
        if (step < nsteps) {
            load_f32(step);
            store_f32(0);
        }
    } else {
        if (step < nsteps) stage(step, 0);
    }

    for (; step < nsteps; step += gridDim.x, cur ^= 1) {
        if constexpr (!F32IN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int next = step + gridDim.x;
        if constexpr (F32IN) {
            if (next < nsteps) load_f32(next);            // in flight during the compute below
        } else {
            if (next < nsteps) stage(next, cur ^ 1);
        }

        const uint8_t *tile = lds + cur * BUF;
        uint4 *dst = (uint4 *)out + (size_t)step * G * OH * PAIRS;
        const int nvalid = min(G, batch - step * G) * OH * PAIRS;
#pragma unroll 1
        for (int i = 0; i < NTASK; ++i) {
            const int o = tid + 256 * i;
            if (o < TASKS && o < nvalid) {
                const int g = o / (OH * PAIRS), rem = o % (OH * PAIRS);
                const int oy = rem / PAIRS, j = rem % PAIRS;
                // pixels ox = 2j, 2j+1 need input cols 4j-1 .. 4j+3 of rows 2oy-1 .. 2oy+1 =
                // tile rows 2oy .. 2oy+2 (tile row 0 is the izp row): dwords at cols 4j-4 and 4j
                const uint32_t *rowp = (const uint32_t *)(tile + g * TILE + GUARD + (oy * S) * W + 4 * j - 4);
                uint32_t ta[3], tb[3];
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    uint32_t d0 = rowp[ky * (W / 4)];
                    const uint32_t d1 = rowp[ky * (W / 4) + 1];
                    d0 = j == 0 ? p.izp4 : d0;                          // column -1 is padding
                    ta[ky] = __builtin_amdgcn_perm(d0, d1, 0x0c010007u); // [d0.b3, d1.b0, d1.b1, 0]
                    tb[ky] = d1 >> 8;                                    // [d1.b1, d1.b2, d1.b3, 0]
                }
                int qa[DM], qb[DM];
#pragma unroll
                for (int c = 0; c < DM; ++c) {
                    int a = p.Kc[c] + (MG ? MF_MAGIC_I : 0), b = a;
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
                        a = sdot4(ta[ky], p.wrow[ky][c], a);
                        b = sdot4(tb[ky], p.wrow[ky][c], b);
                    }
                    qa[c] = requant_t<MG>(a, p.A[c], p.S[c], p.lo_f, p.hi_f);
                    qb[c] = requant_t<MG>(b, p.A[c], p.S[c], p.lo_f, p.hi_f);
                }
                uint4 v;
                v.x = pack4x<XR4>(qa[0], qa[1], qa[2], qa[3]);
                v.y = pack4x<XR4>(qa[4], qa[5], qa[6], qa[7]);
                v.z = pack4x<XR4>(qb[0], qb[1], qb[2], qb[3]);
                v.w = pack4x<XR4>(qb[4], qb[5], qb[6], qb[7]);
                dst[o] = v;
            }
        }
        if constexpr (F32IN) {
            if (next < nsteps) store_f32(cur ^ 1); // the other buffer: last read before this step's barrier
        }
    }
}

// ------------------------------------------------------------------------
// FAST PATH 2b -- DepthwiseConv2D with ONE input channel, up to 8 output channels, any filter
// size / stride / padding (speech.tflite op 1: 49x40x1 -> 25x20x8, 10x8 filter, stride 2).
// (src/ops/depthwise_conv_2d.rs:67: every output channel reads input channel 0.)
// With one input channel the taps of a filter row are CONSECUTIVE input bytes, so a row of the
// window is KG = ceil(KW/4) dwords and every (filter row, 4-tap group, output channel) is one
// real 4-MAC v_dot4: 160 dot4 per output pixel for the 10x8 filter instead of 640 multiply-adds.
//   tile   : one workgroup stages one image in LDS inside an izp halo (SAME padding needs no
//            per-tap test); rows are padded to a multiple of 4 bytes (+4 of over-read room).
//   window : a thread owns one output pixel; its row start (ox*sw) is not dword aligned in
//            general, so it reads KG+1 aligned dwords and shifts with v_alignbyte.
//   weights: packed on the host as [ky][group][8 channels] dwords (zero beyond KW / N), read
//            from LDS as broadcast b128s.
// Every input byte is read from HBM once; the kernel is VALU-bound (54 MAC per byte).
// ------------------------------------------------------------------------
template <bool MG, uint32_t XR4>
__global__ __launch_bounds__(512) void dw_c1_lds(const int8_t *__restrict__ in, int8_t *__restrict__ out,
                                                 DwC1Args p, size_t batch) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int shy = p.pad_same ? (p.KH - 1) / 2 : 0, shx = p.pad_same ? (p.KW - 1) / 2 : 0;
    // halo'd tile covering every tap of every output pixel
    const int TH = (p.OH - 1) * p.sh + p.KH;
    const int TWP = p.TWP, KG = p.KG;
    const int tile_bytes = (TH * TWP + 4 + 15) & ~15;
    uint32_t *wl = (uint32_t *)(lds + tile_bytes);
    const int tid = threadIdx.x;
    for (int i = tid; i < p.KH * KG * 8; i += 512) wl[i] = p.wpack[i];
    for (int i = TH * TWP + tid; i < tile_bytes; i += 512) lds[i] = 0; // over-read room past the last row
    int Kc[8];
    float A[8], S[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        Kc[c] = (c < p.N ? p.Kc[c] : 0) + (MG ? MF_MAGIC_I : 0);
        A[c] = c < p.N ? p.A[c] : 0.0f;
        S[c] = c < p.N ? p.S[c] : 0.0f;
    }
    // tile element i = tid + 512 e comes from image byte gofs[e] (or is halo: -1); the same for
    // every image, so the divisions happen once and an image's loads are issued back to back
    constexpr int MAXE = 8;
    int gofs[MAXE];
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
        const int i = tid + 512 * e;
        const int ty = i / TWP, tx = i - ty * TWP;
        const int iy = ty - shy, ix = tx - shx;
        gofs[e] = (i < TH * TWP && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) ? iy * p.W + ix : -1;
    }
    for (size_t img = blockIdx.x; img < batch; img += gridDim.x) {
        const int8_t *x = in + img * (size_t)p.H * p.W;
        int8_t v[MAXE];
#pragma unroll
        for (int e = 0; e < MAXE; ++e) v[e] = x[gofs[e] < 0 ? 0 : gofs[e]]; // clamped, unconditional
        __syncthreads(); // previous image fully consumed
#pragma unroll
        for (int e = 0; e < MAXE; ++e)
            if (tid + 512 * e < TH * TWP) ((int8_t *)lds)[tid + 512 * e] = gofs[e] < 0 ? (int8_t)p.izp : v[e];
        for (int i = tid + 512 * MAXE; i < TH * TWP; i += 512) { // tiles above 4 KiB: the plain way
            const int ty = i / TWP, tx = i - ty * TWP;
            const int iy = ty - shy, ix = tx - shx;
            // columns TW .. TWP-1 only ever meet zero weights; any finite value will do
            ((int8_t *)lds)[i] = (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) ? x[iy * p.W + ix] : (int8_t)p.izp;
        }
        __syncthreads();
        for (int o = tid; o < p.OH * p.OW; o += 512) {
            const int oy = o / p.OW, ox = o - oy * p.OW;
            int acc[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c] = Kc[c];
            const int base0 = (oy * p.sh) * TWP + ox * p.sw;
            for (int ky = 0; ky < p.KH; ++ky) {
                const int base = base0 + ky * TWP;
                const uint32_t *row = (const uint32_t *)(lds + (base & ~3));
                const uint32_t sh = (uint32_t)(base & 3);
                uint32_t lo = row[0];
                for (int g = 0; g < KG; ++g) {
                    const uint32_t hi = row[g + 1];
                    const uint32_t v = __builtin_amdgcn_alignbyte(hi, lo, sh); // bytes base+4g .. base+4g+3
                    lo = hi;
                    const uint4 w0 = *(const uint4 *)(wl + (ky * KG + g) * 8);
                    const uint4 w1 = *(const uint4 *)(wl + (ky * KG + g) * 8 + 4);
                    acc[0] = sdot4(v, w0.x, acc[0]), acc[1] = sdot4(v, w0.y, acc[1]);
                    acc[2] = sdot4(v, w0.z, acc[2]), acc[3] = sdot4(v, w0.w, acc[3]);
                    acc[4] = sdot4(v, w1.x, acc[4]), acc[5] = sdot4(v, w1.y, acc[5]);
                    acc[6] = sdot4(v, w1.z, acc[6]), acc[7] = sdot4(v, w1.w, acc[7]);
                }
            }
            int q[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) q[c] = requant_t<MG>(acc[c], A[c], S[c], p.lo_f, p.hi_f);
            int8_t *dst = out + (img * (size_t)p.OH * p.OW + o) * p.N;
            if (p.N == 8) {
                *(uint2 *)dst = make_uint2(pack4x<XR4>(q[0], q[1], q[2], q[3]), pack4x<XR4>(q[4], q[5], q[6], q[7]));
            } else {
                for (int c = 0; c < p.N; ++c) dst[c] = (int8_t)(q[c] ^ (int)(XR4 & 0xffu));
            }
        }
    }
}

// ------------------------------------------------------------------------
// FAST PATH 3 -- Conv2D 1x1 stride 1 (pointwise) as an int8 MFMA GEMM.
// (src/ops/conv_2d.rs:28-108 with KH = KW = 1; person_detect ops 2,4,...,26)
//
//   D[out channel][pixel] = sum_k Wt[out channel][k] * X[pixel][k]
//   v_mfma_i32_16x16x64_i8:  A = weights (rows = 16 out channels), B = pixels
//   (cols = 16 pixels), so that a lane ends up holding 4 CONSECUTIVE out channels of
//   one pixel per 16x16 tile and can store them packed.
//
// In NHWC with the batch outermost, the activations of the whole batch ARE the
// row-major [pixels][K] matrix -- no im2col, no LDS staging of activations: lane
// (p = lane&15, g = lane>>4) loads its 16 bytes of operand B straight from HBM in
// MFMA layout, and every wave-level load instruction covers whole contiguous 1 KiB.
// K < 64 (early layers) would waste the 64-deep MFMA k-span, so 64/K pixel groups
// share one B register and the host pre-builds Q = 64/K zero-padded copies A_q of the
// weights, each selecting one group's k-bytes (block-diagonal trick): loads stay
// 16 B/lane fully coalesced at every K.  The weight rows are permuted on the host so
// that tile tt row 4g+j is channel base + g*(NB/4) + 4*tt + j: after TB tiles a lane
// holds NB/4 consecutive output bytes -> one 4/8/16-byte store.
// N > 64 is split over the waves of the workgroup (NSPLIT = N/64), which all read the
// same pixels (L1/L2 hits).  HBM-bound: MFMA work is ~1/8 of the memory time.
// ------------------------------------------------------------------------
#ifndef MF_PW_U_LO
#define MF_PW_U_LO 2
#define MF_PW_U_MID 4
#define MF_PW_U_HI 2
#endif
// chunks each wave keeps in flight (loads of the next U issued before the first use)
constexpr int pw_chunks_in_flight(int K) { return K >= 256 ? MF_PW_U_HI : (K >= 64 ? MF_PW_U_MID : MF_PW_U_LO); }
template <int K, int N, bool MG, uint32_t XR4>
__global__ __launch_bounds__(256) void pw_mfma(const int8_t *__restrict__ in,
                                               int8_t *__restrict__ out, PwArgs p,
                                               long long npix) {
    constexpr int NB = N < 64 ? N : 64;        // channels per wave block
    constexpr int TB = NB / 16;                // 16-channel MFMA tiles per block
    constexpr int NSPLIT = N / NB;             // waves sharing one pixel chunk
    constexpr int KS = K < 64 ? 1 : K / 64;    // 64-deep k steps
    constexpr int Q = K < 64 ? 64 / K : 1;     // pixel groups per B register
    constexpr int CPIX = (K < 64) ? (1024 / K) : 16; // pixels per chunk
    constexpr int SLOTS = 4 / NSPLIT;          // pixel chunks processed concurrently per WG
    // chunks per loop iteration: their loads are all issued before the first use, so a wave
    // keeps U*KS KiB in flight (one 16-pixel chunk per iteration left HBM latency exposed)
    constexpr int U = pw_chunks_in_flight(K);
    // narrow outputs (N < 64) go through a per-wave LDS patch so that every global store is
    // 16 bytes per lane and a wave writes whole contiguous KiB
    constexpr bool XPOSE = TB < 4;
    constexpr int CBYTES = CPIX * N;           // output bytes per chunk (XPOSE: 1 or 2 KiB)
    static_assert(N % 16 == 0 && (K == 8 || K % 16 == 0), "pw_mfma shape");
    static_assert(!XPOSE || (NSPLIT == 1 && CBYTES % 1024 == 0), "transposed store geometry");

    __shared__ __attribute__((aligned(16))) uint8_t patch[XPOSE ? 4 * CBYTES : 16];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pcol = lane & 15, g = lane >> 4;
    const int blk = wave % NSPLIT;             // which 64-channel block this wave owns
    const int slot = wave / NSPLIT;

    // operand A (weights), pre-arranged by the host: [blk][q][tt][ks][lane] x 16 bytes
    v4i Aw[Q][TB][KS];
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
        for (int tt = 0; tt < TB; ++tt)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                Aw[q][tt][ks] = ((const v4i *)p.wprep)[((((size_t)blk * Q + q) * TB + tt) * KS + ks) * 64 + lane];
    // epilogue constants of this lane's channels: block base + g*(NB/4) + 4*tt + j
    float4 cA[TB], cS[TB];
    int4 cK[TB];
#pragma unroll
    for (int tt = 0; tt < TB; ++tt) {
        const int ch = blk * NB + g * (NB / 4) + 4 * tt;
        cA[tt] = *(const float4 *)(p.A + ch);
        cS[tt] = *(const float4 *)(p.S + ch);
        cK[tt] = magic4<MG>(*(const int4 *)(p.Kc + ch));
    }

    const long long nchunks = (npix + CPIX - 1) / CPIX;
    const long long stride = (long long)gridDim.x * SLOTS * U;
    long long chunk0 = ((long long)blockIdx.x * SLOTS + slot) * U;

    // loads are never predicated (clamped instead): see the depthwise staging notes
    auto loadB = [&](long long ch, v4i (&b)[KS]) {
        if constexpr (K >= 64) {
            long long pix = ch * 16 + pcol;
            pix = pix < npix ? pix : npix - 1;
            const int8_t *src = in + pix * K + g * 16;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) b[ks] = *(const v4i *)(src + ks * 64);
        } else if constexpr (K == 32) {
            long long pix = ch * CPIX + (g >> 1) * 16 + pcol;
            pix = pix < npix ? pix : npix - 1;
            b[0] = *(const v4i *)(in + pix * 32 + (g & 1) * 16);
        } else if constexpr (K == 16) {
            long long pix = ch * CPIX + g * 16 + pcol;
            pix = pix < npix ? pix : npix - 1;
            b[0] = *(const v4i *)(in + pix * 16);
        } else { // K == 8: 16 bytes = 2 pixels; npix is even (routing precondition)
            long long pix = ch * CPIX + 2 * (g * 16 + pcol);
            pix = pix + 1 < npix ? pix : npix - 2;
            b[0] = *(const v4i *)(in + pix * 8);
        }
    };

    v4i B[U][KS], Bn[U][KS];
#pragma unroll
    for (int u = 0; u < U; ++u) loadB(min(chunk0 + u, nchunks - 1), B[u]);

    for (; chunk0 < nchunks; chunk0 += stride) {
        const long long nxt = chunk0 + stride;
#pragma unroll
        for (int u = 0; u < U; ++u) loadB(min(nxt + u, nchunks - 1), Bn[u]); // harmless re-read at the tail
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long chunk = chunk0 + u;
            if (chunk < nchunks) { // wave-uniform
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    // pixel (within the chunk) this lane's MFMA column belongs to for sub-group q
                    int lpix;
                    if constexpr (K >= 64) lpix = pcol;
                    else if constexpr (K == 8) lpix = 2 * ((q >> 1) * 16 + pcol) + (q & 1);
                    else lpix = q * 16 + pcol;
                    uint32_t packed[TB];
#pragma unroll
                    for (int tt = 0; tt < TB; ++tt) {
                        // the accumulator starts at Kc (the folded zero-point terms): the MFMA's C
                        // operand does the addition for free
                        v4i acc = {cK[tt].x, cK[tt].y, cK[tt].z, cK[tt].w};
#pragma unroll
                        for (int ks = 0; ks < KS; ++ks)
                            acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(Aw[q][tt][ks], B[u][ks], acc, 0, 0, 0);
                        const int q0 = requant_t<MG>(acc[0], cA[tt].x, cS[tt].x, p.lo_f, p.hi_f);
                        const int q1 = requant_t<MG>(acc[1], cA[tt].y, cS[tt].y, p.lo_f, p.hi_f);
                        const int q2 = requant_t<MG>(acc[2], cA[tt].z, cS[tt].z, p.lo_f, p.hi_f);
                        const int q3 = requant_t<MG>(acc[3], cA[tt].w, cS[tt].w, p.lo_f, p.hi_f);
                        packed[tt] = pack4x<XR4>(q0, q1, q2, q3);
                    }
                    if constexpr (XPOSE) {
                        uint8_t *dstp = patch + wave * CBYTES + lpix * N + g * (NB / 4);
                        if constexpr (TB == 1) *(uint32_t *)dstp = packed[0];
                        else *(uint2 *)dstp = make_uint2(packed[0], packed[1]);
                    } else {
                        const long long pix = chunk * CPIX + lpix;
                        if (pix < npix)
                            *(uint4 *)(out + pix * N + blk * NB + g * 16) =
                                make_uint4(packed[0], packed[1], packed[2], packed[3]);
                    }
                }
                if constexpr (XPOSE) {
                    // same wave wrote the patch; LDS ops of a wave complete in order
                    __builtin_amdgcn_wave_barrier();
                    const long long obase = chunk * (long long)CBYTES;
                    const long long obytes = npix * N;
#pragma unroll
                    for (int j = 0; j < CBYTES / 1024; ++j) {
                        const int off = (j * 64 + lane) * 16;
                        const uint4 v = *(const uint4 *)(patch + wave * CBYTES + off);
                        if (obase + off < obytes) *(uint4 *)(out + obase + off) = v;
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) B[u][ks] = Bn[u][ks];
    }
}

// ------------------------------------------------------------------------
// FAST PATH 3b -- fused DepthwiseConv2D 3x3 -> Conv2D 1x1.
//
// The depthwise kernels are VALU-bound and the large pointwise kernels HBM-bound; running
// them as one kernel removes the depthwise output / pointwise input round trip through HBM
// (38 % of the layer-wise traffic) and lets the depthwise VALU work hide under the
// pointwise output stream.  Per step a workgroup
//   1. has G images staged in LDS by LDS-DMA (double or single buffered, as in FAST PATH 1),
//   2. runs the depthwise conv exactly as dw3x3_nhwc does, but writes its packed int8
//      results to an LDS tile MID laid out [pixel][C] -- which IS the row-major operand
//      matrix the pointwise MFMA kernel reads,
//   3. barrier, then each wave runs pw_mfma's chunk loop with its B operand fetched from MID
//      by ds_read_b128 (same lane -> (pixel, k-block) map) and stores the pointwise outputs
//      to HBM (16 B per lane; narrow N through the per-wave LDS patch).
// Both requantisations stay exactly the reference's (the intermediate tensor is a real int8
// tensor, it just never leaves the CU).  Single-buffered variants issue the next step's DMA
// right after the second barrier, so it still overlaps the pointwise phase.
// ------------------------------------------------------------------------
template <int H, int W, int C, int S, int N, int G, int NTHR, bool DBUF, bool MG, uint32_t XR4>
__global__ __launch_bounds__(NTHR) void dwpw3x3(const int8_t *__restrict__ in,
                                                int8_t *__restrict__ out, DwPwArgs p, int batch) {
    // ---- depthwise geometry (as dw3x3_nhwc) ----
    constexpr int C4 = C / 4;
    constexpr int OH = (H + S - 1) / S, OW = (W + S - 1) / S;
    constexpr int LP = C < 16 ? 16 : C;
    constexpr int ROWB = W * C, ROW = LP + ROWB + LP, TILE = (H + 2) * ROW, BUF = G * TILE;
    constexpr int IMG = H * ROWB, ROWCH = ROWB / 16, NROWS = G * H;
    constexpr int NWAVE = NTHR / 64;
    constexpr int NBUF = DBUF ? 2 : 1;
    constexpr int OPIX = OH * OW;                 // pixels per image after the depthwise
    constexpr int MIDB = G * OPIX * C;            // bytes of the intermediate tensor per step
    static_assert(NTHR % C4 == 0 && ROWB % 16 == 0 && ROWCH <= 64, "depthwise geometry");
    // ---- pointwise geometry (as pw_mfma<K = C, N>) ----
    constexpr int K = C;
    constexpr int NB = N < 64 ? N : 64, TB = NB / 16, NSPLIT = N / NB;
    constexpr int KS = K < 64 ? 1 : K / 64, Q = K < 64 ? 64 / K : 1;
    constexpr int CPIX = (K < 64) ? (1024 / K) : 16;
    constexpr int SLOTS = NWAVE / NSPLIT;
    constexpr bool XPOSE = TB < 4;
    constexpr int CBYTES = CPIX * N;
    static_assert(NWAVE % NSPLIT == 0 && N % 16 == 0 && (K == 8 || K % 16 == 0), "pointwise geometry");
    static_assert(!XPOSE || (NSPLIT == 1 && CBYTES % 1024 == 0), "transposed store geometry");
    static_assert(K != 8 || (OPIX % 2 == 0), "K = 8 loads two pixels per lane");
    // LDS: [staging x NBUF][slack 256][MID (+64 slack)][patch]
    constexpr int MID_OFF = NBUF * BUF + 256;
    constexpr int PATCH_OFF = MID_OFF + MIDB + 64;

    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    for (int i = tid; i < (NBUF * BUF + 256) / 16; i += NTHR)
        ((uint4 *)lds)[i] = make_uint4(p.dw.izp4, p.dw.izp4, p.dw.izp4, p.dw.izp4);

    // ---- depthwise per-lane constants ----
    const int cg = tid & (C4 - 1);
    uint32_t wA[3][4], wB[3][4]; // (w0,w1,w2,0) and (0,w0,w1,w2) per filter row and channel; wB: stride 1 only
    {
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const uint32_t w0 = ((const uint32_t *)p.dw.w)[(ky * 3 + 0) * C4 + cg];
        const uint32_t w1 = ((const uint32_t *)p.dw.w)[(ky * 3 + 1) * C4 + cg];
        const uint32_t w2 = ((const uint32_t *)p.dw.w)[(ky * 3 + 2) * C4 + cg];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            wA[ky][k] = ((w0 >> (8 * k)) & 0xffu) | (((w1 >> (8 * k)) & 0xffu) << 8) |
                        (((w2 >> (8 * k)) & 0xffu) << 16);
            wB[ky][k] = wA[ky][k] << 8;
        }
    }
    }
    const float4 dA = ((const float4 *)p.dw.A)[cg], dS = ((const float4 *)p.dw.S)[cg];
    const int4 dK = magic4<MG>(((const int4 *)p.dw.Kc)[cg]);

    // ---- pointwise per-lane constants ----
    const int pcol = lane & 15, pg = lane >> 4;
    const int blk = wave % NSPLIT, slot = wave / NSPLIT;
    v4i Aw[Q][TB][KS];
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
        for (int tt = 0; tt < TB; ++tt)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                Aw[q][tt][ks] = ((const v4i *)p.pw.wprep)[((((size_t)blk * Q + q) * TB + tt) * KS + ks) * 64 + lane];
    float4 cA[TB], cS[TB];
    int4 cK[TB];
#pragma unroll
    for (int tt = 0; tt < TB; ++tt) {
        const int ch = blk * NB + pg * (NB / 4) + 4 * tt;
        cA[tt] = *(const float4 *)(p.pw.A + ch);
        cS[tt] = *(const float4 *)(p.pw.S + ch);
        cK[tt] = magic4<MG>(*(const int4 *)(p.pw.Kc + ch));
    }
    __syncthreads(); // halo fill complete before any DMA lands

    auto stage = [&](int st, int buf) {
#pragma unroll
        for (int k = 0; k < (NROWS + NWAVE - 1) / NWAVE; ++k) {
            const int r = k * NWAVE + wave;
            const int g = r / H, y = r % H;
            if (r < NROWS && st * G + g < batch && lane < ROWCH)
                dma16(in + ((size_t)(st * G + g) * IMG + y * ROWB + lane * 16),
                      lds + buf * BUF + g * TILE + (y + 1) * ROW + LP);
        }
    };

    uint8_t *mid = lds + MID_OFF;
    const int nsteps = (batch + G - 1) / G;
    int step = blockIdx.x, cur = 0;
    if (step < nsteps) stage(step, 0);

    for (; step < nsteps; step += gridDim.x) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads(); // B1: staged tile complete; previous pointwise phase done with MID
        const int next = step + gridDim.x;
        if constexpr (DBUF) {
            if (next < nsteps) stage(next, cur ^ 1);
        }
        const uint8_t *tile = lds + cur * BUF;
        const int gvalid = min(G, batch - step * G);

        // ---------------- depthwise phase: staged tile -> MID ----------------
        {
            constexpr int R = dw_rows_per_task(OH, S);
            constexpr int OWP = (OW + 1) / 2, OHR = OH / R;
            constexpr int TASKS = G * OHR * OWP * C4, NTASK = (TASKS + NTHR - 1) / NTHR;
#pragma unroll 1
            for (int i = 0; i < NTASK; ++i) {
                const int t = tid + NTHR * i;
                const int pp = t / C4;
                const int g = pp / (OHR * OWP), rem = pp % (OHR * OWP);
                const int oy0 = R * (rem / OWP), ox0 = 2 * (rem % OWP);
                if (t < TASKS && g < gvalid) {
                    int o0[R][4], o1[R][4];
                    const uint8_t *base = tile + g * TILE + (oy0 * S) * ROW + LP + (ox0 * S - 1) * C + cg * 4;
                    if constexpr (S == 2) dw_s2_task<R, ROW, C>(base, wA, dK, o0, o1);
                    else dw_s1_task<R, ROW, C>(base, wA, wB, dK, o0, o1);
#pragma unroll
                    for (int j = 0; j < R; ++j) {
                        uint32_t *dp = (uint32_t *)mid + ((size_t)(g * OH + oy0 + j) * OW + ox0) * C4 + cg;
                        dp[0] = pack4x<XR4>(requant_t<MG>(o0[j][0], dA.x, dS.x, p.dw.lo_f, p.dw.hi_f), requant_t<MG>(o0[j][1], dA.y, dS.y, p.dw.lo_f, p.dw.hi_f),
                                      requant_t<MG>(o0[j][2], dA.z, dS.z, p.dw.lo_f, p.dw.hi_f), requant_t<MG>(o0[j][3], dA.w, dS.w, p.dw.lo_f, p.dw.hi_f));
                        if (ox0 + 1 < OW)
                            dp[C4] = pack4x<XR4>(requant_t<MG>(o1[j][0], dA.x, dS.x, p.dw.lo_f, p.dw.hi_f), requant_t<MG>(o1[j][1], dA.y, dS.y, p.dw.lo_f, p.dw.hi_f),
                                           requant_t<MG>(o1[j][2], dA.z, dS.z, p.dw.lo_f, p.dw.hi_f), requant_t<MG>(o1[j][3], dA.w, dS.w, p.dw.lo_f, p.dw.hi_f));
                    }
                }
            }
        }
        __syncthreads(); // B2: MID complete; everyone is done reading the staged tile
        if constexpr (!DBUF) {
            if (next < nsteps) stage(next, 0); // flies during the pointwise phase
        }

        // ---------------- pointwise phase: MID -> HBM ----------------
        const int npix = gvalid * OPIX;                       // valid pixels of this step
        const int nchunks = (npix + CPIX - 1) / CPIX;
        int8_t *obase = out + (size_t)step * G * OPIX * N;
        // One unit of pointwise work: sub-blocks [QLO, QHI) of a chunk -- compile-time bounds, so the
        // MFMAs and epilogues of a unit stay one straight-line block.
        auto pw_unit = [&](int chunk, auto qlo_c, auto qhi_c) {
            constexpr int QLO = decltype(qlo_c)::value, QHI = decltype(qhi_c)::value;
            v4i B[KS];
            if constexpr (K >= 64) {
                int pix = chunk * 16 + pcol;
                pix = pix < npix ? pix : npix - 1;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) B[ks] = *(const v4i *)(mid + pix * K + pg * 16 + ks * 64);
            } else if constexpr (K == 32) {
                int pix = chunk * CPIX + (pg >> 1) * 16 + pcol;
                pix = pix < npix ? pix : npix - 1;
                B[0] = *(const v4i *)(mid + pix * 32 + (pg & 1) * 16);
            } else if constexpr (K == 16) {
                int pix = chunk * CPIX + pg * 16 + pcol;
                pix = pix < npix ? pix : npix - 1;
                B[0] = *(const v4i *)(mid + pix * 16);
            } else {
                int pix = chunk * CPIX + 2 * (pg * 16 + pcol);
                pix = pix + 1 < npix ? pix : npix - 2;
                B[0] = *(const v4i *)(mid + pix * 8);
            }
#pragma unroll
            for (int q = QLO; q < QHI; ++q) {
                int lpix;
                if constexpr (K >= 64) lpix = pcol;
                else if constexpr (K == 8) lpix = 2 * ((q >> 1) * 16 + pcol) + (q & 1);
                else lpix = q * 16 + pcol;
                uint32_t packed[TB];
#pragma unroll
                for (int tt = 0; tt < TB; ++tt) {
                    v4i acc = {cK[tt].x, cK[tt].y, cK[tt].z, cK[tt].w};
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks)
                        acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(Aw[q][tt][ks], B[ks], acc, 0, 0, 0);
                    packed[tt] = pack4x<XR4>(requant_t<MG>(acc[0], cA[tt].x, cS[tt].x, p.pw.lo_f, p.pw.hi_f),
                                       requant_t<MG>(acc[1], cA[tt].y, cS[tt].y, p.pw.lo_f, p.pw.hi_f),
                                       requant_t<MG>(acc[2], cA[tt].z, cS[tt].z, p.pw.lo_f, p.pw.hi_f),
                                       requant_t<MG>(acc[3], cA[tt].w, cS[tt].w, p.pw.lo_f, p.pw.hi_f));
                }
                if constexpr (XPOSE) {
                    uint8_t *dstp = lds + PATCH_OFF + wave * CBYTES + lpix * N + pg * (NB / 4);
                    if constexpr (TB == 1) *(uint32_t *)dstp = packed[0];
                    else *(uint2 *)dstp = make_uint2(packed[0], packed[1]);
                } else {
                    const int pix = chunk * CPIX + lpix;
                    if (pix < npix)
                        *(uint4 *)(obase + (size_t)pix * N + blk * NB + pg * 16) =
                            make_uint4(packed[0], packed[1], packed[2], packed[3]);
                }
            }
            if constexpr (XPOSE) {
                __builtin_amdgcn_wave_barrier();
                const int cb = chunk * CBYTES, obytes = npix * N;
                // sub-blocks [QLO, QHI) are the bytes [QLO, QHI) * CBYTES / Q of the chunk image
                constexpr int LO = QLO * (CBYTES / Q), HI = QHI * (CBYTES / Q);
#pragma unroll
                for (int j = 0; j < (HI - LO + 1023) / 1024; ++j) {
                    const int off = LO + (j * 64 + lane) * 16;
                    if (off < HI) {
                        const uint4 v = *(const uint4 *)(lds + PATCH_OFF + wave * CBYTES + off);
                        if (cb + off < obytes) *(uint4 *)(obase + cb + off) = v;
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        };
        // Whole chunks dealt round-robin leave the last round partly empty (18 chunks on 8 waves:
        // 3 rounds for 2.25 rounds of work).  For K < 64 a chunk has Q >= 2 independent sub-blocks,
        // so the unit of work is HALF a chunk (the B operand is loaded by both halves' waves): 36
        // units on 8 waves = 4.5 half-rounds -> 5.  With an even number of slots a wave always
        // draws the same half, i.e. it runs only one of the two code copies.
        using std::integral_constant;
        // (measured per shape, r01: -6 % and -9 % on the K = 16 and K = 32 stride-2 pairs, neutral on
        // 24x24x32 stride 1; the K = 8 pair got 13 % slower, so it keeps whole chunks)
        if constexpr (Q >= 2 && SLOTS % 2 == 0 && K >= 16) {
            for (int u = slot; u < 2 * nchunks; u += SLOTS) {
                if ((u & 1) == 0) pw_unit(u >> 1, integral_constant<int, 0>{}, integral_constant<int, Q / 2>{});
                else pw_unit(u >> 1, integral_constant<int, Q / 2>{}, integral_constant<int, Q>{});
            }
        } else {
            for (int chunk = slot; chunk < nchunks; chunk += SLOTS)
                pw_unit(chunk, integral_constant<int, 0>{}, integral_constant<int, Q>{});
        }
        if constexpr (DBUF) cur ^= 1;
    }
}

// ------------------------------------------------------------------------
// FAST PATH 3c -- fused network tail: AveragePool2D whose output is 1x1
// (src/ops/average_pool_2d.rs:29-66) -> Conv2D 1x1 with N <= 8 outputs
// (src/ops/conv_2d.rs:28-108) -> [Reshape] -> Softmax over the N values
// (src/ops/softmax.rs:15-27).  person_detect ops 27..30: 2304 bytes in, 2 bytes out.
// One wavefront per inference: lane l owns channels 4l..4l+3 (+256 per extra pass), sums its
// taps with byte-masked sdot4, requantises the pool (an int8 tensor, like the reference's),
// takes its share of the N dot products, butterfly-reduces them across the wave, and lanes
// 0..N-1 finish the head epilogue and the table softmax.  Every intermediate tensor keeps the
// reference's exact arithmetic; they just stay in registers.
// ------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(256) void tail_pool_head_softmax(const int8_t *__restrict__ in,
                                                              int8_t *__restrict__ out, TailArgs p,
                                                              size_t batch) {
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * 256) >> 6;
    const int C4 = p.C >> 2;
    const size_t img_elems = (size_t)p.H * p.W * p.C;
    for (size_t b = wave; b < batch; b += nwaves) {
        const int8_t *x = in + b * img_elems;
        int dot[N], vs = 0;
#pragma unroll
        for (int n = 0; n < N; ++n) dot[n] = 0;
        for (int c4 = lane; c4 < C4; c4 += 64) {
            // ---- average pool of 4 channels ----
            int s0 = 0, s1 = 0, s2 = 0, s3 = 0;
            for (int t = 0; t < p.ntaps; ++t) {
                const uint32_t v = *(const uint32_t *)(x + p.tap_off[t] + 4 * c4);
                s0 = sdot4(v, 0x00000001u, s0);
                s1 = sdot4(v, 0x00000100u, s1);
                s2 = sdot4(v, 0x00010000u, s2);
                s3 = sdot4(v, 0x01000000u, s3);
            }
            int q[4];
            const int sums[4] = {s0, s1, s2, s3};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float xf = __fmul_rn(p.inv_len, (float)sums[k]);           // (1/len) * f32(sum)
                const float y = __fadd_rn(__fmul_rn(p.pool_c0, xf), p.pool_c1);  // c0 * x + c1
                const float r = __fadd_rn(y, __builtin_copysignf(0x1.fffffep-2f, y));
                int v = (r != r) ? 0 : (int)__builtin_amdgcn_fmed3f(r, -128.0f, 127.0f);
                v = max(v, p.pool_lo);
                q[k] = min(v, p.pool_hi);
            }
            const uint32_t qp = pack4(q[0], q[1], q[2], q[3]);
            // ---- this lane's share of the head dot products ----
            vs = sdot4(qp, 0x01010101u, vs);
#pragma unroll
            for (int n = 0; n < N; ++n)
                dot[n] = sdot4(qp, *(const uint32_t *)(p.w + (size_t)n * p.C + 4 * c4), dot[n]);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            vs += __shfl_xor(vs, off, 64);
#pragma unroll
            for (int n = 0; n < N; ++n) dot[n] += __shfl_xor(dot[n], off, 64);
        }
        // ---- head epilogue (all lanes compute all N: cheap and keeps the softmax uniform) ----
        int h[N];
        float e[N], sum = 0.0f;
#pragma unroll
        for (int n = 0; n < N; ++n) {
            const int acc = dot[n] - p.wzp[n] * vs + p.Kc[n];
            h[n] = requant(acc, p.A[n], p.S[n], p.lo_f, p.hi_f);
            e[n] = p.exp_table[h[n] + 128];
            sum = __fadd_rn(sum, e[n]); // one row: column-major order == index order
        }
#pragma unroll
        for (int n = 0; n < N; ++n) {
            const float prob = __fdiv_rn(e[n], sum);
            const float qf = __fadd_rn(__fdiv_rn(prob, p.sm_oscale), p.sm_ozp_f);
            const float r = __fadd_rn(qf, __builtin_copysignf(0x1.fffffep-2f, qf));
            const int y = (r != r) ? 0 : (int)__builtin_amdgcn_fmed3f(r, -128.0f, 127.0f);
            if (lane == n) out[b * N + n] = (int8_t)y;
        }
    }
}

// ------------------------------------------------------------------------
// FAST PATH 4 -- FullyConnected as a dense int8 MFMA GEMM (BASELINE config 5).
// (src/ops/fully_connected.rs:24-82; rows of all inferences form one [M][K] matrix)
//
//   Y[m][n] = requant( sum_k X[m][k] * W[n][k]  - wzp * rowsum(X[m])  + (c3 - c2[n]) )
//
// Both operands are K-contiguous ("NT" GEMM), the natural layout for
// v_mfma_i32_32x32x32_i8 whose lanes each hold 16 consecutive k-bytes of one row.
//   tile     : 256 x 256 per workgroup, 8 waves as 2 (m) x 4 (n), each wave 128 x 64 = 4 x 2
//              MFMA tiles (128 accumulator VGPRs); 128 x 128 with 4 waves when the problem has
//              too few 256^2 tiles to fill the chip.
//   staging  : BK = 128 bytes per step; X and W tiles go HBM/L2 -> LDS by LDS-DMA
//              (global_load_lds_dwordx4), double buffered.
//   schedule : 256^2 tile (one workgroup per CU, two waves per SIMD): the two wave rows (wm = 0 / 1,
//              one wave of each per SIMD) run ONE BARRIER apart, so one row's MFMA section
//              coincides with the other row's ds_read section instead of both stalling on the
//              LDS at once; the DMAs of the next tile are issued between the MFMAs.  Details
//              and the hazard argument are at the loop.  Measured on 4096^3 (scripts/ubench/
//              gemm_i8.hip): +10 % with random operands, +18 % with zero operands over the
//              lockstep loop (vmcnt(0) + barrier per step), which the 128^2 tile keeps.
//              The gap between zero and random operands (3.0 vs 2.2 POP/s) is the chip's power
//              management, not the schedule: a bare MFMA loop issues at 4.5 POP/s.
//   LDS image: [row][128 B], 16-byte slot index XOR ((row >> 1) & 7) -- with that key the 16
//              lanes of every ds_read_b128 service group ({0-3,12-15,20-27}, ...) hit 16
//              distinct 16-byte bank slots (row & 7 would be 2-way).  A DMA writes LDS linearly
//              (base + lane*16), so the swizzle is applied to the GLOBAL source chunk each
//              lane fetches and again on the fragment reads (guide rule 21: both sides).
//   operands : MFMA "A" = W rows (n), MFMA "B" = X rows (m), so that D[n][m] leaves every
//              lane with 16 results of ONE output row m.  The lane -> W-row map is permuted
//              (rho -> 16*((rho>>2)&1) + 4*(rho>>3) + (rho&3)) so those 16 results are 16
//              CONSECUTIVE n: one packed 16-byte store per lane per tile, no transposition.
//   grid     : XCD-aware remap so the 8 tiles that share panels sit behind the same L2.
// The f32 epilogue is the reference's, fused; |acc| can exceed 2^24 at K = 4096, where
// f32(acc) rounds to nearest even exactly like Rust's `as f32`.
// ------------------------------------------------------------------------
typedef int v16i __attribute__((ext_vector_type(16)));

template <int BM, int BN, int WM, int WN, bool STAGGER>
__global__ __launch_bounds__(64 * WM * WN) void fc_mfma(const int8_t *__restrict__ X,
                                                        int8_t *__restrict__ Y, FcGemmArgs p) {
    constexpr int BK = 128;
    constexpr int NW = WM * WN;                              // waves per workgroup
    constexpr int MT = BM / WM / 32, NT = BN / WN / 32;      // 32x32 MFMA tiles per wave
    constexpr int XT = BM * BK, WT = BN * BK, BUF = XT + WT; // one staging buffer (two allocated)
    constexpr int XP = XT / 1024 / NW, WP = WT / 1024 / NW;  // 1 KiB DMA pieces per wave
    static_assert(XT % (1024 * NW) == 0 && WT % (1024 * NW) == 0, "DMA pieces must divide over the waves");
    // swizzle key: with (row >> 1) & 7 the 16 lanes of every ds_read_b128 service group
    // ({0-3,12-15,20-27}, ...) hit 16 distinct 16-byte bank slots (row & 7 would be 2-way)
    auto key = [](int row) { return (row >> 1) & 7; };
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // XCD-aware 2-D tile order.  Dispatch puts block b on XCD b % 8 and each XCD runs its
    // blocks in order, one residency-full at a time.  Those PM*PN co-resident tiles are
    // mapped to a PM x PN patch, which needs only PM X-panels + PN W-panels per k-step
    // through that XCD's L2 instead of ~1 + PM*PN for a row-major order.
    constexpr int PM = (BM == 128) ? 8 : 4, PN = 8;           // 64 resident 128^2 tiles, 32 256^2 tiles
    const int tiles_m = p.M / BM, tiles_n = p.N / BN;
    int tm, tn;
    {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;   // j-th block of this XCD
        const int patches_n = tiles_n / PN, npatch = (tiles_m / PM) * patches_n;
        if (tiles_m % PM == 0 && tiles_n % PN == 0 && (npatch & 7) == 0) {
            const int patch = xcd * (npatch >> 3) + j / (PM * PN), t = j % (PM * PN);
            tm = (patch / patches_n) * PM + t / PN;
            tn = (patch % patches_n) * PN + t % PN;
        } else {
            tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
        }
    }
    const int K = p.K;
    const int8_t *Xt = X + (size_t)tm * BM * K;
    const int8_t *Wt = p.w + (size_t)tn * BN * K;

    // DMA piece i (1 KiB) of a tile = rows 8i .. 8i+7; lane -> (row, swizzled 16-byte slot)
    auto stage = [&](int kt, int buf) {
        const int r8 = lane >> 3, s8 = lane & 7;
#pragma unroll
        for (int j = 0; j < XP; ++j) {
            const int i = wave * XP + j, row = 8 * i + r8;
            dma16(Xt + (size_t)row * K + (size_t)kt * BK + ((s8 ^ key(row)) << 4), lds + buf * BUF + i * 1024);
        }
#pragma unroll
        for (int j = 0; j < WP; ++j) {
            const int i = wave * WP + j, row = 8 * i + r8;
            dma16(Wt + (size_t)row * K + (size_t)kt * BK + ((s8 ^ key(row)) << 4), lds + buf * BUF + XT + i * 1024);
        }
    };

    v16i acc[NT][MT];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0;

    // fragment rows of this lane
    const int rho = lane & 31, half = lane >> 5;
    const int nloc = 16 * ((rho >> 2) & 1) + 4 * (rho >> 3) + (rho & 3);
    int xoff[MT], woff[NT], xkey[MT], wkey[NT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int row = wm * (BM / WM) + t * 32 + rho;
        xoff[t] = row * BK, xkey[t] = key(row);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int row = wn * (BN / WN) + t * 32 + nloc;
        woff[t] = XT + row * BK, wkey[t] = key(row);
    }

    auto load_frags = [&](const uint8_t *lb, int ks, v4i (&a)[NT], v4i (&b)[MT]) {
#pragma unroll
        for (int t = 0; t < MT; ++t) b[t] = *(const v4i *)(lb + xoff[t] + (((ks * 2 + half) ^ xkey[t]) << 4));
#pragma unroll
        for (int t = 0; t < NT; ++t) a[t] = *(const v4i *)(lb + woff[t] + (((ks * 2 + half) ^ wkey[t]) << 4));
    };

    const int nk = K / BK;
    if constexpr (!STAGGER) {
        // lockstep loop: the DMAs of step t+1 fly during the MFMAs of step t; one vmcnt(0) +
        // barrier per step
        int cur = 0;
        stage(0, 0);
        for (int kt = 0; kt < nk; ++kt, cur ^= 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (kt + 1 < nk) stage(kt + 1, cur ^ 1);
            const uint8_t *lb = lds + cur * BUF;
#pragma unroll
            for (int ks = 0; ks < BK / 32; ++ks) {
                v4i a[NT], b[MT];
                load_frags(lb, ks, a, b);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        acc[nt][mt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[nt], b[mt], acc[nt][mt], 0, 0, 0);
            }
        }
    } else {
        // Staggered wave rows.  Per tile kt every wave runs, in program order,
        //     L0: ds_read the fragments of k-substeps 0,1 of buffer cur
        //     P0  (barrier) ; lgkmcnt(0)
        //     M0: 16 MFMAs, with the 8 DMA pieces of tile kt+1 (-> buffer cur^1) issued between them
        //     P1  (barrier)
        //     L1: ds_read the fragments of k-substeps 2,3 of buffer cur ; vmcnt(0)
        //     P2  (barrier) ; lgkmcnt(0)
        //     M1: 16 MFMAs
        //     P3  (barrier)
        // and wave row 1 executes one extra barrier first, so it is always ONE physical barrier
        // behind row 0: row 0's M sections line up with row 1's L sections and vice versa.
        // Let n be the physical index of row 0's P0(kt); row 1's P0(kt) is n+1.
        //   WAR (DMA of tile kt+1 overwrites tile kt-1's buffer): tile kt-1 is last read in
        //     L1(kt-1) and those reads retire at the lgkmcnt(0) after P2(kt-1) -- physical n-2
        //     for row 0, n-1 for row 1.  DMAs are issued after P0(kt), i.e. after physical n
        //     (row 0) / n+1 (row 1); a wave passes barrier n only once every wave has ARRIVED
        //     at n, and row 1 executes its lgkmcnt(0) between n-1 and its arrival at n.
        //   RAW (tile kt+1 is first read in L0(kt+1), after P3(kt) = physical n+3 / n+4): every
        //     wave waits vmcnt(0) -- its own DMAs have landed in LDS -- before P2(kt), which is
        //     physical n+2 (row 0) / n+3 (row 1); so by the time any wave passes n+3 all
        //     waves' DMAs of tile kt+1 have landed.
        // sched_barrier(0) pins the compiler's instruction order around the barriers.
        constexpr int PIECES = XP + WP, NMF = 2 * NT * MT, GAP = NMF / (PIECES + 1) > 0 ? NMF / (PIECES + 1) : 1;
        static_assert(BK == 128 && WM == 2, "phase plan: 4 k-substeps per tile, two wave rows");
        auto stage_piece = [&](int kt, int buf, int j) {
            const int r8 = lane >> 3, s8 = lane & 7;
            if (j < XP) {
                const int i = wave * XP + j, row = 8 * i + r8;
                dma16(Xt + (size_t)row * K + (size_t)kt * BK + ((s8 ^ key(row)) << 4), lds + buf * BUF + i * 1024);
            } else {
                const int i = wave * WP + (j - XP), row = 8 * i + r8;
                dma16(Wt + (size_t)row * K + (size_t)kt * BK + ((s8 ^ key(row)) << 4), lds + buf * BUF + XT + i * 1024);
            }
        };
        stage(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (wm == 1) __builtin_amdgcn_s_barrier(); // the stagger
        int cur = 0;
        for (int kt = 0; kt < nk; ++kt, cur ^= 1) {
            const uint8_t *lb = lds + cur * BUF;
            const bool more = kt + 1 < nk;
            int piece = 0;
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
                v4i a[2][NT], b[2][MT];
                load_frags(lb, 2 * ph, a[0], b[0]);
                load_frags(lb, 2 * ph + 1, a[1], b[1]);
                if (more && ph == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
                int cnt = 0;
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            acc[nt][mt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[u][nt], b[u][mt], acc[nt][mt], 0, 0, 0);
                            ++cnt;
                            if (ph == 0 && cnt % GAP == 0 && piece < PIECES) {
                                __builtin_amdgcn_sched_barrier(0);
                                if (more) stage_piece(kt + 1, cur ^ 1, piece);
                                ++piece;
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (wm == 0) __builtin_amdgcn_s_barrier(); // row 0 absorbs row 1's extra barrier
    }

    // epilogue: lane (m column = lane & 31, half) holds n = tile + 16*half + r, r = 0..15
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n0 = tn * BN + wn * (BN / WN) + nt * 32 + 16 * half;
        float cA[16];
        int cK[16];
#pragma unroll
        for (int r = 0; r < 16; r += 4) {
            const float4 fa = *(const float4 *)(p.A + n0 + r);
            const int4 ik = *(const int4 *)(p.Kc + n0 + r);
            cA[r] = fa.x, cA[r + 1] = fa.y, cA[r + 2] = fa.z, cA[r + 3] = fa.w;
            cK[r] = ik.x, cK[r + 1] = ik.y, cK[r + 2] = ik.z, cK[r + 3] = ik.w;
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = tm * BM + wm * (BM / WM) + mt * 32 + rho;
            const int corr = p.rowsum ? p.wzp * p.rowsum[m] : 0; // x1 = wzp * row-sum of the input
            uint32_t d[4];
#pragma unroll
            for (int r = 0; r < 16; r += 4) {
                const int q0 = requant(acc[nt][mt][r] + cK[r] - corr, cA[r], p.S, p.lo_f, p.hi_f);
                const int q1 = requant(acc[nt][mt][r + 1] + cK[r + 1] - corr, cA[r + 1], p.S, p.lo_f, p.hi_f);
                const int q2 = requant(acc[nt][mt][r + 2] + cK[r + 2] - corr, cA[r + 2], p.S, p.lo_f, p.hi_f);
                const int q3 = requant(acc[nt][mt][r + 3] + cK[r + 3] - corr, cA[r + 3], p.S, p.lo_f, p.hi_f);
                d[r >> 2] = pack4(q0, q1, q2, q3) ^ p.xr4;
            }
            *(uint4 *)(Y + (size_t)m * p.N + n0) = make_uint4(d[0], d[1], d[2], d[3]);
        }
    }
}

// sum_k x[row][k] for the weight-zero-point term of FullyConnected (fully_connected.rs:60-64);
// one wave per row, 16-byte loads.  Only launched when wzp != 0.
__global__ __launch_bounds__(256) void fc_rowsum(const int8_t *__restrict__ in, int *__restrict__ rowsum,
                                                 size_t rows, int K) {
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * 256) >> 6;
    for (size_t row = wave; row < rows; row += nwaves) {
        const uint4 *x = (const uint4 *)(in + row * (size_t)K);
        int rs = 0;
        for (int k = lane; k < (K >> 4); k += 64) {
            const uint4 v = x[k];
            rs = sdot4(v.x, 0x01010101u, rs);
            rs = sdot4(v.y, 0x01010101u, rs);
            rs = sdot4(v.z, 0x01010101u, rs);
            rs = sdot4(v.w, 0x01010101u, rs);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) rs += __shfl_xor(rs, off, 64);
        if (lane == 0) rowsum[row] = rs;
    }
}

// ------------------------------------------------------------------------
// launchers (host)
// ------------------------------------------------------------------------
// Persistent-workgroup kernels launch exactly as many workgroups as are resident (LDS-,
// VGPR- or wave-limited, asked of the runtime once per kernel), so every workgroup walks the
// same number of steps and none queues behind a finished one.
// Function attributes and occupancy are per DEVICE, and one process may drive several GPUs, so
// each launcher keeps one slot per device (benign race: two threads may both prepare a slot).
struct LaunchState {
    static constexpr int MAX_DEV = 64;
    std::atomic<int> per_cu[MAX_DEV];
};
template <typename Kern> static int prepared(LaunchState &st, Kern kern, int threads, int lds_bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= LaunchState::MAX_DEV) dev = LaunchState::MAX_DEV - 1;
    int n = st.per_cu[dev].load(std::memory_order_relaxed);
    if (n > 0 && dev != LaunchState::MAX_DEV - 1) return n;
    if (lds_bytes > 64 * 1024)
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, threads, (size_t)lds_bytes) != hipSuccess || n < 1) {
        (void)hipGetLastError();
        n = 1;
    }
    st.per_cu[dev].store(n, std::memory_order_relaxed);
    return n;
}

static inline int grid_for(size_t total, int per_block = 256, int cap = 256 * 8) {
    size_t g = (total + per_block - 1) / per_block;
    if (g < 1) g = 1;
    return (int)(g < (size_t)cap ? g : (size_t)cap);
}

void launch_conv2d_generic(const int8_t *in, int8_t *out, const ConvArgs &a, size_t batch, hipStream_t s) {
    const size_t total = batch * a.OH * a.OW * a.N;
    hipLaunchKernelGGL(conv2d_generic, dim3(grid_for(total)), dim3(256), 0, s, in, out, a, total);
}
void launch_dwconv_generic(const int8_t *in, int8_t *out, const ConvArgs &a, size_t batch, hipStream_t s) {
    const size_t total = batch * a.OH * a.OW * a.N;
    hipLaunchKernelGGL(dwconv_generic, dim3(grid_for(total)), dim3(256), 0, s, in, out, a, total);
}
void launch_avgpool_generic(const int8_t *in, int8_t *out, const PoolArgs &a, size_t batch, hipStream_t s) {
    const size_t total = batch * a.OH * a.OW * a.C;
    hipLaunchKernelGGL(avgpool_generic, dim3(grid_for(total)), dim3(256), 0, s, in, out, a, total);
}
void launch_fc_generic(const int8_t *in, int8_t *out, const FcArgs &a, size_t rows, hipStream_t s) {
    const size_t total = rows * a.N;
    hipLaunchKernelGGL(fc_generic, dim3(grid_for(total)), dim3(256), 0, s, in, out, a, total);
}
bool launch_fc_rowwave(const int8_t *in, int8_t *out, const FcArgs &a, size_t rows, hipStream_t s) {
    const int grid = grid_for(rows, 4);
    switch (a.N) {
    case 1: hipLaunchKernelGGL(fc_rowwave<1>, dim3(grid), dim3(256), 0, s, in, out, a, rows); return true;
    case 2: hipLaunchKernelGGL(fc_rowwave<2>, dim3(grid), dim3(256), 0, s, in, out, a, rows); return true;
    case 4: hipLaunchKernelGGL(fc_rowwave<4>, dim3(grid), dim3(256), 0, s, in, out, a, rows); return true;
    case 8: hipLaunchKernelGGL(fc_rowwave<8>, dim3(grid), dim3(256), 0, s, in, out, a, rows); return true;
    default: return false;
    }
}
bool fc_mfma_supported(size_t rows, int N, int K) {
    return rows > 0 && rows % 128 == 0 && N % 128 == 0 && K % 128 == 0 && rows / 128 * (size_t)(N / 128) < (1u << 30);
}
void launch_fc_rowsum(const int8_t *in, int *rowsum, size_t rows, int K, hipStream_t s) {
    hipLaunchKernelGGL(fc_rowsum, dim3(grid_for(rows, 4)), dim3(256), 0, s, in, rowsum, rows, K);
}
template <int BM, int BN, int WM, int WN, bool STAGGER>
static void launch_fc_mfma_t(const int8_t *in, int8_t *out, const FcGemmArgs &a, hipStream_t s) {
    constexpr int lds = 2 * (BM + BN) * 128;
    static LaunchState st;
    (void)prepared(st, fc_mfma<BM, BN, WM, WN, STAGGER>, 64 * WM * WN, lds);
    const int grid = (a.M / BM) * (a.N / BN);
    hipLaunchKernelGGL((fc_mfma<BM, BN, WM, WN, STAGGER>), dim3(grid), dim3(64 * WM * WN), lds, s, in, out, a);
}
void launch_fc_mfma(const int8_t *in, int8_t *out, const FcGemmArgs &a, hipStream_t s) {
    static const int force = [] { const char *e = getenv("MF_FC_TILE"); return e ? atoi(e) : 0; }();
    // 256 x 256 tiles halve the L2 -> LDS traffic per MAC; they need >= 256 tiles to fill the chip
    const bool big = a.M % 256 == 0 && a.N % 256 == 0 && (size_t)(a.M / 256) * (a.N / 256) >= 192;
    if ((big && force != 128) || (force == 256 && a.M % 256 == 0 && a.N % 256 == 0))
        launch_fc_mfma_t<256, 256, 2, 4, true>(in, out, a, s);
    else
        launch_fc_mfma_t<128, 128, 2, 2, false>(in, out, a, s);
}
bool launch_fc_rowwave_softmax(const int8_t *in, int8_t *out, const FcArgs &a, const SoftmaxArgs &sm, size_t rows,
                               hipStream_t s) {
    const int grid = grid_for(rows, 4);
    switch (a.N) {
    case 2: hipLaunchKernelGGL(fc_rowwave_softmax<2>, dim3(grid), dim3(256), 0, s, in, out, a, sm, rows); return true;
    case 4: hipLaunchKernelGGL(fc_rowwave_softmax<4>, dim3(grid), dim3(256), 0, s, in, out, a, sm, rows); return true;
    case 8: hipLaunchKernelGGL(fc_rowwave_softmax<8>, dim3(grid), dim3(256), 0, s, in, out, a, sm, rows); return true;
    default: return false;
    }
}
void launch_softmax(const int8_t *in, int8_t *out, const SoftmaxArgs &a, size_t batch, hipStream_t s) {
    hipLaunchKernelGGL(softmax_table, dim3(grid_for(batch)), dim3(256), 0, s, in, out, a, batch);
}
void launch_quantize(const float *in, int8_t *out, size_t n, float scale, float zp_f, bool u8, hipStream_t s) {
    hipLaunchKernelGGL(quantize_f32, dim3(grid_for((n + 3) / 4)), dim3(256), 0, s, in, out, n, scale, zp_f,
                       u8 ? 0.0f : -128.0f, u8 ? 255.0f : 127.0f, u8 ? 0x80 : 0);
}
void launch_xor80(const int8_t *in, int8_t *out, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(xor80_bytes, dim3(grid_for((n + 15) / 16)), dim3(256), 0, s, in, out, n);
}
void launch_dequantize(const int8_t *in, float *out, size_t n, float scale, float zp_f, bool raw_u8, hipStream_t s) {
    hipLaunchKernelGGL(dequantize_i8, dim3(grid_for(n)), dim3(256), 0, s, in, out, n, scale, zp_f, raw_u8 ? 1 : 0);
}
void launch_synth(int8_t *out, size_t n, uint64_t seed, uint64_t first, hipStream_t s) {
    hipLaunchKernelGGL(synth_i8, dim3(grid_for(n, 256, 256 * 16)), dim3(256), 0, s, out, n, seed, first);
}
void launch_checksum(const int8_t *in, size_t n, unsigned long long *result, hipStream_t s) {
    hipLaunchKernelGGL(checksum_i8, dim3(grid_for(n, 256 * 16, 1024)), dim3(256), 0, s, in, n, result);
}

// ---- fast-path dispatch tables ------------------------------------------------
// the four instances of a fast kernel: {v_cvt, bit-pattern} int->f32  x  {i8, u8} element type
#define MF_DISPATCH4(magic, xr, FN, ARGS, ...)                         \
    do {                                                               \
        if (xr) {                                                      \
            if (magic) FN<__VA_ARGS__, true, 0x80808080u> ARGS;        \
            else FN<__VA_ARGS__, false, 0x80808080u> ARGS;             \
        } else {                                                       \
            if (magic) FN<__VA_ARGS__, true, 0u> ARGS;                 \
            else FN<__VA_ARGS__, false, 0u> ARGS;                      \
        }                                                              \
    } while (0);
template <int H, int W, int C, int S, int G, int NTHR, bool MG, uint32_t XR4>
static void launch_dw(const int8_t *in, int8_t *out, const DwFastArgs &a, int batch, hipStream_t s) {
    constexpr int LP = C < 16 ? 16 : C;
    constexpr int lds = 2 * G * (H + 2) * (LP + W * C + LP) + 256; // two staging buffers + read slack
    static LaunchState st;
    const int per_cu = prepared(st, dw3x3_nhwc<H, W, C, S, G, NTHR, MG, XR4>, NTHR, lds);
    const int nsteps = (batch + G - 1) / G;
    const int grid = nsteps < 256 * per_cu ? nsteps : 256 * per_cu;
    hipLaunchKernelGGL((dw3x3_nhwc<H, W, C, S, G, NTHR, MG, XR4>), dim3(grid), dim3(NTHR), lds, s, in, out, a, batch);
}

const char *dw_fast_name(int H, int W, int C, int S) {
#define MF_DW(h, w, c, s, g, t) \
    if (H == h && W == w && C == c && S == s) return "dw3x3_nhwc<" #h "," #w "," #c "," #s "," #g "," #t ">";
    MF_DW_SHAPES(MF_DW)
#undef MF_DW
    return nullptr;
}
bool launch_dw_fast(int H, int W, int C, int S, const int8_t *in, int8_t *out, const DwFastArgs &a,
                    int batch, hipStream_t s) {
    static const int alt = [] { const char *e = getenv("MF_DW_ALT"); return e ? atoi(e) : -1; }();
    if (alt >= 0) { // tuning candidates, see MF_DW_ALT_SHAPES
        int idx = 0;
        (void)idx;
#define MF_DW(h, w, c, st, g, t)                                                \
    if (idx++ == alt && H == h && W == w && C == c && S == st) {                \
        MF_DISPATCH4(a.magic, a.xr, launch_dw, (in, out, a, batch, s), h, w, c, st, g, t) \
        return true;                                                            \
    }
        MF_DW_ALT_SHAPES(MF_DW)
#undef MF_DW
    }
#define MF_DW(h, w, c, st, g, t)                          \
    if (H == h && W == w && C == c && S == st) {          \
        MF_DISPATCH4(a.magic, a.xr, launch_dw, (in, out, a, batch, s), h, w, c, st, g, t) \
        return true;                                      \
    }
    MF_DW_SHAPES(MF_DW)
#undef MF_DW
    return false;
}

static int dw_c1_lds_bytes(const DwC1Args &a) {
    const int TH = (a.OH - 1) * a.sh + a.KH;
    return ((TH * a.TWP + 4 + 15) & ~15) + a.KH * a.KG * 8 * 4;
}
bool dw_c1_supported(const DwC1Args &a) {
    return a.N >= 1 && a.N <= 8 && dw_c1_lds_bytes(a) <= 64 * 1024;
}
void launch_dw_c1(const int8_t *in, int8_t *out, const DwC1Args &a, size_t batch, hipStream_t s) {
    const int grid = (int)(batch < 256 * 8 ? batch : 256 * 8);
    const int lds = dw_c1_lds_bytes(a);
#define MF_C1(MG, XR) hipLaunchKernelGGL((dw_c1_lds<MG, XR>), dim3(grid), dim3(512), lds, s, in, out, a, batch)
    if (a.xr) { if (a.magic) MF_C1(true, 0x80808080u); else MF_C1(false, 0x80808080u); }
    else { if (a.magic) MF_C1(true, 0u); else MF_C1(false, 0u); }
#undef MF_C1
}

const char *dw_stem_name(int H, int W, int DM, int S) {
    if (H == 96 && W == 96 && DM == 8 && S == 2) return "dw3x3_stem8<96,96,2>";
    return nullptr;
}
bool launch_dw_stem(int H, int W, int DM, int S, const int8_t *in, int8_t *out, const DwStemArgs &a,
                    int batch, hipStream_t s, bool f32_input) {
    if (H == 96 && W == 96 && DM == 8 && S == 2) {
        constexpr int G = 2, lds = 2 * G * (16 + (96 + 2) * 96);
        static LaunchState st, stf;
        const int per_cu = f32_input ? prepared(stf, dw3x3_stem8<96, 96, G, false, 0u, true>, 256, lds)
                                     : prepared(st, dw3x3_stem8<96, 96, G, false, 0u, false>, 256, lds);
        const int nsteps = (batch + G - 1) / G;
        const int grid = nsteps < 256 * per_cu ? nsteps : 256 * per_cu;
#define MF_STEM(MG, XR, F) hipLaunchKernelGGL((dw3x3_stem8<96, 96, G, MG, XR, F>), dim3(grid), dim3(256), lds, s, in, out, a, batch)
#define MF_STEM2(F)                                                                          \
    if (a.xr) { if (a.magic) MF_STEM(true, 0x80808080u, F); else MF_STEM(false, 0x80808080u, F); } \
    else { if (a.magic) MF_STEM(true, 0u, F); else MF_STEM(false, 0u, F); }
        if (f32_input) { MF_STEM2(true) } else { MF_STEM2(false) }
#undef MF_STEM2
#undef MF_STEM
        return true;
    }
    return false;
}

template <int H, int W, int C, int S, int N, int G, int NTHR, int DB, bool MG, uint32_t XR4>
static void launch_dwpw_t(const int8_t *in, int8_t *out, const DwPwArgs &a, int batch, hipStream_t s) {
    constexpr int LP = C < 16 ? 16 : C;
    constexpr int OH = (H + S - 1) / S, OW = (W + S - 1) / S;
    constexpr int BUF = G * (H + 2) * (LP + W * C + LP);
    constexpr int NB = N < 64 ? N : 64, CPIX = C < 64 ? 1024 / C : 16;
    constexpr int patch = (NB / 16 < 4) ? (NTHR / 64) * CPIX * N : 0;
    constexpr int lds = (DB ? 2 : 1) * BUF + 256 + G * OH * OW * C + 64 + patch;
    static_assert(lds <= 163840, "fused tile does not fit the LDS");
    static LaunchState st;
    const int per_cu = prepared(st, dwpw3x3<H, W, C, S, N, G, NTHR, (DB != 0), MG, XR4>, NTHR, lds);
    const int nsteps = (batch + G - 1) / G;
    const int grid = nsteps < 256 * per_cu ? nsteps : 256 * per_cu;
    hipLaunchKernelGGL((dwpw3x3<H, W, C, S, N, G, NTHR, (DB != 0), MG, XR4>), dim3(grid), dim3(NTHR), lds, s, in, out, a, batch);
}
const char *dwpw_name(int H, int W, int C, int S, int N) {
#define MF_DWPW(h, w, c, s, n, g, t, d) \
    if (H == h && W == w && C == c && S == s && N == n) return "dwpw3x3<" #h "," #w "," #c "," #s "," #n "," #g "," #t "," #d ">";
    MF_DWPW_SHAPES(MF_DWPW)
#undef MF_DWPW
    return nullptr;
}
bool launch_dwpw(int H, int W, int C, int S, int N, const int8_t *in, int8_t *out, const DwPwArgs &a,
                 int batch, hipStream_t s) {
    static const int alt = [] { const char *e = getenv("MF_DWPW_ALT"); return e ? atoi(e) : -1; }();
    if (alt >= 0) {
        int idx = 0;
        (void)idx;
#define MF_DWPW(h, w, c, st, n, g, t, d)                                        \
    if (idx++ == alt && H == h && W == w && C == c && S == st && N == n) {      \
        MF_DISPATCH4(a.dw.magic && a.pw.magic, a.pw.xr, launch_dwpw_t, (in, out, a, batch, s), h, w, c, st, n, g, t, d) \
        return true;                                                            \
    }
        MF_DWPW_ALT_SHAPES(MF_DWPW)
#undef MF_DWPW
    }
#define MF_DWPW(h, w, c, st, n, g, t, d)                         \
    if (H == h && W == w && C == c && S == st && N == n) {       \
        MF_DISPATCH4(a.dw.magic && a.pw.magic, a.pw.xr, launch_dwpw_t, (in, out, a, batch, s), h, w, c, st, n, g, t, d) \
        return true;                                             \
    }
    MF_DWPW_SHAPES(MF_DWPW)
#undef MF_DWPW
    return false;
}

bool tail_supported(int C, int N, int ntaps) {
    return C % 4 == 0 && C >= 4 && ntaps >= 1 && ntaps <= 64 && (N == 1 || N == 2 || N == 4 || N == 8);
}
void launch_tail(const int8_t *in, int8_t *out, const TailArgs &a, size_t batch, hipStream_t s) {
    const int grid = grid_for(batch, 4);
    switch (a.N) {
    case 1: hipLaunchKernelGGL(tail_pool_head_softmax<1>, dim3(grid), dim3(256), 0, s, in, out, a, batch); break;
    case 2: hipLaunchKernelGGL(tail_pool_head_softmax<2>, dim3(grid), dim3(256), 0, s, in, out, a, batch); break;
    case 4: hipLaunchKernelGGL(tail_pool_head_softmax<4>, dim3(grid), dim3(256), 0, s, in, out, a, batch); break;
    default: hipLaunchKernelGGL(tail_pool_head_softmax<8>, dim3(grid), dim3(256), 0, s, in, out, a, batch); break;
    }
}

template <int K, int N, bool MG, uint32_t XR4>
static void launch_pw_t(const int8_t *in, int8_t *out, const PwArgs &a, long long npix, int grid, hipStream_t s) {
    hipLaunchKernelGGL((pw_mfma<K, N, MG, XR4>), dim3(grid), dim3(256), 0, s, in, out, a, npix);
}
// workgroups of the grid-strided pointwise kernel (r01 sweep, MF_PW_GRID overrides): the wide early
// layers (K < 64, most pixels) like many short-lived workgroups, the deep late ones few
static long long pw_grid_cap(int K) {
    static const long long forced = [] { const char *e = getenv("MF_PW_GRID"); return e ? atoll(e) : 0LL; }();
    if (forced > 0) return forced;
    return K < 64 ? 256LL * 32 : (K >= 128 ? 256LL * 4 : 256LL * 8);
}
const char *pw_name(int K, int N) {
#define MF_PW(k, n) \
    if (K == k && N == n) return "pw_mfma<" #k "," #n ">";
    MF_PW_SHAPES(MF_PW)
#undef MF_PW
    return nullptr;
}
bool launch_pw(int K, int N, const int8_t *in, int8_t *out, const PwArgs &a, long long npix, hipStream_t s) {
#define MF_PW(k, n)                                                                             \
    if (K == k && N == n) {                                                                     \
        constexpr int NB = n < 64 ? n : 64, SLOTS = 4 / (n / NB), CPIX = k < 64 ? 1024 / k : 16; \
        constexpr int U = pw_chunks_in_flight(k);                                               \
        const long long nchunks = (npix + CPIX - 1) / CPIX;                                     \
        long long grid = (nchunks + SLOTS * U - 1) / (SLOTS * U);                               \
        if (grid > pw_grid_cap(k)) grid = pw_grid_cap(k);                                       \
        if (grid < 1) grid = 1;                                                                 \
        MF_DISPATCH4(a.magic, a.xr, launch_pw_t, (in, out, a, npix, (int)grid, s), k, n)                  \
        return true;                                                                            \
    }
    MF_PW_SHAPES(MF_PW)
#undef MF_PW
    return false;
}

} // namespace k
} // namespace mf
