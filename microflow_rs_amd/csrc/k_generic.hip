// k_generic.hip -- shape-generic kernels (one thread per output element) and the boundary/utility kernels.
//
// conv_2d / depthwise_conv_2d / average_pool_2d / fully_connected for ANY shape (API completeness, the
// reference's unit KATs, non-zero weight zero points, in-product cross-check of the fast paths), softmax,
// quantize / dequantize, the u8 boundary XOR, the synthetic input generator and the checksum.
// Arithmetic contract, shared device helpers and launch plumbing: k_common.hpp.
#include "k_common.hpp"

#include <mutex>
#include <unordered_map>

namespace mf {
namespace k {

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

// The same operator for a 1x1 filter with few outputs (N <= 8, C % 4 == 0; person_detect's head: 256 -> 2): the
// pixels of the batch are the rows of a [rows][C] matrix; one wavefront per row, 4 channels per lane and step
// (coalesced dword loads), v_dot4 against the N filter rows, butterfly reduction, lanes 0..N-1 finish the epilogue.
// Every input byte is read once; same arithmetic as conv2d_generic, value by value (a 1x1 window has no padding).
__global__ __launch_bounds__(256) void conv1x1_rowwave(const int8_t *__restrict__ in, int8_t *__restrict__ out, ConvArgs p,
                                                       size_t rows) {
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * 256) >> 6;
    const int C4 = p.C >> 2;
    for (size_t row = wave; row < rows; row += nwaves) {
        const uint32_t *x = (const uint32_t *)(in + row * (size_t)p.C);
        int dot[8] = {0, 0, 0, 0, 0, 0, 0, 0}, vs = 0;
        for (int c4 = lane; c4 < C4; c4 += 64) {
            const uint32_t v = x[c4];
            vs = sdot4(v, 0x01010101u, vs);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < p.N) dot[j] = sdot4(v, ((const uint32_t *)(p.w + (size_t)j * p.C))[c4], dot[j]);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            vs += __shfl_xor(vs, off, 64);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < p.N) dot[j] += __shfl_xor(dot[j], off, 64);
        }
        int d = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) d = (lane == j) ? dot[j] : d;
        if (lane < p.N) {
            const int acc = d - p.wzp[lane] * vs + p.Kc[lane];
            out[row * p.N + lane] = (int8_t)(requant_any(acc, p.A[lane], p.S[lane], p.lo_f, p.hi_f) ^ p.xr);
        }
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

// The same operator with 4 channels per thread (C % 4 == 0): a tap of a window is one dword of 4 consecutive channels,
// summed per byte with masked v_dot4; consecutive lanes take consecutive channel quads of one output pixel, so every
// wave-level load is a contiguous 256-byte row piece and every store a packed dword.  Same arithmetic, value by value.
__global__ __launch_bounds__(256) void avgpool_c4(const int8_t *__restrict__ in, int8_t *__restrict__ out, PoolArgs p,
                                                  size_t total4) {
    const int C4 = p.C >> 2;
    const int shy = p.pad_same ? (p.KH - 1) / 2 : 0, shx = p.pad_same ? (p.KW - 1) / 2 : 0;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total4; idx += (size_t)gridDim.x * 256) {
        const int c4 = (int)(idx % C4);
        size_t t = idx / C4;
        const int ox = (int)(t % p.OW);
        t /= p.OW;
        const int oy = (int)(t % p.OH);
        const size_t img = t / p.OH;
        const int8_t *ip = in + img * (size_t)p.H * p.W * p.C + 4 * c4;
        int s0 = 0, s1 = 0, s2 = 0, s3 = 0, len = 0;
        for (int ky = 0; ky < p.KH; ++ky) {
            const int iy = oy * p.sh + ky - shy;
            for (int kx = 0; kx < p.KW; ++kx) {
                const int ix = ox * p.sw + kx - shx;
                if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) {
                    const uint32_t v = *(const uint32_t *)(ip + ((size_t)iy * p.W + ix) * p.C);
                    s0 = sdot4(v, 0x00000001u, s0), s1 = sdot4(v, 0x00000100u, s1);
                    s2 = sdot4(v, 0x00010000u, s2), s3 = sdot4(v, 0x01000000u, s3);
                    ++len;
                }
            }
        }
        const float inv = __fdiv_rn(1.0f, (float)len);
        const int sums[4] = {s0, s1, s2, s3};
        int q[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float x = __fmul_rn(inv, (float)(sums[k] + p.bias * len));
            const float y = __fadd_rn(__fmul_rn(p.c0, x), p.c1);
            const float r = __fadd_rn(y, __builtin_copysignf(0x1.fffffep-2f, y));
            int v = (r != r) ? 0 : (int)__builtin_amdgcn_fmed3f(r, p.sat_lo, p.sat_hi); // NaN (len == 0) -> 0
            v = max(v, p.lo);
            q[k] = min(v, p.hi);
        }
        ((uint32_t *)out)[idx] = pack4(q[0], q[1], q[2], q[3]) ^ (0x01010101u * (uint32_t)p.xr);
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

// Exhaustive check behind quant_div's fast form (k_common.hpp): every one of the 2^32 float bit patterns is quantised
// with the true division and with the 3-instruction form; *bad receives the number of patterns whose BYTE differs.
__global__ __launch_bounds__(256) void verify_quant_div_kernel(float scale, float rcp, float zp_f, float sat_lo, float sat_hi,
                                                               unsigned long long *bad) {
    const unsigned long long stride = (unsigned long long)gridDim.x * 256;
    unsigned cnt = 0;
    for (unsigned long long b = (unsigned long long)blockIdx.x * 256 + threadIdx.x; b < (1ull << 32); b += stride) {
        const float x = __uint_as_float((uint32_t)b);
        int q[2];
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const float t = __fadd_rn(quant_div(x, scale, rcp, v == 1), zp_f);
            const float r = __fadd_rn(t, __builtin_copysignf(0x1.fffffep-2f, t));
            q[v] = (r != r) ? 0 : (int)__builtin_amdgcn_fmed3f(r, sat_lo, sat_hi);
        }
        cnt += q[0] != q[1];
    }
    if (cnt) atomicAdd(bad, (unsigned long long)cnt);
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
    wg_sync();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) part[threadIdx.x] += part[threadIdx.x + off];
        wg_sync();
    }
    if (threadIdx.x == 0) atomicAdd(result, part[0]);
}

// ---- launchers ----
void launch_conv2d_generic(const int8_t *in, int8_t *out, const ConvArgs &a, size_t batch, hipStream_t s) {
    const size_t total = batch * a.OH * a.OW * a.N;
    hipLaunchKernelGGL(conv2d_generic, dim3(grid_for(total)), dim3(256), 0, s, in, out, a, total);
}
bool conv1x1_rowwave_supported(const ConvArgs &a) {
    return a.KH == 1 && a.KW == 1 && a.sh == 1 && a.sw == 1 && a.OH == a.H && a.OW == a.W && a.N >= 1 && a.N <= 8 && a.C % 4 == 0;
}
void launch_conv1x1_rowwave(const int8_t *in, int8_t *out, const ConvArgs &a, size_t batch, hipStream_t s) {
    const size_t rows = batch * a.H * a.W;
    hipLaunchKernelGGL(conv1x1_rowwave, dim3(grid_for(rows, 4)), dim3(256), 0, s, in, out, a, rows);
}
void launch_dwconv_generic(const int8_t *in, int8_t *out, const ConvArgs &a, size_t batch, hipStream_t s) {
    const size_t total = batch * a.OH * a.OW * a.N;
    hipLaunchKernelGGL(dwconv_generic, dim3(grid_for(total)), dim3(256), 0, s, in, out, a, total);
}
void launch_avgpool_c4(const int8_t *in, int8_t *out, const PoolArgs &a, size_t batch, hipStream_t s) {
    const size_t total4 = batch * a.OH * a.OW * (a.C / 4);
    hipLaunchKernelGGL(avgpool_c4, dim3(grid_for(total4)), dim3(256), 0, s, in, out, a, total4);
}
void launch_avgpool_generic(const int8_t *in, int8_t *out, const PoolArgs &a, size_t batch, hipStream_t s) {
    const size_t total = batch * a.OH * a.OW * a.C;
    hipLaunchKernelGGL(avgpool_generic, dim3(grid_for(total)), dim3(256), 0, s, in, out, a, total);
}
void launch_fc_generic(const int8_t *in, int8_t *out, const FcArgs &a, size_t rows, hipStream_t s) {
    const size_t total = rows * a.N;
    hipLaunchKernelGGL(fc_generic, dim3(grid_for(total)), dim3(256), 0, s, in, out, a, total);
}
void launch_softmax(const int8_t *in, int8_t *out, const SoftmaxArgs &a, size_t batch, hipStream_t s) {
    hipLaunchKernelGGL(softmax_table, dim3(grid_for(batch)), dim3(256), 0, s, in, out, a, batch);
}
// ~0 = the check itself could not run (allocation / launch / copy failed); the sticky error is consumed here
unsigned long long verify_quant_div(float scale, float rcp, float zp_f, float sat_lo, float sat_hi, hipStream_t s) {
    unsigned long long *d = nullptr, h = ~0ull;
    if (hipMalloc((void **)&d, sizeof(h)) != hipSuccess) {
        (void)hipGetLastError();
        return h;
    }
    if (hipMemsetAsync(d, 0, sizeof(h), s) == hipSuccess) {
        hipLaunchKernelGGL(verify_quant_div_kernel, dim3(256 * 16), dim3(256), 0, s, scale, rcp, zp_f, sat_lo, sat_hi, d);
        if (hipGetLastError() != hipSuccess || hipMemcpyAsync(&h, d, sizeof(h), hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess) {
            (void)hipGetLastError();
            h = ~0ull;
        }
    }
    (void)hipFree(d);
    return h;
}

// ---- self-tests of the epilogue forms of k_common.hpp (mf_selftest_rounding / mf_selftest_requant) ----
// Reference tail of the epilogue on a value x (= round 2's form, read against libm::roundf + `as T`):
__device__ __forceinline__ uint32_t tail_ref(float x, float lo, float hi, uint32_t xr) {
    const float r = __fadd_rn(x, __builtin_copysignf(0x1.fffffep-2f, x));
    return ((uint32_t)(int)__builtin_amdgcn_fmed3f(r, lo, hi) & 0xffu) ^ xr;
}
// mode 1: v_med3 + sticky-bit RNE pack; mode 2: saturating pack (whole range of the element type)
template <int MODE, uint32_t XR4> __device__ __forceinline__ uint32_t tail_new4(float x0, float x1, float x2, float x3, float lo, float hi) {
    if constexpr (MODE == 2) return sat_pack4(x0, x1, x2, x3, XR4 ? 12582912.0f : 12583040.0f) ^ 0x80808080u;
    else if constexpr (MODE == 3) // negative control: round-to-nearest-EVEN without the sticky bit (differs from roundf on ties)
        return pack4((int)__builtin_rintf(__builtin_amdgcn_fmed3f(x0, lo, hi)), (int)__builtin_rintf(__builtin_amdgcn_fmed3f(x1, lo, hi)),
                     (int)__builtin_rintf(__builtin_amdgcn_fmed3f(x2, lo, hi)), (int)__builtin_rintf(__builtin_amdgcn_fmed3f(x3, lo, hi))) ^ XR4;
    else return rne_pack4(__builtin_amdgcn_fmed3f(x0, lo, hi), __builtin_amdgcn_fmed3f(x1, lo, hi), __builtin_amdgcn_fmed3f(x2, lo, hi),
                          __builtin_amdgcn_fmed3f(x3, lo, hi)) ^ XR4;
}
// every float bit pattern as x (four per thread and pass); patterns outside the form's domain (|x| >= 2^22, mode 2:
// |x| >= 30000; NaN) are replaced by 0.5
template <int MODE, uint32_t XR4>
__global__ __launch_bounds__(256) void selftest_rounding_kernel(float lo, float hi, unsigned long long *bad) {
    const unsigned long long stride = (unsigned long long)gridDim.x * 256;
    const float lim = MODE == 2 ? 30000.0f : 4194304.0f;
    unsigned cnt = 0;
    for (unsigned long long b = (unsigned long long)blockIdx.x * 256 + threadIdx.x; b < (1ull << 30); b += stride) {
        float x[4];
        uint32_t want = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            x[k] = __uint_as_float((uint32_t)(b * 4 + k));
            if (!(__builtin_fabsf(x[k]) < lim)) x[k] = 0.5f;
            want |= tail_ref(x[k], lo, hi, XR4 & 0xffu) << (8 * k);
        }
        const uint32_t got = tail_new4<MODE, XR4>(x[0], x[1], x[2], x[3], lo, hi);
        const uint32_t d = got ^ want;
        cnt += (d & 0xffu ? 1 : 0) + (d & 0xff00u ? 1 : 0) + (d & 0xff0000u ? 1 : 0) + (d & 0xff000000u ? 1 : 0);
    }
    if (cnt) atomicAdd(bad, (unsigned long long)cnt);
}
// the whole epilogue (requant_pack4, as the kernels call it) against requant_t + pack4 for every accumulator in
// (-2^22, 2^22) (mode 2: those with |x| < 30000) at one (A, S)
template <int MODE, uint32_t XR4>
__global__ __launch_bounds__(256) void selftest_requant_kernel(float A, float S, float lo, float hi, unsigned long long *bad) {
    const long long stride = (long long)gridDim.x * 256;
    unsigned cnt = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < (1ll << 21); i += stride) {
        int acc[4];
        uint32_t want = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int a = (int)(i * 4 + k) - (1 << 22) + 1;
            if (MODE == 2 && !(__builtin_fabsf(__fadd_rn(A, __fmul_rn(S, (float)a))) < 30000.0f)) a = 0;
            acc[k] = a + MF_MAGIC_I;
            want |= ((uint32_t)requant_t<1>(acc[k], A, S, lo, hi) & 0xffu) << (8 * k);
        }
        want ^= XR4;
        const uint32_t got = requant_pack4<MODE, XR4>(acc[0], acc[1], acc[2], acc[3], make_float4(A, A, A, A), make_float4(S, S, S, S), lo, hi);
        const uint32_t d = got ^ want;
        cnt += (d & 0xffu ? 1 : 0) + (d & 0xff00u ? 1 : 0) + (d & 0xff0000u ? 1 : 0) + (d & 0xff000000u ? 1 : 0);
    }
    if (cnt) atomicAdd(bad, (unsigned long long)cnt);
}
// ---- the gate of epilogue mode 3 (k_common.hpp; host search: epi_fma.cpp) ----
// For every channel c of an operator and EVERY accumulator the channel can produce, acc in [amin[c], amax[c]]: the byte the kernels'
// own requant_pack4<3> stores (v_fma_f32 on the accumulator's bit pattern with the pivot folded in, v_cvt_pk_u8_f32, XOR; executed
// in round-toward-zero as in the kernels) against the reference's two-rounding tail (requant_any: v_cvt_f32_i32, v_mul, v_add,
// roundf, clamp, truncating convert; executed in the default round-to-nearest).  bad[c] counts the mismatches; the operator may use
// the form only if all of them are zero.  blockIdx.y = channel.
// The two halves of an iteration run in different rounding modes.  The compiler does not model MODE, so each half is fenced by
// data: its inputs come out of a volatile asm placed after the mode switch, its results go into one placed before the next.
__device__ __forceinline__ void fp_round_mode(uint32_t m) {
    if (m) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\ts_nop 3" ::: "memory");
    else asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0\n\ts_nop 3" ::: "memory");
}
__global__ __launch_bounds__(256) void verify_fma_form_kernel(const float *A, const float *S, const float *C3, const float *S3, const int *piv,
                                                             const int *amin, const int *amax, const int *patchP, const int *patchR,
                                                             float lo, float hi, uint32_t xr, unsigned long long *bad) {
    const int c = blockIdx.y;
    float a_ = A[c], s_ = S[c], c3 = C3[c], s3 = S3[c];
    const int d = piv[c];
    const int pP = patchP[c], pR = patchR[c]; // the channel's one replaced bit pattern (0: none), as the kernels apply it (epi_patch_apply)
    const long long a0 = amin[c], a1 = amax[c];
    unsigned cnt = 0;
    for (long long base = a0 + 4ll * ((long long)blockIdx.x * 256 + threadIdx.x); base <= a1; base += 4ll * gridDim.x * 256) {
        int av[4], bits[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) av[k] = (int)(base + k <= a1 ? base + k : a1); // (the tail repeats the last accumulator)
        // reference half: round to nearest
        fp_round_mode(0);
        asm volatile("" : "+v"(av[0]), "+v"(av[1]), "+v"(av[2]), "+v"(av[3]), "+v"(a_), "+v"(s_));
        uint32_t want = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) want |= (((uint32_t)requant_any(av[k], a_, s_, lo, hi) & 0xffu) ^ xr) << (8 * k);
        asm volatile("" : "+v"(want));
        // the form's half: toward zero
        fp_round_mode(3);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            bits[k] = MF_MAGIC_I + av[k] + d;
            bits[k] = bits[k] == pP ? pR : bits[k];
        }
        asm volatile("" : "+v"(bits[0]), "+v"(bits[1]), "+v"(bits[2]), "+v"(bits[3]), "+v"(c3), "+v"(s3));
        uint32_t got = requant_pack4<3, 0u>(bits[0], bits[1], bits[2], bits[3], make_float4(c3, c3, c3, c3), make_float4(s3, s3, s3, s3), lo, hi);
        asm volatile("" : "+v"(got));
        const uint32_t df = got ^ want;
        cnt += (df & 0xffu ? 1 : 0) + (df & 0xff00u ? 1 : 0) + (df & 0xff0000u ? 1 : 0) + (df & 0xff000000u ? 1 : 0);
    }
    fp_round_mode(0);
    if (cnt) atomicAdd(bad + c, (unsigned long long)cnt);
}
// all pointers are DEVICE arrays of n entries (bad: zeroed by the caller); returns false when the launch failed
bool verify_fma_form(const float *A, const float *S, const float *C3, const float *S3, const int *piv, const int *amin, const int *amax,
                     const int *patchP, const int *patchR, int n, float lo, float hi, bool u8, unsigned long long *bad, hipStream_t s) {
    hipLaunchKernelGGL(verify_fma_form_kernel, dim3(64, n), dim3(256), 0, s, A, S, C3, S3, piv, amin, amax, patchP, patchR, lo, hi,
                       u8 ? 0x80u : 0u, bad);
    return hipGetLastError() == hipSuccess;
}
// v_cvt_pk_u8_f32 itself, over all 2^32 bit patterns, against what epi_fma.cpp assumes of it (in integer arithmetic, so that the
// model does not depend on the mode it is checked in): RZ = 1, the kernels' mode: truncation toward zero; RZ = 0, the default mode:
// round half to even; either way saturation to [0, 255], NaN -> 0, the other three bytes of the destination untouched
__device__ __forceinline__ uint32_t cvt_pk_model(float x, bool trunc_) {
    if (!(x > 0.0f)) return 0u; // (comparisons are exact in every mode)
    if (x >= 256.0f) return 255u;
    const uint32_t b = __float_as_uint(x), e = b >> 23, man = (b & 0x7fffffu) | 0x800000u;
    if (e < 126) return 0u;     // x < 0.5
    const int sh = 150 - (int)e; // x = man 2^-sh, sh in 16 .. 24
    const uint32_t ip = man >> sh, frac = man & ((1u << sh) - 1u), half = 1u << (sh - 1);
    uint32_t r = ip;
    if (!trunc_) r += (frac > half || (frac == half && (ip & 1u))) ? 1u : 0u;
    return r > 255u ? 255u : r;
}
template <int RZ> __global__ __launch_bounds__(256) void selftest_cvt_pk_kernel(unsigned long long *bad) {
    if (RZ) __builtin_amdgcn_s_setreg(0x801, 3u);
    const unsigned long long stride = (unsigned long long)gridDim.x * 256;
    unsigned cnt = 0;
    for (unsigned long long b = (unsigned long long)blockIdx.x * 256 + threadIdx.x; b < (1ull << 32); b += stride) {
        const float x = __uint_as_float((uint32_t)b);
        const uint32_t want = cvt_pk_model(x, RZ != 0);
        const uint32_t sel = (uint32_t)b & 3u, keep = 0xA5C3F17Eu;
        uint32_t got;
        switch (sel) { // (the byte selector is an immediate in the kernels)
        case 0: got = __builtin_amdgcn_cvt_pk_u8_f32(x, 0u, keep); break;
        case 1: got = __builtin_amdgcn_cvt_pk_u8_f32(x, 1u, keep); break;
        case 2: got = __builtin_amdgcn_cvt_pk_u8_f32(x, 2u, keep); break;
        default: got = __builtin_amdgcn_cvt_pk_u8_f32(x, 3u, keep); break;
        }
        const uint32_t exp = (keep & ~(0xffu << (8 * sel))) | (want << (8 * sel));
        cnt += got != exp;
    }
    if (cnt) atomicAdd(bad, (unsigned long long)cnt);
}
template <typename Launch> static unsigned long long run_selftest(hipStream_t s, Launch launch) {
    unsigned long long *d = nullptr, h = ~0ull;
    if (hipMalloc((void **)&d, sizeof(h)) != hipSuccess) {
        (void)hipGetLastError();
        return h;
    }
    if (hipMemsetAsync(d, 0, sizeof(h), s) == hipSuccess) {
        launch(d);
        if (hipGetLastError() != hipSuccess || hipMemcpyAsync(&h, d, sizeof(h), hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess) {
            (void)hipGetLastError();
            h = ~0ull;
        }
    }
    (void)hipFree(d);
    return h;
}
unsigned long dq_next_launch() {
    static std::atomic<unsigned long> n{0};
    return n.fetch_add(1);
}
unsigned long long selftest_rounding(int mode, bool u8, float lo, float hi, hipStream_t s) {
    return run_selftest(s, [&](unsigned long long *d) {
        const dim3 g(256 * 16), b(256);
        if (mode == 2) {
            if (u8) hipLaunchKernelGGL((selftest_rounding_kernel<2, 0x80808080u>), g, b, 0, s, lo, hi, d);
            else hipLaunchKernelGGL((selftest_rounding_kernel<2, 0u>), g, b, 0, s, lo, hi, d);
        } else if (mode == 3) {
            hipLaunchKernelGGL((selftest_rounding_kernel<3, 0u>), g, b, 0, s, lo, hi, d);
        } else {
            if (u8) hipLaunchKernelGGL((selftest_rounding_kernel<1, 0x80808080u>), g, b, 0, s, lo, hi, d);
            else hipLaunchKernelGGL((selftest_rounding_kernel<1, 0u>), g, b, 0, s, lo, hi, d);
        }
    });
}
unsigned long long selftest_cvt_pk(hipStream_t s) { // both modes in one count
    return run_selftest(s, [&](unsigned long long *d) {
        hipLaunchKernelGGL(selftest_cvt_pk_kernel<1>, dim3(256 * 16), dim3(256), 0, s, d);
        hipLaunchKernelGGL(selftest_cvt_pk_kernel<0>, dim3(256 * 16), dim3(256), 0, s, d);
    });
}
unsigned long long selftest_requant(int mode, bool u8, float A, float S, float lo, float hi, hipStream_t s) {
    return run_selftest(s, [&](unsigned long long *d) {
        const dim3 g(256 * 8), b(256);
        if (mode == 2) {
            if (u8) hipLaunchKernelGGL((selftest_requant_kernel<2, 0x80808080u>), g, b, 0, s, A, S, lo, hi, d);
            else hipLaunchKernelGGL((selftest_requant_kernel<2, 0u>), g, b, 0, s, A, S, lo, hi, d);
        } else {
            if (u8) hipLaunchKernelGGL((selftest_requant_kernel<1, 0x80808080u>), g, b, 0, s, A, S, lo, hi, d);
            else hipLaunchKernelGGL((selftest_requant_kernel<1, 0u>), g, b, 0, s, A, S, lo, hi, d);
        }
    });
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

} // namespace k
} // namespace mf
