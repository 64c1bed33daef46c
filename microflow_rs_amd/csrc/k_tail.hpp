// k_tail.hpp -- the network tail of one inference, shared by the standalone tail kernel (k_tail3.hip) and the
// late-stage kernel (k_stage.hip): AveragePool2D whose output is 1x1 (src/ops/average_pool_2d.rs:29-66) ->
// Conv2D 1x1 with N <= 8 outputs (src/ops/conv_2d.rs:28-108) -> [Reshape] -> Softmax over the N values
// (src/ops/softmax.rs:15-27).  One wavefront per inference: lane l owns channels 4l..4l+3 (+256 per extra pass),
// sums its taps with byte-masked sdot4, requantises the pool (an int8 tensor, like the reference's), takes its
// share of the N dot products, butterfly-reduces them across the wave, and lanes 0..N-1 finish the head
// epilogue and the table softmax.
#pragma once
#include "k_common.hpp"

namespace mf {
namespace k {

// x: the [H][W][C] int8 pool input of ONE inference (HBM or LDS); y: its N output bytes
template <int N>
__device__ __forceinline__ void tail_one(const int8_t *x, int8_t *y, const TailArgs &p, int lane) {
    const int C4 = p.C >> 2;
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
            const float xf = __fmul_rn(p.inv_len, (float)(sums[k] + p.pool_bias * p.ntaps)); // (1/len) * f32(sum of values)
            const float yv = __fadd_rn(__fmul_rn(p.pool_c0, xf), p.pool_c1); // c0 * x + c1
            const float r = __fadd_rn(yv, __builtin_copysignf(0x1.fffffep-2f, yv));
            int v = (r != r) ? 0 : (int)__builtin_amdgcn_fmed3f(r, p.pool_sat_lo, p.pool_sat_hi);
            v = max(v, p.pool_lo);
            q[k] = min(v, p.pool_hi) ^ p.xr; // the stored byte (pack4 keeps the low byte)
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
        e[n] = p.exp_table[(int)(int8_t)(h[n] ^ p.xr) + 128]; // table index = stored byte + 128
        sum = __fadd_rn(sum, e[n]); // one row: column-major order == index order
    }
#pragma unroll
    for (int n = 0; n < N; ++n) {
        const float prob = __fdiv_rn(e[n], sum);
        const float qf = __fadd_rn(__fdiv_rn(prob, p.sm_oscale), p.sm_ozp_f);
        const float r = __fadd_rn(qf, __builtin_copysignf(0x1.fffffep-2f, qf));
        const int yq = (r != r) ? 0 : (int)__builtin_amdgcn_fmed3f(r, p.sm_sat_lo, p.sm_sat_hi);
        if (lane == n) y[n] = (int8_t)(yq ^ p.xr);
    }
}

} // namespace k
} // namespace mf
