// k_dwtask.hpp -- the per-lane depthwise 3x3 tasks (stride 1 and stride 2) shared by the layer-wise
// depthwise kernel (k_depthwise.hip) and the run-time-geometry depthwise kernel (k_rt.hip).
// (src/ops/depthwise_conv_2d.rs:50-104)
#pragma once
#include "k_common.hpp"

namespace mf {
namespace k {

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

} // namespace k
} // namespace mf
