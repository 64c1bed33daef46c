// k_rt.hip -- the RUN-TIME-GEOMETRY fast kernels: DepthwiseConv2D 3x3 and Conv2D 1x1 for ANY height, width and
// (multiple-of-4 / multiple-of-16) channel count, with or without weight zero points.
//
// (src/ops/depthwise_conv_2d.rs:28-105, src/ops/conv_2d.rs:28-108.)  The reference compiles for any shape -- its
// shapes are const generics (depthwise_conv_2d.rs:28-49, conv_2d.rs:28-49).  The kernels in k_depthwise.hip /
// k_pointwise.hip / k_fused*.hip are instantiated per (H, W, C, stride[, N]) row of the tables in kernels.hpp -- tuned
// for person_detect.tflite.  An operator whose shape is in no table lands HERE instead of on the byte-wise `*_generic`
// kernels: same staging (LDS-DMA into halo'd tiles), same v_perm / v_dot4 tap arithmetic, same MFMA pixel matrix, same
// epilogue -- but every extent is a kernel argument, a large image is cut into row bands, and the weights of the 1x1
// convolution live in LDS instead of registers.  Arithmetic contract and helpers: k_common.hpp.
#include "k_common.hpp"

#include <algorithm>
#include <map>
#include <mutex>
#include <vector>

namespace mf {
namespace k {

// ------------------------------------------------------------------------
// dw3x3_rt -- DepthwiseConv2D 3x3, SAME, stride 1 or 2 (both axes), NHWC, C % 4 == 0.
//
//   step    : G whole images (small tensors) or ONE band of BH output rows of one image (large tensors); the dynamic
//             step queue of k_common.hpp deals the steps.
//   staging : the input rows a step needs (+ the halo rows above / below) by LDS-DMA, 1 KiB pieces, into tiles whose
//             side pads hold the input zero point; a row outside the image is written with the zero point instead
//             (band mode) or was filled once (whole-image mode).  Double buffered: the DMAs of the next step fly during
//             this step's taps.
//   task    : 2 output rows x 2 adjacent output pixels x 4 channels per lane (the dw_s1 / dw_s2 tasks of k_dwtask.hpp
//             with the row pitch and the pixel pitch in registers).  A lane's 4 channels never change (the active
//             thread count is a multiple of C / 4), so its 9 tap weights are registers.
//   WZ      : non-zero weight zero points (depthwise_conv_2d.rs:57-63, :70-75): acc -= wzp[c] * sum of the window, the
//             window sum being one more v_dot4 per filter row against the byte mask of the taps.
// ------------------------------------------------------------------------
template <int S, int R, bool WZ, int MG, uint32_t XR4>
__global__ __launch_bounds__(512) void dw3x3_rt(const int8_t *__restrict__ in, int8_t *__restrict__ out, DwRtArgs p, int batch) {
    // R = output rows per task (2, or 3 when the band height divides by 3); 256 or 512 threads (DwRtArgs::NTHR)
    const int NTHR = (int)blockDim.x, NWAVE = NTHR >> 6;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int C = p.C, C4 = C >> 2, ROW = p.ROW, LP = p.LP, TILE = p.TILE, BUF = p.BUF, G = p.G, RB = p.RB;
    const int H = p.H, OH = p.OH, OW = p.OW, ROWB = p.W * C, BH = p.BH, NBANDS = p.NBANDS;
    const int OWP = (OW + 1) >> 1, OHR = BH / R;

    DynSteps dq;
    dq.init(lds + 2 * BUF + 256, p.dw.queue, tid, p.dw.qcfg);
    for (int i = tid; i < (2 * BUF + 256) / 16; i += NTHR) ((uint4 *)lds)[i] = make_uint4(p.dw.izp4, p.dw.izp4, p.dw.izp4, p.dw.izp4);

    // ---- per-lane constants ----
    const int PPT = NTHR / C4;                       // pixel-pair tasks per pass of the workgroup
    const bool active = tid < PPT * C4;
    const int cg = tid % C4, pp0 = tid / C4;
    uint32_t wA[3][4], wB[3][4];                     // (w0,w1,w2,0) and (0,w0,w1,w2) per filter row and channel (wB: stride 1)
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const uint32_t w0 = ((const uint32_t *)p.dw.w)[(ky * 3 + 0) * C4 + cg];
        const uint32_t w1 = ((const uint32_t *)p.dw.w)[(ky * 3 + 1) * C4 + cg];
        const uint32_t w2 = ((const uint32_t *)p.dw.w)[(ky * 3 + 2) * C4 + cg];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            wA[ky][k] = ((w0 >> (8 * k)) & 0xffu) | (((w1 >> (8 * k)) & 0xffu) << 8) | (((w2 >> (8 * k)) & 0xffu) << 16);
            wB[ky][k] = wA[ky][k] << 8;
        }
    }
    const float4 A = ((const float4 *)p.dw.A)[cg], Sc = ((const float4 *)p.dw.S)[cg];
    const int4 Kc = magic4<MG>(((const int4 *)p.dw.Kc)[cg]);
    int4 wz = make_int4(0, 0, 0, 0);
    if constexpr (WZ) wz = ((const int4 *)p.wzp)[cg];
    // first task of this lane in a step, and the carry-free increments of one pass (PPT pixel pairs further)
    const int per_img = OHR * OWP;
    int g0 = pp0 / per_img, rp0 = (pp0 % per_img) / OWP, xp0 = pp0 % OWP;
    const int dx = PPT % OWP, drow = PPT / OWP, dr = drow % OHR, dg = drow / OHR;
    // strides of the walk in the staged tile (bytes) and in the output (dwords), and what a carry adds
    const int t_x = 2 * S * C, t_row = R * S * ROW;
    const int t_step = dg * TILE + dr * t_row + dx * t_x, t_cx = t_row - OWP * t_x, t_cr = TILE - OHR * t_row;
    const int o_x = 2 * C4, o_row = R * OW * C4, o_img = OH * OW * C4;
    const int o_step = dg * o_img + dr * o_row + dx * o_x, o_cx = o_row - OWP * o_x, o_cr = o_img - OHR * o_row;
    const int tbase0 = LP - C + cg * 4;              // window start of pixel pair 0: one pixel left of the image
    wg_sync(); // zero-point fill complete before any DMA lands

    auto stage = [&](int st, int buf) {
        const int band = st % NBANDS, ist = st / NBANDS;
        const int yfirst = band * BH * S - 1;        // input row held by tile row 0
        for (int g = 0; g < G; ++g) {
            const long img = (long)ist * G + g;
            if (img >= batch) break;
            for (int r = wave; r < RB; r += NWAVE) {
                const int y = yfirst + r;
                uint8_t *dst = lds + buf * BUF + g * TILE + r * ROW + LP;
                if (y >= 0 && y < H) {
                    const int8_t *src = in + (img * H + y) * (long)ROWB;
                    for (int o = 0; o < ROWB; o += 1024)
                        if (o + lane * 16 < ROWB) dma16(src + o + lane * 16, dst + o);
                } else if (NBANDS > 1) {             // band mode: this tile row is padding in this step only
                    for (int o = lane * 16; o < ROWB; o += 1024) *(uint4 *)(dst + o) = make_uint4(p.dw.izp4, p.dw.izp4, p.dw.izp4, p.dw.izp4);
                }
            }
        }
    };

    const int nsteps = ((batch + G - 1) / G) * NBANDS;
    int cur = 0;
    if (dq.step < nsteps) stage(dq.step, 0);
    for (; dq.step < nsteps; dq.advance(tid), cur ^= 1) {
        const int step = dq.step;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wg_sync();
        dq.top(tid);
        if (dq.nxt < nsteps) stage(dq.nxt, cur ^ 1);

        const int band = step % NBANDS, ist = step / NBANDS;
        const int gvalid = min(G, batch - ist * G);
        const int oyb = band * BH;
        const uint8_t *tile = lds + cur * BUF;
        uint32_t *dst = (uint32_t *)out + (size_t)ist * G * OH * OW * C4;
        const int obase0 = oyb * OW * C4 + cg;       // (band mode: the band's first output row)
        // task walk with incremental addresses: no multiplication per task (the strides are runtime values)
        int g = g0, rp = rp0, xp = xp0;
        int toff = g0 * TILE + rp0 * t_row + xp0 * t_x;          // byte offset of the task's window in the staged tile
        int ooff = (g0 * OH + R * rp0) * OW * C4 + xp0 * 2 * C4; // dword offset of its first output in the step's output
        while (active && g < gvalid) {
            const int oy0 = oyb + R * rp, ox0 = 2 * xp;
            if (oy0 < OH) {
                const uint8_t *base = tile + tbase0 + toff;
                int o0[R][4], o1[R][4], s0[R][4], s1[R][4];
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    o0[j][0] = o1[j][0] = Kc.x, o0[j][1] = o1[j][1] = Kc.y, o0[j][2] = o1[j][2] = Kc.z, o0[j][3] = o1[j][3] = Kc.w;
#pragma unroll
                    for (int k = 0; k < 4; ++k) s0[j][k] = s1[j][k] = 0;
                }
                constexpr int NIN = S == 1 ? R + 2 : 2 * R + 1; // input rows of the task
                const uint8_t *rowp = base;
#pragma unroll
                for (int r = 0; r < NIN; ++r, rowp += ROW) {
                    const uint32_t v0 = *(const uint32_t *)(rowp);
                    const uint32_t v1 = *(const uint32_t *)(rowp + C);
                    const uint32_t v2 = *(const uint32_t *)(rowp + 2 * C);
                    const uint32_t v3 = *(const uint32_t *)(rowp + 3 * C);
                    const uint32_t ab_lo = __builtin_amdgcn_perm(v1, v0, 0x05010400u), ab_hi = __builtin_amdgcn_perm(v1, v0, 0x07030602u);
                    const uint32_t cd_lo = __builtin_amdgcn_perm(v3, v2, 0x05010400u), cd_hi = __builtin_amdgcn_perm(v3, v2, 0x07030602u);
                    uint32_t win[4], winb[4];
                    win[0] = __builtin_amdgcn_perm(cd_lo, ab_lo, 0x05040100u), win[1] = __builtin_amdgcn_perm(cd_lo, ab_lo, 0x07060302u);
                    win[2] = __builtin_amdgcn_perm(cd_hi, ab_hi, 0x05040100u), win[3] = __builtin_amdgcn_perm(cd_hi, ab_hi, 0x07060302u);
                    if constexpr (S == 2) {          // [v2, v3, v4 (byte k of the fifth pixel), 0]
                        const uint32_t v4 = *(const uint32_t *)(rowp + 4 * C);
                        winb[0] = __builtin_amdgcn_perm(v4, win[0], 0x0c040302u), winb[1] = __builtin_amdgcn_perm(v4, win[1], 0x0c050302u);
                        winb[2] = __builtin_amdgcn_perm(v4, win[2], 0x0c060302u), winb[3] = __builtin_amdgcn_perm(v4, win[3], 0x0c070302u);
                    }
#pragma unroll
                    for (int j = 0; j < R; ++j) {
                        const int ky = r - S * j;    // filter row this input row plays for output row j
                        if (ky >= 0 && ky <= 2) {
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                if constexpr (S == 1) {
                                    o0[j][k] = ky == 0 ? sdot4_first(win[k], wA[0][k], o0[j][k]) : sdot4(win[k], wA[ky][k], o0[j][k]);
                                    o1[j][k] = ky == 0 ? sdot4_first(win[k], wB[0][k], o1[j][k]) : sdot4(win[k], wB[ky][k], o1[j][k]);
                                    if constexpr (WZ) s0[j][k] = sdot4(win[k], 0x00010101u, s0[j][k]), s1[j][k] = sdot4(win[k], 0x01010100u, s1[j][k]);
                                } else {
                                    o0[j][k] = ky == 0 ? sdot4_first(win[k], wA[0][k], o0[j][k]) : sdot4(win[k], wA[ky][k], o0[j][k]);
                                    o1[j][k] = ky == 0 ? sdot4_first(winb[k], wA[0][k], o1[j][k]) : sdot4(winb[k], wA[ky][k], o1[j][k]);
                                    if constexpr (WZ) s0[j][k] = sdot4(win[k], 0x00010101u, s0[j][k]), s1[j][k] = sdot4(winb[k], 0x00010101u, s1[j][k]);
                                }
                            }
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    if (oy0 + j < OH) {
                        if constexpr (WZ) {
                            o0[j][0] -= wz.x * s0[j][0], o0[j][1] -= wz.y * s0[j][1], o0[j][2] -= wz.z * s0[j][2], o0[j][3] -= wz.w * s0[j][3];
                            o1[j][0] -= wz.x * s1[j][0], o1[j][1] -= wz.y * s1[j][1], o1[j][2] -= wz.z * s1[j][2], o1[j][3] -= wz.w * s1[j][3];
                        }
                        uint32_t *dp = dst + obase0 + ooff + j * OW * C4;
                        dp[0] = requant_pack4<MG, XR4>(o0[j][0], o0[j][1], o0[j][2], o0[j][3], A, Sc, p.dw.lo_f, p.dw.hi_f);
                        if (ox0 + 1 < OW) dp[C4] = requant_pack4<MG, XR4>(o1[j][0], o1[j][1], o1[j][2], o1[j][3], A, Sc, p.dw.lo_f, p.dw.hi_f);
                    }
                }
            }
            // next task of this lane: PPT pixel pairs further (single carries by construction)
            xp += dx, rp += dr, g += dg;
            toff += t_step, ooff += o_step;
            if (xp >= OWP) xp -= OWP, ++rp, toff += t_cx, ooff += o_cx;
            if (rp >= OHR) rp -= OHR, ++g, toff += t_cr, ooff += o_cr;
        }
    }
    dq.finish(tid);
}

// ------------------------------------------------------------------------
// pw_rt_lds -- Conv2D 1x1, stride 1, as an int8 MFMA product over the batch's pixel matrix: K % 16 == 0 input channels
// (the host presents K = 8 / K = 4 as K = 16 on pixel pairs / quads with block-diagonal weights), N % 4 == 0 outputs.
//
//   weights : operand A of v_mfma_i32_16x16x64_i8 for every (16-channel tile, 64-deep k step), built by the host
//             ([tile][k step][lane] x 16 bytes, zero beyond K and N), copied to LDS once per workgroup.
//   pixels  : a wave takes 16 pixels at a time; lane (column, g) loads its 16 operand bytes of each k step straight
//             from HBM in MFMA layout (the [pixel][K] matrix IS the batch), all k steps before the first MFMA.
//   results : a lane ends a tile with 4 consecutive channels of its pixel = one packed dword, written to a per-wave
//             LDS patch [16 pixels][N]; the patch is 16 N contiguous output bytes and leaves with 16-byte stores.
//   WZ      : weight zero points (conv_2d.rs:57-63): acc -= wzp[n] * sum_k x[pixel][k], the row sum coming from one
//             more MFMA tile of ones.
// ------------------------------------------------------------------------
template <bool WZ, int MG, uint32_t XR4>
__global__ __launch_bounds__(256) void pw_rt_lds(const int8_t *__restrict__ in, int8_t *__restrict__ out, PwRtArgs p, long long npix) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K = p.K, N = p.N, KS = p.KS, NT = p.NT;
    const int WBYTES = NT * KS * 1024;               // operand A image (+ the ones tile when WZ)
    const int CONST_OFF = WBYTES + (WZ ? KS * 1024 : 0);
    const int NP = NT * 16;                          // padded channel count of the constant tables
    const int PATCH_OFF = CONST_OFF + NP * 16;       // A, S, Kc, wzp: four dwords per channel
    const int PITCH = p.patch_pitch;                 // bytes per pixel row of a patch (N rounded up to 16)
    for (int i = tid; i < CONST_OFF / 16; i += 256) ((uint4 *)lds)[i] = ((const uint4 *)p.wprep)[i];
    for (int i = tid; i < NP; i += 256) {
        const bool in_range = i < N;
        ((float *)(lds + CONST_OFF))[i] = in_range ? p.A[i] : 0.0f;
        ((float *)(lds + CONST_OFF + NP * 4))[i] = in_range ? p.S[i] : 0.0f;
        ((int *)(lds + CONST_OFF + NP * 8))[i] = (in_range ? p.Kc[i] : 0) + (MG != 0 ? MF_MAGIC_I : 0);
        ((int *)(lds + CONST_OFF + NP * 12))[i] = (WZ && in_range) ? p.wzp[i] : 0;
    }
    wg_sync();
    const int col = lane & 15, g = lane >> 4;
    uint8_t *patch = lds + PATCH_OFF + wave * 16 * PITCH;
    const long long nchunks = (npix + 15) / 16;
    constexpr int KSMAX = 8;                         // K <= 512
    for (long long chunk = (long long)blockIdx.x * 4 + wave; chunk < nchunks; chunk += (long long)gridDim.x * 4) {
        long long pix = chunk * 16 + col;
        pix = pix < npix ? pix : npix - 1;           // a ragged last chunk re-reads the last pixel
        const int8_t *src = in + pix * K;
        v4i B[KSMAX];
#pragma unroll
        for (int ks = 0; ks < KSMAX; ++ks) {
            if (ks < KS) {
                // the last k step may hang over K: those lanes' weights are zero, any readable bytes of the row will do
                const int k0 = ks * 64 + g * 16;
                B[ks] = *(const v4i *)(src + (k0 < K ? k0 : 0));
            }
        }
        int rowsum = 0;
        if constexpr (WZ) {
            v4i acc = {0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < KSMAX; ++ks)
                if (ks < KS) acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(*(const v4i *)(lds + WBYTES + ks * 1024 + lane * 16), B[ks], acc, 0, 0, 0);
            rowsum = acc[0];                         // every row of the ones tile is sum_k x[pixel][k]
        }
        for (int nt = 0; nt < NT; ++nt) {
            const int ch = nt * 16 + g * 4;
            const int4 kc = *(const int4 *)(lds + CONST_OFF + NP * 8 + ch * 4);
            v4i acc = {kc.x, kc.y, kc.z, kc.w};
#pragma unroll
            for (int ks = 0; ks < KSMAX; ++ks)
                if (ks < KS) acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(*(const v4i *)(lds + (nt * KS + ks) * 1024 + lane * 16), B[ks], acc, 0, 0, 0);
            if constexpr (WZ) {
                const int4 wz = *(const int4 *)(lds + CONST_OFF + NP * 12 + ch * 4);
                acc[0] -= wz.x * rowsum, acc[1] -= wz.y * rowsum, acc[2] -= wz.z * rowsum, acc[3] -= wz.w * rowsum;
            }
            const float4 a = *(const float4 *)(lds + CONST_OFF + ch * 4), s = *(const float4 *)(lds + CONST_OFF + NP * 4 + ch * 4);
            const uint32_t d = requant_pack4<MG, XR4>(acc[0], acc[1], acc[2], acc[3], a, s, p.lo_f, p.hi_f);
            if (ch < N) *(uint32_t *)(patch + col * PITCH + ch) = d;
        }
        __builtin_amdgcn_wave_barrier();
        // patch rows are N bytes of consecutive pixels: 16 N contiguous output bytes (the last chunk may be short)
        const long long first = chunk * 16;
        const int valid = (int)((npix - first) < 16 ? (npix - first) : 16);
        int8_t *o = out + first * N;
        if (PITCH == N) {
            for (int off = lane * 16; off < valid * N; off += 1024) *(uint4 *)(o + off) = *(const uint4 *)(patch + off);
        } else {                                     // N not a multiple of 16: dword granularity, row by row
            const int n4 = N >> 2;
            for (int e = lane; e < valid * n4; e += 64) {
                const int r = e / n4, c = e - r * n4;
                *(uint32_t *)(o + (size_t)r * N + c * 4) = *(const uint32_t *)(patch + r * PITCH + c * 4);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------------
// pw_rt -- the same product with the weights in REGISTERS (weight zero points == 0): the form pw_mfma<K, N> has, with K
// and N as arguments.
//
//   rows    : `group` consecutive pixels form one row of the product when K < 64 (block-diagonal weights built by the
//             host), so that the 64-deep k step of the MFMA is filled and a row's output is >= 64 contiguous bytes.
//   split   : the N' = N * group output channels are cut into blocks of TB <= 4 tiles; NSPLIT = 1, 2 or 4 waves share a
//             16-row chunk, each holding ITS block's operand A for every k step in registers (KSC = 1, 2, 4 or 8 k steps
//             is the template parameter) and reading the same operand B (L1 / L2 hits).
//   rows of a tile are permuted on the host (row 4 gr + i of tile t = channel base + 4 TB gr + 4 t + i) so that a lane
//             ends with 4 TB CONSECUTIVE output bytes of its row: one 4 .. 16-byte store, 64 .. 1024 contiguous bytes
//             per wave-level store.
//   U chunks are in flight per wave: all their loads are issued before the first MFMA.
// ------------------------------------------------------------------------
template <int KSC, int MG, uint32_t XR4>
__global__ __launch_bounds__(256) void pw_rt(const int8_t *__restrict__ in, int8_t *__restrict__ out, PwRtArgs p, long long nrows) {
    constexpr int U = KSC <= 2 ? 4 : 2;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int K = p.K, N = p.N, KS = p.KS, TB = p.TB, NSPLIT = p.NSPLIT;
    const int col = lane & 15, g = lane >> 4;
    const int blk = wave % NSPLIT, slot = wave / NSPLIT, SLOTS = 4 / NSPLIT;
    v4i Aw[4][KSC];
    float4 cA[4], cS[4];
    int4 cK[4];
    const int ch0 = blk * 16 * TB + g * 4 * TB;      // this lane's first channel
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int ks = 0; ks < KSC; ++ks) {
            Aw[t][ks] = v4i{0, 0, 0, 0};
            if (t < TB && ks < KS) Aw[t][ks] = ((const v4i *)p.wprep)[(((size_t)blk * TB + t) * KS + ks) * 64 + lane];
        }
        const int ch = ch0 + 4 * t;
        const bool live = t < TB && ch < N;
        cA[t] = live ? *(const float4 *)(p.A + ch) : make_float4(0.f, 0.f, 0.f, 0.f);
        cS[t] = live ? *(const float4 *)(p.S + ch) : make_float4(0.f, 0.f, 0.f, 0.f);
        cK[t] = magic4<MG>(live ? *(const int4 *)(p.Kc + ch) : make_int4(0, 0, 0, 0));
    }
    const long long nchunks = (nrows + 15) / 16;
    for (long long cb = ((long long)blockIdx.x * SLOTS + slot) * U; cb < nchunks; cb += (long long)gridDim.x * SLOTS * U) {
        v4i B[U][KSC];
        long long row[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            row[u] = (cb + u) * 16 + col;
            const long long r = row[u] < nrows ? row[u] : nrows - 1; // chunks past the end re-read the last row
            const int8_t *src = in + r * K;
#pragma unroll
            for (int ks = 0; ks < KSC; ++ks) {
                const int k0 = ks * 64 + g * 16;                     // a k step hanging over K meets zero weights
                if (ks < KS) B[u][ks] = *(const v4i *)(src + (k0 < K ? k0 : 0));
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            uint32_t packed[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (t < TB) {
                    v4i acc = {cK[t].x, cK[t].y, cK[t].z, cK[t].w};
#pragma unroll
                    for (int ks = 0; ks < KSC; ++ks)
                        if (ks < KS) acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(Aw[t][ks], B[u][ks], acc, 0, 0, 0);
                    packed[t] = requant_pack4<MG, XR4>(acc[0], acc[1], acc[2], acc[3], cA[t], cS[t], p.lo_f, p.hi_f);
                }
            }
            if (row[u] < nrows) {
                int8_t *o = out + row[u] * N + ch0;
                if (TB == 4 && (N & 15) == 0 && ch0 + 16 <= N) { // (16-byte stores need rows that are whole 16-byte groups)
                    *(uint4 *)o = make_uint4(packed[0], packed[1], packed[2], packed[3]);
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        if (t < TB && ch0 + 4 * t < N) *(uint32_t *)(o + 4 * t) = packed[t];
                }
            }
        }
    }
}

// ------------------------------------------------------------------------
// conv_rows_lds -- Conv2D with FEW input channels, any filter / stride / padding (a network's first convolution:
// 3x3x3 -> 16 stride 2 and the like; src/ops/conv_2d.rs:28-108), and DepthwiseConv2D with ONE input channel and a
// depth multiplier (src/ops/depthwise_conv_2d.rs:67: every output channel reads input channel 0).
//
// With few channels the taps of one filter row are KW * C CONSECUTIVE bytes of the NHWC image row, and the same holds
// for the filter ([N][KH][KW][C]): a window row is KG = ceil(KW C / 4) dwords and every (filter row, dword group,
// output channel) is one real 4-MAC v_dot4.
//   step    : G whole images per workgroup step; the NEXT step's images are loaded into registers (dwords, 16 per thread)
//             before this step's compute and written to the LDS tiles after it -- inside a halo of the input zero point,
//             so that SAME padding needs no per-tap test.
//   item    : one output pixel x 8 output channels per thread: the window row is read as KG + 1 aligned dwords and
//             shifted with v_alignbyte; the packed weights ([ky][group][8 channels] dwords) are LDS broadcasts.
//   WZ      : filter zero points (conv_2d.rs:57-63): acc -= fzp[n] * (sum of the window), one more v_dot4 per row group
//             against the byte mask of the real taps.
// ------------------------------------------------------------------------
template <bool WZ, int MG, uint32_t XR4>
__global__ __launch_bounds__(512) void conv_rows_lds(const int8_t *__restrict__ in, int8_t *__restrict__ out, ConvRowsArgs p, int batch) {
    constexpr int NTHR = 512, MAXE = 16;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int tid = threadIdx.x;
    const int KG = p.KG, KH = p.KH, TWP = p.TWP, TILE = p.TILE, G = p.G, NP = p.NP, NG8 = NP >> 3, N = p.N;
    const int W_OFF = G * TILE;                          // [KH][KG][NP] weight dwords
    const int M_OFF = W_OFF + KH * KG * NP * 4;          // [KG] byte masks of the real taps (WZ)
    const int C_OFF = M_OFF + ((KG * 4 + 15) & ~15);     // A, S, Kc, fzp: [NP] each
    for (int i = tid; i < G * TILE / 4; i += NTHR) ((uint32_t *)lds)[i] = p.izp4;
    for (int i = tid; i < KH * KG * NP; i += NTHR) ((uint32_t *)(lds + W_OFF))[i] = p.wpack[i];
    for (int i = tid; i < KG; i += NTHR) ((uint32_t *)(lds + M_OFF))[i] = p.mask[i];
    for (int i = tid; i < NP; i += NTHR) {
        const bool live = i < N;
        ((float *)(lds + C_OFF))[i] = live ? p.A[i] : 0.0f;
        ((float *)(lds + C_OFF + NP * 4))[i] = live ? p.S[i] : 0.0f;
        ((int *)(lds + C_OFF + NP * 8))[i] = (live ? p.Kc[i] : 0) + (MG != 0 ? MF_MAGIC_I : 0);
        ((int *)(lds + C_OFF + NP * 12))[i] = (WZ && live) ? p.wzp[i] : 0;
    }
    // staging map (the same for every step): dword e of this thread is image g, row y, dword x of the step
    const int ROWD = p.ROWB >> 2, IMGD = p.H * ROWD, STEPD = G * IMGD;
    int gofs[MAXE], lofs[MAXE];
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
        const int idx = tid + NTHR * e;
        const int g = idx / IMGD, r = idx - g * IMGD, y = r / ROWD, x = r - y * ROWD;
        gofs[e] = idx < STEPD ? idx : -1;
        lofs[e] = g * TILE + (y + p.shy) * TWP + p.XO + 4 * x;
    }
    const int nsteps = (batch + G - 1) / G;
    uint32_t v[MAXE];
    auto fetch = [&](int st) {
        const uint32_t *src = (const uint32_t *)in + (size_t)st * STEPD;
        const long lim = ((long)batch - (long)st * G) * IMGD; // dwords of the step that exist (a ragged last step)
#pragma unroll
        for (int e = 0; e < MAXE; ++e) {
            // (select the VALUE, not the address: `cond ? src[i] : p.izp4` made the compiler choose between a global and a
            // kernel-argument address, i.e. a flat load that also counts on lgkmcnt)
            const bool ok = gofs[e] >= 0 && gofs[e] < lim;
            const uint32_t got = src[ok ? gofs[e] : 0];
            v[e] = ok ? got : p.izp4;
        }
    };
    const float inv_ow = 1.0f / (float)p.OW, inv_opix = 1.0f / (float)(p.OH * p.OW);
    const int OPIX = p.OH * p.OW;
    int step = blockIdx.x;
    if (step < nsteps) fetch(step);
    for (; step < nsteps; step += gridDim.x) {
        wg_sync();                                 // the previous step's compute is done with the tiles
#pragma unroll
        for (int e = 0; e < MAXE; ++e)
            if (gofs[e] >= 0) *(uint32_t *)(lds + lofs[e]) = v[e];
        wg_sync();
        if (step + gridDim.x < nsteps) fetch(step + gridDim.x); // in flight during the compute below
        const int gvalid = min(G, batch - step * G);
        for (int po = tid; po < G * OPIX; po += NTHR) {
            // pixel item -> (image, row, column); exact float divisions (indices < 2^22)
            const int g = (int)(((float)po + 0.5f) * inv_opix);
            const int o = po - g * OPIX;
            const int oy = (int)(((float)o + 0.5f) * inv_ow), ox = o - oy * p.OW;
            if (g >= gvalid) continue;
            for (int cgp = 0; cgp < NG8; ++cgp) {        // 8 output channels at a time
            int acc[8], wsum = 0;
            const int4 k0 = *(const int4 *)(lds + C_OFF + NP * 8 + cgp * 32), k1 = *(const int4 *)(lds + C_OFF + NP * 8 + cgp * 32 + 16);
            acc[0] = k0.x, acc[1] = k0.y, acc[2] = k0.z, acc[3] = k0.w, acc[4] = k1.x, acc[5] = k1.y, acc[6] = k1.z, acc[7] = k1.w;
            const int base0 = g * TILE + (oy * p.sh) * TWP + ox * p.sw * p.C + p.X0;
            for (int ky = 0; ky < KH; ++ky) {
                const int base = base0 + ky * TWP;
                const uint32_t *row = (const uint32_t *)(lds + (base & ~3));
                const uint32_t sh = (uint32_t)(base & 3);
                uint32_t lo = row[0];
                for (int kg = 0; kg < KG; ++kg) {
                    const uint32_t hi = row[kg + 1];
                    const uint32_t val = __builtin_amdgcn_alignbyte(hi, lo, sh); // window bytes 4 kg .. 4 kg + 3 of this row
                    lo = hi;
                    const uint4 w0 = *(const uint4 *)(lds + W_OFF + ((ky * KG + kg) * NP + cgp * 8) * 4);
                    const uint4 w1 = *(const uint4 *)(lds + W_OFF + ((ky * KG + kg) * NP + cgp * 8) * 4 + 16);
                    acc[0] = sdot4(val, w0.x, acc[0]), acc[1] = sdot4(val, w0.y, acc[1]);
                    acc[2] = sdot4(val, w0.z, acc[2]), acc[3] = sdot4(val, w0.w, acc[3]);
                    acc[4] = sdot4(val, w1.x, acc[4]), acc[5] = sdot4(val, w1.y, acc[5]);
                    acc[6] = sdot4(val, w1.z, acc[6]), acc[7] = sdot4(val, w1.w, acc[7]);
                    if constexpr (WZ) wsum = sdot4(val, ((const uint32_t *)(lds + M_OFF))[kg], wsum);
                }
            }
            if constexpr (WZ) {
                const int4 z0 = *(const int4 *)(lds + C_OFF + NP * 12 + cgp * 32), z1 = *(const int4 *)(lds + C_OFF + NP * 12 + cgp * 32 + 16);
                acc[0] -= z0.x * wsum, acc[1] -= z0.y * wsum, acc[2] -= z0.z * wsum, acc[3] -= z0.w * wsum;
                acc[4] -= z1.x * wsum, acc[5] -= z1.y * wsum, acc[6] -= z1.z * wsum, acc[7] -= z1.w * wsum;
            }
            const float4 a0 = *(const float4 *)(lds + C_OFF + cgp * 32), a1 = *(const float4 *)(lds + C_OFF + cgp * 32 + 16);
            const float4 s0 = *(const float4 *)(lds + C_OFF + NP * 4 + cgp * 32), s1 = *(const float4 *)(lds + C_OFF + NP * 4 + cgp * 32 + 16);
            const uint32_t d0 = requant_pack4<MG, XR4>(acc[0], acc[1], acc[2], acc[3], a0, s0, p.lo_f, p.hi_f);
            const uint32_t d1 = requant_pack4<MG, XR4>(acc[4], acc[5], acc[6], acc[7], a1, s1, p.lo_f, p.hi_f);
            int8_t *dst = out + ((size_t)(step * G + g) * OPIX + o) * N + cgp * 8;
            if ((N & 7) == 0) {
                st_out_t<false>(dst, make_uint2(d0, d1));
            } else {                                     // N not a multiple of 8: byte stores of the channels that exist
                const int nleft = N - cgp * 8;
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    if (c < nleft) dst[c] = (int8_t)(((c < 4 ? d0 : d1) >> (8 * (c & 3))) & 0xffu);
            }
            }
        }
    }
}

// ------------------------------------------------------------------------
// conv_mm_rt -- Conv2D with any filter KH x KW, strides, SAME / VALID padding, C % 16 == 0 input channels, N % 4 == 0
// outputs (src/ops/conv_2d.rs:28-108), as an int8 MFMA product over K = KH * KW * C.
//
// In NHWC the filter [N][KH][KW][C] IS the row-major [N][K] matrix, and the 16 operand bytes a lane supplies for k-bytes
// 64 ks + 16 g .. of its pixel are 16 consecutive channels of ONE tap (C % 16 == 0) = 16 consecutive bytes of the staged
// tile: address = (the pixel's window start) + (a per-(k step, lane group) offset from a small table).  No im2col.
//   step    : G whole images or one band of output rows, staged by LDS-DMA into halo'd tiles as in dw3x3_rt (one staging
//             buffer: the product is matrix-pipe work, several workgroups per CU cover each other's DMA waits).
//   weights : operand A for every (block of <= 4 tiles, tile, k step) in LDS, rows permuted so that a lane ends with
//             4 TB consecutive output channels (as pw_rt); read once per 16-pixel chunk and block.
//   WZ      : filter zero points (conv_2d.rs:57-63): one more tile of ones gives the window sum.
// ------------------------------------------------------------------------
//   NTHR    : 256, several workgroups per CU -- or, when the weight image leaves room for ONE workgroup per CU only, 1024: the
//             16 waves share that one copy of the weights and a step of up to 32 images (round 3 ran such shapes with four
//             waves per CU, each step a DMA wait and two barriers for 4 - 7 images: 85 - 139 TMAC/s, 0.04 VALU instructions
//             per clock and SIMD).
template <bool WZ, int MG, uint32_t XR4, int NTHR>
__global__ __launch_bounds__(NTHR) void conv_mm_rt(const int8_t *__restrict__ in, int8_t *__restrict__ out, ConvMmArgs p, int batch) {
    constexpr int NWAVE = NTHR / 64;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int C = p.C, N = p.N, KS = p.KS, TB = p.TB, NBLK = p.NBLK, ROW = p.ROW, TILE = p.TILE, G = p.G, RB = p.RB;
    const int H = p.H, OH = p.OH, OW = p.OW, ROWB = p.W * C, BH = p.BH, NBANDS = p.NBANDS;
    const int T_OFF = 0, W_OFF = G * TILE + 256;
    const int WBYTES = NBLK * TB * KS * 1024;
    const int ONES_OFF = W_OFF + WBYTES;
    const int O_OFF = ONES_OFF + (WZ ? KS * 1024 : 0);  // [KS][4] tap offsets
    for (int i = tid; i < (G * TILE + 256) / 16; i += NTHR) ((uint4 *)lds)[i] = make_uint4(p.izp4, p.izp4, p.izp4, p.izp4);
    for (int i = tid; i < (WBYTES + (WZ ? KS * 1024 : 0)) / 16; i += NTHR) ((uint4 *)(lds + W_OFF))[i] = ((const uint4 *)p.wprep)[i];
    for (int i = tid; i < KS * 4; i += NTHR) ((int *)(lds + O_OFF))[i] = p.tap_off[i];
    const int col = lane & 15, g = lane >> 4;
    const float inv_ow = 1.0f / (float)OW, inv_bp = 1.0f / (float)(BH * OW);
    const int nsteps = ((batch + G - 1) / G) * NBANDS;
    wg_sync();
    for (int step = blockIdx.x; step < nsteps; step += gridDim.x) {
        const int band = step % NBANDS, ist = step / NBANDS;
        const int yfirst = band * BH * p.sh - p.padt;   // input row held by tile row 0
        wg_sync();                                 // the previous step's reads of the tile are done
        for (int gi = 0; gi < G; ++gi) {
            const long img = (long)ist * G + gi;
            if (img >= batch) break;
            for (int r = wave; r < RB; r += NWAVE) {
                const int y = yfirst + r;
                uint8_t *dst = lds + T_OFF + gi * TILE + r * ROW + p.LP;
                if (y >= 0 && y < H) {
                    const int8_t *src = in + (img * H + y) * (long)ROWB;
                    for (int o = 0; o < ROWB; o += 1024)
                        if (o + lane * 16 < ROWB) dma16(src + o + lane * 16, dst + o);
                } else if (NBANDS > 1) {
                    for (int o = lane * 16; o < ROWB; o += 1024) *(uint4 *)(dst + o) = make_uint4(p.izp4, p.izp4, p.izp4, p.izp4);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wg_sync();
        const int gvalid = min(G, batch - ist * G);
        const int rows_here = min(BH, OH - band * BH);
        const int npix = gvalid * BH * OW;               // (rows past the image are masked below)
        for (int chunk = wave; chunk * 16 < npix; chunk += NWAVE) {
            const int pp = chunk * 16 + col;
            const int pc = pp < npix ? pp : npix - 1;
            const int gi = (int)(((float)pc + 0.5f) * inv_bp);
            const int rr = pc - gi * BH * OW;
            const int oyl = (int)(((float)rr + 0.5f) * inv_ow), ox = rr - oyl * OW;
            const bool live = pp < npix && oyl < rows_here;
            const uint8_t *win = lds + T_OFF + gi * TILE + (oyl * p.sh) * ROW + p.LP + (ox * p.sw - p.padl) * C;
            int rowsum = 0;
            if constexpr (WZ) {
                v4i acc = {0, 0, 0, 0};
                for (int ks = 0; ks < KS; ++ks) {
                    const v4i b = *(const v4i *)(win + ((const int *)(lds + O_OFF))[ks * 4 + g]);
                    acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(*(const v4i *)(lds + ONES_OFF + ks * 1024 + lane * 16), b, acc, 0, 0, 0);
                }
                rowsum = acc[0];
            }
            const size_t opix = ((size_t)(ist * G + gi) * OH + band * BH + oyl) * OW + ox;
            for (int blk = 0; blk < NBLK; ++blk) {
                const int ch0 = blk * 16 * TB + g * 4 * TB;
                v4i acc[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int ch = ch0 + 4 * t;
                    int4 kc = make_int4(0, 0, 0, 0);
                    if (t < TB && ch < N) kc = *(const int4 *)(p.Kc + ch);
                    kc = magic4<MG>(kc);
                    acc[t] = v4i{kc.x, kc.y, kc.z, kc.w};
                }
                // two-deep software pipeline: the tap offset of step ks + 2 and the operand B of step ks + 1 are in flight
                // while step ks multiplies
                const int *tab = (const int *)(lds + O_OFF) + g;
                int off1 = KS > 1 ? tab[4] : 0;
                v4i b = *(const v4i *)(win + tab[0]);
                for (int ks = 0; ks < KS; ++ks) {
                    const int off2 = ks + 2 < KS ? tab[(ks + 2) * 4] : 0;
                    const v4i bn = *(const v4i *)(win + off1);
                    const uint8_t *wa = lds + W_OFF + ((size_t)(blk * TB) * KS + ks) * 1024 + lane * 16;
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        if (t < TB) acc[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(*(const v4i *)(wa + (size_t)t * KS * 1024), b, acc[t], 0, 0, 0);
                    b = bn, off1 = off2;
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int ch = ch0 + 4 * t;
                    if (t < TB && ch < N) {
                        if constexpr (WZ) {
                            const int4 wz = *(const int4 *)(p.wzp + ch);
                            acc[t][0] -= wz.x * rowsum, acc[t][1] -= wz.y * rowsum, acc[t][2] -= wz.z * rowsum, acc[t][3] -= wz.w * rowsum;
                        }
                        const uint32_t d = requant_pack4<MG, XR4>(acc[t][0], acc[t][1], acc[t][2], acc[t][3], *(const float4 *)(p.A + ch),
                                                                  *(const float4 *)(p.S + ch), p.lo_f, p.hi_f);
                        if (live) *(uint32_t *)(out + opix * N + ch) = d;
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------
// dw_mm_rt -- DepthwiseConv2D with ANY filter KH x KW (<= 7 x 7), strides and SAME / VALID padding, C % 16 == 0, one output per input
// channel, zero filter zero points (src/ops/depthwise_conv_2d.rs:28-105; the 3x3 SAME stride-1 / 2 family has its own kernels).
//
// conv_mm_rt's staging and pixel walk with the depthwise contraction of k_fused_mm.hip: for a 16-channel group the taps are a
// block-diagonal K = 16 KH KW product -- lane group g of k step ks supplies the group's 16 bytes of tap 4 ks + g (one ds_read_b128
// at (window start) + (table offset of the tap) + 16 (group)), operand A holds the tap's weight of channel r in byte r of row r
// (ops.hip build_dw_mm_rt_weights).  A wave works through a contiguous range of (group, 16-pixel chunk) items with the group's operand
// A, tap offsets and epilogue constants in registers; a lane ends an item with 4 consecutive channels of its pixel = one packed dword,
// which goes through a 256-byte per-wave LDS patch so that 16 lanes store a pixel's 16 bytes each.  (First versions, 24x24x32 5x5 at
// batch 65 536: constants and stores per lane from / to device memory 1.08 TB/s; operand A per k step from LDS 1.58 TB/s.)
// ------------------------------------------------------------------------
// Knock-out timing experiments (WRONG results, never shipped): 1 conflict-free linear operand-B reads, 2 no HBM stores, 4 no
// requantisation, 8 no staging after a workgroup's first step.  24x24x32 5x5 (profiles/r06/n_dwmm_layout_variants.txt): 1.00-1.03 ms ->
// 0.88-0.91 / 0.86 / 0.95 / 0.88, all four 0.46: no single cost dominates -- a step is stage -> barrier -> compute -> barrier with
// nothing overlapping inside a workgroup, occupancy (five workgroups per CU) is what hides it.  A channel-group planar tile
// ([group][row][pixel] x 16 bytes, row pitch from a bank model: 4.9 modelled LDS cycles per tap read instead of 7.5) was built,
// bit-exact and 10-16 % SLOWER: its staging is a 16-byte gather at stride C with W of 64 lanes active per instruction.
#ifndef MF_DWMM_KO
#define MF_DWMM_KO 0
#endif
int dw_mm_lds_bytes(const ConvMmArgs &a) {
    return a.G * a.TILE + 256 + a.KS * 16 + 64 + 3 * a.C * 4 + 4 * 256;
}
// KSMAX == p.KS: the k steps (4 taps each) the instance holds in registers -- 1, 2, 3, 4, 7 or 13; the planner pads a filter's k steps up to
// the next of these with zero weights, so the loops over them have no guards
template <int MG, uint32_t XR4, int KSMAX>
__global__ __launch_bounds__(256) void dw_mm_rt(const int8_t *__restrict__ in, int8_t *__restrict__ out, ConvMmArgs p, int batch) {
    constexpr int NTHR = 256, NWAVE = 4;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int C = p.C, KS = p.KS, NBLK = p.NBLK, ROW = p.ROW, TILE = p.TILE, G = p.G, RB = p.RB;
    const int H = p.H, OH = p.OH, OW = p.OW, ROWB = p.W * C, BH = p.BH, NBANDS = p.NBANDS;
    // One staging buffer and nothing else of size in LDS: occupancy is what hides a step's DMA wait and its two barriers (measured:
    // a second buffer with the next step's images in flight, 1.58 against 1.30 ms on 24x24x32 5x5 -- three workgroups per CU became
    // two).  Operand A is read from device memory (L2) at a wave's few group switches, not kept in LDS.
    const int TBUF = G * TILE + 256;
    const int O_OFF = TBUF;                              // [KS][4] tap offsets
    const int C_OFF = O_OFF + KS * 16 + 64;              // A | S | Kc (+ the bit-pattern offset), C entries each
    const int P_OFF = C_OFF + 3 * C * 4;                 // per-wave output patches: 16 pixels x 16 bytes
    for (int i = tid; i < TBUF / 16; i += NTHR) ((uint4 *)lds)[i] = make_uint4(p.izp4, p.izp4, p.izp4, p.izp4);
    for (int i = tid; i < KS * 4; i += NTHR) ((int *)(lds + O_OFF))[i] = p.tap_off[i];
    for (int i = tid; i < C; i += NTHR) {
        ((float *)(lds + C_OFF))[i] = p.A[i], ((float *)(lds + C_OFF))[C + i] = p.S[i];
        ((int *)(lds + C_OFF))[2 * C + i] = p.Kc[i] + (MG != 0 ? MF_MAGIC_I : 0);
    }
    const int col = lane & 15, g = lane >> 4;
    const float inv_ow = 1.0f / (float)OW, inv_bp = 1.0f / (float)(BH * OW);
    const int nsteps = ((batch + G - 1) / G) * NBANDS;
    uint8_t *patch = lds + P_OFF + wave * 256;
    wg_sync();
    int toff[KSMAX]; // this lane group's tap offset in every k step
#pragma unroll
    for (int ks = 0; ks < KSMAX; ++ks) toff[ks] = ((const int *)(lds + O_OFF))[ks * 4 + g];
    auto stage = [&](int st, int buf) {
        const int band = st % NBANDS, ist = st / NBANDS;
        const int yfirst = band * BH * p.sh - p.padt;   // input row held by tile row 0
        for (int gi = 0; gi < G; ++gi) {
            const long img = (long)ist * G + gi;
            if (img >= batch) break;
            for (int r = wave; r < RB; r += NWAVE) {
                const int y = yfirst + r;
                uint8_t *dst = lds + buf * TBUF + gi * TILE + r * ROW + p.LP;
                if (y >= 0 && y < H) {
                    const int8_t *src = in + (img * H + y) * (long)ROWB;
                    for (int o = 0; o < ROWB; o += 1024)
                        if (o + lane * 16 < ROWB) dma16(src + o + lane * 16, dst + o);
                } else if (NBANDS > 1) {                 // band mode: this tile row is padding in this step only
                    for (int o = lane * 16; o < ROWB; o += 1024) *(uint4 *)(dst + o) = make_uint4(p.izp4, p.izp4, p.izp4, p.izp4);
                }
            }
        }
    };
    for (int step = blockIdx.x; step < nsteps; step += gridDim.x) {
        const int band = step % NBANDS, ist = step / NBANDS;
        wg_sync();                                 // the previous step's reads of the tile are done
        if (!(MF_DWMM_KO & 8) || step == (int)blockIdx.x) stage(step, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wg_sync();
        const uint8_t *tile = lds;
        const int gvalid = min(G, batch - ist * G);
        const int rows_here = min(BH, OH - band * BH);
        const int npix = gvalid * BH * OW;               // the step's pixels; a chunk = 16 consecutive ones = 16 C consecutive output bytes
        const int nlive = NBANDS > 1 ? rows_here * OW : npix; // (band mode, G == 1: the rows past the image are the band's last pixels)
        int8_t *obase = out + ((size_t)ist * G * OH + (size_t)band * BH) * OW * C;
        // Work items = (16-channel group, chunk of 16 pixels), group-major; every wave takes one contiguous quarter of them, so that a
        // group's operand A (KS x 16 bytes per lane) and the tap offsets live in REGISTERS across the wave's chunks of that group
        // (from LDS per k step they were half of this kernel's LDS traffic: 14.5 KiB per 256 output bytes).
        const int nchunks = (npix + 15) >> 4, nitems = NBLK * nchunks;
        const int i0 = (int)(((long)nitems * wave) / NWAVE), i1 = (int)(((long)nitems * (wave + 1)) / NWAVE);
        int blk_have = -1;
        v4i wA[KSMAX];
        int4 kc = make_int4(0, 0, 0, 0);
        float4 cA = make_float4(0.f, 0.f, 0.f, 0.f), cS = cA;
        for (int it = i0; it < i1; ++it) {
            const int blk = it / nchunks, chunk = it - blk * nchunks;
            if (blk != blk_have) { // (wave-uniform; at most NBLK / NWAVE + 1 times per step)
                blk_have = blk;
                const int ch = 16 * blk + 4 * g;
#pragma unroll
                for (int ks = 0; ks < KSMAX; ++ks)
                    wA[ks] = ((const v4i *)p.wprep)[((size_t)blk * KSMAX + ks) * 64 + lane];
                kc = *(const int4 *)(lds + C_OFF + (2 * C + ch) * 4);
                cA = *(const float4 *)(lds + C_OFF + ch * 4), cS = *(const float4 *)(lds + C_OFF + (C + ch) * 4);
            }
            const int pp = chunk * 16 + col;
            const int pc = pp < npix ? pp : npix - 1;
            const int gi = (int)(((float)pc + 0.5f) * inv_bp);
            const int rr = pc - gi * BH * OW;
            const int oyl = (int)(((float)rr + 0.5f) * inv_ow), ox = rr - oyl * OW;
            const uint8_t *winb = (MF_DWMM_KO & 1) ? tile + lane * 16 : tile + gi * TILE + (oyl * p.sh) * ROW + p.LP + (ox * p.sw - p.padl) * C + 16 * blk;
            v4i acc = {kc.x, kc.y, kc.z, kc.w};
            // a ring of RING operand-B registers: tap chunk ks + RING is fetched behind the MFMA of chunk ks (all of them at once cost
            // 28 registers at 5x5 -- one wave per SIMD of occupancy)
            constexpr int RING = KSMAX < 4 ? KSMAX : 4;
            v4i b[RING];
#pragma unroll
            for (int ks = 0; ks < RING; ++ks) b[ks] = *(const v4i *)(winb + ((MF_DWMM_KO & 1) ? ks * 1024 : toff[ks]));
#pragma unroll
            for (int ks = 0; ks < KSMAX; ++ks) {
                acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(wA[ks], b[ks % RING], acc, 0, 0, 0);
                if (ks + RING < KSMAX) b[ks % RING] = *(const v4i *)(winb + ((MF_DWMM_KO & 1) ? (ks + RING) * 1024 : toff[ks + RING]));
            }
            const uint32_t d = (MF_DWMM_KO & 4) ? (uint32_t)(acc[0] ^ acc[1] ^ acc[2] ^ acc[3])
                                                : requant_pack4<MG, XR4>(acc[0], acc[1], acc[2], acc[3], cA, cS, p.lo_f, p.hi_f);
            // 16 pixels x this group's 16 bytes through the wave's patch: lane l < 16 stores pixel l's 16 bytes.  (Collecting four chunks
            // so that all 64 lanes store, and non-temporal stores: no faster / slower, profiles/r06/n2_dwmm_store_variants.txt.)
            *(uint32_t *)(patch + col * 16 + 4 * g) = d;
            if (lane < 16 && chunk * 16 + lane < nlive && (!(MF_DWMM_KO & 2) || d == 0x12345678u)) st_out_t<false>(obase + ((size_t)chunk * 16 + lane) * C + 16 * blk, *(const uint4 *)(patch + lane * 16));
        }
    }
}
template <int MG, uint32_t XR4, int KSMAX>
static void launch_dw_mm_k(const int8_t *in, int8_t *out, const ConvMmArgs &a, int batch, hipStream_t s) {
    const int lds = dw_mm_lds_bytes(a);
    int per_cu = 1;
    {
        static std::mutex mu;
        static std::map<std::pair<int, int>, int> cache;
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::lock_guard<std::mutex> lock(mu);
        auto it = cache.find({dev, lds});
        if (it == cache.end()) {
            (void)hipFuncSetAttribute((const void *)dw_mm_rt<MG, XR4, KSMAX>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            int n = 1;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, dw_mm_rt<MG, XR4, KSMAX>, 256, (size_t)lds) != hipSuccess || n < 1) {
                (void)hipGetLastError();
                n = 1;
            }
            it = cache.emplace(std::make_pair(dev, lds), n).first;
        }
        per_cu = it->second;
    }
    const int nsteps = ((batch + a.G - 1) / a.G) * a.NBANDS;
    const int grid = nsteps < 256 * per_cu ? nsteps : 256 * per_cu;
    hipLaunchKernelGGL((dw_mm_rt<MG, XR4, KSMAX>), dim3(grid), dim3(256), lds, s, in, out, a, batch);
}
template <int MG, uint32_t XR4>
static void launch_dw_mm_t(const int8_t *in, int8_t *out, const ConvMmArgs &a, int batch, hipStream_t s) {
    switch (a.KS) { // (conv_mm_plan pads the k steps of its depthwise mode to one of these)
    case 1: launch_dw_mm_k<MG, XR4, 1>(in, out, a, batch, s); break;
    case 2: launch_dw_mm_k<MG, XR4, 2>(in, out, a, batch, s); break;
    case 3: launch_dw_mm_k<MG, XR4, 3>(in, out, a, batch, s); break;
    case 4: launch_dw_mm_k<MG, XR4, 4>(in, out, a, batch, s); break;
    case 7: launch_dw_mm_k<MG, XR4, 7>(in, out, a, batch, s); break;
    default: launch_dw_mm_k<MG, XR4, 13>(in, out, a, batch, s); break;
    }
}
void launch_dw_mm(const int8_t *in, int8_t *out, const ConvMmArgs &a, int batch, hipStream_t s) {
    const int mg = a.magic;
    if (a.xr) {
        if (mg == 2) launch_dw_mm_t<2, 0x80808080u>(in, out, a, batch, s);
        else if (mg) launch_dw_mm_t<1, 0x80808080u>(in, out, a, batch, s);
        else launch_dw_mm_t<0, 0x80808080u>(in, out, a, batch, s);
    } else {
        if (mg == 2) launch_dw_mm_t<2, 0u>(in, out, a, batch, s);
        else if (mg) launch_dw_mm_t<1, 0u>(in, out, a, batch, s);
        else launch_dw_mm_t<0, 0u>(in, out, a, batch, s);
    }
}

// ---- launchers ----
bool dw_rt_plan(DwRtArgs &a, int H, int W, int C, int S, int OH, int OW) {
    if (C % 4 != 0 || C / 4 > 512 || (S != 1 && S != 2) || (W * C) % 16 != 0) return false;
    const int LP = ((C < 16 ? 16 : C) + 15) & ~15;
    const int ROW = LP + W * C + LP;
    constexpr int BUDGET = 36 * 1024;                // per staging buffer: two buffers, two workgroups per CU
    // output rows per task: 3 shares more input-row transposes (4.67 instead of 5 tap instructions per output byte at
    // stride 1) when the rows divide by 3; else 2 with a masked last row
    const int R = (OH % 3 == 0) ? 3 : 2;
    a.R = R;
    auto rows_for = [&](int bh) { return S == 1 ? bh + 2 : 2 * bh + 1; }; // tile rows behind bh output rows (masked rows included)
    const int OHE = (OH + R - 1) / R * R;
    a.H = H, a.W = W, a.C = C, a.OH = OH, a.OW = OW, a.ROW = ROW, a.LP = LP;
    if (rows_for(OHE) * ROW <= BUDGET) {
        a.NBANDS = 1, a.BH = OHE, a.RB = rows_for(OHE), a.TILE = a.RB * ROW;
        // images per step: as many as fit, but prefer a count whose tasks fill whole passes of the 512 threads
        const int gmax = std::min(16, BUDGET / a.TILE), C4 = C / 4, pass = (512 / C4) * C4;
        const int per_img = (OHE / R) * ((OW + 1) / 2) * C4;
        int best = 1;
        double best_eff = 0.0;
        for (int g = 1; g <= gmax; ++g) {
            const int tasks = g * per_img;
            const double eff = (double)tasks / (double)(((tasks + pass - 1) / pass) * pass);
            if (eff >= best_eff - 0.02) best = g, best_eff = std::max(best_eff, eff);
        }
        a.G = best;
    } else {
        int bh = OHE;
        while (bh > R && rows_for(bh) * ROW > BUDGET) bh -= R;
        if (rows_for(bh) * ROW > 64 * 1024) return false; // a single task row does not fit: the row is too wide
        const int nb = (OH + bh - 1) / bh;                  // bands of (nearly) equal height
        bh = ((OH + nb - 1) / nb + R - 1) / R * R;
        a.BH = bh, a.NBANDS = (OH + bh - 1) / bh, a.RB = rows_for(bh), a.TILE = a.RB * ROW, a.G = 1;
    }
    a.BUF = a.G * a.TILE;
    // threads per workgroup: 256 for the large stride-2 layers (measured on person_detect's shapes with the tables off, 512 ->
    // 256 threads: 48x48x16 s2 0.69 -> 0.56 ms, 24x24x32 s2 0.35 -> 0.32; the small tensors lose 15 - 20 %, stride 1 is indifferent)
    {
        const int forced = switches().dw_rt_threads;
        a.NTHR = (S == 2 && H * W * C >= 16384 && C / 4 <= 256) ? 256 : 512;
        if (forced == 256 && C / 4 <= 256) a.NTHR = 256;
        if (forced == 512) a.NTHR = 512;
    }
    return 2 * a.BUF + 256 + 16 <= 160 * 1024;
}
template <int S, int R, bool WZ, int MG, uint32_t XR4>
static void launch_dw_rt_t(const int8_t *in, int8_t *out, const DwRtArgs &a, int batch, hipStream_t s) {
    const int lds = 2 * a.BUF + 256 + 16;
    // occupancy depends on the run-time LDS size: asked once per (instance, size, device)
    int per_cu = 1;
    {
        static std::mutex mu;
        static std::map<std::pair<int, int>, int> cache;
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::lock_guard<std::mutex> lock(mu);
        auto it = cache.find({dev, lds * 2 + (a.NTHR == 256)});
        if (it == cache.end()) {
            (void)hipFuncSetAttribute((const void *)dw3x3_rt<S, R, WZ, MG, XR4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            int n = 1;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, dw3x3_rt<S, R, WZ, MG, XR4>, a.NTHR, (size_t)lds) != hipSuccess || n < 1) {
                (void)hipGetLastError();
                n = 1;
            }
            it = cache.emplace(std::make_pair(dev, lds * 2 + (a.NTHR == 256)), n).first;
        }
        per_cu = it->second;
    }
    const int nsteps = ((batch + a.G - 1) / a.G) * a.NBANDS;
    const int grid = nsteps < 256 * per_cu ? nsteps : 256 * per_cu;
    DwRtArgs b = a;
    const double opix = (double)a.OH * a.OW;
    b.dw.qcfg = dq_config(nsteps, grid, dq_est_us((double)batch * ((double)a.H * a.W * a.C + opix * a.C), (double)batch * opix * a.C));
    b.dw.queue = dq_slot(b.dw.queue, b.dw.qlaunch);
    hipLaunchKernelGGL((dw3x3_rt<S, R, WZ, MG, XR4>), dim3(grid), dim3(a.NTHR), lds, s, in, out, b, batch);
}
void launch_dw_rt(const int8_t *in, int8_t *out, const DwRtArgs &a, int S, bool wz, int batch, hipStream_t s) {
    const int mg = a.dw.magic;
#define MF_RT_GO2(SS, RR, WZZ)                                                                     \
    do {                                                                                           \
        if (a.dw.xr) {                                                                             \
            if (mg == 2) launch_dw_rt_t<SS, RR, WZZ, 2, 0x80808080u>(in, out, a, batch, s);        \
            else if (mg) launch_dw_rt_t<SS, RR, WZZ, 1, 0x80808080u>(in, out, a, batch, s);        \
            else launch_dw_rt_t<SS, RR, WZZ, 0, 0x80808080u>(in, out, a, batch, s);                \
        } else {                                                                                   \
            if (mg == 2) launch_dw_rt_t<SS, RR, WZZ, 2, 0u>(in, out, a, batch, s);                 \
            else if (mg) launch_dw_rt_t<SS, RR, WZZ, 1, 0u>(in, out, a, batch, s);                 \
            else launch_dw_rt_t<SS, RR, WZZ, 0, 0u>(in, out, a, batch, s);                         \
        }                                                                                          \
    } while (0)
#define MF_RT_GO(SS, WZZ) do { if (a.R == 3) MF_RT_GO2(SS, 3, WZZ); else MF_RT_GO2(SS, 2, WZZ); } while (0)
    if (S == 1) { if (wz) MF_RT_GO(1, true); else MF_RT_GO(1, false); }
    else { if (wz) MF_RT_GO(2, true); else MF_RT_GO(2, false); }
#undef MF_RT_GO2
#undef MF_RT_GO
}

int pw_rt_lds_bytes(int K, int N, bool wz) {
    const int KS = (K + 63) / 64, NT = (N + 15) / 16;
    return NT * KS * 1024 + (wz ? KS * 1024 : 0) + NT * 16 * 16 + 4 * 16 * ((N + 15) & ~15);
}
bool pw_rt_supported(int K, int N, bool wz) {
    return K % 16 == 0 && K >= 16 && K <= 512 && N % 4 == 0 && N >= 4 && pw_rt_lds_bytes(K, N, wz) <= 96 * 1024;
}
template <bool WZ, int MG, uint32_t XR4>
static void launch_pw_rt_t(const int8_t *in, int8_t *out, const PwRtArgs &a, long long npix, hipStream_t s) {
    const int lds = pw_rt_lds_bytes(a.K, a.N, WZ);
    // occupancy per (device, LDS size in KiB): asked of the runtime once, not per launch (and never during a graph capture)
    static LaunchState st[97];
    const int per_cu = prepared(st[(lds + 1023) / 1024], pw_rt_lds<WZ, MG, XR4>, 256, lds);
    const long long nchunks = (npix + 15) / 16;
    const long long want = (nchunks + 3) / 4;
    const int cap = 256 * per_cu;                    // persistent: the weight image is copied to LDS once per workgroup
    const int grid = (int)(want < cap ? want : cap);
    hipLaunchKernelGGL((pw_rt_lds<WZ, MG, XR4>), dim3(grid), dim3(256), lds, s, in, out, a, npix);
}
template <int KSC, int MG, uint32_t XR4>
static void launch_pw_rt_reg_t(const int8_t *in, int8_t *out, const PwRtArgs &a, long long nrows, hipStream_t s) {
    static LaunchState st;
    const int per_cu = prepared(st, pw_rt<KSC, MG, XR4>, 256, 0);
    const int U = KSC <= 2 ? 4 : 2, SLOTS = 4 / a.NSPLIT;
    const long long nchunks = (nrows + 15) / 16, want = (nchunks + (long long)SLOTS * U - 1) / ((long long)SLOTS * U);
    // persistent: a workgroup fetches its operand A once; a few waves of workgroups per CU keep the loads deep
    const long long cap = 256LL * per_cu * 2;
    const int grid = (int)(want < cap ? (want < 1 ? 1 : want) : cap);
    hipLaunchKernelGGL((pw_rt<KSC, MG, XR4>), dim3(grid), dim3(256), 0, s, in, out, a, nrows);
}
void launch_pw_rt(const int8_t *in, int8_t *out, const PwRtArgs &a, bool wz, long long npix, hipStream_t s) {
    const int mg = a.magic;
    if (!wz && a.TB > 0) { // weights in registers
#define MF_RT_GO(KSC)                                                                              \
    do {                                                                                           \
        if (a.xr) {                                                                                \
            if (mg == 2) launch_pw_rt_reg_t<KSC, 2, 0x80808080u>(in, out, a, npix, s);             \
            else if (mg) launch_pw_rt_reg_t<KSC, 1, 0x80808080u>(in, out, a, npix, s);             \
            else launch_pw_rt_reg_t<KSC, 0, 0x80808080u>(in, out, a, npix, s);                     \
        } else {                                                                                   \
            if (mg == 2) launch_pw_rt_reg_t<KSC, 2, 0u>(in, out, a, npix, s);                      \
            else if (mg) launch_pw_rt_reg_t<KSC, 1, 0u>(in, out, a, npix, s);                      \
            else launch_pw_rt_reg_t<KSC, 0, 0u>(in, out, a, npix, s);                              \
        }                                                                                          \
    } while (0)
        if (a.KS <= 1) MF_RT_GO(1);
        else if (a.KS <= 2) MF_RT_GO(2);
        else if (a.KS <= 4) MF_RT_GO(4);
        else MF_RT_GO(8);
#undef MF_RT_GO
        return;
    }
#define MF_RT_GO(WZZ)                                                                              \
    do {                                                                                           \
        if (a.xr) {                                                                                \
            if (mg == 2) launch_pw_rt_t<WZZ, 2, 0x80808080u>(in, out, a, npix, s);                 \
            else if (mg) launch_pw_rt_t<WZZ, 1, 0x80808080u>(in, out, a, npix, s);                 \
            else launch_pw_rt_t<WZZ, 0, 0x80808080u>(in, out, a, npix, s);                         \
        } else {                                                                                   \
            if (mg == 2) launch_pw_rt_t<WZZ, 2, 0u>(in, out, a, npix, s);                          \
            else if (mg) launch_pw_rt_t<WZZ, 1, 0u>(in, out, a, npix, s);                          \
            else launch_pw_rt_t<WZZ, 0, 0u>(in, out, a, npix, s);                                  \
        }                                                                                          \
    } while (0)
    if (wz) MF_RT_GO(true); else MF_RT_GO(false);
#undef MF_RT_GO
}

// ------------------------------------------------------------------------
// dw3x3_stem_rt -- DepthwiseConv2D 3x3 stride 2 SAME with ONE input channel and DM = 4 or 8 outputs at any H x W (W % 16 == 0):
// a MobileNet stem at any input resolution and width (src/ops/depthwise_conv_2d.rs:28-105 with Cin = 1: every output channel reads
// input channel 0).  The run-time-geometry form of dw3x3_stem8_mm (k_depthwise.hip): the 9 taps are ONE MFMA per 256 output bytes.
//   column  : one 16-byte group of the NHWC output = 16 / DM adjacent pixels x DM channels (group j of output row oy); its windows
//             span input columns XS j - 1 .. XS j + XS - 1 of rows 2 oy - 1 .. 2 oy + 1, XS = 32 / DM
//   K bytes : lane group g = filter row; the lane supplies the aligned bytes at columns XS j - 4 .. of tile row 2 oy + g (8 bytes
//             and v_mfma_i32_16x16x32_i8 for DM = 8, 16 bytes and 16x16x64 for DM = 4); pixel p of the group uses bytes
//             3 + 2 p .. 5 + 2 p, where the host put the weights in operand A (group 3: zeros)
//   rows    : (pixel of the group, channel): a lane ends with 4 consecutive channels = one packed dword; four tiles are
//             transposed across the lane groups, so that a lane stores 16 bytes and 16 lanes 256 contiguous ones
//   step    : G whole images, copied verbatim by LDS-DMA between izp rows (row pitch W: column -1 is patched with a select,
//             no other padding exists for even W), double buffered, dynamic step queue
// ------------------------------------------------------------------------
template <int DM, int MG, uint32_t XR4>
__global__ __launch_bounds__(256) void dw3x3_stem_rt(const int8_t *__restrict__ in, int8_t *__restrict__ out, DwStemRtArgs p, int batch) {
    constexpr int GUARD = 16, XS = 32 / DM;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, g = lane >> 4;
    const int W = p.W, G = p.G, TILE = p.TILE, BUF = G * TILE, IMG = p.H * W, QR = p.QR, QTOT = p.QTOT, NQUAD = p.NQUAD;
    for (int i = tid; i < (2 * BUF + 64) / 16; i += 256) ((uint4 *)lds)[i] = make_uint4(p.izp4, p.izp4, p.izp4, p.izp4);
    const long Aw8 = (long)(((unsigned long)p.wmm[lane][1] << 32) | (unsigned long)p.wmm[lane][0]);
    const v4i Aw16 = {(int)p.wmm[lane][0], (int)p.wmm[lane][1], (int)p.wmm[lane][2], (int)p.wmm[lane][3]};
    const int cq = DM == 8 ? (g & 1) * 4 : 0; // this lane's channels within its pixel
    const float4 cA = make_float4(p.A[cq], p.A[cq + 1], p.A[cq + 2], p.A[cq + 3]);
    const float4 cS = make_float4(p.S[cq], p.S[cq + 1], p.S[cq + 2], p.S[cq + 3]);
    const v4i cK = {p.Kc[cq] + (MG ? MF_MAGIC_I : 0), p.Kc[cq + 1] + (MG ? MF_MAGIC_I : 0), p.Kc[cq + 2] + (MG ? MF_MAGIC_I : 0),
                    p.Kc[cq + 3] + (MG ? MF_MAGIC_I : 0)};
    DynSteps dq;
    dq.init(lds + 2 * BUF + 64, p.queue, tid, p.qcfg);
    wg_sync();

    const int NI = (IMG + 1023) >> 10; // 1 KiB DMA instructions per image (the last one partly masked)
    auto stage = [&](int st, int buf) {
        for (int gg = 0; gg < G; ++gg) {
            if (st * G + gg >= batch) break;
            const int8_t *src = in + (size_t)(st * G + gg) * IMG;
            uint8_t *dst = lds + buf * BUF + gg * TILE + GUARD + W;
            for (int c = wave; c < NI; c += 4)
                if (c * 1024 + lane * 16 < IMG) dma16(src + c * 1024 + lane * 16, dst + c * 1024);
        }
    };
    const int nsteps = (batch + G - 1) / G;
    int cur = 0;
    if (dq.step < nsteps) stage(dq.step, 0);
    const int lane_row = GUARD + g * W - 4; // + XS q + oy W = tile row 2 oy + g, column XS j - 4
    for (; dq.step < nsteps; dq.advance(tid), cur ^= 1) {
        const int step = dq.step;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wg_sync();
        dq.top(tid);
        if (dq.nxt < nsteps) stage(dq.nxt, cur ^ 1);
        const int gv = min(G, batch - step * G);
        for (int gg = 0; gg < gv; ++gg) {
            const uint8_t *tile = lds + cur * BUF + gg * TILE + lane_row;
            uint4 *dst = (uint4 *)out + (size_t)(step * G + gg) * QTOT;
#pragma unroll 1
            for (int k = (wave + gg) & 3; k < NQUAD; k += 4) {
                uint32_t r[4];
#pragma unroll
                for (int n = 0; n < 4; n += 2) {
                    v4i acc[2];
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        const int q = min((4 * k + n + m) * 16 + col, QTOT - 1); // (a ragged last quad re-reads the last group)
                        const int oy = (int)(((float)q + 0.5f) * p.inv_qr);
                        const bool first = q == oy * QR;                   // j == 0: column -1 is padding
                        const uint32_t *src = (const uint32_t *)(tile + XS * q + oy * W);
                        uint32_t d0 = src[0];
                        d0 = first ? p.izp4 : d0;
                        if constexpr (DM == 8) {
                            const long B = (long)(((unsigned long)src[1] << 32) | (unsigned long)d0);
                            acc[m] = __builtin_amdgcn_mfma_i32_16x16x32_i8(Aw8, B, cK, 0, 0, 0);
                        } else {
                            const v4i B = {(int)d0, (int)src[1], (int)src[2], (int)src[3]};
                            acc[m] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Aw16, B, cK, 0, 0, 0);
                        }
                    }
                    requant_pack4x2<MG, XR4>(acc[0], cA, cS, acc[1], cA, cS, p.lo_f, p.hi_f, r[n], r[n + 1]);
                }
                lane_group_transpose4(r[0], r[1], r[2], r[3]); // lane (col, g): the 16 bytes of group col of tile 4 k + g
                const int qs = (4 * k + g) * 16 + col;
                if (qs < QTOT) st_out(dst + qs, make_uint4(r[0], r[1], r[2], r[3]));
            }
        }
    }
    dq.finish(tid);
}

int conv_rows_lds_bytes(const ConvRowsArgs &a) {
    return a.G * a.TILE + a.KH * a.KG * a.NP * 4 + ((a.KG * 4 + 15) & ~15) + a.NP * 16 + 64;
}
// fills the geometry of a; false: the shape is not for this kernel
bool conv_rows_plan(ConvRowsArgs &a, int H, int W, int C, int N, int KH, int KW, int sh, int sw, int OH, int OW, bool pad_same) {
    const int RWB = KW * C;
    if (RWB > 64 || KH > 16 || N < 1 || N > 64 || (W * C) % 4 != 0 || C >= 16) return false; // (16 and more channels: conv_mm_rt)
    a.H = H, a.W = W, a.C = C, a.N = N, a.KH = KH, a.KW = KW, a.sh = sh, a.sw = sw, a.OH = OH, a.OW = OW;
    a.ROWB = W * C, a.KG = (RWB + 3) / 4, a.NP = (N + 7) & ~7;
    const int padl = pad_same ? (KW - 1) / 2 : 0, padt = pad_same ? (KH - 1) / 2 : 0;
    a.shy = padt;
    a.XO = (padl * C + 3) & ~3;                       // image column 0 at a dword boundary
    a.X0 = a.XO - padl * C;                           // window of output column 0 starts here
    const int need = a.X0 + ((OW - 1) * sw) * C + 4 * (a.KG + 1); // last window's last aligned dword
    a.TWP = (std::max(need, a.XO + W * C) + 3 + 4) & ~3;
    const int TH = std::max((OH - 1) * sh + KH, padt + H);
    a.TILE = (TH * a.TWP + 4 + 15) & ~15;
    const int img_dwords = H * (W * C / 4);
    int g = std::min(8, (512 * 16) / std::max(img_dwords, 1));   // 16 prefetched dwords per thread
    while (g > 1 && g * a.TILE > 64 * 1024) --g;
    if (g < 1 || a.TILE > 64 * 1024) return false;
    a.G = g;
    return conv_rows_lds_bytes(a) <= 96 * 1024;
}
template <bool WZ, int MG, uint32_t XR4>
static void launch_conv_rows_t(const int8_t *in, int8_t *out, const ConvRowsArgs &a, int batch, hipStream_t s) {
    const int lds = conv_rows_lds_bytes(a);
    int per_cu = 1;
    {
        static std::mutex mu;
        static std::map<std::pair<int, int>, int> cache;
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::lock_guard<std::mutex> lock(mu);
        auto it = cache.find({dev, lds});
        if (it == cache.end()) {
            (void)hipFuncSetAttribute((const void *)conv_rows_lds<WZ, MG, XR4>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            int n = 1;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv_rows_lds<WZ, MG, XR4>, 512, (size_t)lds) != hipSuccess || n < 1) {
                (void)hipGetLastError();
                n = 1;
            }
            it = cache.emplace(std::make_pair(dev, lds), n).first;
        }
        per_cu = it->second;
    }
    const int nsteps = (batch + a.G - 1) / a.G;
    const int grid = nsteps < 256 * per_cu ? nsteps : 256 * per_cu;
    hipLaunchKernelGGL((conv_rows_lds<WZ, MG, XR4>), dim3(grid), dim3(512), lds, s, in, out, a, batch);
}
bool dw_stem_rt_plan(DwStemRtArgs &a, int H, int W, int DM, int OH, int OW) {
    if ((DM != 4 && DM != 8) || W % 16 != 0 || W < 16 || H < 2 || OH != (H + 1) / 2 || OW != W / 2) return false;
    a.H = H, a.W = W, a.OH = OH, a.OW = OW, a.DM = DM;
    a.QR = OW * DM / 16, a.QTOT = OH * a.QR, a.NQUAD = (a.QTOT + 63) / 64;
    if (a.QTOT + 64 >= (1 << 22)) return false;       // (the float index arithmetic)
    a.inv_qr = 1.0f / (float)a.QR;
    a.TILE = (16 + (H + 3) * W + 15) & ~15;            // guard, izp row, H rows, izp rows (lane group 3 reads row 2 oy + 3 against zero weights)
    if (2 * a.TILE + 80 > 150 * 1024) return false;
    int g = 1;
    while (g < 8 && 2 * (2 * g) * a.TILE <= 40 * 1024) g *= 2;
    a.G = g;
    return true;
}
template <int DM, int MG, uint32_t XR4>
static void launch_dw_stem_rt_t(const int8_t *in, int8_t *out, const DwStemRtArgs &a_in, int batch, hipStream_t s) {
    DwStemRtArgs a = a_in;
    const int lds = 2 * a.G * a.TILE + 64 + 16;
    static LaunchState st[161];
    const int per_cu = prepared(st[(lds + 1023) / 1024], dw3x3_stem_rt<DM, MG, XR4>, 256, lds);
    const int nsteps = (batch + a.G - 1) / a.G;
    const int grid = nsteps < 256 * per_cu ? nsteps : 256 * per_cu;
    const double out_bytes = (double)batch * a.OH * a.OW * a.DM;
    a.qcfg = dq_config(nsteps, grid, dq_est_us((double)batch * a.H * a.W + out_bytes, out_bytes));
    a.queue = dq_slot(a.queue, a.qlaunch);
    hipLaunchKernelGGL((dw3x3_stem_rt<DM, MG, XR4>), dim3(grid), dim3(256), lds, s, in, out, a, batch);
}
void launch_dw_stem_rt(const int8_t *in, int8_t *out, const DwStemRtArgs &a, int batch, hipStream_t s) {
    const int mg = a.magic;
#define MF_RT_GO(DMM)                                                                              \
    do {                                                                                           \
        if (a.xr) {                                                                                \
            if (mg == 2) launch_dw_stem_rt_t<DMM, 2, 0x80808080u>(in, out, a, batch, s);           \
            else if (mg) launch_dw_stem_rt_t<DMM, 1, 0x80808080u>(in, out, a, batch, s);           \
            else launch_dw_stem_rt_t<DMM, 0, 0x80808080u>(in, out, a, batch, s);                   \
        } else {                                                                                   \
            if (mg == 2) launch_dw_stem_rt_t<DMM, 2, 0u>(in, out, a, batch, s);                    \
            else if (mg) launch_dw_stem_rt_t<DMM, 1, 0u>(in, out, a, batch, s);                    \
            else launch_dw_stem_rt_t<DMM, 0, 0u>(in, out, a, batch, s);                            \
        }                                                                                          \
    } while (0)
    if (a.DM == 8) MF_RT_GO(8); else MF_RT_GO(4);
#undef MF_RT_GO
}
void launch_conv_rows(const int8_t *in, int8_t *out, const ConvRowsArgs &a, bool wz, int batch, hipStream_t s) {
    const int mg = a.magic;
#define MF_RT_GO(WZZ)                                                                              \
    do {                                                                                           \
        if (a.xr) {                                                                                \
            if (mg == 2) launch_conv_rows_t<WZZ, 2, 0x80808080u>(in, out, a, batch, s);            \
            else if (mg) launch_conv_rows_t<WZZ, 1, 0x80808080u>(in, out, a, batch, s);            \
            else launch_conv_rows_t<WZZ, 0, 0x80808080u>(in, out, a, batch, s);                    \
        } else {                                                                                   \
            if (mg == 2) launch_conv_rows_t<WZZ, 2, 0u>(in, out, a, batch, s);                     \
            else if (mg) launch_conv_rows_t<WZZ, 1, 0u>(in, out, a, batch, s);                     \
            else launch_conv_rows_t<WZZ, 0, 0u>(in, out, a, batch, s);                             \
        }                                                                                          \
    } while (0)
    if (wz) MF_RT_GO(true); else MF_RT_GO(false);
#undef MF_RT_GO
}

int conv_mm_lds_bytes(const ConvMmArgs &a, bool wz) {
    return a.G * a.TILE + 256 + a.NBLK * a.TB * a.KS * 1024 + (wz ? a.KS * 1024 : 0) + a.KS * 16 + 64;
}
// fills the geometry and the tap-offset table (host copy in `tap`); false: not for this kernel
bool conv_mm_plan(ConvMmArgs &a, std::vector<int> &tap, int H, int W, int C, int N, int KH, int KW, int sh, int sw, int OH, int OW,
                  bool pad_same, bool wz, bool dwise) {
    if (C % 16 != 0 || N % 4 != 0 || KH > 7 || KW > 7 || (W * C) % 16 != 0) return false;
    if (dwise && (N != C || wz)) return false;           // one output per input channel; filter zero points take the generic kernel
    const int Ktot = dwise ? KH * KW * 16 : KH * KW * C; // depthwise: 16 k-bytes per tap and block
    int KS = (Ktot + 63) / 64;
    if (dwise) KS = KS <= 4 ? KS : (KS <= 7 ? 7 : 13);  // (dw_mm_rt's instances; the steps beyond the filter meet zero weights)
    const int NT = (N + 15) / 16;
    const int TB = dwise ? 1 : (NT < 4 ? NT : 4), NBLK = (NT + TB - 1) / TB;
    a.dwise = dwise ? 1 : 0;
    const int wbytes = NBLK * TB * KS * 1024 + (wz ? KS * 1024 : 0);
    if (wbytes > 96 * 1024) return false;
    const int padl = pad_same ? (KW - 1) / 2 : 0, padt = pad_same ? (KH - 1) / 2 : 0;
    const int LP = (padl * C + 15) & ~15;
    int over = ((OW - 1) * sw - padl + KW - W) * C;      // bytes read right of the image row
    if (over < 0) over = 0;
    const int RP = ((over + 15) & ~15) + 16;
    const int ROW = LP + W * C + RP;
    a.H = H, a.W = W, a.C = C, a.N = N, a.KH = KH, a.KW = KW, a.sh = sh, a.sw = sw, a.OH = OH, a.OW = OW;
    a.padl = padl, a.padt = padt, a.LP = LP, a.ROW = ROW, a.KS = KS, a.TB = TB, a.NBLK = NBLK;
    const int budget = 150 * 1024 - wbytes - 1024;
    auto rows_for = [&](int bh) { return (bh - 1) * sh + KH; };
    a.NTHR = 256;
    if (rows_for(OH) * ROW <= budget && rows_for(OH) * ROW <= 48 * 1024) {
        a.NBANDS = 1, a.BH = OH, a.RB = rows_for(OH), a.TILE = a.RB * ROW;
        int g = std::min(48 * 1024, budget) / a.TILE;
        a.G = g < 1 ? 1 : (g > 16 ? 16 : g);
        // Does a second workgroup fit beside this one?  If not, one 16-wave workgroup with a step as large as the LDS allows
        // (whole chunks per wave: 16 waves x 16 pixels).
        const bool small_wg = switches().conv_mm_256; // A/B: round 3's four-wave workgroups
        if (!small_wg && 2 * (a.G * a.TILE + wbytes + 1024) > 160 * 1024) {
            int gg = budget / a.TILE;
            gg = gg > 32 ? 32 : gg;
            const int opix = OH * OW;
            while (gg > 1 && (gg * opix) % 256 != 0 && ((gg - 1) * opix + 255) / 256 == (gg * opix + 255) / 256) --gg; // no smaller step with as many rounds
            if (gg >= 1) a.G = gg, a.NTHR = 1024;
        }
    } else {
        int bh = OH;
        const int cap = std::min(budget, 48 * 1024);
        while (bh > 1 && rows_for(bh) * ROW > cap) --bh;
        if (rows_for(bh) * ROW > cap) return false;
        const int nb = (OH + bh - 1) / bh;
        bh = (OH + nb - 1) / nb;
        a.BH = bh, a.NBANDS = (OH + bh - 1) / bh, a.RB = rows_for(bh), a.TILE = a.RB * ROW, a.G = 1;
    }
    tap.assign((size_t)KS * 4, 0);
    for (int ks = 0; ks < KS; ++ks)
        for (int g = 0; g < 4; ++g) {
            const int kk0 = ks * 64 + g * 16;
            if (kk0 >= Ktot) continue;                   // beyond K: zero weights, offset 0
            const int CK = dwise ? 16 : C;               // k-bytes per tap
            const int t = kk0 / CK, c0 = kk0 % CK, ky = t / KW, kx = t % KW;
            tap[(size_t)ks * 4 + g] = ky * ROW + kx * C + c0;
        }
    if (dwise) {
        a.NTHR = 256; // (dw_mm_rt: four-wave workgroups; a step of at most what leaves room for a second workgroup)
        while (a.G > 1 && 2 * dw_mm_lds_bytes(a) > 160 * 1024) --a.G;
        return dw_mm_lds_bytes(a) <= 160 * 1024;
    }
    return conv_mm_lds_bytes(a, wz) <= 160 * 1024;
}
template <bool WZ, int MG, uint32_t XR4, int NTHR>
static void launch_conv_mm_n(const int8_t *in, int8_t *out, const ConvMmArgs &a, int batch, hipStream_t s) {
    const int lds = conv_mm_lds_bytes(a, WZ);
    int per_cu = 1;
    {
        static std::mutex mu;
        static std::map<std::pair<int, int>, int> cache;
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::lock_guard<std::mutex> lock(mu);
        auto it = cache.find({dev, lds});
        if (it == cache.end()) {
            (void)hipFuncSetAttribute((const void *)conv_mm_rt<WZ, MG, XR4, NTHR>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            int n = 1;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv_mm_rt<WZ, MG, XR4, NTHR>, NTHR, (size_t)lds) != hipSuccess || n < 1) {
                (void)hipGetLastError();
                n = 1;
            }
            it = cache.emplace(std::make_pair(dev, lds), n).first;
        }
        per_cu = it->second;
    }
    const int nsteps = ((batch + a.G - 1) / a.G) * a.NBANDS;
    const int grid = nsteps < 256 * per_cu ? nsteps : 256 * per_cu;
    hipLaunchKernelGGL((conv_mm_rt<WZ, MG, XR4, NTHR>), dim3(grid), dim3(NTHR), lds, s, in, out, a, batch);
}
template <bool WZ, int MG, uint32_t XR4>
static void launch_conv_mm_t(const int8_t *in, int8_t *out, const ConvMmArgs &a, int batch, hipStream_t s) {
    if (a.NTHR == 1024) launch_conv_mm_n<WZ, MG, XR4, 1024>(in, out, a, batch, s);
    else launch_conv_mm_n<WZ, MG, XR4, 256>(in, out, a, batch, s);
}
void launch_conv_mm(const int8_t *in, int8_t *out, const ConvMmArgs &a, bool wz, int batch, hipStream_t s) {
    const int mg = a.magic;
#define MF_RT_GO(WZZ)                                                                              \
    do {                                                                                           \
        if (a.xr) {                                                                                \
            if (mg == 2) launch_conv_mm_t<WZZ, 2, 0x80808080u>(in, out, a, batch, s);              \
            else if (mg) launch_conv_mm_t<WZZ, 1, 0x80808080u>(in, out, a, batch, s);              \
            else launch_conv_mm_t<WZZ, 0, 0x80808080u>(in, out, a, batch, s);                      \
        } else {                                                                                   \
            if (mg == 2) launch_conv_mm_t<WZZ, 2, 0u>(in, out, a, batch, s);                       \
            else if (mg) launch_conv_mm_t<WZZ, 1, 0u>(in, out, a, batch, s);                       \
            else launch_conv_mm_t<WZZ, 0, 0u>(in, out, a, batch, s);                               \
        }                                                                                          \
    } while (0)
    if (wz) MF_RT_GO(true); else MF_RT_GO(false);
#undef MF_RT_GO
}

} // namespace k
} // namespace mf
