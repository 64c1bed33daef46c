// k_depthwise.hip -- DepthwiseConv2D fast paths (src/ops/depthwise_conv_2d.rs:28-105).
//
// dw3x3_nhwc (3x3 SAME, any stride-1/2 NHWC shape in MF_DW_SHAPES), dw3x3_stem8 (one input channel -> 8,
// the network stem, optionally fused with the f32 boundary quantisation) and dw_c1_lds (one input
// channel, any filter: speech.tflite).
// Arithmetic contract, shared device helpers and launch plumbing: k_common.hpp.
#include "k_common.hpp"
#include "k_dwtask.hpp"

namespace mf {
namespace k {

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
template <int H, int W, int C, int S, int G, int NTHR, int MG, uint32_t XR4>
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
    DynSteps dq;
    dq.init(lds + 2 * BUF + 256, p.queue, tid, p.qcfg); // (the launcher allocates 16 bytes behind the read slack)
    wg_sync(); // halo fill complete before any DMA lands

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
    int cur = 0;
    if (dq.step < nsteps) stage(dq.step, 0);

    for (; dq.step < nsteps; dq.advance(tid), cur ^= 1) {
        const int step = dq.step;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's DMAs (and old stores) done
        wg_sync();                                  // ... and everyone else's
        dq.top(tid);
        const int next = dq.nxt;
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
                    constexpr bool NT = !(H == 12 && S == 1); // (k_common.hpp st_out_t; the 12x12x64 stride-1 layer loses 6 % with it)
                    int o0[R][4], o1[R][4];
                    const uint8_t *base = tile + g * TILE + (oy0 * S) * ROW + LP + (ox0 * S - 1) * C + cg * 4;
                    if constexpr (S == 2) dw_s2_task<R, ROW, C>(base, wA, Kc, o0, o1);
                    else dw_s1_task<R, ROW, C>(base, wA, wB, Kc, o0, o1);
#pragma unroll
                    for (int j = 0; j < R; ++j) {
                        uint32_t *dp = dst + ((size_t)(g * OH + oy0 + j) * OW + ox0) * C4 + cg;
                        st_out_t<NT>(dp, requant_pack4<MG, XR4>(o0[j][0], o0[j][1], o0[j][2], o0[j][3], A, Sc, p.lo_f, p.hi_f));
                        if (ox0 + 1 < OW)
                            st_out_t<NT>(dp + C4, requant_pack4<MG, XR4>(o1[j][0], o1[j][1], o1[j][2], o1[j][3], A, Sc, p.lo_f, p.hi_f));
                    }
                }
            }
        }
    }
    dq.finish(tid);
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

template <int H, int W, int G, int MG, uint32_t XR4, bool F32IN>
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
    DynSteps dq;
    dq.init(lds + 2 * BUF, p.queue, tid, p.qcfg); // (16 bytes behind the two staging buffers)
    wg_sync();

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
                const float t = __fadd_rn(quant_div(pre[k][e], p.in_scale, p.in_rcp, p.in_fast != 0), p.in_zp_f);
                const float r = __fadd_rn(t, __builtin_copysignf(0x1.fffffep-2f, t));
                q[e] = (r != r) ? 0 : (int)__builtin_amdgcn_fmed3f(r, p.in_sat_lo, p.in_sat_hi);
            }
            *(uint32_t *)(lds + buf * BUF + g * TILE + GUARD + W + c * 4) = pack4(q[0], q[1], q[2], q[3]) ^ p.in_xr4;
        }
    };

    const int nsteps = (batch + G - 1) / G;
    int cur = 0;
    if constexpr (F32IN) {
        if (dq.step < nsteps) {
            load_f32(dq.step);
            store_f32(0);
        }
    } else {
        if (dq.step < nsteps) stage(dq.step, 0);
    }

    for (; dq.step < nsteps; dq.advance(tid), cur ^= 1) {
        const int step = dq.step;
        if constexpr (!F32IN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wg_sync();
        dq.top(tid);
        const int next = dq.nxt;
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
                    int a = p.Kc[c] + (MG != 0 ? MF_MAGIC_I : 0), b = a;
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
                st_out(dst + o, v);
            }
        }
        if constexpr (F32IN) {
            if (next < nsteps) store_f32(cur ^ 1); // the other buffer: last read before this step's barrier
        }
    }
    dq.finish(tid);
}

// ------------------------------------------------------------------------
// FAST PATH 2a' -- the same stem with its 9 taps on the matrix pipe (dw3x3_stem8_mm; int8 input).
// dw3x3_stem8 spends 6 v_dot4 (half rate) per 8 output bytes before the requantisation; here the taps of
// 16 pixel pairs x (2 pixels x 8 channels) are ONE v_mfma_i32_16x16x32_i8:
//   column j  : pixel pair (2j, 2j+1) of an output row; its 3x3 windows span input columns 4j-1 .. 4j+3
//   K bytes   : lane group g = filter row ky (g = 3: zero weights); the lane's 8 bytes are the ALIGNED dwords at
//               columns 4j-4 and 4j of tile row 2oy + ky (one ds_read2_b32): pixel 2j uses bytes 3..5, pixel 2j+1
//               bytes 5..7 -- the weights sit at those byte positions of operand A (DwStemArgs::wmm, host-built)
//   rows      : (pixel of the pair, channel) -> a lane ends with 4 consecutive channels of one pixel = one packed
//               dword; the 64 dwords of a tile are 256 contiguous output bytes
//   tiles     : the 24 pairs x 2 rows of an output row pair are exactly 3 tiles; a wave walks row pairs, all lane
//               offsets are three per-type constants
// Same tile, staging and (reference) padding as dw3x3_stem8; what is left on the VALU is the requantisation.
// ------------------------------------------------------------------------
template <int H, int W, int G, int MG, uint32_t XR4, bool F32IN>
__global__ __launch_bounds__(256) void dw3x3_stem8_mm(const int8_t *__restrict__ in, int8_t *__restrict__ out, DwStemArgs p,
                                                      int batch) {
    static_assert(!(F32IN && MG == 3), "the boundary quantisation of the f32 entry needs round-to-nearest: no single-fma epilogue there");
    epi_enter<MG>();
    constexpr int S = 2;
    constexpr int OH = (H + S - 1) / S, OW = (W + S - 1) / S;
    constexpr int GUARD = 16;
    constexpr int TILE = GUARD + (H + 2) * W;     // [guard][izp row][H rows][izp row]
    constexpr int BUF = G * TILE;
    constexpr int IMG = H * W;
    constexpr int NI = IMG / 1024;                // 1 KiB DMA instructions per image
    constexpr int PAIRS = OW / 2;                 // pixel pairs per output row
    constexpr int RP = OH / 2;                    // output row pairs per image (3 tiles each)
    constexpr int WPI = 4 / G;                    // waves per image
    constexpr int RPW = RP / WPI;                 // row pairs per wave
    static_assert(IMG % 1024 == 0 && W % 16 == 0 && PAIRS == 24 && OH % 2 == 0 && TILE % 16 == 0, "stem geometry");
    static_assert((G == 1 || G == 2 || G == 4) && RP % WPI == 0, "wave split");

    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, g = lane >> 4;
    for (int i = tid; i < 2 * BUF / 16; i += 256)
        ((uint4 *)lds)[i] = make_uint4(p.izp4, p.izp4, p.izp4, p.izp4);
    const long Aw = (long)(((unsigned long)p.wmm[lane][1] << 32) | (unsigned long)p.wmm[lane][0]);
    const int cq = (g & 1) * 4; // this lane's channels within its pixel
    const float4 cA = make_float4(p.A[cq], p.A[cq + 1], p.A[cq + 2], p.A[cq + 3]);
    const float4 cS = make_float4(p.S[cq], p.S[cq + 1], p.S[cq + 2], p.S[cq + 3]);
    const v4i cK = {p.Kc[cq] + (MG ? MF_MAGIC_I : 0), p.Kc[cq + 1] + (MG ? MF_MAGIC_I : 0), p.Kc[cq + 2] + (MG ? MF_MAGIC_I : 0),
                    p.Kc[cq + 3] + (MG ? MF_MAGIC_I : 0)};
    // the three tile types of a row pair (rows r0, r0 + 1; 24 pairs each): pairs 0..15 of r0 | 16..23 of r0 and
    // 0..7 of r0 + 1 | 8..23 of r0 + 1.  Lane byte offset of its operand inside the image tile, relative to the
    // row pair: tile row 2 oy + ky (tile row 0 is the izp row = input row -1), column 4j - 4
    int off[3];
    bool first[3]; // j == 0: column -1 is padding
    {
        const int oy1 = col >= 8 ? 1 : 0, j1 = col >= 8 ? col - 8 : 16 + col;
        off[0] = GUARD + g * W + 4 * col - 4;
        off[1] = GUARD + (2 * oy1 + g) * W + 4 * j1 - 4;
        off[2] = GUARD + (2 + g) * W + 4 * (8 + col) - 4;
        first[0] = col == 0, first[1] = col == 8, first[2] = false;
    }
    const int img_g = wave / WPI, rp0 = (wave % WPI) * RPW;
    DynSteps dq;
    dq.init(lds + 2 * BUF, p.queue, tid, p.qcfg); // (16 bytes behind the two staging buffers)
    wg_sync();

    auto stage = [&](int st, int buf) {
#pragma unroll
        for (int k = 0; k < (G * NI + 3) / 4; ++k) {
            const int r = k * 4 + wave;           // wave-uniform 1 KiB piece of the step
            const int gg = r / NI, c = r % NI;
            if (r < G * NI && st * G + gg < batch)
                dma16(in + ((size_t)(st * G + gg) * IMG + c * 1024 + lane * 16),
                      lds + buf * BUF + gg * TILE + GUARD + W + c * 1024);
        }
    };

    // f32 staging (F32IN; same arithmetic and order as dw3x3_stem8's): float4 k of this thread is pixels
    // 4 (k * 256 + tid) .. + 3 of the step's G images, loaded before this step's compute, quantised and written after it
    constexpr int NF = F32IN ? G * IMG / 4 / 256 : 1;
    static_assert(!F32IN || (G * IMG) % 1024 == 0, "f32 staging geometry");
    f32x4 pre[NF];
    auto load_f32 = [&](int st) {
        const f32x4 *src = (const f32x4 *)in;
#pragma unroll
        for (int k = 0; k < NF; ++k) {
            const int idx = k * 256 + tid, gg = idx / (IMG / 4);
            const size_t img = (size_t)st * G + gg;
            const size_t at = (img < (size_t)batch ? img : (size_t)batch - 1) * (IMG / 4) + idx % (IMG / 4);
            pre[k] = src[at]; // clamped, unconditional: a ragged last step re-reads the last image
        }
    };
    auto store_f32 = [&](int buf) {
#pragma unroll
        for (int k = 0; k < NF; ++k) {
            const int idx = k * 256 + tid, gg = idx / (IMG / 4), c = idx % (IMG / 4);
            int qv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t = __fadd_rn(quant_div(pre[k][e], p.in_scale, p.in_rcp, p.in_fast != 0), p.in_zp_f);
                const float r = __fadd_rn(t, __builtin_copysignf(0x1.fffffep-2f, t));
                qv[e] = (r != r) ? 0 : (int)__builtin_amdgcn_fmed3f(r, p.in_sat_lo, p.in_sat_hi);
            }
            *(uint32_t *)(lds + buf * BUF + gg * TILE + GUARD + W + c * 4) = pack4(qv[0], qv[1], qv[2], qv[3]) ^ p.in_xr4;
        }
    };

    const int nsteps = (batch + G - 1) / G;
    int cur = 0;
    if constexpr (F32IN) {
        if (dq.step < nsteps) {
            load_f32(dq.step);
            store_f32(0);
        }
    } else {
        if (dq.step < nsteps) stage(dq.step, 0);
    }
    for (; dq.step < nsteps; dq.advance(tid), cur ^= 1) {
        const int step = dq.step;
        if constexpr (!F32IN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wg_sync();
        dq.top(tid);
        const int next = dq.nxt;
        if constexpr (F32IN) {
            if (next < nsteps) load_f32(next); // in flight during the compute below
        } else {
            if (next < nsteps) stage(next, cur ^ 1);
        }
        if (step * G + img_g < batch) { // (wave-uniform)

        const uint8_t *tile = lds + cur * BUF + img_g * TILE + (2 * rp0) * S * W; // row pair rp: output rows 2rp, 2rp+1
        uint4 *dst = (uint4 *)out + ((size_t)(step * G + img_g) * OH * PAIRS + (size_t)(2 * rp0) * PAIRS) + g * 16 + col;
        // per row pair: 3 tiles = 48 pixel pairs = 768 output bytes.  Four row pairs = 12 tiles per iteration; the
        // four dwords of a pixel pair sit in the four lane groups of ONE tile, so tiles are taken in groups of four
        // and transposed across the lane groups (2 x 2 v_permlane swaps): lane (col, g) then holds all 16 bytes of
        // pair `col` of the group's tile g -> one 16-byte store per lane, 256 contiguous bytes per 16 lanes
        static_assert(RPW % 4 == 0, "row pairs per wave in fours");
#pragma unroll 1
        for (int it = 0; it < RPW / 4; ++it) {
            const uint8_t *t0 = tile + it * (16 * W); // 4 row pairs = 8 output rows = 16 input rows further down
            long B[12];
#pragma unroll
            for (int n = 0; n < 12; ++n) {
                const uint32_t *q = (const uint32_t *)(t0 + (n / 3) * (4 * W) + off[n % 3]);
                uint32_t d0 = q[0];
                const uint32_t d1 = q[1];
                d0 = first[n % 3] ? p.izp4 : d0;
                B[n] = (long)(((unsigned long)d1 << 32) | (unsigned long)d0);
            }
            uint32_t q[12];
#pragma unroll
            for (int n = 0; n < 12; n += 2) {
                const v4i a0 = __builtin_amdgcn_mfma_i32_16x16x32_i8(Aw, B[n], cK, 0, 0, 0);
                const v4i a1 = __builtin_amdgcn_mfma_i32_16x16x32_i8(Aw, B[n + 1], cK, 0, 0, 0);
                requant_pack4x2<MG, XR4>(a0, cA, cS, a1, cA, cS, p.lo_f, p.hi_f, q[n], q[n + 1]);
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                uint32_t r0 = q[4 * k], r1 = q[4 * k + 1], r2 = q[4 * k + 2], r3 = q[4 * k + 3];
                lane_group_transpose4(r0, r1, r2, r3);
                st_out(dst + (size_t)it * 192 + (4 * k) * 16, make_uint4(r0, r1, r2, r3)); // tile 4k + g: + g * 16 is in dst
            }
        }
        }
        if constexpr (F32IN) {
            if (next < nsteps) store_f32(cur ^ 1); // the other buffer: last read before this step's barrier
        }
    }
    dq.finish(tid);
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
template <int MG, uint32_t XR4>
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
        wg_sync(); // previous image fully consumed
#pragma unroll
        for (int e = 0; e < MAXE; ++e)
            if (tid + 512 * e < TH * TWP) ((int8_t *)lds)[tid + 512 * e] = gofs[e] < 0 ? (int8_t)p.izp : v[e];
        for (int i = tid + 512 * MAXE; i < TH * TWP; i += 512) { // tiles above 4 KiB: the plain way
            const int ty = i / TWP, tx = i - ty * TWP;
            const int iy = ty - shy, ix = tx - shx;
            // columns TW .. TWP-1 only ever meet zero weights; any finite value will do
            ((int8_t *)lds)[i] = (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) ? x[iy * p.W + ix] : (int8_t)p.izp;
        }
        wg_sync();
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

// ---- launchers ----
template <int H, int W, int C, int S, int G, int NTHR, int MG, uint32_t XR4>
static void launch_dw(const int8_t *in, int8_t *out, const DwFastArgs &a, int batch, hipStream_t s) {
    constexpr int LP = C < 16 ? 16 : C;
    constexpr int lds = 2 * G * (H + 2) * (LP + W * C + LP) + 256 + 16; // two staging buffers + read slack + step queue
    static LaunchState st;
    const int per_cu = prepared(st, dw3x3_nhwc<H, W, C, S, G, NTHR, MG, XR4>, NTHR, lds);
    const int nsteps = (batch + G - 1) / G;
    const int grid = nsteps < 256 * per_cu ? nsteps : 256 * per_cu;
    constexpr int OPIX = ((H + S - 1) / S) * ((W + S - 1) / S);
    DwFastArgs b = a;
    b.qcfg = dq_config(nsteps, grid, dq_est_us((double)batch * (H * W * C + OPIX * C), (double)batch * OPIX * C));
    b.queue = dq_slot(b.queue, b.qlaunch);
    hipLaunchKernelGGL((dw3x3_nhwc<H, W, C, S, G, NTHR, MG, XR4>), dim3(grid), dim3(NTHR), lds, s, in, out, b, batch);
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
    const int alt = switches().dw_alt;
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
bool launch_dw_stem(int H, int W, int DM, int S, const int8_t *in, int8_t *out, const DwStemArgs &a_in,
                    int batch, hipStream_t s, bool f32_input) {
    if (H == 96 && W == 96 && DM == 8 && S == 2) {
        constexpr int G = 2, lds = 2 * G * (16 + (96 + 2) * 96) + 16; // + step queue
        DwStemArgs a = a_in;
        a.qcfg = dq_config((batch + G - 1) / G, 512, dq_est_us((double)batch * (96 * 96 * (f32_input ? 4 : 1) + 48 * 48 * 8), (double)batch * 48 * 48 * 8));
        a.queue = dq_slot(a.queue, a.qlaunch);
        static LaunchState st, stf;
        const int per_cu = f32_input ? prepared(stf, dw3x3_stem8<96, 96, G, false, 0u, true>, 256, lds)
                                     : prepared(st, dw3x3_stem8<96, 96, G, false, 0u, false>, 256, lds);
        const int nsteps = (batch + G - 1) / G;
        const int grid = nsteps < 256 * per_cu ? nsteps : 256 * per_cu;
#define MF_STEM(MG, XR, F) hipLaunchKernelGGL((dw3x3_stem8<96, 96, G, MG, XR, F>), dim3(grid), dim3(256), lds, s, in, out, a, batch)
#define MF_STEM2(F)                                                                          \
    if (a.xr) { if (a.magic) MF_STEM(true, 0x80808080u, F); else MF_STEM(false, 0x80808080u, F); } \
    else { if (a.magic) MF_STEM(true, 0u, F); else MF_STEM(false, 0u, F); }
        const bool valu = switches().stem_valu;
        if (a.magic == 3 && valu) return false; // (the single-fma epilogue exists in the matrix-pipe form only; the caller does not ask for it then)
        if (!valu) { // taps on the matrix pipe
            static LaunchState stm, stmf;
            const int pcu = f32_input ? prepared(stmf, dw3x3_stem8_mm<96, 96, G, false, 0u, true>, 256, lds)
                                      : prepared(stm, dw3x3_stem8_mm<96, 96, G, false, 0u, false>, 256, lds);
            const int gridm = nsteps < 256 * pcu ? nsteps : 256 * pcu;
#define MF_STEMM(MG, XR, F) hipLaunchKernelGGL((dw3x3_stem8_mm<96, 96, G, MG, XR, F>), dim3(gridm), dim3(256), lds, s, in, out, a, batch)
#define MF_STEMM2(F)                                                                               \
    if (a.xr) { if (a.magic == 2) MF_STEMM(2, 0x80808080u, F); else if (a.magic) MF_STEMM(1, 0x80808080u, F); else MF_STEMM(0, 0x80808080u, F); } \
    else { if (a.magic == 2) MF_STEMM(2, 0u, F); else if (a.magic) MF_STEMM(1, 0u, F); else MF_STEMM(0, 0u, F); }
            if (a.magic == 3) { // (the caller asks for the single-fma form only with int8 input: ops.hip)
                if (f32_input) return false;
                MF_STEMM(3, 0u, false);
            } else if (f32_input) { MF_STEMM2(true) } else { MF_STEMM2(false) }
#undef MF_STEMM2
#undef MF_STEMM
            return true;
        }
        if (f32_input) { MF_STEM2(true) } else { MF_STEM2(false) }
#undef MF_STEM2
#undef MF_STEM
        return true;
    }
    return false;
}

} // namespace k
} // namespace mf
