// k_tail3.hip -- the end of a MobileNet-v1 style network in ONE launch (person_detect ops 25..30):
//   DepthwiseConv2D 3x3 stride 1 on 3x3x256 -> Conv2D 1x1 256 -> 256 -> AveragePool2D over all 9 pixels ->
//   Conv2D 1x1 256 -> N (head) -> [Reshape] -> Softmax
// (src/ops/depthwise_conv_2d.rs:28-105, src/ops/conv_2d.rs:28-108, src/ops/average_pool_2d.rs:29-66,
//  src/ops/softmax.rs:15-27)
//
// Arithmetic contract, shared device helpers and launch plumbing: k_common.hpp.
//
// A 3x3 tensor has 9 pixels: tiles of 16 PIXELS (dwpw_mm) leave MFMA columns idle and a step of 8 images is too
// little work between two barriers.  Here the 16 columns of every MFMA are 16 IMAGES (as in k_dwfc.hip): every
// column does the same thing at the same pixel, all addresses are (lane constant) + (compile-time offset), no
// column is ever idle, and a workgroup step is 16 images.
//   waves    : 16; wave w owns 16-channel group w of the depthwise conv and output tile w (16 channels) of the
//              pointwise conv, with both operand sets (3 + 4 MFMA A registers) and epilogue constants in registers.
//   depthwise: unit = (pixel, channel group): 3 MFMAs (taps on the matrix pipe, k_fused_mm.hip); operand B of lane
//              (image, kx) is the 16 channels of input pixel (y + ky - 1, x + kx - 1) -- one ds_read_b128; a row
//              or a column outside the tensor is read from a 17th, all-zero-point image slot (row: compile-time,
//              column: one per-lane select per x, hoisted).  Result -> requantise -> MID
//              [image][pixel][256] in LDS.
//   pointwise: per pixel 4 K-steps of one MFMA each, operand B straight from MID; the requantised outputs are not
//              stored at all: AveragePool2D over the whole 3x3 tensor is the sum over the 9 units a lane runs, kept
//              in registers (the element values -- i8 or u8 -- exactly as the reference's pool input).
//   tail     : pool epilogue, the lane's share of the head dot products (4 channels x N), lane-group and wave
//              reduction through LDS, 16 threads finish head epilogue + table softmax for their image.
// Round 4: C = 256 or 128 channels (16 or 8 waves); the tensor is H x W = 3x3, 2x2 or 4x4 (a 96-, 64- or 128-pixel input; compile-time), the epilogue form MG 0 (v_cvt: 256-deep
// products whose accumulators may leave (-2^22, 2^22)) or 1; 4x4 keeps ONE staged image set (135 KB of LDS) and refills it behind the
// depthwise phase's barrier.
// HBM traffic: 2304 B in, N bytes out per inference.  Image pitch 2320 B: the 16 lanes of a b128 service group hit
// 16 distinct 16-byte bank slots (2320 / 4 = 4 mod 64 words).
#include "k_common.hpp"
#include "k_tail.hpp"
#ifndef MF_TAIL3_RAW_BARRIER
#define MF_TAIL3_RAW_BARRIER 0 // (tuning)
#endif

#ifndef MF_TAIL3_DIAG
#define MF_TAIL3_DIAG 0 // 1: cycle stamps of block 0 / wave 0 at the phase boundaries of its first two steps (never shipped)
#endif

namespace mf {
namespace k {

#if MF_TAIL3_DIAG
__device__ long long g_tail3_trace[32];
#define MF_TR(k) do { if (blockIdx.x == 0 && wave == (MF_TAIL3_DIAG - 1) && lane == 0 && trace_step < 2) g_tail3_trace[(k) + 8 * trace_step] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define MF_TR(k) do { } while (0)
#endif

// FRONT (round 6, person_detect ops 23..30 in one launch): the pair in front of this pair runs in the same step -- DepthwiseConv2D 3x3
// stride 2 on 2H x 2W x C/2 + Conv2D 1x1 C/2 -> C -- with the same 16-images-per-MFMA-column scheme: the staged tensor is the FRONT
// pair's input X6 [16][2H 2W C/2] (one set, refilled behind the front depthwise's barrier), its depthwise output goes to the (still
// idle) MID buffer, its 1x1 output -- this pair's input -- to X3, which is then never staged from HBM.  Wave w owns channel group
// w % (C/32) of the front depthwise for every second pixel and output tile w of the front 1x1 (operands resident: 3 + C/128
// registers x 4; the epilogue constants of the two front operators are read from an LDS copy per phase).  Two more barriers per step.
template <int H, int W, int C, int N, int NTHR, bool DBUF, int MG, uint32_t XR4, bool FRONT = false>
__global__ __launch_bounds__(NTHR) void pair3_tail(const int8_t *__restrict__ in, int8_t *__restrict__ out, PairTailArgs p, PairFrontArgs fr,
                                                   size_t batch) {
    constexpr int IMGS = 16, PIX = H * W, IMG = PIX * C, KS = C / 64;
    constexpr int NPIECE = (IMG + 1023) / 1024;            // 1 KiB DMA pieces per image (the last one may be short)
    constexpr int XP = IMG + 16;                           // image pitch in LDS (X3 and MID)
    constexpr int NW = NTHR / 64;
    static_assert((C == 128 || C == 256) && C / 16 == NW, "one channel group and one output tile per wave");
    static_assert(!FRONT || !DBUF, "the front pair writes X3 itself: one set");
    // the front pair's tensor: H1 x W1 x C1, 16 images in X6
    constexpr int H1 = 2 * H, W1 = 2 * W, C1 = C / 2, IMG1 = H1 * W1 * C1, XP1 = IMG1 + 16, NPIECE1 = (IMG1 + 1023) / 1024;
    constexpr int KS1 = C1 / 64, NQ1 = C1 / 16, WPG = NW / NQ1; // k steps of the front 1x1; channel groups and waves per group of the front depthwise
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    constexpr int SET = 16 * XP;                           // one set of 16 staged images
    uint8_t *x6 = lds;                                     // FRONT: [16][XP1]
    uint8_t *x3 = lds + (FRONT ? 16 * XP1 : 0);            // [1 or 2 sets][16][XP] + one all-zero-point image slot
    uint8_t *zslot = x3 + (DBUF ? 2 : 1) * SET;
    constexpr int ZS = FRONT ? 256 : XP;                   // (only its first 16 NW bytes are ever read)
    uint8_t *mid = zslot + ZS;                             // [16][XP]
    int *part = (int *)(mid + 16 * XP);                    // [2][16 images][4]: head sums + value sum, added up by LDS atomics
    float *expt = (float *)(part + 2 * 16 * 4);            // softmax's 256-entry table
    int *hci = (int *)(expt + 256);                        // head constants [N][4]: wzp, Kc, A, S (kept out of registers)
    uint8_t *zslot1 = (uint8_t *)(hci + 4 * N);            // FRONT: 256 bytes of the front depthwise's input zero point
    uint8_t *cst = zslot1 + 256;                           // FRONT: A | S | Kc of the front depthwise (C1 each), the front 1x1, this pair's depthwise and 1x1 (C each)
    constexpr int CS_FPW = 3 * C1, CS_DW = 3 * C1 + 3 * C, CS_PW = 3 * C1 + 6 * C; // (in 4-byte words)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, g = lane >> 4;
#if MF_TAIL3_DIAG
    int trace_step = 0;
    if (blockIdx.x == 0 && tid == 0) g_tail3_trace[31] = (long long)__builtin_readcyclecounter();
#endif

    for (int i = tid; i < ZS / 16; i += NTHR) ((uint4 *)zslot)[i] = make_uint4(p.izp4, p.izp4, p.izp4, p.izp4);
    if (tid < 2 * 16 * 4) part[tid] = 0;
    for (int i = tid; i < 256; i += NTHR) expt[i] = p.tail.exp_table[i];
    if (tid < N) {
        hci[4 * tid] = p.tail.wzp[tid], hci[4 * tid + 1] = p.tail.Kc[tid];
        hci[4 * tid + 2] = __float_as_int(p.tail.A[tid]), hci[4 * tid + 3] = __float_as_int(p.tail.S[tid]);
    }
    if constexpr (FRONT) {
        if (tid < 16) ((uint4 *)zslot1)[tid] = make_uint4(fr.izp4, fr.izp4, fr.izp4, fr.izp4);
        for (int i = tid; i < C1; i += NTHR) {
            ((float *)cst)[i] = fr.dwA[i], ((float *)cst)[C1 + i] = fr.dwS[i], ((int *)cst)[2 * C1 + i] = fr.dwK[i];
        }
        for (int i = tid; i < C; i += NTHR) {
            ((float *)cst)[CS_FPW + i] = fr.pwA[i], ((float *)cst)[CS_FPW + C + i] = fr.pwS[i], ((int *)cst)[CS_FPW + 2 * C + i] = fr.pwK[i];
            ((float *)cst)[CS_DW + i] = p.dwA[i], ((float *)cst)[CS_DW + C + i] = p.dwS[i], ((int *)cst)[CS_DW + 2 * C + i] = p.dwK[i];
            ((float *)cst)[CS_PW + i] = p.pwA[i], ((float *)cst)[CS_PW + C + i] = p.pwS[i], ((int *)cst)[CS_PW + 2 * C + i] = p.pwK[i];
        }
    }
    // operands and constants of this wave's channel group / output tile (channels 16 wave + 4 g .. + 3 for this lane)
    v4i Adw[3], Apw[KS];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) Adw[ky] = ((const v4i *)p.dw_wmm)[(wave * 3 + ky) * 64 + lane];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) Apw[ks] = ((const v4i *)p.pw_w)[(wave * KS + ks) * 64 + lane];
    // FRONT: channel group q1 of the front depthwise for the pixels px % WPG == h1, output tile `wave` of the front 1x1
    const int q1 = wave % NQ1, h1 = wave / NQ1;
    // The front pair's operands are NOT resident (the instance spilled with them: 128 registers at 16 waves per CU): they are
    // fetched per step from L2 -- the depthwise's at the end of the step before (they return with the explicit vmcnt(0) there), the
    // 1x1's at the top of the front depthwise phase, consumed behind its barrier and IN FRONT of the staging DMAs' issue (vmcnt
    // retires in order: a compiler-placed wait for them behind the DMAs would wait for the HBM round trip).  The pointers are
    // laundered so that the loads are not hoisted out of the step loop again.
    v4i Adw1[3], Apw1[KS1];
    // (the OFFSET is laundered, not the pointer: a pointer that went through an asm statement is a generic one, and flat loads count
    // on lgkmcnt too -- every LDS wait of the phase would then wait for them)
    auto fetch_dw1 = [&]() {
        uint32_t off = (uint32_t)((q1 * 3) * 64 + lane) * 16u;
        asm volatile("" : "+v"(off));
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) Adw1[ky] = *(const v4i *)((const uint8_t *)fr.dw_wmm + off + ky * 1024);
    };
    auto fetch_pw1 = [&]() {
        uint32_t off = (uint32_t)((wave * KS1) * 64 + lane) * 16u;
        asm volatile("" : "+v"(off));
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) Apw1[ks] = *(const v4i *)((const uint8_t *)fr.pw_w + off + ks * 1024);
    };
    if constexpr (FRONT) fetch_dw1();
    const int ch = 16 * wave + 4 * g;
    // epilogue constants of this pair: resident -- or, with the front pair's operands in the register file too, read from the LDS
    // copy at the top of each phase (the FRONT instance spilled otherwise, and a spill reload is a vector-memory load that waits
    // for the staging DMAs in flight: vmcnt retires in order)
    float4 dA, dS, pA, pS;
    v4i dK, pK;
    auto consts_at = [&](int base, int n, int c0, float4 &A, float4 &S, v4i &K) { // block at word `base`, n channels, channels c0 .. c0 + 3
        A = *(const float4 *)(cst + (base + c0) * 4), S = *(const float4 *)(cst + (base + n + c0) * 4);
        const int4 k4 = *(const int4 *)(cst + (base + 2 * n + c0) * 4);
        K = v4i{k4.x, k4.y, k4.z, k4.w};
    };
    if constexpr (!FRONT) {
        dA = *(const float4 *)(p.dwA + ch), dS = *(const float4 *)(p.dwS + ch);
        const int4 dK4 = *(const int4 *)(p.dwK + ch);
        dK = v4i{dK4.x, dK4.y, dK4.z, dK4.w};
        pA = *(const float4 *)(p.pwA + ch), pS = *(const float4 *)(p.pwS + ch);
        const int4 pK4 = *(const int4 *)(p.pwK + ch);
        pK = v4i{pK4.x, pK4.y, pK4.z, pK4.w};
    }
    uint32_t hw[N];
#pragma unroll
    for (int n = 0; n < N; ++n) hw[n] = *(const uint32_t *)(p.tail.w + (size_t)n * C + ch);
    // lane part of the depthwise operand address for output column x: input column x + g - 1, or the zero-point slot
    int laneoff[W];
#pragma unroll
    for (int x = 0; x < W; ++x) {
        const int cx = x + g - 1;
        laneoff[x] = (cx >= 0 && cx <= W - 1) ? col * XP + cx * C : -1; // (-1: the zero-point slot, see below)
    }

    // FRONT: input column 2 x + g - 1 of output column x (stride 2; the halo column is on the left only), or the zero-point slot
    int flo[W];
#pragma unroll
    for (int x = 0; x < W; ++x) {
        const int cx = 2 * x + g - 1;
        flo[x] = (cx >= 0 && cx <= W1 - 1) ? col * XP1 + cx * C1 : -1;
    }
    const size_t nblk = (batch + IMGS - 1) / IMGS;
    // staging by LDS-DMA, no registers: an image is 2304 B = two 1 KiB pieces + one of 256 B (16 lanes); 48 pieces per
    // step, 3 per wave.  A ragged last step re-reads the last image.
    auto stage = [&](size_t blk, int set) {
#pragma unroll
        for (int k = 0; k < (IMGS * NPIECE + NW - 1) / NW; ++k) { // 16 x NPIECE pieces per step, dealt over the NW waves
            const int j = wave + NW * k, img = j / NPIECE, piece = j - img * NPIECE;
            size_t image = blk * IMGS + img;
            image = image < batch ? image : batch - 1;
            if (j < IMGS * NPIECE && piece * 1024 + lane * 16 < IMG)
                dma16(in + image * IMG + piece * 1024 + lane * 16, x3 + set * SET + img * XP + piece * 1024);
        }
    };
    auto stage6 = [&](size_t blk) { // FRONT: 16 images of IMG1 bytes into X6
#pragma unroll
        for (int k = 0; k < (IMGS * NPIECE1 + NW - 1) / NW; ++k) {
            const int j = wave + NW * k, img = j / NPIECE1, piece = j - img * NPIECE1;
            size_t image = blk * IMGS + img;
            image = image < batch ? image : batch - 1;
            if (j < IMGS * NPIECE1 && piece * 1024 + lane * 16 < IMG1) // (scalar source base: image and piece are wave-uniform)
                dma16_sb(in + image * IMG1 + piece * 1024, (uint32_t)lane * 16u, x6 + img * XP1 + piece * 1024);
        }
    };
    if (blockIdx.x >= nblk) return;
    if constexpr (FRONT) stage6(blockIdx.x);
    else stage(blockIdx.x, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wg_sync(); // zero-point slot, tables, zeroed sums and the first images are in place
    int cur = 0;
    for (size_t blk = blockIdx.x; blk < nblk; blk += gridDim.x, cur ^= 1) {
        // the other image set was last read before the previous step's barriers: refill it, a whole step ahead
        if constexpr (DBUF) {
            if (blk + gridDim.x < nblk) stage(blk + gridDim.x, cur ^ 1);
        }
        if constexpr (FRONT) {
            fetch_pw1(); // (in flight under the front depthwise)
            // ---- front depthwise 3x3 stride 2: X6 -> MID (as [image][pixel][C1]) ----
            {
                const int ch1 = 16 * q1 + 4 * g;
                float4 fA, fS;
                v4i fK;
                consts_at(0, C1, ch1, fA, fS, fK);
#pragma unroll
                for (int px = 0; px < PIX; ++px) {
                    if (px % WPG != h1) continue; // (wave-uniform)
                    const int y = px / W, x = px % W;
                    v4i acc = fK;
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
                        const int iy = 2 * y + ky - 1; // (never below the tensor: 2 (H - 1) + 1 = H1 - 1)
                        const uint8_t *src = (flo[x] >= 0 && iy >= 0) ? x6 + flo[x] + iy * (W1 * C1) : zslot1;
                        const v4i B = *(const v4i *)(src + 16 * q1);
                        acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(Adw1[ky], B, acc, 0, 0, 0);
                    }
                    *(uint32_t *)(mid + col * XP + px * C1 + ch1) = requant_pack4<MG, XR4>(acc[0], acc[1], acc[2], acc[3], fA, fS, fr.dw_lo, fr.dw_hi);
                }
            }
            wg_sync(); // the front depthwise's tensor is complete; X6 has been read
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) asm volatile("" : "+v"(Apw1[ks])); // the compiler's wait for the 1x1's operands goes HERE
            if (blk + gridDim.x < nblk) stage6(blk + gridDim.x); // lands under the four phases that follow
            // ---- front 1x1 C1 -> C: MID -> X3 (this pair's input, [image][pixel][C]) ----
            {
                float4 qA, qS;
                v4i qK;
                consts_at(CS_FPW, C, ch, qA, qS, qK);
#pragma unroll
                for (int px = 0; px < PIX; ++px) {
                    v4i acc = qK;
#pragma unroll
                    for (int ks = 0; ks < KS1; ++ks) {
                        const v4i B = *(const v4i *)(mid + col * XP + px * C1 + 64 * ks + 16 * g);
                        acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(Apw1[ks], B, acc, 0, 0, 0);
                    }
                    *(uint32_t *)(x3 + col * XP + px * C + ch) = requant_pack4<MG, XR4>(acc[0], acc[1], acc[2], acc[3], qA, qS, fr.pw_lo, fr.pw_hi);
                }
            }
            wg_sync(); // X3 complete; MID is free for this pair's depthwise
        }
        const uint8_t *xs = x3 + (DBUF ? cur : 0) * SET;
        MF_TR(0);
        // ---- depthwise 3x3: channel group `wave`, all 9 pixels ----
        if constexpr (FRONT) consts_at(CS_DW, C, ch, dA, dS, dK);
#pragma unroll
        for (int px = 0; px < PIX; ++px) {
            const int y = px / W, x = px % W;
            v4i acc = dK;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int iy = y + ky - 1;
                // a row outside the tensor: the zero-point slot too (compile-time choice)
                const uint8_t *src = (laneoff[x] >= 0 && iy >= 0 && iy <= H - 1) ? xs + laneoff[x] + iy * (W * C) : zslot;
                const v4i B = *(const v4i *)(src + 16 * wave);
                acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(Adw[ky], B, acc, 0, 0, 0);
            }
            *(uint32_t *)(mid + col * XP + px * C + ch) =
                requant_pack4<MG, XR4>(acc[0], acc[1], acc[2], acc[3], dA, dS, p.dw_lo, p.dw_hi);
        }
        MF_TR(1);
#if MF_TAIL3_RAW_BARRIER
        // (the bare instruction: __syncthreads() carries a fence that hipcc completes with vmcnt(0) while an LDS-DMA is outstanding --
        // the next step's images, staged a whole step ahead, would have to land by here)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); // MID complete
#else
        wg_sync(); // MID complete
#endif
        if constexpr (!DBUF && !FRONT) { // one image set: it has been read, refill it under the rest of the step
            if (blk + gridDim.x < nblk) stage(blk + gridDim.x, 0);
        }
        MF_TR(2);
        MF_TR(3);
        // ---- pointwise 256 -> 256: output tile `wave`; AveragePool2D = the sum over the 9 pixels ----
        if constexpr (FRONT) consts_at(CS_PW, C, ch, pA, pS, pK);
        int pool[4] = {0, 0, 0, 0};
#pragma unroll
        for (int px = 0; px < PIX; ++px) {
            v4i acc = pK;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const v4i B = *(const v4i *)(mid + col * XP + px * C + 64 * ks + 16 * g);
                acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(Apw[ks], B, acc, 0, 0, 0);
            }
            pool[0] += (int)requant_clamped<MG>(acc[0], pA.x, pS.x, p.pw_lo, p.pw_hi);
            pool[1] += (int)requant_clamped<MG>(acc[1], pA.y, pS.y, p.pw_lo, p.pw_hi);
            pool[2] += (int)requant_clamped<MG>(acc[2], pA.z, pS.z, p.pw_lo, p.pw_hi);
            pool[3] += (int)requant_clamped<MG>(acc[3], pA.w, pS.w, p.pw_lo, p.pw_hi);
        }
        MF_TR(4);
        // ---- pool epilogue (average_pool_2d.rs:52-57) and this lane's share of the head dot products ----
        int q[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float xf = __fmul_rn(p.tail.inv_len, (float)pool[k]);                     // (1/len) * f32(sum)
            const float yv = __fadd_rn(__fmul_rn(p.tail.pool_c0, xf), p.tail.pool_c1);      // c0 * x + c1
            const float r = __fadd_rn(yv, __builtin_copysignf(0x1.fffffep-2f, yv));
            int vq = (r != r) ? 0 : (int)__builtin_amdgcn_fmed3f(r, p.tail.pool_sat_lo, p.tail.pool_sat_hi);
            vq = max(vq, p.tail.pool_lo);
            q[k] = min(vq, p.tail.pool_hi) ^ p.tail.xr; // the stored byte (pack4 keeps the low byte)
        }
        const uint32_t qp = pack4(q[0], q[1], q[2], q[3]);
        int dot[N], vs = sdot4(qp, 0x01010101u, 0);
#pragma unroll
        for (int n = 0; n < N; ++n) dot[n] = sdot4(qp, hw[n], 0);
        // lane groups of one image, then the waves
        vs += __shfl_xor(vs, 16, 64), vs += __shfl_xor(vs, 32, 64);
#pragma unroll
        for (int n = 0; n < N; ++n) dot[n] += __shfl_xor(dot[n], 16, 64), dot[n] += __shfl_xor(dot[n], 32, 64);
        if (g == 0) { // 16 waves add into the step's 16 x (N + 1) sums
            int *dst = part + (cur * 16 + col) * 4;
#pragma unroll
            for (int n = 0; n < N; ++n) atomicAdd(dst + n, dot[n]);
            atomicAdd(dst + N, vs);
        }
        MF_TR(5);
        if constexpr (FRONT) fetch_dw1(); // the next step's front depthwise operands
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's pieces of the next step have landed
        if constexpr (FRONT) {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) asm volatile("" : "+v"(Adw1[ky]));
        }
        wg_sync(); // sums complete; MID consumed; the next images are in place for everyone
        MF_TR(6);
        if (tid < 16 * N) { // head epilogue + softmax: thread (image, n) = (tid / N, tid % N)  (conv_2d.rs:93-98, softmax.rs:20-27)
            const int img = tid / N, n = tid - img * N;
            int *src = part + (cur * 16 + img) * 4;
            const int acc = src[n] - hci[4 * n] * src[N] + hci[4 * n + 1];
            const int h = requant(acc, __int_as_float(hci[4 * n + 2]), __int_as_float(hci[4 * n + 3]), p.tail.lo_f, p.tail.hi_f);
            const float e = expt[(int)(int8_t)(h ^ p.tail.xr) + 128]; // table index = stored byte + 128
            float sum = 0.0f;
#pragma unroll
            for (int j = 0; j < N; ++j) sum = __fadd_rn(sum, __shfl(e, img * N + j, 64)); // one row: index order
            const float prob = __fdiv_rn(e, sum);
            const float qf = __fadd_rn(__fdiv_rn(prob, p.tail.sm_oscale), p.tail.sm_ozp_f);
            const float r = __fadd_rn(qf, __builtin_copysignf(0x1.fffffep-2f, qf));
            const size_t image = blk * IMGS + img;
            if (image < batch)
                out[image * N + n] = (int8_t)(((r != r) ? 0 : (int)__builtin_amdgcn_fmed3f(r, p.tail.sm_sat_lo, p.tail.sm_sat_hi)) ^ p.tail.xr);
            // (the sums of this buffer are next added to two steps from now, after two more barriers)
            __builtin_amdgcn_wave_barrier();
            src[n] = 0;
            if (n == 0) src[N] = 0;
        }
        MF_TR(7);
#if MF_TAIL3_DIAG
        ++trace_step;
#endif
    }
}

bool pair_tail_supported(int H, int W, int C, int N_pw, int N_head, int ntaps) {
    return H == W && (H == 2 || H == 3 || H == 4) && (C == 256 || C == 128) && N_pw == C && N_head == 2 && ntaps == H * W;
}
const char *pair_tail_name(int H, int C) {
    static const char *names[2][3] = {{"pair3_tail<2,2,128,2>", "pair3_tail<3,3,128,2>", "pair3_tail<4,4,128,2>"},
                                      {"pair3_tail<2,2,256,2>", "pair3_tail<3,3,256,2>", "pair3_tail<4,4,256,2>"}};
    return names[C == 256][H - 2];
}
template <int H, int C, bool DBUF, int MG, uint32_t XR4>
static int launch_pair_tail_t(const int8_t *in, int8_t *out, const PairTailArgs &a, size_t batch, hipStream_t s) {
    constexpr int NTHR = C * 4, XP = H * H * C + 16; // one wave per 16 channels
    constexpr int lds = ((DBUF ? 2 : 1) * 16 + 1 + 16) * XP + 2 * 16 * 4 * 4 + 256 * 4 + 2 * 4 * 4;
    const size_t nblk = (batch + 15) / 16;
    static LaunchState st;
    const int per_cu = prepared(st, pair3_tail<H, H, C, 2, NTHR, DBUF, MG, XR4>, NTHR, lds);
    const size_t cap = (size_t)256 * per_cu;
    hipLaunchKernelGGL((pair3_tail<H, H, C, 2, NTHR, DBUF, MG, XR4>), dim3((unsigned)(nblk < cap ? nblk : cap)), dim3(NTHR), lds, s, in, out, a, PairFrontArgs{}, batch);
    return per_cu;
}
// ops 23..30 of person_detect: the front pair (6x6x128 stride 2 -> 3x3x256) + pair3_tail<3,3,256,2> in one launch
bool pair_front_supported(int H, int W, int C, int S, int N, int tailH, int tailC) {
    return H == 6 && W == 6 && C == 128 && S == 2 && N == 256 && tailH == 3 && tailC == 256;
}
template <int MG, uint32_t XR4>
static void launch_pair_front_tail_t(const int8_t *in, int8_t *out, const PairTailArgs &a, const PairFrontArgs &fr, size_t batch, hipStream_t s) {
    constexpr int H = 3, C = 256, NTHR = C * 4, XP = H * H * C + 16, XP1 = 4 * H * H * (C / 2) + 16;
    constexpr int lds = 16 * XP1 + (16 + 16) * XP + 256 + 2 * 16 * 4 * 4 + 256 * 4 + 2 * 4 * 4 + 256 + 3 * (C / 2 + 3 * C) * 4;
    static_assert(lds <= 160 * 1024, "one workgroup per CU");
    const size_t nblk = (batch + 15) / 16;
    static LaunchState st;
    const int per_cu = prepared(st, pair3_tail<H, H, C, 2, NTHR, false, MG, XR4, true>, NTHR, lds);
    const size_t cap = (size_t)256 * per_cu;
    hipLaunchKernelGGL((pair3_tail<H, H, C, 2, NTHR, false, MG, XR4, true>), dim3((unsigned)(nblk < cap ? nblk : cap)), dim3(NTHR), lds, s, in, out, a, fr, batch);
}
void launch_pair_front_tail(const int8_t *in, int8_t *out, const PairTailArgs &a, const PairFrontArgs &fr, size_t batch, hipStream_t s) {
    if (a.tail.xr) {
        if (a.magic) launch_pair_front_tail_t<1, 0x80808080u>(in, out, a, fr, batch, s);
        else launch_pair_front_tail_t<0, 0x80808080u>(in, out, a, fr, batch, s);
    } else {
        if (a.magic) launch_pair_front_tail_t<1, 0u>(in, out, a, fr, batch, s);
        else launch_pair_front_tail_t<0, 0u>(in, out, a, fr, batch, s);
    }
}
void launch_pair_tail(const int8_t *in, int8_t *out, const PairTailArgs &a, size_t batch, hipStream_t s) {
    int per_cu = 1;
#define MF_PT_GO(HH, CC, DB)                                                                          \
    do {                                                                                              \
        if (a.tail.xr) {                                                                              \
            if (a.magic) per_cu = launch_pair_tail_t<HH, CC, DB, 1, 0x80808080u>(in, out, a, batch, s); \
            else per_cu = launch_pair_tail_t<HH, CC, DB, 0, 0x80808080u>(in, out, a, batch, s);       \
        } else {                                                                                      \
            if (a.magic) per_cu = launch_pair_tail_t<HH, CC, DB, 1, 0u>(in, out, a, batch, s);        \
            else per_cu = launch_pair_tail_t<HH, CC, DB, 0, 0u>(in, out, a, batch, s);                \
        }                                                                                             \
    } while (0)
    if (a.C == 256) {
        if (a.H == 3) MF_PT_GO(3, 256, true);
        else if (a.H == 2) MF_PT_GO(2, 256, true);
        else MF_PT_GO(4, 256, false);
    } else {
        if (a.H == 3) MF_PT_GO(3, 128, true);
        else if (a.H == 2) MF_PT_GO(2, 128, true);
        else MF_PT_GO(4, 128, true);
    }
#undef MF_PT_GO
    (void)per_cu;
#if MF_TAIL3_DIAG
    {
        static int calls = 0;
        if (++calls == 12) {
            (void)hipStreamSynchronize(s);
            long long h[32];
            (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_tail3_trace), sizeof(h));
            fprintf(stderr, "[tail3 trace] batch %zu per_cu %d; cycles since kernel entry:", batch, per_cu);
            for (int i = 0; i < 16; ++i) fprintf(stderr, " %d:%lld", i, h[i] - h[31]);
            fprintf(stderr, "\n");
        }
    }
#endif
}

// ------------------------------------------------------------------------
// (folded in from k_fused.hip, round 6: the standalone tail kernel -- any pooled shape -- and the routing of the pair launches)
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
    const size_t img_elems = (size_t)p.H * p.W * p.C;
    for (size_t b = wave; b < batch; b += nwaves) tail_one<N>(in + b * img_elems, out + b * N, p, lane);
}

// ---- launchers ----
// The DepthwiseConv2D 3x3 + Conv2D 1x1 pair kernels live in k_fused_mm.hip (depthwise taps on the matrix pipe): dwpw_rr keeps the
// intermediate tensor in registers (C <= 32), dwpw_mm in LDS (every table shape).  MF_DWPW_IMPL=mm takes dwpw_mm for every pair.
// (Round 1's dwpw3x3 -- taps on the VALU -- was retired in round 5: every table shape has had a matrix-pipe kernel since round 2.)
int dwpw_impl() {
    return switches().dwpw_mm_only ? 1 : 2;
}
const char *dwpw_name(int H, int W, int C, int S, int N) {
    if (dwpw_impl() == 2 && dwpw_rr_name(H, W, C, S, N)) return dwpw_rr_name(H, W, C, S, N);
    return dwpw_mm_name(H, W, C, S, N);
}
bool launch_dwpw(int H, int W, int C, int S, int N, const int8_t *in, int8_t *out, const DwPwArgs &a,
                 int batch, hipStream_t s) {
    if (dwpw_impl() == 2 && launch_dwpw_rr(H, W, C, S, N, in, out, a, batch, s)) return true;
    return launch_dwpw_mm(H, W, C, S, N, in, out, a, batch, s);
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

} // namespace k
} // namespace mf
