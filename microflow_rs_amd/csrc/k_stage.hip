// k_stage.hip -- the late stage of a MobileNet-v1 style network as ONE persistent kernel.
//
// person_detect ops 13..30: five DepthwiseConv2D 3x3 + Conv2D 1x1 pairs on 6x6x128, the stride-2 pair
// 6x6x128 -> 3x3x256, the pair on 3x3x256, AveragePool2D -> Conv2D (2 outputs) -> Softmax.
// (src/ops/depthwise_conv_2d.rs:28-105, conv_2d.rs:28-108, average_pool_2d.rs:29-66, softmax.rs:15-27.)
// Per inference these tensors are 2.3 .. 4.6 KB, so the pair kernels spend their time on launches, HBM round
// trips and half-empty workgroups; here G images enter LDS once (4.6 KB each) and 2 bytes leave.  Every
// intermediate tensor is the reference's int8 tensor, requantised with the reference's arithmetic -- it just
// lives in LDS.
//
// A step (G = 4 images per workgroup) is a fixed sequence of phases; a workgroup barrier follows every DEPTHWISE
// phase only (pointwise(L) -> depthwise(L+1) needs none: a wave reads back the channels it wrote, see below):
//     depthwise (tile -> MID)   as in dwpw_mm: taps on the matrix pipe, unit = 16 columns x 16 channels,
//                               wave w owns channel group w (and w + 8 when C = 256)
//     pointwise (MID -> tile)   wave w owns output channels 16w .. 16w+15 (and 16(w+8) ..) for ALL pixels: its A
//                               operands and epilogue constants are 8 + 12 VGPRs, and its 4-byte results go
//                               straight into the NEXT depthwise's halo tile (same swizzle), or -- last pair --
//                               into a plain [pixel][256] buffer for the tail
//     tail                      wave g < G: pool + head + softmax of image g (k_tail.hpp), 2 bytes to HBM
// Both phase kinds walk 9 (6 or 3 for the 3x3 tensors) equal items per wave with lane-constant + immediate
// LDS addresses.  The weights of the NEXT phase are fetched (L2-resident, 12 .. 56 VGPRs) before the current
// phase's arithmetic, and the next step's images are DMA-staged as soon as the 6x6 tile is dead (after the
// stride-2 depthwise), under the last four phases.
// LDS: 6x6x128 halo tiles 33 KB + 3x3x256 halo tiles 27 KB + MID 18 KB = 80 KB -> two workgroups per CU.
#include "k_common.hpp"
#include "k_tail.hpp"

#include <cstdio>

namespace mf {
namespace k {

#ifndef MF_STAGE_DIAG
#define MF_STAGE_DIAG 0
#endif
#ifndef MF_STAGE_SB
#define MF_STAGE_SB 1 // items between two scheduling barriers in the 9-item phases (tuning)
#endif

#if MF_STAGE_DIAG == 2 // diagnostics build: cycle stamps of block 0 / wave 0 at every phase boundary of its 2nd step
__device__ long long g_stage_trace[32];
#define MF_TR(k) do { if (blockIdx.x == 0 && wave == 0 && lane == 0 && trace_step == 1) g_stage_trace[k] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define MF_TR(k) do { } while (0)
#endif

namespace {
struct DwW {        // depthwise operands of one 16-channel group
    v4i A[3];
    float4 a, s;
    int4 k;
};
template <int KS> struct PwW { // pointwise operands of one 16-output-channel tile
    v4i A[KS];
    float4 a, s;
    int4 k;
};
} // namespace

template <int G, int NTHR, int NREP, int NOUT>
__global__ __launch_bounds__(NTHR, 4) void late_stage_6x6x128(const int8_t *__restrict__ in, int8_t *__restrict__ out,
                                                              StageArgs p, int batch) {
    static_assert(G == 4 && NTHR == 512, "column grids below are written for 4 images and 8 waves");
    constexpr int NWAVE = 8;
    // 6x6x128 halo tile.  Depthwise column grid: 2 rows x 2 x x 4 images (y fastest).  Row pitch +32, image pitch
    // +64 and the group index XOR (x & 1) make every tap read conflict-free and the other three access patterns of
    // a pair 1.5x / 2x / 1x their ideal LDS cycles (scripts/model/stage_banks.py: 270 cycles per wave and pair
    // against 464 for the dwpw_mm<6,6,128> layout this kernel started with).
    constexpr int LP6 = 128, ROW6 = 128 + 768 + 128 + 32, TILE6 = 8 * ROW6 + 64, TS6 = 0x001;
    // 3x3x256 halo tile (column grid 4 images x 1 row x 4 x: one x position is padding)
    constexpr int LP3 = 256, ROW3 = 256 + 768 + 256 + 64, TILE3 = 5 * ROW3, TS3 = 0x021;
    constexpr int PIX6 = 36, PIX3 = 9, NP6 = G * PIX6, NP3 = G * PIX3;
    constexpr int PLANE6 = NP6 * 16, PLANE3 = NP3 * 16; // MID planes [16-channel group][pixel][16 B]
    constexpr int IMG6 = PIX6 * 128;
    // LDS regions.  B: MID of the even 6x6 pairs, later the 3x3x256 halo tile.  A: MID of the odd 6x6 pairs, later
    // the 3x3x256 MID planes followed by [3x3x128 MID planes | the tail's plain [pixel][256] input] (the second
    // takes over when the first is dead).  Two MID buffers let a wave run pointwise(L) -> depthwise(L+1) without a
    // barrier: it only reads the channels it wrote itself, and writes the buffer nobody is still reading.
    constexpr int OFF_B = G * TILE6 + 512, OFF_T3 = OFF_B;
    constexpr int BBYTES = G * TILE3 > 8 * PLANE6 ? G * TILE3 : 8 * PLANE6;
    constexpr int OFF_A = OFF_B + BBYTES + 512;
    constexpr int OFF_M3B = OFF_A, OFF_M3A = OFF_A + 16 * PLANE3, OFF_X3 = OFF_M3A;
    constexpr int ABYTES = (8 * PLANE6 > 16 * PLANE3 + NP3 * 256) ? 8 * PLANE6 : 16 * PLANE3 + NP3 * 256;
    static_assert(OFF_A + ABYTES <= 81920, "two workgroups per CU");
    static_assert(NREP % 2 == 1, "the last 6x6 pair must use region B's MID (region A is the stride-2 pair's)");

    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    for (int i = tid; i < OFF_A / 16; i += NTHR) ((uint4 *)lds)[i] = make_uint4(p.izp4, p.izp4, p.izp4, p.izp4);

    // ---- lane constants ----
    const int col = lane & 15, g = lane >> 4;
    // depthwise 6x6 stride 1: columns = (row parity cy, x parity cx, image cg), units = 3 row pairs x 3 x pairs
    const int a_cy = col & 1, a_cx = (col >> 1) & 1, a_cg = col >> 2;
    const int a_xl = a_cx + g - 1;
    const int tb6 = a_cg * TILE6 + a_cy * ROW6 + LP6 + a_xl * 128 + 16 * (wave ^ tile_swz<TS6>(a_xl));
    const int mb6 = wave * PLANE6 + (a_cg * PIX6 + a_cy * 6 + a_cx) * 16 + 4 * g;
    // depthwise stride 2 (6x6 -> 3x3) and depthwise 3x3: columns = (x 0..3 [3 = padding], image), units = 3 rows
    const int b_cx = col & 3, b_cg = col >> 2;
    const bool b_valid = b_cx < 3;
    const int s_xl = 2 * b_cx + g - 1;
    const int tb6s = b_cg * TILE6 + LP6 + s_xl * 128 + 16 * (wave ^ tile_swz<TS6>(s_xl));
    const int c_xl = b_cx + g - 1;
    const int tb3 = OFF_T3 + b_cg * TILE3 + LP3 + c_xl * 256 + 16 * (wave ^ tile_swz<TS3>(c_xl));
    const int mb3 = wave * PLANE3 + (b_cg * PIX3 + b_cx) * 16 + 4 * g;
    // pointwise: lane (pixel column pcol, pg) of chunk c handles pixel 16c + pcol, output channels 16 tt + 4 pg ..
    const int pcol = lane & 15, pg = lane >> 4;
    int o6[9]; // where chunk c's result goes in the 6x6x128 halo tile (this wave's 16-channel group, swizzled)
#pragma unroll
    for (int c = 0; c < 9; ++c) {
        const int pix = c * 16 + pcol, img = pix / PIX6, r = pix % PIX6, y = r / 6, x = r % 6;
        o6[c] = img * TILE6 + (y + 1) * ROW6 + LP6 + x * 128 + 16 * (wave ^ tile_swz<TS6>(x)) + 4 * pg;
    }
    int o3[3]; // the same for the 3x3x256 tile, tile tt = wave (tile wave + 8 is 128 bytes further); -1: no such pixel
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int pix = c * 16 + pcol, img = pix / PIX3, r = pix % PIX3, y = r / 3, x = r % 3;
        o3[c] = pix < NP3 ? OFF_T3 + img * TILE3 + (y + 1) * ROW3 + LP3 + x * 256 + 16 * (wave ^ tile_swz<TS3>(x)) + 4 * pg : -1;
    }

    // Operand fetches: wave-uniform base (SGPRs: the pair's pointer + the wave's group / tile) + a 32-bit lane offset,
    // so that no per-lane 64-bit addresses are kept alive across the step loop.
    const uint32_t l16 = (uint32_t)lane * 16u, g16 = (uint32_t)g * 16u, pg16 = (uint32_t)pg * 16u;
    // (the pointers come out of a table in memory, so the compiler no longer knows they are global: say so, or
    // every fetch becomes a flat load with a 64-bit per-lane address)
    typedef __attribute__((address_space(1))) const v4i g_v4i;
    auto ld16 = [](const void *base, uint32_t off) { return *(g_v4i *)((uintptr_t)base + off); };
    auto ldf4 = [&](const void *base, uint32_t off) {
        const v4i v = ld16(base, off);
        return make_float4(__int_as_float(v[0]), __int_as_float(v[1]), __int_as_float(v[2]), __int_as_float(v[3]));
    };
    auto ldi4 = [&](const void *base, uint32_t off) {
        const v4i v = ld16(base, off);
        return make_int4(v[0], v[1], v[2], v[3]);
    };
    // The pair table is read with SCALAR loads (constant address space): a phase's operand fetches then wait for a
    // scalar-cache hit, not for a vector load of their own pointers.  The pointer is re-laundered every step
    // (below) so that operand addresses are formed where they are used instead of being hoisted out of the step
    // loop into ~60 VGPRs.
    typedef __attribute__((address_space(4))) const StagePair c_pair;
    c_pair *pairs = (c_pair *)(uintptr_t)p.pairs;
    auto load_dw = [&](int pair, int q) {
        c_pair &sp = pairs[pair];
        DwW w;
#pragma unroll
        for (int ty = 0; ty < 3; ++ty) w.A[ty] = ld16((const uint8_t *)sp.dw_wmm + (q * 3 + ty) * 1024, l16);
        w.a = ldf4((const uint8_t *)sp.dwA + q * 64, g16);
        w.s = ldf4((const uint8_t *)sp.dwS + q * 64, g16);
        w.k = ldi4((const uint8_t *)sp.dwK + q * 64, g16); // (the host added the bit-pattern offset: no dependent VALU here)
        return w;
    };
    auto load_pw2 = [&](int pair, int tt) {
        c_pair &sp = pairs[pair];
        PwW<2> w;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) w.A[ks] = ld16((const uint8_t *)sp.pw_w + (tt * 2 + ks) * 1024, l16);
        w.a = ldf4((const uint8_t *)sp.pwA + tt * 64, pg16);
        w.s = ldf4((const uint8_t *)sp.pwS + tt * 64, pg16);
        w.k = ldi4((const uint8_t *)sp.pwK + tt * 64, pg16);
        return w;
    };
    auto load_pw4 = [&](int pair, int tt) {
        c_pair &sp = pairs[pair];
        PwW<4> w;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) w.A[ks] = ld16((const uint8_t *)sp.pw_w + (tt * 4 + ks) * 1024, l16);
        w.a = ldf4((const uint8_t *)sp.pwA + tt * 64, pg16);
        w.s = ldf4((const uint8_t *)sp.pwS + tt * 64, pg16);
        w.k = ldi4((const uint8_t *)sp.pwK + tt * 64, pg16);
        return w;
    };
    // Items (depthwise units, pointwise chunks) run as a two-deep software pipeline: the operand loads of item
    // i + 1 are issued, then item i is multiplied, requantised and written; sched_barrier keeps the compiler from
    // hoisting every item's loads to the front (which spills at the 128-VGPR budget two workgroups per CU need).
    struct Taps {
        v4i b[3];
    };
    auto dw_load = [&](int taddr, int rowpitch) {
        Taps t;
        t.b[0] = *(const v4i *)(lds + taddr);
        t.b[1] = *(const v4i *)(lds + taddr + rowpitch);
        t.b[2] = *(const v4i *)(lds + taddr + 2 * rowpitch);
        return t;
    };
    // one depthwise unit: 3 MFMAs on the loaded taps, requantise, 4 result bytes to MID
    auto dw_finish = [&](const DwW &w, const Taps &t, int maddr, float lo, float hi, bool valid) {
        v4i acc = {w.k.x, w.k.y, w.k.z, w.k.w};
        acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(w.A[0], t.b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(w.A[1], t.b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(w.A[2], t.b[2], acc, 0, 0, 0);
        const uint32_t d = requant_pack4<true, 0u>(acc[0], acc[1], acc[2], acc[3], w.a, w.s, lo, hi);
        if (valid) *(uint32_t *)(lds + maddr) = d;
    };
    // Depthwise items i = 0 .. N-1 at tile offsets taddr + toff(i), MID offsets maddr + moff(i), as a three-stage
    // pipeline: the taps of item i + 2 are loaded, the three MFMAs of item i + 1 are issued BETWEEN the pieces of item
    // i's requantisation (an MFMA is asynchronous: its ~40 cycles pass under the next ~16 VALU instructions instead
    // of stalling the wave, which executes in order), then item i's packed dword is written.
    auto dw_pipe = [&](auto n_c, const DwW &w, int taddr, int rowpitch, int maddr, float lo, float hi, bool valid, auto toff,
                       auto moff) {
        constexpr int NI = decltype(n_c)::value;
        Taps t1 = dw_load(taddr + toff(0), rowpitch), t2 = t1;
        if (NI > 1) t2 = dw_load(taddr + toff(1), rowpitch);
        v4i acc = {w.k.x, w.k.y, w.k.z, w.k.w};
        acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(w.A[0], t1.b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(w.A[1], t1.b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(w.A[2], t1.b[2], acc, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const bool more = i + 1 < NI;
            Taps t3 = t2;
            if (i + 2 < NI) t3 = dw_load(taddr + toff(i + 2), rowpitch);
            v4i nxt = {w.k.x, w.k.y, w.k.z, w.k.w};
            __builtin_amdgcn_sched_barrier(0);
            if (more) nxt = __builtin_amdgcn_mfma_i32_16x16x64_i8(w.A[0], t2.b[0], nxt, 0, 0, 0);
            const float r0 = requant_clamped<true>(acc[0], w.a.x, w.s.x, lo, hi);
            const float r1 = requant_clamped<true>(acc[1], w.a.y, w.s.y, lo, hi);
            __builtin_amdgcn_sched_barrier(0);
            if (more) nxt = __builtin_amdgcn_mfma_i32_16x16x64_i8(w.A[1], t2.b[1], nxt, 0, 0, 0);
            const float r2 = requant_clamped<true>(acc[2], w.a.z, w.s.z, lo, hi);
            const float r3 = requant_clamped<true>(acc[3], w.a.w, w.s.w, lo, hi);
            __builtin_amdgcn_sched_barrier(0);
            if (more) nxt = __builtin_amdgcn_mfma_i32_16x16x64_i8(w.A[2], t2.b[2], nxt, 0, 0, 0);
            const uint32_t d = cvt_pack4(r0, r1, r2, r3);
            if (valid) *(uint32_t *)(lds + maddr + moff(i)) = d;
            __builtin_amdgcn_sched_barrier(0);
            acc = nxt, t2 = t3;
        }
    };
    // three units one row step apart (the 3x3 outputs): taddr / maddr advance by trow / 48 bytes
    auto dw_rows3 = [&](const DwW &w, int taddr, int trow, int rowpitch, int maddr, float lo, float hi, bool valid) {
        dw_pipe(std::integral_constant<int, 3>{}, w, taddr, rowpitch, maddr, lo, hi, valid, [trow](int u) { return u * trow; },
                [](int u) { return u * 3 * 16; });
    };

    // Region B serves as a MID buffer during the 6x6 pairs, which overwrites the halo of the 3x3x256 tile living
    // there afterwards.  Every wave therefore re-fills, for ITS two channel groups (the only ones it will read), the
    // 16 halo pixel slots of each image -- one slot per lane: rows 0 and 4 (x = -1 .. 3), x = -1 and 3 of rows 1 .. 3.
    int h3;
    {
        const int img = lane >> 4, sl = lane & 15;
        const int row = sl < 5 ? 0 : (sl < 10 ? 4 : 1 + (sl - 10) / 2);
        const int x = sl < 5 ? sl - 1 : (sl < 10 ? sl - 6 : (((sl - 10) & 1) ? 3 : -1));
        h3 = OFF_T3 + img * TILE3 + row * ROW3 + LP3 + x * 256 + 16 * (wave ^ tile_swz<TS3>(x));
    }

    auto stage = [&](int st) { // G images, 6 rows each, one 768-byte DMA per row, group index swizzled like TS6
        const int src_lane = lane ^ tile_swz<TS6>(lane >> 3);
#pragma unroll
        for (int k = 0; k < (G * 6 + NWAVE - 1) / NWAVE; ++k) {
            const int r = k * NWAVE + wave;
            const int gi = r / 6, y = r % 6;
            if (r < G * 6 && st * G + gi < batch && lane < 48)
                dma16(in + ((size_t)(st * G + gi) * IMG6 + y * 768 + src_lane * 16), lds + gi * TILE6 + (y + 1) * ROW6 + LP6);
        }
    };

    __syncthreads(); // halo fill complete before any DMA lands
    const int nsteps = (batch + G - 1) / G;
    int step = blockIdx.x;
    if (step < nsteps) stage(step);
    DwW wd = load_dw(0, wave);
#if MF_STAGE_DIAG == 1
    const PwW<2> wp_diag = load_pw2(0, wave);
#endif

#if MF_STAGE_DIAG == 2
    int trace_step = 0;
#endif
    for (; step < nsteps; step += gridDim.x) {
        MF_TR(24);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads(); // this step's images are in the 6x6 tile; the previous step's tail is done with its input
        asm volatile("" : "+s"(pairs));
        const int gvalid = min(G, batch - step * G);
        MF_TR(0);

        // ---------------- NREP pairs on 6x6x128 ----------------
        for (int rep = 0; rep < NREP; ++rep) {
            const int mid = (rep & 1) ? OFF_A : OFF_B; // this pair's MID buffer
#if MF_STAGE_DIAG == 1 // diagnostics build (wrong results): no per-pair operand fetches inside the 6x6 loop
            const PwW<2> wp = wp_diag;
#else
            const PwW<2> wp = load_pw2(rep, wave);     // lands during the depthwise phase
#endif
            {
                const float lo = pairs[rep].dw_lo, hi = pairs[rep].dw_hi;
                const int mb = mid + mb6;
                dw_pipe(std::integral_constant<int, 9>{}, wd, tb6, ROW6, mb, lo, hi, true,
                        [](int u) { return (u / 3) * 2 * ROW6 + (u % 3) * 2 * 128; },
                        [](int u) { return ((u / 3) * 12 + (u % 3) * 2) * 16; });
            }
            MF_TR(1 + 3 * rep);
            __syncthreads(); // MID complete (every channel group); every wave is done reading the tile
            MF_TR(2 + 3 * rep);
#if MF_STAGE_DIAG != 1
            wd = load_dw(rep + 1, wave); // the next pair's depthwise (pair NREP = the stride-2 pair), lands during the pointwise phase
#endif
            {
                const float lo = pairs[rep].pw_lo, hi = pairs[rep].pw_hi;
                const int rb = mid + pg * PLANE6 + pcol * 16;
                // same pipeline: operands of chunk c + 2 loaded, the two MFMAs of chunk c + 1 between the halves of
                // chunk c's requantisation
                v4i b0 = *(const v4i *)(lds + rb), b1 = *(const v4i *)(lds + rb + 4 * PLANE6);
                v4i c0 = *(const v4i *)(lds + rb + 256), c1 = *(const v4i *)(lds + rb + 4 * PLANE6 + 256);
                v4i acc = {wp.k.x, wp.k.y, wp.k.z, wp.k.w};
                acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(wp.A[0], b0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(wp.A[1], b1, acc, 0, 0, 0);
#pragma unroll
                for (int c = 0; c < 9; ++c) {
                    const bool more = c + 1 < 9;
                    v4i e0 = c0, e1 = c1;
                    if (c + 2 < 9) {
                        e0 = *(const v4i *)(lds + rb + (c + 2) * 256);
                        e1 = *(const v4i *)(lds + rb + 4 * PLANE6 + (c + 2) * 256);
                    }
                    v4i nxt = {wp.k.x, wp.k.y, wp.k.z, wp.k.w};
                    __builtin_amdgcn_sched_barrier(0);
                    if (more) nxt = __builtin_amdgcn_mfma_i32_16x16x64_i8(wp.A[0], c0, nxt, 0, 0, 0);
                    const float r0 = requant_clamped<true>(acc[0], wp.a.x, wp.s.x, lo, hi);
                    const float r1 = requant_clamped<true>(acc[1], wp.a.y, wp.s.y, lo, hi);
                    __builtin_amdgcn_sched_barrier(0);
                    if (more) nxt = __builtin_amdgcn_mfma_i32_16x16x64_i8(wp.A[1], c1, nxt, 0, 0, 0);
                    const float r2 = requant_clamped<true>(acc[2], wp.a.z, wp.s.z, lo, hi);
                    const float r3 = requant_clamped<true>(acc[3], wp.a.w, wp.s.w, lo, hi);
                    *(uint32_t *)(lds + o6[c]) = cvt_pack4(r0, r1, r2, r3);
                    __builtin_amdgcn_sched_barrier(0);
                    acc = nxt, c0 = e0, c1 = e1;
                }
            }
            MF_TR(3 + 3 * rep);
            // NO barrier: the next depthwise of this wave reads channel group `wave` of the tile -- exactly the bytes
            // this wave has just written (LDS operations of a wave complete in order) -- and writes the other MID buffer
            asm volatile("" ::: "memory");
        }

        // ---------------- stride-2 pair: 6x6x128 -> 3x3x128 -> 3x3x256 ----------------
        // (from here on a wave owns two tiles / channel groups, tt = wave and wave + 8: they are processed one
        // after the other, each with the operands of the next one in flight, so that at most two sets are live)
        PwW<2> wp2 = load_pw2(NREP, wave);
        {
            const float lo = pairs[NREP].dw_lo, hi = pairs[NREP].dw_hi;
            dw_rows3(wd, tb6s, 2 * ROW6, ROW6, OFF_M3A + mb3, lo, hi, b_valid);
        }
        MF_TR(16);
        __syncthreads(); // 3x3x128 MID complete; the 6x6 tile and the last 6x6 MID (region B) are dead
        MF_TR(17);
        {
            const int next = step + gridDim.x;
            if (next < nsteps) stage(next); // flies during the remaining phases
        }
        {
            const uint4 z = make_uint4(p.izp4, p.izp4, p.izp4, p.izp4);
            *(uint4 *)(lds + h3) = z;       // channel group `wave`
            *(uint4 *)(lds + h3 + 128) = z; // channel group `wave + 8`
        }
        auto pw24 = [&](const PwW<2> &w, int t) {
            const float lo = pairs[NREP].pw_lo, hi = pairs[NREP].pw_hi;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                int pix = c * 16 + pcol;
                pix = pix < NP3 ? pix : NP3 - 1;
                const v4i b0 = *(const v4i *)(lds + OFF_M3A + pg * PLANE3 + pix * 16);
                const v4i b1 = *(const v4i *)(lds + OFF_M3A + (pg + 4) * PLANE3 + pix * 16);
                v4i acc = {w.k.x, w.k.y, w.k.z, w.k.w};
                acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(w.A[0], b0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(w.A[1], b1, acc, 0, 0, 0);
                const uint32_t d = requant_pack4<true, 0u>(acc[0], acc[1], acc[2], acc[3], w.a, w.s, lo, hi);
                if (o3[c] >= 0) *(uint32_t *)(lds + o3[c] + 128 * t) = d;
            }
        };
        {
            const PwW<2> wb = load_pw2(NREP, wave + 8);
            pw24(wp2, 0);
            wd = load_dw(NREP + 1, wave);
            pw24(wb, 1);
        }
        MF_TR(18);
        asm volatile("" ::: "memory"); // no barrier: channel groups wave and wave + 8 of the 3x3x256 tile are this wave's own

        // ---------------- pair on 3x3x256 ----------------
        {
            const float lo = pairs[NREP + 1].dw_lo, hi = pairs[NREP + 1].dw_hi;
            const DwW wdb = load_dw(NREP + 1, wave + 8);
            dw_rows3(wd, tb3, ROW3, ROW3, OFF_M3B + mb3, lo, hi, b_valid);
            const PwW<4> wqa = load_pw4(NREP + 1, wave);
            dw_rows3(wdb, tb3 + 8 * 16, ROW3, ROW3, OFF_M3B + mb3 + 8 * PLANE3, lo, hi, b_valid);
            MF_TR(19);
            __syncthreads(); // 3x3x256 MID complete; the 3x3x128 MID is dead (its space becomes the tail's input)
            MF_TR(20);
            auto pw26 = [&](const PwW<4> &w, int tt) {
                const float plo = pairs[NREP + 1].pw_lo, phi = pairs[NREP + 1].pw_hi;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    int pix = c * 16 + pcol;
                    const bool ok = pix < NP3;
                    pix = ok ? pix : NP3 - 1;
                    v4i acc = {w.k.x, w.k.y, w.k.z, w.k.w};
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        const v4i b = *(const v4i *)(lds + OFF_M3B + (pg + 4 * ks) * PLANE3 + pix * 16);
                        acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(w.A[ks], b, acc, 0, 0, 0);
                    }
                    const uint32_t d = requant_pack4<true, 0u>(acc[0], acc[1], acc[2], acc[3], w.a, w.s, plo, phi);
                    if (ok) *(uint32_t *)(lds + OFF_X3 + pix * 256 + 16 * tt + 4 * pg) = d;
                }
            };
            const PwW<4> wqb = load_pw4(NREP + 1, wave + 8);
            pw26(wqa, wave);
            wd = load_dw(0, wave); // for the next step
            pw26(wqb, wave + 8);
        }
        MF_TR(21);
        __syncthreads(); // the tail's input is complete
        MF_TR(22);

        // ---------------- tail: pool + head + softmax, one wave per image ----------------
        if (wave < gvalid)
            tail_one<NOUT>((const int8_t *)lds + OFF_X3 + wave * PIX3 * 256, out + ((size_t)step * G + wave) * NOUT, p.tail, lane);
        MF_TR(23);
#if MF_STAGE_DIAG == 2
        ++trace_step;
#endif
    }
}

// ---- launcher ----
bool launch_late_stage(const int8_t *in, int8_t *out, const StageArgs &a, int batch, hipStream_t s) {
    constexpr int G = 4, NTHR = 512, NREP = 5;
#ifndef MF_STAGE_LDS_KB
#define MF_STAGE_LDS_KB 80 // (tuning: 100 forces one workgroup per CU)
#endif
    constexpr int lds = MF_STAGE_LDS_KB * 1024;
    static LaunchState st2, st4;
    const int nsteps = (batch + G - 1) / G;
#define MF_STAGE(NOUT, ST)                                                                                       \
    {                                                                                                             \
        const int per_cu = prepared(ST, late_stage_6x6x128<G, NTHR, NREP, NOUT>, NTHR, lds);                      \
        const int grid = nsteps < 256 * per_cu ? nsteps : 256 * per_cu;                                           \
        hipLaunchKernelGGL((late_stage_6x6x128<G, NTHR, NREP, NOUT>), dim3(grid), dim3(NTHR), lds, s, in, out, a, batch); \
        return true;                                                                                              \
    }
#if MF_STAGE_DIAG == 2
    {
        static int calls = 0;
        if (++calls == 3 && a.tail.N == 2) {
            const int per_cu = prepared(st2, late_stage_6x6x128<G, NTHR, NREP, 2>, NTHR, lds);
            const int grid = nsteps < 256 * per_cu ? nsteps : 256 * per_cu;
            hipLaunchKernelGGL((late_stage_6x6x128<G, NTHR, NREP, 2>), dim3(grid), dim3(NTHR), lds, s, in, out, a, batch);
            (void)hipStreamSynchronize(s);
            long long h[32];
            (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stage_trace), sizeof(h));
            fprintf(stderr, "[stage trace] grid %d per_cu %d; cycles since step start:", grid, per_cu);
            for (int i = 0; i < 25; ++i) fprintf(stderr, " %d:%lld", i, h[i] - h[24]);
            fprintf(stderr, "\n");
            return true;
        }
    }
#endif
    if (a.tail.N == 2) MF_STAGE(2, st2)
    if (a.tail.N == 4) MF_STAGE(4, st4)
#undef MF_STAGE
    return false;
}

} // namespace k
} // namespace mf
