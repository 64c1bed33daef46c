// k_quad_mm.hip -- person_detect ops 9..12 in one launch: DepthwiseConv2D 3x3 + Conv2D 1x1 on 12x12x64 (stride 1, 64 outputs), then
// DepthwiseConv2D 3x3 stride 2 + Conv2D 1x1 -> 6x6x128, the 12x12x64 tensor between the two pairs in LDS.
// (src/ops/depthwise_conv_2d.rs:28-105, src/ops/conv_2d.rs:28-108; four reference operators, each with its own requantisation.)
//
// Why.  As two dwpw_mm launches these operators run at what a plain copy reaches on this chip (0.70 and 0.65 of the HBM peak, 0.39 ms
// together) and two thirds of what they move is the tensor between them (9 216 B per image written and read back).  Their instruction
// issue -- matrix pipe + VALU, which do not overlap on a SIMD -- is 0.18 ms.
//
// How.  The stage kernel's scheme (k_stage.hip) with two different geometries: a step is G = 2 images, 8 waves; in every phase a wave
// owns ONE 16-channel group or 16-output tile (its operands and epilogue constants: 24 + 16 + 24 + 16 registers for the four phases,
// resident for the whole launch) and half of the pixels:
//     A.dw  staged tile A -> MID A        wave (q, h): group q of image h, 9 units (column grid 4 rows x 4 x: dwpw_mm<12,12,64,1,64>'s)
//     A.pw  MID A -> tile B               wave (t, h): output tile t of image h's 9 pixel chunks; the 4-byte results go straight
//                                         into pair B's halo tile (its swizzle)
//     B.dw  tile B -> MID B (stride 2)    wave (q, h): 3 units (column grid 2 rows x 8 x, 6 of the 8 columns live)
//     B.pw  MID B -> OUT                  wave w: output tile w (of 8) for all 72 pixels; OUT = the plain [pixel][128] tensor, kept in
//                                         the interior of tile B's first image (dead by then) and copied to HBM with 16-byte stores at
//                                         the top of the next step
// Four workgroup barriers per step.  The next step's images are DMA-staged into tile A as soon as A.dw has read it.  (Knock-outs, same
// box: 0.323 ms; staged from cache-resident addresses 0.295; no staging after the first step 0.280; no pair B 0.258; no requantisation
// 0.288; without the two barriers inside pair B 0.326.  An L2 prefetch of the images two steps ahead -- one dword per 64 bytes -- bought
// 1.5 % and was not kept.)
// LDS: 25.5 + 18.5 + 25 + 5.2 + 3.8 (epilogue constants) KB = 78 KB -> two workgroups per CU.
#include "k_common.hpp"

#ifndef MF_QMM_UB
#define MF_QMM_UB 2 // units / chunks of pair A's phases a wave has in flight
#endif
#ifndef MF_QMM_KO
#define MF_QMM_KO 0 // knock-out timing experiments (WRONG results, never shipped): 1 no barriers behind A.pw and B.dw, 2 no copy-out, 4 no pair B at
                    // all, 8 no staging after a workgroup's first step, 16 no requantisation
#endif
namespace mf {
namespace k {

namespace {
struct Q64 {
    static constexpr int G = 2, NTHR = 512, NWAVE = 8, NQ = 4, LP = 64, ROWB = 768;
    // pair A: dwpw_mm<12,12,64,1,64>'s tile (row pad 16, no swizzle), column grid 4 rows x 4 x
    static constexpr int ROW_A = LP + ROWB + LP + 16, TILE_A = 14 * ROW_A, BUF_A = G * TILE_A;
    static constexpr int OPIX_A = 144, NPIX_A = G * OPIX_A, PLANE_A = NPIX_A * 16 + 16;
    // pair B (stride 2): row pad 0, 16-byte group index ^ (x bit 2 | x bit 3 << 1), column grid 2 rows x 8 x (x fastest):
    // conflict-free tap reads by the bank model (scripts/model/dwmm_search.py)
    static constexpr int ROW_B = LP + ROWB + LP, TILE_B = 14 * ROW_B, BUF_B = G * TILE_B, TS_B = 0x043;
    static constexpr int OPIX_B = 36, NPIX_B = G * OPIX_B, P16_B = 80, PLANE_B = P16_B * 16 + 16;
    static constexpr int OFF_TA = 0;
    static constexpr int OFF_MA = OFF_TA + BUF_A + 512;
    static constexpr int OFF_TB = OFF_MA + NQ * PLANE_A + 64;
    static constexpr int OFF_MB = OFF_TB + BUF_B + 512;
    // the four operators' epilogue constants (A | S | Kc per operator): read per phase, 3 x 16 bytes per lane -- resident in registers
    // next to the four operand sets they made the kernel spill (a spill reload is a vector-memory load, and vmcnt retires in order:
    // it waits for the staging DMAs in flight)
    static constexpr int OFF_C = OFF_MB + NQ * PLANE_B + 64;
    static constexpr int C_A_DW = 0, C_A_PW = 64 * 12, C_B_DW = 2 * 64 * 12, C_B_PW = 3 * 64 * 12, C_BYTES = 3 * 64 * 12 + 128 * 12;
    static constexpr int OFF_Q = OFF_C + C_BYTES;
    static constexpr int LDS = OFF_Q + 16;
};
static_assert(2 * Q64::LDS <= 160 * 1024, "two workgroups per CU");
static_assert(Q64::OFF_MA % 16 == 0 && Q64::OFF_TB % 16 == 0 && Q64::OFF_MB % 16 == 0, "16-byte aligned regions");
} // namespace

template <int MG, uint32_t XR4>
__global__ __launch_bounds__(512, 4) void quad_mm_12x12x64(const int8_t *__restrict__ in, int8_t *__restrict__ out, QuadArgs p, int batch) {
    using Z = Q64;
    constexpr int UB = MF_QMM_UB;
    epi_enter<MG>();
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    DynSteps dq;
    dq.init(lds + Z::OFF_Q, p.a.dw.queue, tid, p.a.dw.qcfg);

    // halos: tile A carries pair A's input zero point, tile B pair B's (= the zero point of the tensor between the pairs)
    for (int i = tid; i < (Z::BUF_A + 512) / 16; i += Z::NTHR)
        ((uint4 *)(lds + Z::OFF_TA))[i] = make_uint4(p.a.dw.izp4, p.a.dw.izp4, p.a.dw.izp4, p.a.dw.izp4);
    for (int i = tid; i < (Z::BUF_B + 512) / 16; i += Z::NTHR)
        ((uint4 *)(lds + Z::OFF_TB))[i] = make_uint4(p.b.dw.izp4, p.b.dw.izp4, p.b.dw.izp4, p.b.dw.izp4);

    const int col = lane & 15, g = lane >> 4;   // depthwise: column, tap column;  pointwise: pixel column, 4-channel sub-block
    const int qa = wave & 3, ha = wave >> 2;    // this wave's channel group / output tile (pair A, depthwise B) and image
    // ---- A.dw ----
    const int cyA = col & 3, cxA = col >> 2;
    const int tbA = Z::OFF_TA + ha * Z::TILE_A + cyA * Z::ROW_A + Z::LP + (cxA + g - 1) * 64 + 16 * qa;
    const int mbA = Z::OFF_MA + qa * Z::PLANE_A + (ha * Z::OPIX_A + cyA * 12 + cxA) * 16 + 4 * g;
    // ---- B.dw ----
    const int cxB = col & 7, cyB = col >> 3;
    const int xlB = cxB * 2 + g - 1;
    const int tbB = Z::OFF_TB + ha * Z::TILE_B + cyB * 2 * Z::ROW_B + Z::LP + xlB * 64 + 16 * (qa ^ tile_swz<Z::TS_B>(xlB));
    const int mbB = Z::OFF_MB + qa * Z::PLANE_B + (ha * Z::OPIX_B + cyB * 6 + cxB) * 16 + 4 * g;
    const bool liveB = cxB < 6;
    // ---- operands and epilogue constants of the four phases (resident) ----
    v4i AdA[3], AdB[3];
#pragma unroll
    for (int ty = 0; ty < 3; ++ty) {
        AdA[ty] = ((const v4i *)p.a.dw.wmm)[(qa * 3 + ty) * 64 + lane];
        AdB[ty] = ((const v4i *)p.b.dw.wmm)[(qa * 3 + ty) * 64 + lane];
    }
    const v4i AwA = ((const v4i *)p.a.pw.wprep)[qa * 64 + lane]; // pointwise A: tile qa of 4 (lane (pixel, g) ends with channels 16 g + 4 qa ..)
    const int blk = wave >> 2, tt = wave & 3;                    // pointwise B: tile `wave` of 8 = (blk, tt): channels 64 blk + 16 g + 4 tt ..
    const v4i AwB = ((const v4i *)p.b.pw.wprep)[wave * 64 + lane];
    // epilogue constants -> LDS (Kc with the bit-pattern offset); a phase reads its lane's three 16-byte blocks
    {
        auto put = [&](int off, int n, const float *A, const float *S, const int *K) {
            for (int i = tid; i < n; i += Z::NTHR) {
                ((float *)(lds + Z::OFF_C + off))[i] = A[i], ((float *)(lds + Z::OFF_C + off))[n + i] = S[i];
                ((int *)(lds + Z::OFF_C + off))[2 * n + i] = K[i] + (MG != 0 ? MF_MAGIC_I : 0);
            }
        };
        put(Z::C_A_DW, 64, p.a.dw.A, p.a.dw.S, p.a.dw.Kc), put(Z::C_A_PW, 64, p.a.pw.A, p.a.pw.S, p.a.pw.Kc);
        put(Z::C_B_DW, 64, p.b.dw.A, p.b.dw.S, p.b.dw.Kc), put(Z::C_B_PW, 128, p.b.pw.A, p.b.pw.S, p.b.pw.Kc);
    }
    struct Consts {
        float4 a, s;
        int4 k;
    };
    auto consts = [&](int off, int n, int ch) { // channels ch .. ch + 3 of the operator whose block starts at off (n channels)
        Consts c;
        c.a = *(const float4 *)(lds + Z::OFF_C + off + ch * 4), c.s = *(const float4 *)(lds + Z::OFF_C + off + (n + ch) * 4);
        c.k = *(const int4 *)(lds + Z::OFF_C + off + (2 * n + ch) * 4);
        return c;
    };
    const int ch_dw = 16 * qa + 4 * g, chA = g * 16 + 4 * qa, chB = blk * 64 + g * 16 + 4 * tt;
    // where pointwise A's chunk c goes in tile B (pair B's halo tile, its swizzle)
    int oB[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) {
        const int pix = c * 16 + col, y = pix / 12, x = pix - y * 12;
        oB[c] = Z::OFF_TB + ha * Z::TILE_B + (y + 1) * Z::ROW_B + Z::LP + x * 64 + 16 * (g ^ tile_swz<Z::TS_B>(x)) + 4 * qa;
    }
    // mode 3: this wave's patched channel per phase, if any (kernels.hpp EpiPatchRec; dwpw_mm's tables)
    const EpiPatchRec prDA = epi_patch_load(MG == 3 ? p.a.dw.patch : nullptr, qa), prPA = epi_patch_load(MG == 3 ? p.a.pw.patch : nullptr, qa);
    const EpiPatchRec prDB = epi_patch_load(MG == 3 ? p.b.dw.patch : nullptr, qa), prPB = epi_patch_load(MG == 3 ? p.b.pw.patch : nullptr, wave);
    wg_sync(); // halo fill complete before any DMA lands

    auto stage = [&](int st) { // 24 image rows of 768 bytes over 8 waves, one DMA instruction per row (48 of 64 lanes)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int r = k * Z::NWAVE + wave, gi = r / 12, y = r % 12;
            if (st * Z::G + gi < batch && lane < 48)
                dma16(in + ((size_t)(st * Z::G + gi) * 9216 + y * Z::ROWB + lane * 16), lds + Z::OFF_TA + gi * Z::TILE_A + (y + 1) * Z::ROW_A + Z::LP);
        }
    };
    // OUT: pixel p (of 72) x 128 bytes in the interior of tile B's first image -- row p / 6, 6 pixels per 768-byte row; the 16-byte slot
    // index is XOR-ed with the pixel's low bits (the 16 lanes of a column group write the same slot of 16 pixels 128 bytes apart)
    auto out_addr = [](int pix, int slot) { return Z::OFF_TB + (pix / 6 + 1) * Z::ROW_B + Z::LP + (pix % 6) * 128 + 16 * (slot ^ (pix & 7)); };
    auto copy_out = [&](int st, int gv) {
        int8_t *ob = out + (size_t)st * Z::G * Z::OPIX_B * 128;
        for (int j = tid; j < gv * Z::OPIX_B * 8; j += Z::NTHR) {
            const int pix = j >> 3, sp = j & 7; // physical slot sp holds logical slot sp ^ (pix & 7)
            st_out(ob + pix * 128 + ((sp ^ (pix & 7)) << 4), *(const uint4 *)(lds + Z::OFF_TB + (pix / 6 + 1) * Z::ROW_B + Z::LP + (pix % 6) * 128 + sp * 16));
        }
    };

    const int nsteps = (batch + Z::G - 1) / Z::G;
    if (dq.step < nsteps) stage(dq.step);
    int prev = -1, prev_gv = 0;
    for (; dq.step < nsteps; dq.advance(tid)) {
        const int step = dq.step;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wg_sync(); // B1: tile A has landed; the previous step's OUT is complete
        dq.top(tid);
        const int next = dq.nxt;
        const int gvalid = min(Z::G, batch - step * Z::G);
        if (prev >= 0 && !(MF_QMM_KO & 2)) copy_out(prev, prev_gv); // (done before B2: A.pw overwrites tile B behind it)

        // Units / chunks go through in batches of UB: the MFMA chains of a batch are interleaved, the operands of the NEXT batch are
        // fetched behind them and in front of this batch's requantisation (a wave has one dependency chain per unit: tap rows -> three
        // chained MFMAs -> fma / convert chain -> store).
        // ---------------- A.dw: tile A -> MID A ----------------
        {
            constexpr int NU = 9;
            const Consts k = consts(Z::C_A_DW, 64, ch_dw);
            auto toff = [](int t) constexpr { return (t / 3) * 4 * Z::ROW_A + (t % 3) * 4 * 64; };
            v4i bq[UB][3], bn[UB][3];
#pragma unroll
            for (int u = 0; u < UB; ++u)
#pragma unroll
                for (int ty = 0; ty < 3; ++ty) bq[u][ty] = *(const v4i *)(lds + tbA + toff(u < NU ? u : 0) + ty * Z::ROW_A);
#pragma unroll
            for (int t0 = 0; t0 < NU; t0 += UB) {
                v4i acc[UB];
#pragma unroll
                for (int u = 0; u < UB; ++u) acc[u] = v4i{k.k.x, k.k.y, k.k.z, k.k.w};
#pragma unroll
                for (int ty = 0; ty < 3; ++ty)
#pragma unroll
                    for (int u = 0; u < UB; ++u)
                        if (t0 + u < NU) acc[u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(AdA[ty], bq[u][ty], acc[u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < UB; ++u)
                    if (t0 + UB + u < NU) {
#pragma unroll
                        for (int ty = 0; ty < 3; ++ty) bn[u][ty] = *(const v4i *)(lds + tbA + toff(t0 + UB + u) + ty * Z::ROW_A);
                    }
#pragma unroll
                for (int u = 0; u < UB; ++u)
                    if (t0 + u < NU) {
                        const int t = t0 + u;
                        if constexpr (MG == 3) epi_patch_apply(acc[u], prDA, g);
                        const uint32_t d = (MF_QMM_KO & 16) ? (uint32_t)(acc[u][0] ^ acc[u][1] ^ acc[u][2] ^ acc[u][3])
                                                            : requant_pack4<MG, XR4>(acc[u][0], acc[u][1], acc[u][2], acc[u][3], k.a, k.s, p.a.dw.lo_f, p.a.dw.hi_f);
                        *(uint32_t *)(lds + mbA + ((t / 3) * 4 * 12 + (t % 3) * 4) * 16) = d;
                    }
#pragma unroll
                for (int u = 0; u < UB; ++u)
#pragma unroll
                    for (int ty = 0; ty < 3; ++ty) bq[u][ty] = bn[u][ty];
            }
        }
        wg_sync(); // B2: MID A complete; tile A and OUT are dead
        if (next < nsteps && !((MF_QMM_KO & 8) && prev >= 0)) stage((MF_QMM_KO & 32) ? (int)(blockIdx.x & 63) : next); // lands during the three phases that follow (knock-out 32: always cache-resident images)

        // ---------------- A.pw: MID A -> tile B (pair B's halo tile) ----------------
        {
            constexpr int NC = 9;
            const Consts k = consts(Z::C_A_PW, 64, chA);
            const uint8_t *mid = lds + Z::OFF_MA + g * Z::PLANE_A + (ha * Z::OPIX_A + col) * 16;
            v4i bq[UB], bn[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) bq[u] = *(const v4i *)(mid + (u < NC ? u : 0) * 256);
#pragma unroll
            for (int c0 = 0; c0 < NC; c0 += UB) {
                v4i acc[UB];
#pragma unroll
                for (int u = 0; u < UB; ++u)
                    if (c0 + u < NC) acc[u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(AwA, bq[u], v4i{k.k.x, k.k.y, k.k.z, k.k.w}, 0, 0, 0);
#pragma unroll
                for (int u = 0; u < UB; ++u)
                    if (c0 + UB + u < NC) bn[u] = *(const v4i *)(mid + (c0 + UB + u) * 256);
#pragma unroll
                for (int u = 0; u < UB; ++u)
                    if (c0 + u < NC) {
                        if constexpr (MG == 3) epi_patch_apply(acc[u], prPA, g);
                        const uint32_t d = (MF_QMM_KO & 16) ? (uint32_t)(acc[u][0] ^ acc[u][1] ^ acc[u][2] ^ acc[u][3])
                                                            : requant_pack4<MG, XR4>(acc[u][0], acc[u][1], acc[u][2], acc[u][3], k.a, k.s, p.a.pw.lo_f, p.a.pw.hi_f);
                        *(uint32_t *)(lds + oB[c0 + u]) = d;
                    }
#pragma unroll
                for (int u = 0; u < UB; ++u) bq[u] = bn[u];
            }
        }
        if (!(MF_QMM_KO & 1)) wg_sync(); // B3: tile B complete

        // ---------------- B.dw (stride 2): tile B -> MID B ----------------
        if (!(MF_QMM_KO & 4)) {
            constexpr int NU = 3;
            const Consts k = consts(Z::C_B_DW, 64, ch_dw);
            v4i bq[NU][3]; // all three units' tap rows at once
#pragma unroll
            for (int t = 0; t < NU; ++t)
#pragma unroll
                for (int ty = 0; ty < 3; ++ty) bq[t][ty] = *(const v4i *)(lds + tbB + (t * 4 + ty) * Z::ROW_B);
            v4i acc[NU];
#pragma unroll
            for (int t = 0; t < NU; ++t) acc[t] = v4i{k.k.x, k.k.y, k.k.z, k.k.w};
#pragma unroll
            for (int ty = 0; ty < 3; ++ty)
#pragma unroll
                for (int t = 0; t < NU; ++t) acc[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(AdB[ty], bq[t][ty], acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NU; ++t) {
                if constexpr (MG == 3) epi_patch_apply(acc[t], prDB, g);
                const uint32_t d = (MF_QMM_KO & 16) ? (uint32_t)(acc[t][0] ^ acc[t][1] ^ acc[t][2] ^ acc[t][3])
                                                    : requant_pack4<MG, XR4>(acc[t][0], acc[t][1], acc[t][2], acc[t][3], k.a, k.s, p.b.dw.lo_f, p.b.dw.hi_f);
                if (liveB) *(uint32_t *)(lds + mbB + t * 2 * 6 * 16) = d;
            }
        }
        if (!(MF_QMM_KO & 1)) wg_sync(); // B4: MID B complete; tile B is dead

        // ---------------- B.pw: MID B -> OUT ----------------
        if (!(MF_QMM_KO & 4)) {
            constexpr int NC = 5;
            const Consts k = consts(Z::C_B_PW, 128, chB);
            v4i b[NC], acc[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int pix = c * 16 + col, pc = pix < Z::NPIX_B ? pix : Z::NPIX_B - 1;
                b[c] = *(const v4i *)(lds + Z::OFF_MB + g * Z::PLANE_B + pc * 16);
            }
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(AwB, b[c], v4i{k.k.x, k.k.y, k.k.z, k.k.w}, 0, 0, 0);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int pix = c * 16 + col;
                if constexpr (MG == 3) epi_patch_apply(acc[c], prPB, g);
                const uint32_t d = (MF_QMM_KO & 16) ? (uint32_t)(acc[c][0] ^ acc[c][1] ^ acc[c][2] ^ acc[c][3])
                                                    : requant_pack4<MG, XR4>(acc[c][0], acc[c][1], acc[c][2], acc[c][3], k.a, k.s, p.b.pw.lo_f, p.b.pw.hi_f);
                if (pix < Z::NPIX_B) *(uint32_t *)(lds + out_addr(pix, blk * 4 + g) + 4 * tt) = d;
            }
        }
        prev = step, prev_gv = gvalid;
    }
    wg_sync();
    if (prev >= 0) copy_out(prev, prev_gv);
    dq.finish(tid);
}

template <int MG, uint32_t XR4> static void launch_quad_mm_t(const int8_t *in, int8_t *out, const QuadArgs &a, int batch, hipStream_t s) {
    static LaunchState st;
    const int per_cu = prepared(st, quad_mm_12x12x64<MG, XR4>, Q64::NTHR, Q64::LDS);
    const int nsteps = (batch + Q64::G - 1) / Q64::G;
    const int grid = nsteps < 256 * per_cu ? nsteps : 256 * per_cu;
    QuadArgs b = a;
    b.a.dw.qcfg = dq_config(nsteps, grid, dq_est_us((double)batch * (9216 + 4608), (double)batch * (9216 * 2 + 2304 + 4608)));
    b.a.dw.queue = dq_slot(b.a.dw.queue, b.a.dw.qlaunch);
    hipLaunchKernelGGL((quad_mm_12x12x64<MG, XR4>), dim3(grid), dim3(Q64::NTHR), Q64::LDS, s, in, out, b, batch);
}
bool quad_mm_shape(int H, int W, int C, int S, int N, int H2, int W2, int C2, int S2, int N2) {
    return H == 12 && W == 12 && C == 64 && S == 1 && N == 64 && H2 == 12 && W2 == 12 && C2 == 64 && S2 == 2 && N2 == 128;
}
void launch_quad_mm(const int8_t *in, int8_t *out, const QuadArgs &a, int batch, hipStream_t s) {
    const int mg = std::min(std::min(a.a.dw.magic, a.a.pw.magic), std::min(a.b.dw.magic, a.b.pw.magic));
    const bool u8 = a.a.dw.xr != 0;
    if (mg == 3) launch_quad_mm_t<3, 0u>(in, out, a, batch, s); // (the single-fma form's code does not depend on the element type)
    else if (u8) {
        if (mg == 2) launch_quad_mm_t<2, 0x80808080u>(in, out, a, batch, s);
        else launch_quad_mm_t<1, 0x80808080u>(in, out, a, batch, s);
    } else {
        if (mg == 2) launch_quad_mm_t<2, 0u>(in, out, a, batch, s);
        else launch_quad_mm_t<1, 0u>(in, out, a, batch, s);
    }
}

} // namespace k
} // namespace mf
