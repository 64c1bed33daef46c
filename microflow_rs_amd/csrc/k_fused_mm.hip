// k_fused_mm.hip -- fused DepthwiseConv2D 3x3 -> Conv2D 1x1 with the depthwise taps on the MATRIX pipe.
//
// (src/ops/depthwise_conv_2d.rs:28-105 + src/ops/conv_2d.rs:28-108; same operators, same results as
// dwpw3x3 in k_fused.hip -- only the way the depthwise accumulators are formed differs.)
//
// Why.  r01's fused kernels are VALU-bound, not HBM-bound: per output byte the reference's f32
// requantisation costs ~24 VALU cycles and cannot shrink (every step is individually rounded), and
// the depthwise multiply-adds cost as much again -- 2.75 v_perm byte transposes + 3 v_dot4 per output
// byte -- while the MFMA pipe idles.  Here the 9 taps of 16 channels x 16 pixels are ONE K = 192
// contraction against a block-diagonal weight matrix, i.e. three v_mfma_i32_16x16x64_i8:
//
//     D[channel r][pixel p] = sum over (ty, tx, c')  A[r][(ty,tx,c')] * B[(ty,tx,c')][p]
//     A[r][(ty,tx,c')] = w[ty][tx][16q + r] if c' == r (and tx < 3) else 0      (host-built, wmm)
//     B[(ty,tx,c')][p] = input pixel (y_p + ty - 1, x_p + tx - 1), channel 16q + c'
//
// In NHWC the 16 K-bytes a lane must supply -- tap (ty, tx = lane >> 4), channels 16q .. 16q+15 of
// its pixel -- are 16 CONSECUTIVE bytes of the staged tile: one ds_read_b128, no transposes, no
// dot4s, and the address is lane constant + compile-time unit offset (zero VALU).  15/16 of the
// MACs multiply by zero; that is the price for using a pipe that was idle (3 MFMAs = 48 matrix-pipe
// cycles per 256 outputs, against 92 VALU cycles for their requantisation, which stays the
// bottleneck).  The kernel remains a byte-streaming kernel bounded by HBM / the f32 epilogue; the
// contraction itself is NOT dense and is not priced as MFMA work anywhere.
//
// C = 8 (48x48x8): a 16-byte K-block is two adjacent pixels x 8 channels, so an MFMA column is a
// PAIR of output pixels (2j, 2j+1) and the rows are (pixel parity, channel); three blocks per
// filter row cover input pixels 2j-2 .. 2j+3.
//
// Work decomposition of the depthwise phase: a UNIT = 16 MFMA columns x one 16-channel group.
// The 16 columns of a unit are a CG x CY x CX sub-grid over (image, output row, output x) chosen per
// shape (scripts/model/dwmm_search.py) so that the lanes of every ds_read_b128 service group hit
// distinct 16-byte bank slots; where a pixel is wider than 16 bytes the chunk index inside the
// pixel is XOR-swizzled with low bits of x -- applied on the DMA *source* address, because a DMA
// writes LDS linearly -- and again on the reads.  Every wave owns one channel group (its 12 VGPRs of
// A operands and 12 of epilogue constants) and walks its units with immediates only.
// The depthwise result goes to MID as planes [16-channel group][pixel][16 B]: the writes are
// lane constant + immediate, and the pointwise phase's ds_read_b128 (16 consecutive pixels of one
// plane) is conflict-free without a swizzle.  The pointwise phase is dwpw3x3's.
#include "k_common.hpp"


namespace mf {
namespace k {


// LDS bytes of one workgroup.  WPE (template parameter of the kernel) = waves per SIMD the register allocation
// must leave room for: (workgroups the LDS admits per CU) x (waves per workgroup) / 4, chosen per shape.
constexpr int dwmm_lds_bytes(int H, int W, int C, int S, int N, int G, int NTHR, bool DBUF, int ROWPAD) {
    const int LP = C < 16 ? 16 : C, OH = (H + S - 1) / S, OW = (W + S - 1) / S;
    const int BUF = G * (H + 2) * (LP + W * C + LP + ROWPAD);
    const int NB = N < 64 ? N : 64, CPIX = C < 64 ? 1024 / C : 16;
    const int patch = (NB / 16 < 4) ? (NTHR / 64) * CPIX * N : 0;
    const int NPIX = G * OH * OW, P16 = (NPIX + 15) / 16 * 16;
    const int MIDB = C == 8 ? NPIX * 8 : (C / 16) * (P16 * 16 + 16);
    return (DBUF ? 2 : 1) * BUF + 512 + MIDB + 64 + patch + 16; // (+ 16: the step queue's two ints)
}
// DWONLY: the depthwise operator alone (layer-wise execution): the pointwise phase is replaced by a copy of MID --
// which then IS the operator's output tensor -- to HBM with 16-byte loads and stores.
template <int H, int W, int C, int S, int N, int G, int NTHR, bool DBUF, int CG, int CY, int ORD, int ROWPAD, int TS,
          int WPE, int MG, uint32_t XR4, bool DWONLY>
__global__ __launch_bounds__(NTHR, WPE) void dwpw_mm(const int8_t *__restrict__ in, int8_t *__restrict__ out, DwPwArgs p,
                                                int batch) {
    epi_enter<MG>();
    // ---- depthwise geometry ----
    constexpr bool PAIR = C == 8;                 // MFMA column = two adjacent output pixels
    constexpr int OH = (H + S - 1) / S, OW = (W + S - 1) / S;
    constexpr int OWC = PAIR ? OW / 2 : OW;       // columns per output row
    constexpr int NQ = PAIR ? 1 : C / 16;         // 16-channel groups
    constexpr int CX = 16 / (CG * CY);
    constexpr int UG = G / CG, UY = OH / CY, UX = (OWC + CX - 1) / CX;
    constexpr int LP = C < 16 ? 16 : C;
    constexpr int ROWB = W * C, ROW = LP + ROWB + LP + ROWPAD, TILE = (H + 2) * ROW, BUF = G * TILE;
    constexpr int IMG = H * ROWB, ROWCH = ROWB / 16, NROWS = G * H;
    constexpr int NWAVE = NTHR / 64;
    constexpr int NBUF = DBUF ? 2 : 1;
    constexpr int OPIX = OH * OW, NPIX = G * OPIX;
    constexpr int P16 = (NPIX + 15) / 16 * 16;
    constexpr int PLANE = P16 * 16 + 16;          // bytes per MID plane (+16: planes start on different bank slots)
    constexpr int MIDB = PAIR ? NPIX * 8 : NQ * PLANE;
    static_assert(CG * CY * CX == 16 && G % CG == 0 && OH % CY == 0, "column grid");
    static_assert(ROWB % 16 == 0 && ROWCH <= 64 && ROW % 16 == 0, "staging geometry");
    static_assert(!PAIR || (S == 1 && OW % 2 == 0), "pair columns");
    // units: QW channel groups per wave, PS wave groups share one channel group and split one unit dimension
    constexpr int QW = NQ > NWAVE ? NQ / NWAVE : 1;
    constexpr int PS = NQ >= NWAVE ? 1 : NWAVE / NQ;
    static_assert(NQ >= NWAVE ? NQ % NWAVE == 0 : NWAVE % NQ == 0, "waves per channel group");
    // the PS wave groups tile the unit grid: PSY x PSX x PSG of them along rows, x and images
    constexpr int PSY = cgcd(UY, PS), PSX = cgcd(UX, PS / PSY), PSG = cgcd(UG, PS / PSY / PSX);
    static_assert(PSY * PSX * PSG == PS, "the unit grid does not divide over the wave groups");
    constexpr int NUG = UG / PSG, NUY = UY / PSY, NUX = UX / PSX; // units per wave and channel group
    static_assert(QW == 1 || tile_swz<TS>(0xff) < NWAVE, "swizzle must stay inside a wave's channel-group stride");
    static_assert(tile_swz<TS>(0xff) < (NQ > 1 ? NQ : 1) || NQ == 1, "swizzle wider than the pixel");
    // byte strides of one unit step in the staged tile and in MID
    constexpr int T_UG = CG * TILE, T_UY = CY * S * ROW, T_UX = PAIR ? CX * 16 : CX * S * C;
    constexpr int M_UG = PAIR ? CG * OPIX * 8 : CG * OPIX * 16, M_UY = PAIR ? CY * OW * 8 : CY * OW * 16, M_UX = CX * 16;
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
    // LDS: [staging x NBUF][slack 512][MID (+64 slack)][patch]
    constexpr int MID_OFF = NBUF * BUF + 512;
    constexpr int PATCH_OFF = MID_OFF + MIDB + 64;
    constexpr int DQ_OFF = dwmm_lds_bytes(H, W, C, S, N, G, NTHR, DBUF, ROWPAD) - 16;

    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    DynSteps dq;
    dq.init(lds + DQ_OFF, p.dw.queue, tid, p.dw.qcfg);

    for (int i = tid; i < (NBUF * BUF + 512) / 16; i += NTHR)
        ((uint4 *)lds)[i] = make_uint4(p.dw.izp4, p.dw.izp4, p.dw.izp4, p.dw.izp4);

    // ---- depthwise per-lane constants ----
    const int col = lane & 15, g = lane >> 4;
    int cg, cy, cx;
    {
        // ORD: which of (image, row, x) varies fastest over the 16 columns: 0 gyx, 1 gxy, 2 ygx, 3 yxg, 4 xgy, 5 xyg
        constexpr int D0 = (ORD == 0 || ORD == 1) ? CG : (ORD == 2 || ORD == 3) ? CY : CX;
        constexpr int D1 = (ORD == 2 || ORD == 4) ? CG : (ORD == 0 || ORD == 5) ? CY : CX;
        const int i0 = col % D0, i1 = (col / D0) % D1, i2 = col / (D0 * D1);
        cg = (ORD == 0 || ORD == 1) ? i0 : (ORD == 2 || ORD == 4) ? i1 : i2;
        cy = (ORD == 2 || ORD == 3) ? i0 : (ORD == 0 || ORD == 5) ? i1 : i2;
        cx = (ORD == 4 || ORD == 5) ? i0 : (ORD == 1 || ORD == 3) ? i1 : i2;
    }
    const int q0 = NQ >= NWAVE ? wave : wave % NQ;    // this wave's (first) channel group
    const int wp = NQ >= NWAVE ? 0 : wave / NQ;       // its position along the split unit dimension
    const int wpy = wp % PSY, wpx = (wp / PSY) % PSX, wpg = wp / (PSY * PSX); // its place in the wave-group grid
    const int wave_t = wpg * T_UG + wpy * T_UY + wpx * T_UX;
    const int wave_m = wpg * M_UG + wpy * M_UY + wpx * M_UX;
    int tbase, mbase;
    if constexpr (PAIR) {
        // blocks of a filter row: pixel pairs (2x-2,2x-1), (2x,2x+1), (2x+2,2x+3) [, pad]
        tbase = cg * TILE + cy * ROW + LP + (2 * cx - 2) * 8 + g * 16 + wave_t;
        mbase = (cg * OPIX + cy * OW) * 8 + cx * 16 + g * 4 + wave_m;
    } else {
        const int xl = cx * S + g - 1; // input pixel of this lane's tap column (tx = g; g == 3 meets zero weights)
        tbase = cg * TILE + cy * S * ROW + LP + xl * C + 16 * (q0 ^ tile_swz<TS>(xl)) + wave_t;
        mbase = q0 * PLANE + (cg * OPIX + cy * OW + cx) * 16 + g * 4 + wave_m;
    }
    v4i Adw[QW][3];
    float4 dA[QW], dS[QW];
    int4 dK[QW];
#pragma unroll
    for (int k = 0; k < QW; ++k) {
        const int q = q0 + k * NWAVE;
#pragma unroll
        for (int ty = 0; ty < 3; ++ty) Adw[k][ty] = ((const v4i *)p.dw.wmm)[(q * 3 + ty) * 64 + lane];
        const int ch4 = PAIR ? (g & 1) : 4 * q + g; // this lane's 4 channels start at 4 * ch4
        dA[k] = ((const float4 *)p.dw.A)[ch4];
        dS[k] = ((const float4 *)p.dw.S)[ch4];
        dK[k] = magic4<MG>(((const int4 *)p.dw.Kc)[ch4]);
    }
    // mode 3: the patched channel (kernels.hpp EpiPatchRec; none, for most operators) of each of this wave's tiles
    EpiPatchRec dpr[QW], cpr[TB > 0 ? TB : 1];
#pragma unroll
    for (int k = 0; k < QW; ++k) dpr[k] = MG == 3 && !PAIR ? epi_patch_load(p.dw.patch, q0 + k * NWAVE) : EpiPatchRec{0, 0};

    // ---- pointwise per-lane constants ----
    const int pcol = lane & 15, pg = lane >> 4;
    const int blk = wave % NSPLIT, slot = wave / NSPLIT;
    v4i Aw[Q][TB][KS];
    float4 cA[TB], cS[TB];
    int4 cK[TB];
    if constexpr (!DWONLY) {
#pragma unroll
        for (int q = 0; q < Q; ++q)
#pragma unroll
            for (int tt = 0; tt < TB; ++tt)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
                    Aw[q][tt][ks] = ((const v4i *)p.pw.wprep)[((((size_t)blk * Q + q) * TB + tt) * KS + ks) * 64 + lane];
#pragma unroll
        for (int tt = 0; tt < TB; ++tt) {
            const int ch = blk * NB + pg * (NB / 4) + 4 * tt;
            cA[tt] = *(const float4 *)(p.pw.A + ch);
            cS[tt] = *(const float4 *)(p.pw.S + ch);
            cK[tt] = magic4<MG>(*(const int4 *)(p.pw.Kc + ch));
            cpr[tt] = MG == 3 ? epi_patch_load(p.pw.patch, blk * TB + tt) : EpiPatchRec{0, 0};
        }
    }
    wg_sync(); // halo fill complete before any DMA lands

    auto stage = [&](int st, int buf) {
        // chunk i of a row = pixel i / NQ, 16-byte group i % NQ; it receives the source chunk whose group is
        // XOR-swizzled with the pixel's low x bits (the tap reads undo it)
        const int src_lane = NQ > 1 ? (lane ^ tile_swz<TS>(lane / (NQ > 1 ? NQ : 1))) : lane;
#pragma unroll
        for (int k = 0; k < (NROWS + NWAVE - 1) / NWAVE; ++k) {
            const int r = k * NWAVE + wave;
            const int gi = r / H, y = r % H;
            if (r < NROWS && st * G + gi < batch && lane < ROWCH)
                dma16(in + ((size_t)(st * G + gi) * IMG + y * ROWB + src_lane * 16),
                      lds + buf * BUF + gi * TILE + (y + 1) * ROW + LP);
        }
    };

    uint8_t *mid = lds + MID_OFF;
    const int nsteps = (batch + G - 1) / G;
    int cur = 0;
    if (dq.step < nsteps) stage(dq.step, 0);

    for (; dq.step < nsteps; dq.advance(tid)) {
        const int step = dq.step;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wg_sync(); // B1: staged tile complete; previous pointwise phase done with MID
        dq.top(tid);
        const int next = dq.nxt;
        if constexpr (DBUF) {
            if (next < nsteps) stage(next, cur ^ 1);
        }
        const int gvalid = min(G, batch - step * G);

        // ---------------- depthwise phase: staged tile -> MID, on the matrix pipe ----------------
        {
            const uint8_t *tb = lds + cur * BUF + tbase;
            uint8_t *mb = mid + mbase;
            // Units go through in batches of UB: the tap loads of the NEXT batch are issued before the
            // requantisation of this one, and the UB MFMA chains of a batch are interleaved (a chain is 3
            // dependent MFMAs; two chains issue back to back without waiting on each other).
            constexpr int NU = NUG * NUY * NUX, NUT = QW * NU;
            constexpr int UB = WPE >= 4 ? 1 : (NUT % 2 == 0 ? 2 : (NUT % 3 == 0 ? 3 : 1)); // 128-VGPR kernels: one chain at a time
            auto toff_of = [](int t) constexpr {
                const int k = t / NU, iu = t % NU;
                // unit coordinates (the wave-group part is in tbase / mbase): interleaved over the wave groups
                const int ug = (iu / (NUY * NUX)) * PSG, uy = ((iu / NUX) % NUY) * PSY, ux = (iu % NUX) * PSX;
                return ug * T_UG + uy * T_UY + ux * T_UX + k * NWAVE * 16;
            };
            auto moff_of = [](int t) constexpr {
                const int k = t / NU, iu = t % NU;
                const int ug = (iu / (NUY * NUX)) * PSG, uy = ((iu / NUX) % NUY) * PSY, ux = (iu % NUX) * PSX;
                return ug * M_UG + uy * M_UY + ux * M_UX + k * NWAVE * PLANE;
            };
            v4i bq[UB][3], bn[UB][3];
#pragma unroll
            for (int u = 0; u < UB; ++u)
#pragma unroll
                for (int ty = 0; ty < 3; ++ty) bq[u][ty] = *(const v4i *)(tb + toff_of(u) + ty * ROW);
#pragma unroll
            for (int t0 = 0; t0 < NUT; t0 += UB) {
                v4i acc[UB];
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const int k = (t0 + u) / NU;
                    acc[u] = v4i{dK[k].x, dK[k].y, dK[k].z, dK[k].w};
                }
#pragma unroll
                for (int ty = 0; ty < 3; ++ty)
#pragma unroll
                    for (int u = 0; u < UB; ++u)
                        acc[u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Adw[(t0 + u) / NU][ty], bq[u][ty], acc[u], 0, 0, 0);
                if (t0 + UB < NUT) {
#pragma unroll
                    for (int u = 0; u < UB; ++u)
#pragma unroll
                        for (int ty = 0; ty < 3; ++ty) bn[u][ty] = *(const v4i *)(tb + toff_of(t0 + UB + u) + ty * ROW);
                }
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const int k = (t0 + u) / NU;
                    if constexpr (MG == 3 && !PAIR) epi_patch_apply(acc[u], dpr[k], g);
                    const uint32_t d = requant_pack4<MG, XR4>(acc[u][0], acc[u][1], acc[u][2], acc[u][3], dA[k], dS[k], p.dw.lo_f, p.dw.hi_f);
                    const int moff = moff_of(t0 + u);
                    if constexpr (UX * CX != OWC) { // the column grid overhangs the row: those lanes have no pixel
                        const int ux = ((t0 + u) % NU % NUX) * PSX;
                        if (cx + (ux + wpx) * CX < OWC) *(uint32_t *)(mb + moff) = d;
                    } else {
                        *(uint32_t *)(mb + moff) = d;
                    }
                }
#pragma unroll
                for (int u = 0; u < UB; ++u)
#pragma unroll
                    for (int ty = 0; ty < 3; ++ty) bq[u][ty] = bn[u][ty];
            }
        }
        wg_sync(); // B2: MID complete; everyone is done reading the staged tile
        if constexpr (!DBUF) {
            if (next < nsteps) stage(next, 0); // flies during the pointwise phase
        }

        if constexpr (DWONLY) {
            // ---------------- the depthwise operator alone: MID -> HBM ----------------
            const int npix = gvalid * OPIX;
            int8_t *ob = out + (size_t)step * G * OPIX * C;
            if constexpr (PAIR) { // MID is [pixel][8 bytes]: the output tensor itself (OPIX is even)
                for (int i = tid; i < npix / 2; i += NTHR) st_out_t<true>(ob + i * 16, *(const uint4 *)(mid + i * 16));
            } else {              // planar [16-channel group][pixel][16 bytes] -> [pixel][C]
                for (int e = tid; e < npix * NQ; e += NTHR)
                    st_out_t<true>(ob + (size_t)e * 16, *(const uint4 *)(mid + (e % NQ) * PLANE + (e / NQ) * 16));
            }
            if constexpr (DBUF) cur ^= 1;
            continue;
        }
        // ---------------- pointwise phase: MID -> HBM (dwpw3x3's, reading the planar MID) ----------------
        const int npix = gvalid * OPIX;
        const int nchunks = (npix + CPIX - 1) / CPIX;
        int8_t *obase = out + (size_t)step * G * OPIX * N;
        auto pw_unit = [&](int chunk, auto qlo_c, auto qhi_c) {
            constexpr int QLO = decltype(qlo_c)::value, QHI = decltype(qhi_c)::value;
            v4i B[KS];
            if constexpr (K >= 64) {
                int pix = chunk * 16 + pcol;
                pix = pix < npix ? pix : npix - 1;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) B[ks] = *(const v4i *)(mid + (pg + 4 * ks) * PLANE + pix * 16);
            } else if constexpr (K == 32) {
                int pix = chunk * CPIX + (pg >> 1) * 16 + pcol;
                pix = pix < npix ? pix : npix - 1;
                B[0] = *(const v4i *)(mid + (pg & 1) * PLANE + pix * 16);
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
                    if constexpr (MG == 3) epi_patch_apply(acc, cpr[tt], pg);
                    packed[tt] = requant_pack4<MG, XR4>(acc[0], acc[1], acc[2], acc[3], cA[tt], cS[tt], p.pw.lo_f, p.pw.hi_f);
                }
                if constexpr (XPOSE) {
                    uint8_t *dstp = lds + PATCH_OFF + wave * CBYTES + lpix * N + pg * (NB / 4);
                    if constexpr (TB == 1) *(uint32_t *)dstp = packed[0];
                    else *(uint2 *)dstp = make_uint2(packed[0], packed[1]);
                } else {
                    const int pix = chunk * CPIX + lpix;
                    if (pix < npix)
                        st_out(obase + (size_t)pix * N + blk * NB + pg * 16, make_uint4(packed[0], packed[1], packed[2], packed[3]));
                }
            }
            if constexpr (XPOSE) {
                __builtin_amdgcn_wave_barrier();
                const int cb = chunk * CBYTES, obytes = npix * N;
                constexpr int LO = QLO * (CBYTES / Q), HI = QHI * (CBYTES / Q);
#pragma unroll
                for (int j = 0; j < (HI - LO + 1023) / 1024; ++j) {
                    const int off = LO + (j * 64 + lane) * 16;
                    if (off < HI) {
                        const uint4 v = *(const uint4 *)(lds + PATCH_OFF + wave * CBYTES + off);
                        if (cb + off < obytes) st_out(obase + cb + off, v);
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        };
        using std::integral_constant;
        if constexpr (Q >= 2 && SLOTS % 2 == 0 && K >= 16) { // half-chunk units (see dwpw3x3)
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
    dq.finish(tid);
}

// ------------------------------------------------------------------------
// dwpw_rr -- the same pair with the intermediate tensor kept in REGISTERS (C = 8, 16, 32).
//
// The depthwise MFMA leaves lane (column, g) with 4 consecutive channels of its pixel -- after the
// reference's requantisation one packed dword.  With C <= 32 a pixel is at most two such dwords per lane,
// and those 8 bytes ARE a valid B operand of v_mfma_i32_16x16x32_i8 (lane g supplies K-bytes 8g .. 8g+7) if
// the pointwise weights are laid out for that K order on the host (wrr: K-byte 8g + b  <->  channel 4g + b for
// b < 4, channel 16 + 4g + b - 4 for C = 32, zero weight otherwise).  So a wave runs a unit -- 16 MFMA
// columns, all channels -- from the staged tile to the HBM store without touching LDS again: no MID tensor,
// no second barrier, no transposition patch; the only workgroup barrier left is the one that publishes the
// next staged images, and between two of them the waves drift freely (one wave's MFMAs under another's
// requantisation).  The pointwise rows are permuted so that lane g' ends with N/4 CONSECUTIVE output bytes
// of its pixel (8 or 16: one global store).  C = 8: columns are pixel pairs, the pointwise K covers both
// pixels (each output row multiplies only its own pixel's half) and lane g' owns pixel parity g' >> 1,
// channels 8 (g' & 1) .. +7.
// ------------------------------------------------------------------------
template <int H, int W, int C, int S, int N, int G, int NTHR, int DB, int CG, int CY, int ORD, int ROWPAD, int TS,
          int WPE, int MG, uint32_t XR4>
__global__ __launch_bounds__(NTHR, WPE) void dwpw_rr(const int8_t *__restrict__ in, int8_t *__restrict__ out, DwPwArgs p,
                                                     int batch) {
    epi_enter<MG>();
    constexpr bool PAIR = C == 8;
    // DB: 0 one staging buffer, 1 two.  (A dedicated loader wave -- the only wave that waits for the DMAs, so that the
    // compute waves' `s_waitcnt vmcnt(0)` never waits for their own output stores -- was measured: no gain.)
    constexpr bool DBUF = DB != 0;
    static_assert(C == 8 || C == 16 || C == 32, "register-resident pairs: C <= 32");
    constexpr int OH = (H + S - 1) / S, OW = (W + S - 1) / S;
    constexpr int OWC = PAIR ? OW / 2 : OW;
    constexpr int NQ = PAIR ? 1 : C / 16;
    constexpr int CX = 16 / (CG * CY);
    constexpr int UG = G / CG, UY = OH / CY, UX = OWC / CX;
    constexpr int LP = C < 16 ? 16 : C;
    constexpr int ROWB = W * C, ROW = LP + ROWB + LP + ROWPAD, TILE = (H + 2) * ROW, BUF = G * TILE;
    constexpr int IMG = H * ROWB, ROWCH = ROWB / 16, NROWS = G * H;
    constexpr int NWAVE = NTHR / 64;
    constexpr int NBUF = DBUF ? 2 : 1;
    constexpr int OPIX = OH * OW;
    static_assert(CG * CY * CX == 16 && G % CG == 0 && OH % CY == 0 && OWC % CX == 0, "column grid");
    static_assert(ROWB % 16 == 0 && ROWCH <= 64 && ROW % 16 == 0, "staging geometry");
    static_assert(!PAIR || (S == 1 && OW % 2 == 0), "pair columns");
    static_assert(NQ == 1 || tile_swz<TS>(0xff) < NQ, "swizzle wider than the pixel");
    // the NWAVE waves tile the unit grid
    constexpr int PSY = cgcd(UY, NWAVE), PSX = cgcd(UX, NWAVE / PSY), PSG = cgcd(UG, NWAVE / PSY / PSX);
    static_assert(PSY * PSX * PSG == NWAVE, "the unit grid does not divide over the waves");
    constexpr int NUG = UG / PSG, NUY = UY / PSY, NUX = UX / PSX, NU = NUG * NUY * NUX;
    constexpr int T_UG = CG * TILE, T_UY = CY * S * ROW, T_UX = PAIR ? CX * 16 : CX * S * C;
    constexpr int O_UG = CG * OPIX * N, O_UY = CY * OW * N, O_UX = (PAIR ? 2 : 1) * CX * N; // output bytes per unit step
    // pointwise: NT 16-row MFMAs per unit; a lane ends with LB consecutive output bytes
    constexpr int NT = (PAIR ? 2 * N : N) / 16, LB = 4 * NT;
    static_assert(LB == 8 || LB == 16, "one 8- or 16-byte store per lane");

    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    DynSteps dq;
    dq.init(lds + NBUF * BUF + 512, p.dw.queue, tid, p.dw.qcfg); // (16 bytes behind the staging buffers' slack)

    for (int i = tid; i < (NBUF * BUF + 512) / 16; i += NTHR)
        ((uint4 *)lds)[i] = make_uint4(p.dw.izp4, p.dw.izp4, p.dw.izp4, p.dw.izp4);

    const int col = lane & 15, g = lane >> 4;
    int cg, cy, cx;
    {
        constexpr int D0 = (ORD == 0 || ORD == 1) ? CG : (ORD == 2 || ORD == 3) ? CY : CX;
        constexpr int D1 = (ORD == 2 || ORD == 4) ? CG : (ORD == 0 || ORD == 5) ? CY : CX;
        const int i0 = col % D0, i1 = (col / D0) % D1, i2 = col / (D0 * D1);
        cg = (ORD == 0 || ORD == 1) ? i0 : (ORD == 2 || ORD == 4) ? i1 : i2;
        cy = (ORD == 2 || ORD == 3) ? i0 : (ORD == 0 || ORD == 5) ? i1 : i2;
        cx = (ORD == 4 || ORD == 5) ? i0 : (ORD == 1 || ORD == 3) ? i1 : i2;
    }
    const int wpy = wave % PSY, wpx = (wave / PSY) % PSX, wpg = wave / (PSY * PSX);
    const int wave_t = wpg * T_UG + wpy * T_UY + wpx * T_UX;
    int tbase[NQ];
    if constexpr (PAIR) {
        tbase[0] = cg * TILE + cy * ROW + LP + (2 * cx - 2) * 8 + g * 16 + wave_t;
    } else {
        const int xl = cx * S + g - 1;
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            tbase[q] = cg * TILE + cy * S * ROW + LP + xl * C + 16 * (q ^ tile_swz<TS>(xl)) + wave_t;
    }
    // output: pixel of this lane's column (C = 8: its parity half), first of its LB channels
    const int opar = PAIR ? (g >> 1) : 0;
    const int n0 = PAIR ? 8 * (g & 1) : (N / 4) * g;
    const int obase_lane = (cg * OPIX + cy * OW + (PAIR ? 2 * cx + opar : cx)) * N + n0 +
                           wpg * O_UG + wpy * O_UY + wpx * O_UX;

    v4i Adw[NQ][3];
    float4 dA[NQ], dS[NQ];
    int4 dK[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
#pragma unroll
        for (int ty = 0; ty < 3; ++ty) Adw[q][ty] = ((const v4i *)p.dw.wmm)[(q * 3 + ty) * 64 + lane];
        const int ch4 = PAIR ? (g & 1) : 4 * q + g;
        dA[q] = ((const float4 *)p.dw.A)[ch4];
        dS[q] = ((const float4 *)p.dw.S)[ch4];
        dK[q] = magic4<MG>(((const int4 *)p.dw.Kc)[ch4]);
    }
    long Apw[NT];
    float4 cA[NT], cS[NT];
    int4 cK[NT];
#pragma unroll
    for (int m = 0; m < NT; ++m) {
        Apw[m] = ((const long *)p.pw.wrr)[m * 64 + lane];
        cA[m] = *(const float4 *)(p.pw.A + n0 + 4 * m);
        cS[m] = *(const float4 *)(p.pw.S + n0 + 4 * m);
        cK[m] = magic4<MG>(*(const int4 *)(p.pw.Kc + n0 + 4 * m));
    }
    wg_sync(); // halo fill complete before any DMA lands

    auto stage = [&](int st, int buf) {
        const int src_lane = NQ > 1 ? (lane ^ tile_swz<TS>(lane / (NQ > 1 ? NQ : 1))) : lane;
#pragma unroll
        for (int k = 0; k < (NROWS + NWAVE - 1) / NWAVE; ++k) {
            const int r = k * NWAVE + wave;
            const int gi = r / H, y = r % H;
            if (r < NROWS && st * G + gi < batch && lane < ROWCH)
                dma16(in + ((size_t)(st * G + gi) * IMG + y * ROWB + src_lane * 16),
                      lds + buf * BUF + gi * TILE + (y + 1) * ROW + LP);
        }
    };

    const int nsteps = (batch + G - 1) / G;
    int cur = 0;
    if (dq.step < nsteps) stage(dq.step, 0);

    for (; dq.step < nsteps; dq.advance(tid)) {
        const int step = dq.step;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wg_sync(); // the staged tile is complete; every wave is done with the buffer the next DMA overwrites
        dq.top(tid);
        const int next = dq.nxt;
        if constexpr (DBUF) {
            if (next < nsteps) stage(next, cur ^ 1);
        }
        const int gvalid = min(G, batch - step * G);
        const uint8_t *tb = lds + cur * BUF;
        int8_t *ob = out + (size_t)step * G * OPIX * N + obase_lane;

        // units in flight per wave.  (r02 same-session A/B of 1 / 2 / 3 per kernel: no effect anywhere except the stride-2
        // pair with two channel groups, 24x24x32: 3 units 0.352 -> 0.341 ms.)
        constexpr int UB = (S == 2 && NQ > 1 && NU % 3 == 0) ? 3 : (WPE >= 4 || NQ > 1) ? 1 : (NU % 2 == 0 ? 2 : (NU % 3 == 0 ? 3 : 1));
        auto coords = [](int iu, int &ug, int &uy, int &ux) constexpr {
            ug = (iu / (NUY * NUX)) * PSG, uy = ((iu / NUX) % NUY) * PSY, ux = (iu % NUX) * PSX;
        };
        auto toff_of = [&](int iu) constexpr {
            int ug = 0, uy = 0, ux = 0;
            coords(iu, ug, uy, ux);
            return ug * T_UG + uy * T_UY + ux * T_UX;
        };
        v4i bq[UB][NQ][3], bn[UB][NQ][3];
#pragma unroll
        for (int u = 0; u < UB; ++u)
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int ty = 0; ty < 3; ++ty) bq[u][q][ty] = *(const v4i *)(tb + tbase[q] + toff_of(u) + ty * ROW);
#pragma unroll
        for (int t0 = 0; t0 < NU; t0 += UB) {
            v4i acc[UB][NQ];
#pragma unroll
            for (int u = 0; u < UB; ++u)
#pragma unroll
                for (int q = 0; q < NQ; ++q) acc[u][q] = v4i{dK[q].x, dK[q].y, dK[q].z, dK[q].w};
#pragma unroll
            for (int ty = 0; ty < 3; ++ty)
#pragma unroll
                for (int u = 0; u < UB; ++u)
#pragma unroll
                    for (int q = 0; q < NQ; ++q)
                        acc[u][q] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Adw[q][ty], bq[u][q][ty], acc[u][q], 0, 0, 0);
            if (t0 + UB < NU) {
#pragma unroll
                for (int u = 0; u < UB; ++u)
#pragma unroll
                    for (int q = 0; q < NQ; ++q)
#pragma unroll
                        for (int ty = 0; ty < 3; ++ty)
                            bn[u][q][ty] = *(const v4i *)(tb + tbase[q] + toff_of(t0 + UB + u) + ty * ROW);
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                // depthwise requantisation: the intermediate int8 tensor, 4 (C = 32: 8) bytes per lane
                uint32_t d[2] = {0u, 0u};
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    // this dword is an MFMA operand a few instructions later: the SDWA byte writes must not be the
                    // instruction right before their reader (k_common.hpp: the packs end with an independent slot)
                    d[q] = requant_pack4<MG, XR4>(acc[u][q][0], acc[u][q][1], acc[u][q][2], acc[u][q][3], dA[q], dS[q],
                                                  p.dw.lo_f, p.dw.hi_f);
                // K-bytes 8g .. 8g+7 of the pointwise contraction (bytes 4..7 meet zero weights when C < 32)
                const long bop = (long)(((unsigned long)d[1] << 32) | (unsigned long)d[0]);
                uint32_t packed[NT];
#pragma unroll
                for (int m = 0; m < NT; ++m) {
                    v4i pa = {cK[m].x, cK[m].y, cK[m].z, cK[m].w};
                    pa = __builtin_amdgcn_mfma_i32_16x16x32_i8(Apw[m], bop, pa, 0, 0, 0);
                    packed[m] = requant_pack4<MG, XR4>(pa[0], pa[1], pa[2], pa[3], cA[m], cS[m], p.pw.lo_f, p.pw.hi_f);
                }
                int ug = 0, uy = 0, ux = 0;
                coords(t0 + u, ug, uy, ux);
                const int ooff = ug * O_UG + uy * O_UY + ux * O_UX;
                if (G == 1 || cg + (ug + wpg) * CG < gvalid) // a ragged last step stages fewer than G images (G = 1: no test, no branch between the units)
                {
                    if constexpr (LB == 8) st_out(ob + ooff, make_uint2(packed[0], packed[1]));
                    else st_out(ob + ooff, make_uint4(packed[0], packed[1], packed[2], packed[3]));
                }
            }
#pragma unroll
            for (int u = 0; u < UB; ++u)
#pragma unroll
                for (int q = 0; q < NQ; ++q)
#pragma unroll
                    for (int ty = 0; ty < 3; ++ty) bq[u][q][ty] = bn[u][q][ty];
        }
        if constexpr (DBUF) {
            cur ^= 1;
        } else {
            // single staging buffer: everyone must be done reading it before the next DMA lands
            wg_sync();
            if (next < nsteps) stage(next, 0);
        }
    }
    dq.finish(tid);
}

// ---- launchers ----
template <int H, int W, int C, int S, int N, int G, int NTHR, int DB, int CG, int CY, int ORD, int ROWPAD, int TS, int WPE,
          int MG, uint32_t XR4>
static void launch_dwpw_mm_t(const int8_t *in, int8_t *out, const DwPwArgs &a, int batch, hipStream_t s) {
    constexpr int lds = dwmm_lds_bytes(H, W, C, S, N, G, NTHR, DB != 0, ROWPAD);
    static_assert(lds <= 163840, "fused tile does not fit the LDS");
    static LaunchState st;
    const int per_cu = prepared(st, dwpw_mm<H, W, C, S, N, G, NTHR, (DB != 0), CG, CY, ORD, ROWPAD, TS, WPE, MG, XR4, false>, NTHR, lds);
    const int nsteps = (batch + G - 1) / G;
    const int grid = nsteps < 256 * per_cu ? nsteps : 256 * per_cu;
    constexpr int OPIX = ((H + S - 1) / S) * ((W + S - 1) / S);
    DwPwArgs b = a;
    b.dw.qcfg = dq_config(nsteps, grid, dq_est_us((double)batch * (H * W * C + OPIX * N), (double)batch * OPIX * (C + N)));
    b.dw.queue = dq_slot(b.dw.queue, b.dw.qlaunch);
    hipLaunchKernelGGL((dwpw_mm<H, W, C, S, N, G, NTHR, (DB != 0), CG, CY, ORD, ROWPAD, TS, WPE, MG, XR4, false>), dim3(grid), dim3(NTHR),
                       lds, s, in, out, b, batch);
}
// the depthwise operator alone (DWONLY instance of the same shape)
template <int H, int W, int C, int S, int N, int G, int NTHR, int DB, int CG, int CY, int ORD, int ROWPAD, int TS, int WPE,
          int MG, uint32_t XR4>
static void launch_dw_mm_t(const int8_t *in, int8_t *out, const DwPwArgs &a, int batch, hipStream_t s) {
    constexpr int lds = dwmm_lds_bytes(H, W, C, S, N, G, NTHR, DB != 0, ROWPAD);
    static LaunchState st;
    const int per_cu = prepared(st, dwpw_mm<H, W, C, S, N, G, NTHR, (DB != 0), CG, CY, ORD, ROWPAD, TS, WPE, MG, XR4, true>, NTHR, lds);
    const int nsteps = (batch + G - 1) / G;
    const int grid = nsteps < 256 * per_cu ? nsteps : 256 * per_cu;
    constexpr int OPIX = ((H + S - 1) / S) * ((W + S - 1) / S);
    DwPwArgs b = a;
    b.dw.qcfg = dq_config(nsteps, grid, dq_est_us((double)batch * (H * W * C + OPIX * C), (double)batch * OPIX * C));
    b.dw.queue = dq_slot(b.dw.queue, b.dw.qlaunch);
    hipLaunchKernelGGL((dwpw_mm<H, W, C, S, N, G, NTHR, (DB != 0), CG, CY, ORD, ROWPAD, TS, WPE, MG, XR4, true>), dim3(grid), dim3(NTHR),
                       lds, s, in, out, b, batch);
}
// Measured per shape against dw3x3_nhwc (v_dot4 taps, no MID round trip): the matrix-pipe form wins on the three
// large early layers (48x48x8: 0.54 -> 0.48 ms, 48x48x16 s2: 0.60 -> 0.57, 24x24x32: 0.49 -> 0.47) and loses on
// the small late ones (12x12x64: 0.26 -> 0.36, 6x6x128: 0.135 -> 0.185: two barriers per small step), so only
// those three are instantiated.
constexpr bool dw_mm_layerwise(int h, int s) { return h >= 48 || (h >= 24 && s == 1); }
const char *dw_mm_name(int H, int W, int C, int S) {
#define MF_DWMM(h, w, c, s, n, g, t, d, cg, cy, ord, rp, ts, wpe) \
    if (dw_mm_layerwise(h, s) && H == h && W == w && C == c && S == s) return "dw3x3_mm<" #h "," #w "," #c "," #s "," #g "," #t ">";
    MF_DWMM_SHAPES(MF_DWMM)
#undef MF_DWMM
    return nullptr;
}
bool launch_dw_mm(int H, int W, int C, int S, const int8_t *in, int8_t *out, const DwFastArgs &dw, int batch, hipStream_t s) {
    if (!dw.wmm) return false;
    DwPwArgs a{};
    a.dw = dw;
#define MF_DWMM(h, w, c, st, n, g, t, d, cg, cy, ord, rp, ts, wpe)                                                  \
    if constexpr (dw_mm_layerwise(h, st)) {                                                                        \
        if (H == h && W == w && C == c && S == st) {                                                               \
            MF_DISPATCH5(dw.magic, dw.xr, launch_dw_mm_t, (in, out, a, batch, s), h, w, c, st, n, g, t, d, cg, cy, ord, rp, ts, wpe) \
            return true;                                                                                           \
        }                                                                                                          \
    }
    MF_DWMM_SHAPES(MF_DWMM)
#undef MF_DWMM
    return false;
}
const char *dwpw_mm_name(int H, int W, int C, int S, int N) {
#define MF_DWMM(h, w, c, s, n, g, t, d, cg, cy, ord, rp, ts, wpe) \
    if (H == h && W == w && C == c && S == s && N == n) return "dwpw_mm<" #h "," #w "," #c "," #s "," #n "," #g "," #t "," #d ">";
    MF_DWMM_SHAPES(MF_DWMM)
#undef MF_DWMM
    return nullptr;
}
bool launch_dwpw_mm(int H, int W, int C, int S, int N, const int8_t *in, int8_t *out, const DwPwArgs &a, int batch,
                    hipStream_t s) {
    if (!a.dw.wmm) return false;
    const int alt = switches().dwmm_alt;
    if (alt >= 0) { // tuning candidates, see MF_DWMM_ALT_SHAPES
        int idx = 0;
        (void)idx;
#define MF_DWMM(h, w, c, st, n, g, t, d, cg, cy, ord, rp, ts, wpe)                                                        \
    if (idx++ == alt && H == h && W == w && C == c && S == st && N == n) {                                           \
        MF_DISPATCH5(a.dw.magic < a.pw.magic ? a.dw.magic : a.pw.magic, a.pw.xr, launch_dwpw_mm_t, (in, out, a, batch, s), h, w, c, st, n, g, t, d, \
                     cg, cy, ord, rp, ts, wpe)                                                                            \
        return true;                                                                                                 \
    }
        MF_DWMM_ALT_SHAPES(MF_DWMM)
#undef MF_DWMM
    }
#define MF_DWMM(h, w, c, st, n, g, t, d, cg, cy, ord, rp, ts, wpe)                                                        \
    if (H == h && W == w && C == c && S == st && N == n) {                                                           \
        MF_DISPATCH5(a.dw.magic < a.pw.magic ? a.dw.magic : a.pw.magic, a.pw.xr, launch_dwpw_mm_t, (in, out, a, batch, s), h, w, c, st, n, g, t, d, \
                     cg, cy, ord, rp, ts, wpe)                                                                            \
        return true;                                                                                                 \
    }
    MF_DWMM_SHAPES(MF_DWMM)
#undef MF_DWMM
    return false;
}

template <int H, int W, int C, int S, int N, int G, int NTHR, int DB, int CG, int CY, int ORD, int ROWPAD, int TS, int WPE,
          int MG, uint32_t XR4>
static void launch_dwpw_rr_t(const int8_t *in, int8_t *out, const DwPwArgs &a, int batch, hipStream_t s) {
    constexpr int LP = C < 16 ? 16 : C;
    constexpr int lds = (DB ? 2 : 1) * G * (H + 2) * (LP + W * C + LP + ROWPAD) + 512 + 16; // + step queue
    static_assert(lds <= 163840, "staged tiles do not fit the LDS");
    static LaunchState st;
    const int per_cu = prepared(st, dwpw_rr<H, W, C, S, N, G, NTHR, DB, CG, CY, ORD, ROWPAD, TS, WPE, MG, XR4>, NTHR, lds);
    const int nsteps = (batch + G - 1) / G;
    const int grid = nsteps < 256 * per_cu ? nsteps : 256 * per_cu;
    constexpr int OPIX = ((H + S - 1) / S) * ((W + S - 1) / S);
    DwPwArgs b = a;
    b.dw.qcfg = dq_config(nsteps, grid, dq_est_us((double)batch * (H * W * C + OPIX * N), (double)batch * OPIX * (C + N)));
    b.dw.queue = dq_slot(b.dw.queue, b.dw.qlaunch);
    hipLaunchKernelGGL((dwpw_rr<H, W, C, S, N, G, NTHR, DB, CG, CY, ORD, ROWPAD, TS, WPE, MG, XR4>), dim3(grid), dim3(NTHR),
                       lds, s, in, out, b, batch);
}
const char *dwpw_rr_name(int H, int W, int C, int S, int N) {
#define MF_DWRR(h, w, c, s, n, g, t, d, cg, cy, ord, rp, ts, wpe) \
    if (H == h && W == w && C == c && S == s && N == n) return "dwpw_rr<" #h "," #w "," #c "," #s "," #n "," #g "," #t "," #d ">";
    MF_DWRR_SHAPES(MF_DWRR)
#undef MF_DWRR
    return nullptr;
}
bool launch_dwpw_rr(int H, int W, int C, int S, int N, const int8_t *in, int8_t *out, const DwPwArgs &a, int batch,
                    hipStream_t s) {
    if (!a.dw.wmm || !a.pw.wrr) return false;
    if (a.dw.patch || a.pw.patch) return false; // (patched accumulators: dwpw_mm's job)
    const int alt = switches().dwrr_alt;
    if (alt >= 0) { // tuning candidates, see MF_DWRR_ALT_SHAPES
        int idx = 0;
        (void)idx;
#define MF_DWRR(h, w, c, st, n, g, t, d, cg, cy, ord, rp, ts, wpe)                                                   \
    if (idx++ == alt && H == h && W == w && C == c && S == st && N == n) {                                           \
        MF_DISPATCH5(a.dw.magic < a.pw.magic ? a.dw.magic : a.pw.magic, a.pw.xr, launch_dwpw_rr_t, (in, out, a, batch, s), h, w, c, st, n, g, t, d, \
                     cg, cy, ord, rp, ts, wpe)                                                                       \
        return true;                                                                                                 \
    }
        MF_DWRR_ALT_SHAPES(MF_DWRR)
#undef MF_DWRR
    }
#define MF_DWRR(h, w, c, st, n, g, t, d, cg, cy, ord, rp, ts, wpe)                                                   \
    if (H == h && W == w && C == c && S == st && N == n) {                                                           \
        MF_DISPATCH5(a.dw.magic < a.pw.magic ? a.dw.magic : a.pw.magic, a.pw.xr, launch_dwpw_rr_t, (in, out, a, batch, s), h, w, c, st, n, g, t, d, \
                     cg, cy, ord, rp, ts, wpe)                                                                       \
        return true;                                                                                                 \
    }
    MF_DWRR_SHAPES(MF_DWRR)
#undef MF_DWRR
    return false;
}

} // namespace k
} // namespace mf
