// k_fused.hip -- fused operator groups: DepthwiseConv2D 3x3 + Conv2D 1x1 pairs and the network tail.
//
// (src/ops/depthwise_conv_2d.rs:28-105, src/ops/conv_2d.rs:28-108, src/ops/average_pool_2d.rs:29-66,
// src/ops/softmax.rs:15-27)
// Arithmetic contract, shared device helpers and launch plumbing: k_common.hpp.
#include "k_common.hpp"
#include "k_dwtask.hpp"
#include "k_tail.hpp"

namespace mf {
namespace k {

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
template <int H, int W, int C, int S, int N, int G, int NTHR, bool DBUF, int MG, uint32_t XR4>
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
    constexpr int DQ_OFF = PATCH_OFF + (XPOSE ? NWAVE * CBYTES : 0); // the step queue's two ints

    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    DynSteps dq;
    dq.init(lds + DQ_OFF, p.dw.queue, tid, p.dw.qcfg);

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
    int cur = 0;
    if (dq.step < nsteps) stage(dq.step, 0);

    for (; dq.step < nsteps; dq.advance(tid)) {
        const int step = dq.step;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads(); // B1: staged tile complete; previous pointwise phase done with MID
        dq.top(tid);
        const int next = dq.nxt;
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
                        dp[0] = requant_pack4<MG, XR4>(o0[j][0], o0[j][1], o0[j][2], o0[j][3], dA, dS, p.dw.lo_f, p.dw.hi_f);
                        if (ox0 + 1 < OW)
                            dp[C4] = requant_pack4<MG, XR4>(o1[j][0], o1[j][1], o1[j][2], o1[j][3], dA, dS, p.dw.lo_f, p.dw.hi_f);
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
                // sub-blocks [QLO, QHI) are the bytes [QLO, QHI) * CBYTES / Q of the chunk image
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
    dq.finish(tid);
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
    const size_t img_elems = (size_t)p.H * p.W * p.C;
    for (size_t b = wave; b < batch; b += nwaves) tail_one<N>(in + b * img_elems, out + b * N, p, lane);
}

// ---- launchers ----
template <int H, int W, int C, int S, int N, int G, int NTHR, int DB, int MG, uint32_t XR4>
static void launch_dwpw_t(const int8_t *in, int8_t *out, const DwPwArgs &a, int batch, hipStream_t s) {
    constexpr int LP = C < 16 ? 16 : C;
    constexpr int OH = (H + S - 1) / S, OW = (W + S - 1) / S;
    constexpr int BUF = G * (H + 2) * (LP + W * C + LP);
    constexpr int NB = N < 64 ? N : 64, CPIX = C < 64 ? 1024 / C : 16;
    constexpr int patch = (NB / 16 < 4) ? (NTHR / 64) * CPIX * N : 0;
    constexpr int lds = (DB ? 2 : 1) * BUF + 256 + G * OH * OW * C + 64 + patch + 16; // + step queue
    static_assert(lds <= 163840, "fused tile does not fit the LDS");
    static LaunchState st;
    const int per_cu = prepared(st, dwpw3x3<H, W, C, S, N, G, NTHR, (DB != 0), MG, XR4>, NTHR, lds);
    const int nsteps = (batch + G - 1) / G;
    const int grid = nsteps < 256 * per_cu ? nsteps : 256 * per_cu;
    DwPwArgs b = a;
    b.dw.qcfg = dq_config(nsteps, grid, dq_est_us((double)batch * (H * W * C + OH * OW * N), (double)batch * OH * OW * (C + N)));
    b.dw.queue = dq_slot(b.dw.queue);
    hipLaunchKernelGGL((dwpw3x3<H, W, C, S, N, G, NTHR, (DB != 0), MG, XR4>), dim3(grid), dim3(NTHR), lds, s, in, out, b, batch);
}
int dwpw_impl();
const char *dwpw_name(int H, int W, int C, int S, int N) {
    if (dwpw_impl() == 2 && dwpw_rr_name(H, W, C, S, N)) return dwpw_rr_name(H, W, C, S, N);
    if (dwpw_impl() >= 1 && dwpw_mm_name(H, W, C, S, N)) return dwpw_mm_name(H, W, C, S, N);
#define MF_DWPW(h, w, c, s, n, g, t, d) \
    if (H == h && W == w && C == c && S == s && N == n) return "dwpw3x3<" #h "," #w "," #c "," #s "," #n "," #g "," #t "," #d ">";
    MF_DWPW_SHAPES(MF_DWPW)
#undef MF_DWPW
    return nullptr;
}
// MF_DWPW_IMPL=valu keeps the depthwise taps on the VALU (dwpw3x3, r01); =mm uses the matrix-pipe form with the
// intermediate tensor in LDS for every pair (dwpw_mm); default (2): dwpw_rr (intermediate in registers) where a
// shape has one, else dwpw_mm, else dwpw3x3
int dwpw_impl() {
    static const int impl = [] {
        const char *e = getenv("MF_DWPW_IMPL");
        return !e ? 2 : (e[0] == 'v' ? 0 : (e[0] == 'm' ? 1 : 2));
    }();
    return impl;
}
bool launch_dwpw(int H, int W, int C, int S, int N, const int8_t *in, int8_t *out, const DwPwArgs &a,
                 int batch, hipStream_t s) {
    if (dwpw_impl() == 2 && launch_dwpw_rr(H, W, C, S, N, in, out, a, batch, s)) return true;
    if (dwpw_impl() >= 1 && launch_dwpw_mm(H, W, C, S, N, in, out, a, batch, s)) return true;
    static const int alt = [] { const char *e = getenv("MF_DWPW_ALT"); return e ? atoi(e) : -1; }();
    if (alt >= 0) {
        int idx = 0;
        (void)idx;
#define MF_DWPW(h, w, c, st, n, g, t, d)                                        \
    if (idx++ == alt && H == h && W == w && C == c && S == st && N == n) {      \
        MF_DISPATCH4(a.dw.magic < a.pw.magic ? a.dw.magic : a.pw.magic, a.pw.xr, launch_dwpw_t, (in, out, a, batch, s), h, w, c, st, n, g, t, d) \
        return true;                                                            \
    }
        MF_DWPW_ALT_SHAPES(MF_DWPW)
#undef MF_DWPW
    }
#define MF_DWPW(h, w, c, st, n, g, t, d)                         \
    if (H == h && W == w && C == c && S == st && N == n) {       \
        MF_DISPATCH4(a.dw.magic < a.pw.magic ? a.dw.magic : a.pw.magic, a.pw.xr, launch_dwpw_t, (in, out, a, batch, s), h, w, c, st, n, g, t, d) \
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

} // namespace k
} // namespace mf
