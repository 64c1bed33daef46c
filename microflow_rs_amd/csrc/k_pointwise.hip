// k_pointwise.hip -- Conv2D 1x1 as an int8 MFMA GEMM over the batch's pixel matrix (src/ops/conv_2d.rs:28-108).
//
// Arithmetic contract, shared device helpers and launch plumbing: k_common.hpp.
#include "k_common.hpp"

namespace mf {
namespace k {

// ------------------------------------------------------------------------
// FAST PATH 3 -- Conv2D 1x1 stride 1 (pointwise) as an int8 MFMA GEMM.
// (src/ops/conv_2d.rs:28-108 with KH = KW = 1; person_detect ops 2,4,...,26)
//
//   D[out channel][pixel] = sum_k Wt[out channel][k] * X[pixel][k]
//   v_mfma_i32_16x16x64_i8:  A = weights (rows = 16 out channels), B = pixels
//   (cols = 16 pixels), so that a lane ends up holding 4 CONSECUTIVE out channels of
//   one pixel per 16x16 tile and can store them packed.
//
// In NHWC with the batch outermost, the activations of the whole batch ARE the
// row-major [pixels][K] matrix -- no im2col, no LDS staging of activations: lane
// (p = lane&15, g = lane>>4) loads its 16 bytes of operand B straight from HBM in
// MFMA layout, and every wave-level load instruction covers whole contiguous 1 KiB.
// K < 64 (early layers) would waste the 64-deep MFMA k-span, so 64/K pixel groups
// share one B register and the host pre-builds Q = 64/K zero-padded copies A_q of the
// weights, each selecting one group's k-bytes (block-diagonal trick): loads stay
// 16 B/lane fully coalesced at every K.  The weight rows are permuted on the host so
// that tile tt row 4g+j is channel base + g*(NB/4) + 4*tt + j: after TB tiles a lane
// holds NB/4 consecutive output bytes -> one 4/8/16-byte store.
// N > 64 is split over the waves of the workgroup (NSPLIT = N/64), which all read the
// same pixels (L1/L2 hits).  HBM-bound: MFMA work is ~1/8 of the memory time.
// ------------------------------------------------------------------------
#ifndef MF_PW_NT
#define MF_PW_NT -1
#endif
#ifndef MF_PW_U_LO
#define MF_PW_U_LO 2
#define MF_PW_U_MID 4
#define MF_PW_U_HI 2
#endif
// chunks each wave keeps in flight (loads of the next U issued before the first use)
constexpr int pw_chunks_in_flight(int K) { return K >= 256 ? MF_PW_U_HI : (K >= 64 ? MF_PW_U_MID : MF_PW_U_LO); }
template <int K, int N, int MG, uint32_t XR4>
__global__ __launch_bounds__(256) void pw_mfma(const int8_t *__restrict__ in,
                                               int8_t *__restrict__ out, PwArgs p,
                                               long long npix) {
    epi_enter<MG>();
    constexpr int NB = N < 64 ? N : 64;        // channels per wave block
    constexpr int TB = NB / 16;                // 16-channel MFMA tiles per block
    constexpr int NSPLIT = N / NB;             // waves sharing one pixel chunk
    constexpr int KS = K < 64 ? 1 : K / 64;    // 64-deep k steps
    constexpr int Q = K < 64 ? 64 / K : 1;     // pixel groups per B register
    constexpr int CPIX = (K < 64) ? (1024 / K) : 16; // pixels per chunk
    constexpr int SLOTS = 4 / NSPLIT;          // pixel chunks processed concurrently per WG
    // chunks per loop iteration: their loads are all issued before the first use, so a wave
    // keeps U*KS KiB in flight (one 16-pixel chunk per iteration left HBM latency exposed)
    constexpr int U = pw_chunks_in_flight(K);
    // narrow outputs (N < 64) go through a per-wave LDS patch so that every global store is
    // 16 bytes per lane and a wave writes whole contiguous KiB
    constexpr bool XPOSE = TB < 4;
    constexpr bool NT = MF_PW_NT < 0 ? N <= K : MF_PW_NT != 0; // non-temporal stores (k_common.hpp st_out_t): not for N = 2K
    constexpr int CBYTES = CPIX * N;           // output bytes per chunk (XPOSE: 1 or 2 KiB)
    static_assert(N % 16 == 0 && (K == 8 || K % 16 == 0), "pw_mfma shape");
    static_assert(!XPOSE || (NSPLIT == 1 && CBYTES % 1024 == 0), "transposed store geometry");

    __shared__ __attribute__((aligned(16))) uint8_t patch[XPOSE ? 4 * CBYTES : 16];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pcol = lane & 15, g = lane >> 4;
    const int blk = wave % NSPLIT;             // which 64-channel block this wave owns
    const int slot = wave / NSPLIT;

    // operand A (weights), pre-arranged by the host: [blk][q][tt][ks][lane] x 16 bytes
    v4i Aw[Q][TB][KS];
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
        for (int tt = 0; tt < TB; ++tt)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                Aw[q][tt][ks] = ((const v4i *)p.wprep)[((((size_t)blk * Q + q) * TB + tt) * KS + ks) * 64 + lane];
    // epilogue constants of this lane's channels: block base + g*(NB/4) + 4*tt + j
    float4 cA[TB], cS[TB];
    int4 cK[TB];
#pragma unroll
    for (int tt = 0; tt < TB; ++tt) {
        const int ch = blk * NB + g * (NB / 4) + 4 * tt;
        cA[tt] = *(const float4 *)(p.A + ch);
        cS[tt] = *(const float4 *)(p.S + ch);
        cK[tt] = magic4<MG>(*(const int4 *)(p.Kc + ch));
    }

    const long long nchunks = (npix + CPIX - 1) / CPIX;
    const long long stride = (long long)gridDim.x * SLOTS * U;
    long long chunk0 = ((long long)blockIdx.x * SLOTS + slot) * U;

    // loads are never predicated (clamped instead): see the depthwise staging notes
    auto loadB = [&](long long ch, v4i (&b)[KS]) {
        if constexpr (K >= 64) {
            long long pix = ch * 16 + pcol;
            pix = pix < npix ? pix : npix - 1;
            const int8_t *src = in + pix * K + g * 16;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) b[ks] = *(const v4i *)(src + ks * 64);
        } else if constexpr (K == 32) {
            long long pix = ch * CPIX + (g >> 1) * 16 + pcol;
            pix = pix < npix ? pix : npix - 1;
            b[0] = *(const v4i *)(in + pix * 32 + (g & 1) * 16);
        } else if constexpr (K == 16) {
            long long pix = ch * CPIX + g * 16 + pcol;
            pix = pix < npix ? pix : npix - 1;
            b[0] = *(const v4i *)(in + pix * 16);
        } else { // K == 8: 16 bytes = 2 pixels; npix is even (routing precondition)
            long long pix = ch * CPIX + 2 * (g * 16 + pcol);
            pix = pix + 1 < npix ? pix : npix - 2;
            b[0] = *(const v4i *)(in + pix * 8);
        }
    };

    v4i B[U][KS], Bn[U][KS];
#pragma unroll
    for (int u = 0; u < U; ++u) loadB(min(chunk0 + u, nchunks - 1), B[u]);

    for (; chunk0 < nchunks; chunk0 += stride) {
        const long long nxt = chunk0 + stride;
#pragma unroll
        for (int u = 0; u < U; ++u) loadB(min(nxt + u, nchunks - 1), Bn[u]); // harmless re-read at the tail
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long chunk = chunk0 + u;
            if (chunk < nchunks) { // wave-uniform
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    // pixel (within the chunk) this lane's MFMA column belongs to for sub-group q
                    int lpix;
                    if constexpr (K >= 64) lpix = pcol;
                    else if constexpr (K == 8) lpix = 2 * ((q >> 1) * 16 + pcol) + (q & 1);
                    else lpix = q * 16 + pcol;
                    uint32_t packed[TB];
#pragma unroll
                    for (int tt = 0; tt < TB; ++tt) {
                        // the accumulator starts at Kc (the folded zero-point terms): the MFMA's C
                        // operand does the addition for free
                        v4i acc = {cK[tt].x, cK[tt].y, cK[tt].z, cK[tt].w};
#pragma unroll
                        for (int ks = 0; ks < KS; ++ks)
                            acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(Aw[q][tt][ks], B[u][ks], acc, 0, 0, 0);
                        packed[tt] = requant_pack4<MG, XR4>(acc[0], acc[1], acc[2], acc[3], cA[tt], cS[tt], p.lo_f, p.hi_f);
                    }
                    if constexpr (XPOSE) {
                        uint8_t *dstp = patch + wave * CBYTES + lpix * N + g * (NB / 4);
                        if constexpr (TB == 1) *(uint32_t *)dstp = packed[0];
                        else *(uint2 *)dstp = make_uint2(packed[0], packed[1]);
                    } else {
                        const long long pix = chunk * CPIX + lpix;
                        if (pix < npix)
                            st_out_t<NT>(out + pix * N + blk * NB + g * 16, make_uint4(packed[0], packed[1], packed[2], packed[3]));
                    }
                }
                if constexpr (XPOSE) {
                    // same wave wrote the patch; LDS ops of a wave complete in order
                    __builtin_amdgcn_wave_barrier();
                    const long long obase = chunk * (long long)CBYTES;
                    const long long obytes = npix * N;
#pragma unroll
                    for (int j = 0; j < CBYTES / 1024; ++j) {
                        const int off = (j * 64 + lane) * 16;
                        const uint4 v = *(const uint4 *)(patch + wave * CBYTES + off);
                        if (obase + off < obytes) st_out_t<NT>(out + obase + off, v);
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) B[u][ks] = Bn[u][ks];
    }
}

// ---- launchers ----
template <int K, int N, int MG, uint32_t XR4>
static void launch_pw_t(const int8_t *in, int8_t *out, const PwArgs &a, long long npix, int grid, hipStream_t s) {
    hipLaunchKernelGGL((pw_mfma<K, N, MG, XR4>), dim3(grid), dim3(256), 0, s, in, out, a, npix);
}
// workgroups of the grid-strided pointwise kernel (r01 sweep, MF_PW_GRID overrides): the wide early
// layers (K < 64, most pixels) like many short-lived workgroups, the deep late ones few
static long long pw_grid_cap(int K, int N) {
    const long long forced = switches().pw_grid;
    if (forced > 0) return forced;
    // r02 sweep (256 x {2 .. 64}, same session): K >= 128: x2 beats x4 by 4 - 8 %; K < 64 with N <= 32: x64 beats x32 by
    // 2 - 12 %; the rest stays
    return K < 64 ? (N <= 32 ? 256LL * 64 : 256LL * 32) : (K >= 128 ? 256LL * 2 : 256LL * 8);
}
const char *pw_name(int K, int N) {
#define MF_PW(k, n) \
    if (K == k && N == n) return "pw_mfma<" #k "," #n ">";
    MF_PW_SHAPES(MF_PW)
#undef MF_PW
    return nullptr;
}
bool launch_pw(int K, int N, const int8_t *in, int8_t *out, const PwArgs &a, long long npix, hipStream_t s) {
#define MF_PW(k, n)                                                                             \
    if (K == k && N == n) {                                                                     \
        constexpr int NB = n < 64 ? n : 64, SLOTS = 4 / (n / NB), CPIX = k < 64 ? 1024 / k : 16; \
        constexpr int U = pw_chunks_in_flight(k);                                               \
        const long long nchunks = (npix + CPIX - 1) / CPIX;                                     \
        long long grid = (nchunks + SLOTS * U - 1) / (SLOTS * U);                               \
        if (grid > pw_grid_cap(k, n)) grid = pw_grid_cap(k, n);                                 \
        if (grid < 1) grid = 1;                                                                 \
        MF_DISPATCH5(a.magic, a.xr, launch_pw_t, (in, out, a, npix, (int)grid, s), k, n)                  \
        return true;                                                                            \
    }
    MF_PW_SHAPES(MF_PW)
#undef MF_PW
    return false;
}

} // namespace k
} // namespace mf
