// k_dwfc.hip -- DepthwiseConv2D with ONE input channel (8 output channels) -> FullyConnected (4 outputs) -> Softmax
// in ONE launch, the depthwise taps on the matrix pipe (speech.tflite ops 1..3: 49x40x1 -> 25x20x8 -> 4 -> 4).
// (src/ops/depthwise_conv_2d.rs:27-110 with one input channel, src/ops/fully_connected.rs:24-82,
//  src/ops/softmax.rs:13-32)
//
// Arithmetic contract, shared device helpers and launch plumbing: k_common.hpp.
//
// Why the matrix pipe: the 10x8 filter is 80 taps per output byte; on the VALU (dw_c1_lds, one v_dot4 per 4 taps and
// channel) that is 20 half-rate instructions per output byte -- 2.5x the requantisation that follows.  As a
// contraction it is tiny: out[c] = sum_taps w[tap][c] x[tap].
//
//   columns  : 16 IMAGES.  One workgroup owns 16 images; column i of every MFMA is image i, so all 16 columns of a
//              tile share the pixel position and every operand address is (lane constant) + (wave-uniform offset).
//   rows     : (p, c) = vertical neighbour p in {0,1} x channel c in 0..7: output pixels (2t, ox) and (2t+1, ox)
//              share 12 input rows (stride 2, 10 filter rows), so one accumulator tile has no idle rows and the 16
//              K-bytes x 4 lane groups x 3 MFMAs are exactly those 12 rows.
//   K bytes  : lane group g of k-step k supplies tile row 4t + 4k + g, bytes [8m, 8m+16): the 8-tap window of pixel
//              ox = 4m + s starts at byte 8m + 2s + 1 of it, wherever s is -- the operand A (weights) is built on the
//              host per s in 0..3 with the taps at that byte offset and zeros elsewhere.  The activation read is
//              therefore always an ALIGNED 16 bytes (one ds_read2_b64), no byte shifting on the device at all, and ONE
//              read serves the four pixels 4m .. 4m+3: a unit (t, m) is 3 operand reads and 4 x 3 MFMAs.
//   tile     : per image [60 rows][56 B] at a pitch of 3376 B, inside a halo of the input zero point (written once);
//              image column 0 sits at byte 4 of a row so that staging is dword writes.  Pitches chosen so that the
//              32 lanes of a b64 read pass hit 64 distinct banks: (image * 844 + g * 14) mod 64 are disjoint pairs.
//              Two tile sets: the next 16 images are written while this step's are being read.
//   epilogue : a lane holds, per pixel, 4 channels of one image: requantise and pack (bytes as the depthwise operator
//              would store them).  The 4 dwords of a unit's 4 pixels are exactly one lane's 16 K-bytes of operand B
//              of a further MFMA, whose operand A holds the FullyConnected weights of those 64 activations in rows
//              0..3 and ones in row 4 (the row sum the weight zero point needs): the FullyConnected is one more MFMA
//              per unit, accumulated across the wave's units.  The depthwise output never exists in memory.
//   tail     : accumulator rows 0..4 -> LDS per wave -> 64 threads finish (image, output): sum over waves,
//              requantise, softmax over the 4 outputs in the reference's order, store 64 bytes.
// HBM traffic: input + 4 output bytes per inference; the kernel is bounded by the requantisation VALU work.
#include "k_common.hpp"

namespace mf {
namespace k {

typedef int v2i __attribute__((ext_vector_type(2)));
#ifndef MF_DWFC_DIAG
#define MF_DWFC_DIAG 0 // 1: cycle stamps of block 0 / wave 0 at the phase boundaries of its first two steps (never shipped)
#endif
#ifndef MF_DWFC_WPE
#define MF_DWFC_WPE 2  // waves per SIMD the register budget allows (launch bound)
#endif

#if MF_DWFC_DIAG
__device__ long long g_dwfc_trace[32];
#define MF_TR(k) do { if (blockIdx.x == 0 && wave == 0 && lane == 0 && trace_step < 2) g_dwfc_trace[(k) + 8 * trace_step] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define MF_TR(k) do { } while (0)
#endif

template <int NTHR, int MG, uint32_t XR4, bool WZP>
__global__ __launch_bounds__(NTHR, MF_DWFC_WPE) void dwc1_fc_softmax(const int8_t *__restrict__ in, int8_t *__restrict__ out, DwFcArgs p,
                                                        size_t batch) {
    using Gm = DwFcGeom;
    constexpr int NW = NTHR / 64;
    constexpr int HW = Gm::H * Gm::W;
    constexpr int NCH = Gm::IMGS * HW / 16;                               // 16-byte chunks of one step's input
    constexpr int NE = (NCH + NTHR - 1) / NTHR;
    constexpr int SET = Gm::IMGS * Gm::TILE;                              // one tile set
    static_assert(HW % 8 == 0 && Gm::W % 4 == 0 && (Gm::IMGS * HW) % 16 == 0, "staging granularity");
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint8_t *fcw = lds + 2 * SET;                                         // [NU][4 g][5 rows][16 B]
    int *part = (int *)(fcw + Gm::FCW_BYTES);                             // [2][NW][16 images][8]: 4 sums, row sum
    float *expt = (float *)(part + 2 * NW * 16 * 8);                      // softmax's 256-entry table
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, g = lane >> 4;
#if MF_DWFC_DIAG
    int trace_step = 0;
    if (blockIdx.x == 0 && tid == 0) g_dwfc_trace[31] = (long long)__builtin_readcyclecounter();
#endif

    for (int i = tid; i < Gm::FCW_BYTES / 16; i += NTHR) ((uint4 *)fcw)[i] = ((const uint4 *)p.wfc)[i];
    for (int i = tid; i < 2 * SET / 16; i += NTHR) ((uint4 *)lds)[i] = make_uint4(p.izp4, p.izp4, p.izp4, p.izp4);
    for (int i = tid; i < 256; i += NTHR) expt[i] = p.sm.exp_table[i];
    v4i Aw[4][3];
#pragma unroll
    for (int sft = 0; sft < 4; ++sft)
#pragma unroll
        for (int k = 0; k < 3; ++k) Aw[sft][k] = ((const v4i *)p.wA)[(sft * 3 + k) * 64 + lane];
    const int cq = (g & 1) * 4; // this lane's channels within the pixel
    const float4 cA = *(const float4 *)(p.dwA + cq), cS = *(const float4 *)(p.dwS + cq);
    const int4 cK = magic4<MG>(*(const int4 *)(p.dwKc + cq));
    const int fcK = p.fc.Kc[lane & 3];  // the finishing threads' (tid < 64) output n = lane & 3
    const float fcA = p.fc.A[lane & 3];
    // units (t, m) of this wave: wave, wave + NW, ...
    const int ni = (Gm::NU - wave + NW - 1) / NW;
    const int tb_off = col * Gm::TILE + g * Gm::RP;
    const bool frow = col < 5;                                           // rows of the FullyConnected operand A that exist
    const uint8_t *fw = fcw + (g * 5 + (frow ? col : 0)) * 16;
    auto loadB = [&](const uint8_t *tb, int j, v4i (&B)[3]) {
        const int t = j / Gm::NM, m = j - t * Gm::NM;
        const uint8_t *a = tb + (4 * t) * Gm::RP + 8 * m;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const v2i b0 = *(const v2i *)(a + (4 * k) * Gm::RP), b1 = *(const v2i *)(a + (4 * k) * Gm::RP + 8);
            B[k] = (v4i){b0.x, b0.y, b1.x, b1.y};
        }
    };
    auto loadF = [&](int j) {
        v4i f = {0, 0, 0, 0};
        if (frow) f = *(const v4i *)(fw + j * 320);
        return f;
    };

    const size_t nblk = (batch + Gm::IMGS - 1) / Gm::IMGS;
    // 16 images are contiguous in HBM: 16 B per lane, clamped at the end of the batch
    uint4 v[NE];
    auto load_images = [&](size_t blk) {
        const int8_t *src = in + blk * (size_t)(Gm::IMGS * HW);
        const size_t left = (batch - blk * Gm::IMGS) * (size_t)HW;
        const int limit = left < (size_t)(Gm::IMGS * HW) ? (int)left : Gm::IMGS * HW; // valid bytes (multiple of 8)
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int off = (tid + NTHR * e) * 16;
            v[e] = make_uint4(0, 0, 0, 0);
            if (off + 16 <= limit) v[e] = *(const uint4 *)(src + off);
            else if (off + 8 <= limit) { // the last image of an odd count ends mid-chunk
                const uint2 h = *(const uint2 *)(src + off);
                v[e].x = h.x, v[e].y = h.y;
            }
        }
    };
    auto write_images = [&](uint8_t *set) {
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int c = tid + NTHR * e;
            if (c < NCH) {
                const uint32_t w4[4] = {v[e].x, v[e].y, v[e].z, v[e].w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int d = 4 * c + q;                       // dword of the 16-image block
                    const int img = d / (HW / 4), r = d - img * (HW / 4);
                    const int row = r / (Gm::W / 4), cw = r - row * (Gm::W / 4);
                    *(uint32_t *)(set + img * Gm::TILE + (row + Gm::PT) * Gm::RP + Gm::XO + 4 * cw) = w4[q];
                }
            }
        }
    };
    if (blockIdx.x >= nblk) return;
    load_images(blockIdx.x);
    wg_sync(); // halo and tables written
    MF_TR(0);
    write_images(lds);
    if (blockIdx.x + gridDim.x < nblk) load_images(blockIdx.x + gridDim.x);
    wg_sync();
    MF_TR(1);
    int cur = 0;
    for (size_t blk = blockIdx.x; blk < nblk; blk += gridDim.x, cur ^= 1) {
        // ---- depthwise taps (MFMA) -> requantise -> FullyConnected (MFMA) on tile set `cur` ----
        const uint8_t *tb = lds + cur * SET + tb_off;
        v4i facc = {0, 0, 0, 0};
        v4i B[3], Bn[3], F, Fn;
        loadB(tb, wave, B);
        F = loadF(wave);
        for (int i = 0; i < ni; ++i) {
            const int jn = wave + NW * (i + 1 < ni ? i + 1 : i); // next unit (harmless re-read at the end)
            loadB(tb, jn, Bn);
            Fn = loadF(jn);
            v4i acc[4];
#pragma unroll
            for (int sft = 0; sft < 4; ++sft) acc[sft] = (v4i){cK.x, cK.y, cK.z, cK.w};
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int sft = 0; sft < 4; ++sft)
                    acc[sft] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Aw[sft][k], B[k], acc[sft], 0, 0, 0);
            uint32_t q0, q1, q2, q3; // pixels 4m .. 4m+3, this lane's 4 channels
            requant_pack4x2<MG, XR4>(acc[0], cA, cS, acc[1], cA, cS, p.dw_lo, p.dw_hi, q0, q1);
            requant_pack4x2<MG, XR4>(acc[2], cA, cS, acc[3], cA, cS, p.dw_lo, p.dw_hi, q2, q3);
            facc = __builtin_amdgcn_mfma_i32_16x16x64_i8(F, (v4i){(int)q0, (int)q1, (int)q2, (int)q3}, facc, 0, 0, 0);
#pragma unroll
            for (int k = 0; k < 3; ++k) B[k] = Bn[k];
            F = Fn;
        }
        MF_TR(2);
        // accumulator rows: lane group 0 holds outputs 0..3 of image `col`, lane group 1 row 4 = the row sum
        int *pw = part + ((cur * NW + wave) * 16 + col) * 8;
        if (g == 0) *(int4 *)pw = make_int4(facc[0], facc[1], facc[2], facc[3]);
        if (g == 1) pw[4] = facc[0];
        // the other tile set was last read one step ago (before the previous barrier): stage the next images now
        if (blk + gridDim.x < nblk) {
            write_images(lds + (cur ^ 1) * SET);
            if (blk + 2 * (size_t)gridDim.x < nblk) load_images(blk + 2 * (size_t)gridDim.x);
        }
        MF_TR(3);
        wg_sync(); // partial sums and the next tile set are visible
        MF_TR(4);
        if (tid < 64) { // (image, output) = (lane >> 2, lane & 3)
            const int img = lane >> 2, n = lane & 3;
            int d = 0, r = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                d += part[((cur * NW + w) * 16 + img) * 8 + n];
                if constexpr (WZP) r += part[((cur * NW + w) * 16 + img) * 8 + 4];
            }
            const int acc = d - p.fc.wzp * r + fcK;
            // the FullyConnected output byte as it would be stored (i8 domain), then softmax's table index
            const int y = (int)(int8_t)(requant(acc, fcA, p.fc.S, p.fc.lo_f, p.fc.hi_f) ^ p.fc.xr);
            const float e = expt[y + 128];
            float sum = 0.0f;
#pragma unroll
            for (int jn = 0; jn < 4; ++jn) sum = __fadd_rn(sum, __shfl(e, (lane & ~3) + jn, 64)); // softmax.rs:20-21 order
            const float prob = __fdiv_rn(e, sum);
            const float qf = __fadd_rn(__fdiv_rn(prob, p.sm.oscale), p.sm.ozp_f);
            const float rr = __fadd_rn(qf, __builtin_copysignf(0x1.fffffep-2f, qf));
            const int qi = (rr != rr) ? 0 : (int)__builtin_amdgcn_fmed3f(rr, p.sm.sat_lo, p.sm.sat_hi);
            const size_t image = blk * Gm::IMGS + img;
            if (image < batch) out[image * 4 + n] = (int8_t)(qi ^ p.sm.xr);
        }
        MF_TR(5);
#if MF_DWFC_DIAG
        ++trace_step;
#endif
    }
}

bool dwfc_supported(int H, int W, int KH, int KW, int sh, int sw, int OH, int OW, int DM, int NFC) {
    using Gm = DwFcGeom;
    return H == Gm::H && W == Gm::W && KH == Gm::KH && KW == Gm::KW && sh == Gm::S && sw == Gm::S && OH == Gm::OH &&
           OW == Gm::OW && DM == 8 && NFC == 4;
}
const char *dwfc_name() { return "dwc1_fc_softmax<49,40,10,8,2>"; }
void launch_dwfc(const int8_t *in, int8_t *out, const DwFcArgs &a, size_t batch, hipStream_t s) {
    using Gm = DwFcGeom;
    constexpr int NTHR = MF_DWFC_THREADS;
    constexpr int lds = 2 * Gm::IMGS * Gm::TILE + Gm::FCW_BYTES + 2 * (NTHR / 64) * 16 * 8 * 4 + 256 * 4;
    const size_t nblk = (batch + Gm::IMGS - 1) / Gm::IMGS;
    const bool wz = a.fc.wzp != 0;
#define MF_DWFC(MG, XR, WZ)                                                                           \
    do {                                                                                              \
        static LaunchState st;                                                                        \
        const int per_cu = prepared(st, dwc1_fc_softmax<NTHR, MG, XR, WZ>, NTHR, lds);                \
        const size_t cap = (size_t)256 * per_cu;                                                      \
        hipLaunchKernelGGL((dwc1_fc_softmax<NTHR, MG, XR, WZ>), dim3((unsigned)(nblk < cap ? nblk : cap)), dim3(NTHR), lds, \
                           s, in, out, a, batch);                                                     \
    } while (0)
#define MF_DWFC2(MG, XR) \
    if (wz) MF_DWFC(MG, XR, true); else MF_DWFC(MG, XR, false)
    if (a.xr) { if (a.magic == 2) { MF_DWFC2(2, 0x80808080u); } else if (a.magic) { MF_DWFC2(1, 0x80808080u); } else { MF_DWFC2(0, 0x80808080u); } }
    else { if (a.magic == 2) { MF_DWFC2(2, 0u); } else if (a.magic) { MF_DWFC2(1, 0u); } else { MF_DWFC2(0, 0u); } }
#undef MF_DWFC2
#undef MF_DWFC
#if MF_DWFC_DIAG
    {
        static int calls = 0;
        if (++calls == 12 || calls == 40) {
            (void)hipStreamSynchronize(s);
            long long h[32];
            (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_dwfc_trace), sizeof(h));
            fprintf(stderr, "[dwfc trace] batch %zu; cycles since kernel entry:", batch);
            for (int i = 0; i < 16; ++i) fprintf(stderr, " %d:%lld", i, h[i] - h[31]);
            fprintf(stderr, "\n");
        }
    }
#endif
}

} // namespace k
} // namespace mf
