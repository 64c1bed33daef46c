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
//              therefore always an ALIGNED 16 bytes (one ds_read2_b64), no byte shifting on the device at all.
//   tile     : per image [60 rows][56 B] at a pitch of 3376 B, inside a halo of the input zero point (written once);
//              image column 0 sits at byte 4 of a row so that staging is dword writes.  Pitches chosen so that the
//              32 lanes of a b64 read pass hit 64 distinct banks: (image * 844 + g * 14) mod 64 are disjoint pairs.
//   epilogue : a lane holds 4 channels of one pixel of one image: requantise, pack (bytes as the depthwise operator
//              would store them), and feed the FullyConnected at once: 4 v_dot4 against that pixel's weights (LDS
//              table, one broadcast b128 per lane group).  The depthwise output never exists in memory.
//   tail     : partial sums -> lane groups (2 shuffles) -> waves (LDS) -> 64 threads finish (image, output):
//              requantise, softmax over the 4 outputs in the reference's order, store 64 bytes.
// HBM traffic: input + 4 output bytes per inference; the kernel is bounded by the requantisation VALU work.
#include "k_common.hpp"

namespace mf {
namespace k {

typedef int v2i __attribute__((ext_vector_type(2)));

template <int NTHR, bool MG, uint32_t XR4, bool WZP>
__global__ __launch_bounds__(NTHR) void dwc1_fc_softmax(const int8_t *__restrict__ in, int8_t *__restrict__ out, DwFcArgs p,
                                                        size_t batch) {
    using Gm = DwFcGeom;
    constexpr int NW = NTHR / 64, PARTS = NW / 4; // waves; waves per tap shift s
    constexpr int HW = Gm::H * Gm::W;
    constexpr int NCH = Gm::IMGS * HW / 16;                               // 16-byte chunks of one step's input
    constexpr int NE = (NCH + NTHR - 1) / NTHR;
    static_assert(HW % 8 == 0 && Gm::W % 4 == 0 && (Gm::IMGS * HW) % 16 == 0, "staging granularity");
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint8_t *fcw = lds + Gm::IMGS * Gm::TILE;                             // [4 s][NU][4 g][4 n] dwords
    int *part = (int *)(fcw + Gm::FCW_BYTES);                             // [NW][16 images][8]: 4 sums, row sum
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, g = lane >> 4;
    const int s = wave & 3, part_i = wave >> 2;

    for (int i = tid; i < Gm::FCW_BYTES / 16; i += NTHR) ((uint4 *)fcw)[i] = ((const uint4 *)p.wfc)[i];
    for (int i = tid; i < Gm::IMGS * Gm::TILE / 16; i += NTHR) ((uint4 *)lds)[i] = make_uint4(p.izp4, p.izp4, p.izp4, p.izp4);
    v4i Aw[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) Aw[k] = ((const v4i *)p.wA)[(s * 3 + k) * 64 + lane];
    const int cq = (g & 1) * 4; // this lane's channels within the pixel
    const float4 cA = *(const float4 *)(p.dwA + cq), cS = *(const float4 *)(p.dwS + cq);
    const int4 cK = magic4<MG>(*(const int4 *)(p.dwKc + cq));
    // units (t, m) of this wave: an equal share of the NU = NT * NM of its tap shift
    constexpr int PER = (Gm::NU + PARTS - 1) / PARTS;
    const int j0 = part_i * PER, j1 = (j0 + PER < Gm::NU) ? j0 + PER : Gm::NU;
    const uint8_t *tb = lds + col * Gm::TILE + g * Gm::RP;
    const uint8_t *fw = fcw + (size_t)s * Gm::NU * 64 + g * 16;

    const size_t nblk = (batch + Gm::IMGS - 1) / Gm::IMGS;
    for (size_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        // ---- stage 16 images: contiguous in HBM, 16 B per lane ----
        const int8_t *src = in + blk * (size_t)(Gm::IMGS * HW);
        const size_t left = (batch - blk * Gm::IMGS) * (size_t)HW;
        const int limit = left < (size_t)(Gm::IMGS * HW) ? (int)left : Gm::IMGS * HW; // valid bytes (multiple of 8)
        uint4 v[NE];
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
        __syncthreads(); // halo / table written (first step); previous step's tiles and partial sums consumed
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
                    *(uint32_t *)(lds + img * Gm::TILE + (row + Gm::PT) * Gm::RP + Gm::XO + 4 * cw) = w4[q];
                }
            }
        }
        __syncthreads();

        // ---- depthwise taps (MFMA) -> requantise -> FullyConnected partial sums ----
        int fc[4] = {0, 0, 0, 0}, rs = 0;
        constexpr int UB = 3;
        for (int j = j0; j < j1; j += UB) {
            v4i B[UB][3];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int jj = (j + u < j1) ? j + u : j1 - 1;
                const int t = jj / Gm::NM, m = jj - t * Gm::NM;
                const uint8_t *a = tb + (4 * t) * Gm::RP + 8 * m;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const v2i b0 = *(const v2i *)(a + (4 * k) * Gm::RP), b1 = *(const v2i *)(a + (4 * k) * Gm::RP + 8);
                    B[u][k] = (v4i){b0.x, b0.y, b1.x, b1.y};
                }
            }
            v4i acc[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) acc[u] = (v4i){cK.x, cK.y, cK.z, cK.w};
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int u = 0; u < UB; ++u) acc[u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Aw[k], B[u][k], acc[u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                if (j + u < j1) { // wave-uniform
                    const int jj = j + u;
                    const uint32_t q = requant_pack4<MG, XR4>(acc[u][0], acc[u][1], acc[u][2], acc[u][3], cA, cS, p.dw_lo, p.dw_hi);
                    const uint4 wv = *(const uint4 *)(fw + jj * 64);
                    fc[0] = sdot4(q, wv.x, fc[0]), fc[1] = sdot4(q, wv.y, fc[1]);
                    fc[2] = sdot4(q, wv.z, fc[2]), fc[3] = sdot4(q, wv.w, fc[3]);
                    if constexpr (WZP) { // row sum of the FullyConnected input: pixel row 2t + p must exist
                        const int t = jj / Gm::NM;
                        rs = sdot4(q, (2 * t + (g >> 1) < Gm::OH) ? 0x01010101u : 0u, rs);
                    }
                }
            }
        }
        // lane groups of one image, then the waves
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            fc[n] += __shfl_xor(fc[n], 16, 64);
            fc[n] += __shfl_xor(fc[n], 32, 64);
        }
        if constexpr (WZP) rs += __shfl_xor(rs, 16, 64), rs += __shfl_xor(rs, 32, 64);
        if (g == 0) {
            int *dst = part + (wave * 16 + col) * 8;
            *(int4 *)dst = make_int4(fc[0], fc[1], fc[2], fc[3]);
            dst[4] = rs;
        }
        __syncthreads();
        if (tid < 64) { // (image, output) = (lane >> 2, lane & 3)
            const int img = lane >> 2, n = lane & 3;
            int d = 0, r = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                d += part[(w * 16 + img) * 8 + n];
                if constexpr (WZP) r += part[(w * 16 + img) * 8 + 4];
            }
            const int acc = d - p.fc.wzp * r + p.fc.Kc[n];
            // the FullyConnected output byte as it would be stored (i8 domain), then softmax's table index
            const int y = (int)(int8_t)(requant(acc, p.fc.A[n], p.fc.S, p.fc.lo_f, p.fc.hi_f) ^ p.fc.xr);
            const float e = p.sm.exp_table[y + 128];
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
    constexpr int lds = Gm::IMGS * Gm::TILE + Gm::FCW_BYTES + (NTHR / 64) * 16 * 8 * 4;
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
    if (a.xr) { if (a.magic) { MF_DWFC2(true, 0x80808080u); } else { MF_DWFC2(false, 0x80808080u); } }
    else { if (a.magic) { MF_DWFC2(true, 0u); } else { MF_DWFC2(false, 0u); } }
#undef MF_DWFC2
#undef MF_DWFC
}

} // namespace k
} // namespace mf
