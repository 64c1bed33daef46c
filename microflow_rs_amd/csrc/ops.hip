// ops.hip -- prepared operators: fold the reference's constants into the
// per-channel device arrays the kernels consume, pick a kernel for the shape,
// launch.  Host code only (HIP runtime API); the kernels are in k_*.hip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "kernels.hpp"
#include "mf_internal.hpp"
#include <array>
#include <mutex>
#include <map>

namespace mf {

#define MF_HIP(call)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess)                                                                 \
            fail(e_ == hipErrorOutOfMemory ? MF_ERR_OOM : MF_ERR_HIP,                         \
                 std::string(#call) + ": " + hipGetErrorString(e_));                          \
    } while (0)

int dev_count() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

void dev_require(int device) {
    const int n = dev_count();
    if (n <= 0)
        fail(MF_ERR_NO_DEVICE, "no HIP device available: libmicroflow_amd has no CPU fallback");
    if (device < 0 || device >= n)
        fail(MF_ERR_INVALID_ARG, "device index " + std::to_string(device) + " out of range (" +
                                     std::to_string(n) + " devices)");
    MF_HIP(hipSetDevice(device));
}

namespace {

inline int32_t wrap_add(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
inline int32_t wrap_sub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
inline int32_t wrap_mul(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    void upload(const void *src, size_t bytes) {
        if (p) {
            (void)hipFree(p);
            p = nullptr;
        }
        if (!bytes) return;
        MF_HIP(hipMalloc(&p, bytes));
        MF_HIP(hipMemcpy(p, src, bytes, hipMemcpyHostToDevice));
    }
    template <typename T> const T *as() const { return (const T *)p; }
};

// activation + `as T` saturation as one clamp [lo, hi] in T's domain  (src/activation.rs:21-34)
void act_bounds(int act, float oscale, int ozp, bool u8, int &lo, int &hi) {
    lo = u8 ? 0 : -128;
    hi = u8 ? 255 : 127;
    if (act == MF_ACT_RELU || act == MF_ACT_RELU6) lo = ozp;        // max(y, zero_point)
    if (act == MF_ACT_RELU6) hi = h_quantize_t(6.0f, oscale, ozp, u8); // min(.., quantize(6.0))
    if (lo > hi) lo = hi; // min(max(y, lo), hi) == hi for every y when lo > hi
}

} // namespace

struct OpImpl {
    int device = 0;
    OpSpec s; // pointers inside are NOT valid after create
    size_t in_elems = 0, out_elems = 0;
    bool force_generic = false;
    bool accepts_f32 = false;  // op_set_input_quant succeeded: op_run_f32 may replace quantize + op_run
    bool finite_consts = true; // A / S all finite (the shape-specialised and fused epilogues assume it)
    std::string generic_name, fast_name;
    enum Fast { NONE, DW_NHWC, DW_STEM, DW_STEM_RT, DW_C1, PW_MFMA, FC_ROWWAVE, FC_MFMA, POOL_C4, CONV1X1_ROW, DW_RT, PW_RT, CONV_ROWS, CONV_MM } fast = NONE;
    int *d_rowsum = nullptr; // FC_MFMA with wzp != 0: per-row input sums
    size_t rowsum_cap = 0, rowsum_rows = 0; // (ints allocated; the row count the counter pairs currently sit behind)
    int8_t *d_ext = nullptr; // op_run_external on a u8 operator: input moved to the i8 domain
    size_t ext_cap = 0;
    // d_rowsum and d_ext are ONE scratch each per operator, while a handle may be launched on several streams: every use waits (on
    // the device) for the previous use's last reader and records the event again behind its own, so concurrent launches of these
    // two paths are serialised instead of racing; growing a buffer waits for the event on the host before the free.
    hipEvent_t scratch_ev = nullptr;
    bool scratch_used = false;
    // Under stream capture (mf_model_set_graph) the handshake is skipped: a captured wait on an event recorded outside the capture
    // is not legal, and an event recorded INTO the graph would leave later eager waits looking at a stale record.  The model runtime
    // captures one stream, on which the launches are ordered anyway, and a graph's buffers never grow (the eager pass before the
    // capture sized them).
    static bool capturing(hipStream_t s) {
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &st) != hipSuccess) (void)hipGetLastError();
        return st != hipStreamCaptureStatusNone;
    }
    void scratch_acquire(hipStream_t s, bool growing) {
        if (capturing(s)) {
            if (growing) fail(MF_ERR_HIP, "a scratch buffer would have to grow inside a stream capture");
            return;
        }
        if (!scratch_ev) MF_HIP(hipEventCreateWithFlags(&scratch_ev, hipEventDisableTiming));
        if (scratch_used) {
            if (growing) MF_HIP(hipEventSynchronize(scratch_ev));
            else MF_HIP(hipStreamWaitEvent(s, scratch_ev, 0));
        }
    }
    void scratch_release(hipStream_t s) {
        if (capturing(s)) return;
        if (!scratch_ev) MF_HIP(hipEventCreateWithFlags(&scratch_ev, hipEventDisableTiming));
        MF_HIP(hipEventRecord(scratch_ev, s));
        scratch_used = true;
    }

    DevBuf d_w, d_wzp, d_A, d_S, d_Kc, d_wprep, d_wsp, d_wrr, d_table;
    unsigned long q_launches = 0; // launches that drew a counter set from d_queue so far (atomic increments: k_common.hpp dq_slot)
    DevBuf d_queue; // zeroed counters: the dynamic step queue of the persistent kernels launched for this operator (k_common.hpp)
    k::DwC1Args dwc1{};
    k::ConvArgs conv{};
    k::PoolArgs pool{};
    k::FcArgs fc{};
    k::SoftmaxArgs sm{};
    k::DwFastArgs dwf{};
    k::DwStemArgs stem{};
    k::DwStemRtArgs stemrt{};
    k::PwArgs pw{};
    // run-time-geometry kernels (k_rt.hip): shapes outside the tables of kernels.hpp
    k::DwRtArgs dwrt{};
    k::PwRtArgs pwrt{};
    k::ConvRowsArgs crows{};
    k::ConvMmArgs cmm{};
    DevBuf d_tap;          // conv_mm_rt: tap offset table
    DevBuf d_crw, d_crm;   // conv_rows_lds: packed weights, tap masks
    bool rt_wz = false;    // non-zero weight zero points
    int magic_mode = 0;    // conv-like operators: epilogue mode the host proved usable (k_common.hpp: 0, 1 or 2)
    // ... and mode 3, the single-fma form: found per channel by the host search (epi_fma.cpp) AND confirmed on the device over every
    // reachable accumulator (k_generic.hip verify_fma_form).  The arrays hold C', S', Kc + pivot; the two-rounding constants stay
    // beside them (a fused launch uses mode 3 only if every operator in it has it).
    bool fma_ok = false;     // ... for every channel, with at most EPI_PATCH_MAX patched accumulators in all (fma_patch)
    k::EpiPatch fma_patch{}; // the channels whose line needs ONE accumulator replaced (epi_fma.cpp); n = 0: none
    bool fma_strict() const { return fma_ok && fma_patch.n == 0; } // what the kernels without patch support need
    DevBuf d_A3, d_S3, d_Kc3;
    std::vector<float> h_A3, h_S3;
    std::vector<int32_t> h_Kc3;
    int pw_group = 1;      // pixels presented as one row of the 1x1 product (K = 8 -> 2, K = 4 -> 4)
    DevBuf d_rtA, d_rtS, d_rtKc, d_rtwzp; // constants replicated per group member
};

namespace {

// per-channel A/S/Kc/wzp arrays for the conv-like operators
// Returns the worst-case |acc| over all inputs: max|v - izp| * max_c sum_taps |w[c] - wzp[c]|
// (acc = sum over ALL taps of (v' - izp)(w - wzp), the halo contributing 0).
int64_t fold_conv_constants(OpImpl &op, const OpSpec &s, bool depthwise, std::vector<float> &A,
                            std::vector<float> &S, std::vector<int32_t> &Kc,
                            std::vector<int32_t> &wzp, std::vector<int64_t> *acc_bound_per_channel = nullptr,
                            std::vector<std::pair<int64_t, int64_t>> *acc_range_per_channel = nullptr) {
    const int N = s.N;
    const int taps = s.KH * s.KW;
    int64_t max_wabs = 0;
    A.resize(N), S.resize(N), Kc.resize(N), wzp.resize(N);
    for (int c = 0; c < N; ++c) {
        volatile float a = (float)s.ozp + s.c0[c]; // f32(ozp) + c0[c], rounded once, as the reference does first
        A[c] = a;
        S[c] = s.c1[c < s.nc1 ? c : 0];            // constants.1.get(b).unwrap_or(constants.1[0])
        wzp[c] = s.wzp[c < s.nq ? c : 0];          // zero_point.get(b).unwrap_or(zero_point[0])
        int32_t wsum = 0;
        int32_t T;
        int64_t wabs = 0;
        // the exact interval of acc = sum_taps (v - izp)(w - wzp) over int8 v (a halo tap contributes 0, which lies inside every
        // tap's own interval): what the single-fma epilogue has to reproduce the reference on (epi_fma.cpp)
        const int64_t vlo = -128 - s.izp, vhi = 127 - s.izp;
        int64_t amin = 0, amax = 0;
        auto tap = [&](int w) {
            wsum = wrap_add(wsum, w);
            const int64_t dw_ = (int64_t)w - wzp[c];
            wabs += dw_ < 0 ? -dw_ : dw_;
            amin += std::min(std::min(vlo * dw_, vhi * dw_), (int64_t)0), amax += std::max(std::max(vlo * dw_, vhi * dw_), (int64_t)0);
        };
        if (depthwise) { // weights [KH][KW][N]
            for (int t = 0; t < taps; ++t) tap(s.weights[(size_t)t * N + c]);
            T = taps;
        } else { // filters [N][KH][KW][C]
            const int8_t *f = s.weights + (size_t)c * taps * s.C;
            for (int t = 0; t < taps * s.C; ++t) tap(f[t]);
            T = taps * s.C;
        }
        if (acc_range_per_channel) acc_range_per_channel->push_back({amin, amax});
        max_wabs = std::max(max_wabs, wabs);
        if (acc_bound_per_channel) acc_bound_per_channel->push_back(wabs * std::max(127 - s.izp, s.izp + 128));
        // Kc = -izp * sum(w) + T * izp * wzp   (k2 and k3 of the reference with the halo == izp)
        Kc[c] = wrap_add(wrap_sub(0, wrap_mul(s.izp, wsum)), wrap_mul(wrap_mul(T, s.izp), wzp[c]));
    }
    (void)op;
    const int64_t vdev = std::max(127 - s.izp, s.izp + 128); // max |v - izp| over int8 v
    return vdev * max_wabs;
}

bool all_finite(const std::vector<float> &v) {
    return std::all_of(v.begin(), v.end(), [](float x) { return std::isfinite(x); });
}
bool all_zero(const std::vector<int32_t> &v) {
    return std::all_of(v.begin(), v.end(), [](int32_t x) { return x == 0; });
}

// operand A of v_mfma_i32_16x16x64_i8 for pw_mfma<K,N>: [blk][q][tt][ks][lane][16 bytes]
std::vector<int8_t> build_pw_weights(const int8_t *w /*[N][K]*/, int K, int N) {
    const int NB = N < 64 ? N : 64, TB = NB / 16, NSPLIT = N / NB;
    const int KS = K < 64 ? 1 : K / 64, Q = K < 64 ? 64 / K : 1;
    std::vector<int8_t> out((size_t)NSPLIT * Q * TB * KS * 64 * 16, 0);
    for (int blk = 0; blk < NSPLIT; ++blk)
        for (int q = 0; q < Q; ++q)
            for (int tt = 0; tt < TB; ++tt)
                for (int ks = 0; ks < KS; ++ks)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int r = lane & 15, g = lane >> 4; // A row, k-block of this lane
                        // row r = 4*gr + j of tile tt is channel base + gr*(NB/4) + 4*tt + j
                        const int gr = r >> 2, j = r & 3;
                        const int ch = blk * NB + gr * (NB / 4) + 4 * tt + j;
                        int8_t *dst = &out[(((((size_t)blk * Q + q) * TB + tt) * KS + ks) * 64 + lane) * 16];
                        for (int i = 0; i < 16; ++i) {
                            int k = -1;
                            if (K >= 64) k = ks * 64 + g * 16 + i;
                            else if (K == 32) k = ((g >> 1) == q) ? (g & 1) * 16 + i : -1;
                            else if (K == 16) k = (g == q) ? i : -1;
                            else if (K == 8) k = (g == (q >> 1) && (i >> 3) == (q & 1)) ? (i & 7) : -1;
                            dst[i] = k >= 0 ? w[(size_t)ch * K + k] : (int8_t)0;
                        }
                    }
    return out;
}

// Depthwise 3x3 weights [3][3][C] as operand A of v_mfma_i32_16x16x64_i8 for dwpw_mm (k_fused_mm.hip):
// [group q][filter row ty][lane][16 bytes].  A lane holds row r = lane & 15 of the 16 x 64 block-diagonal
// matrix, K-block g = lane >> 4.
//   C >= 16: row r = channel 16q + r; block g = tap column tx = g (g == 3: padding), its 16 K-bytes are
//            the 16 channels of that tap's pixel -> the only non-zero byte is c' == r: w[ty][g][16q + r].
//   C == 8 : row r = (output pixel parity r >> 3, channel r & 7); block g = input pixel pair
//            (2x-2+2g, 2x-1+2g), byte (pp, c'): non-zero for c' == channel and tap column
//            tx = 2g + pp - 1 - parity in 0..2.
std::vector<int8_t> build_dw_mm_weights(const int8_t *w /*[3][3][C]*/, int C) {
    const int NQ = C == 8 ? 1 : C / 16;
    std::vector<int8_t> out((size_t)NQ * 3 * 64 * 16, 0);
    for (int q = 0; q < NQ; ++q)
        for (int ty = 0; ty < 3; ++ty)
            for (int lane = 0; lane < 64; ++lane) {
                const int r = lane & 15, g = lane >> 4;
                int8_t *dst = &out[(((size_t)q * 3 + ty) * 64 + lane) * 16];
                if (C == 8) {
                    const int par = r >> 3, c = r & 7;
                    for (int pp = 0; pp < 2; ++pp) {
                        const int tx = 2 * g + pp - 1 - par;
                        if (g < 3 && tx >= 0 && tx <= 2) dst[pp * 8 + c] = w[(ty * 3 + tx) * 8 + c];
                    }
                } else if (g < 3) {
                    dst[r] = w[(ty * 3 + g) * C + 16 * q + r];
                }
            }
    return out;
}

// The same taps for the structured-sparse matrix instruction (k_quad.hip).  A 3x3 depthwise operand A is block diagonal: a row has at
// most two non-zero bytes in any 16-byte chunk, never two in one group of four -- 2:4 sparse with room to spare -- and
// v_smfmac_i32_16x16x128_i8 multiplies a 2:4-sparse 16 x 128 A in the time v_mfma_i32_16x16x64_i8 takes for a dense 16 x 64
// (scripts/ubench/mfma_rates.hip: 16.7 against 17 cycles).  Eight of the nine (filter row, chunk column) blocks of
// build_dw_mm_weights go into ONE sparse instruction, the ninth into a v_mfma_i32_16x16x32_i8: two matrix instructions per unit
// instead of three.  Operand layout as measured by scripts/ubench/smfmac_probe.hip (profiles/r06/h_smfmac_probe.txt):
//   B lane (column, group lb) holds 32 bytes = two 16-byte chunks (half 0 / 1); A lane (row, group ga) holds 16 stored bytes:
//   stored byte s = 8 ha + 2 grp + j is element j of group grp (four dense bytes) of the chunk that B lane group lb = 2 (ga & 1) + ha
//   holds in half ga >> 1, and bits 2 s + 1 : 2 s of the index register say which of the four dense bytes it is.
// Chunk of (lb, half) as (filter row, chunk column): half 0 of lane groups 0..3 = (0,0) (0,1) (0,2) (1,0), half 1 = (1,1) (1,2) (2,0) (2,1);
// the ninth is (2,2).
// Returns [q][lane][32 bytes] = {stored A (16), index (4), ninth block as operand A of v_mfma_i32_16x16x32_i8 (8: lane group g' holds
// bytes 8 g' .. 8 g' + 7 of the chunk, groups 2 and 3 zero), 4 bytes padding}; empty if a group of four holds more than two non-zeros.
const int DW_SP_CHUNK[4][2][2] = {{{0, 0}, {1, 1}}, {{0, 1}, {1, 2}}, {{0, 2}, {2, 0}}, {{1, 0}, {2, 1}}};
std::vector<int8_t> build_dw_sp_weights(const std::vector<int8_t> &dense /* build_dw_mm_weights */, int NQ) {
    std::vector<int8_t> out((size_t)NQ * 64 * 32, 0);
    auto block = [&](int q, int ty, int gch, int r) { return &dense[((((size_t)q * 3 + ty) * 64) + (size_t)(gch * 16 + r)) * 16]; };
    for (int q = 0; q < NQ; ++q)
        for (int lane = 0; lane < 64; ++lane) {
            const int r = lane & 15, ga = lane >> 4;
            int8_t *dst = &out[((size_t)q * 64 + lane) * 32];
            uint32_t idx = 0;
            for (int ha = 0; ha < 2; ++ha) {
                const int lb = 2 * (ga & 1) + ha, half = ga >> 1;
                const int8_t *blk = block(q, DW_SP_CHUNK[lb][half][0], DW_SP_CHUNK[lb][half][1], r);
                for (int grp = 0; grp < 4; ++grp) {
                    int pos[4], n = 0;
                    for (int b = 0; b < 4; ++b)
                        if (blk[4 * grp + b] != 0) pos[n++] = b;
                    if (n > 2) return {};
                    if (n == 0) pos[0] = 0, pos[1] = 1;
                    if (n == 1) pos[1] = (pos[0] + 1) & 3;
                    for (int j = 0; j < 2; ++j) {
                        const int sb = 8 * ha + 2 * grp + j;
                        dst[sb] = j < n ? blk[4 * grp + pos[j]] : (int8_t)0;
                        idx |= (uint32_t)pos[j] << (2 * sb);
                    }
                }
            }
            memcpy(dst + 16, &idx, 4);
            if (ga < 2) memcpy(dst + 20, block(q, 2, 2, r) + 8 * ga, 8);
        }
    return out;
}

// Depthwise weights [KH][KW][C] (C % 16 == 0) as operand A of conv_mm_rt's depthwise mode (k_rt.hip): [16-channel group][k step][lane]
// x 16 bytes; lane (row r, group g) of step ks holds tap t = 4 ks + g: its only non-zero byte is byte r = w[t][16 q + r].
std::vector<int8_t> build_dw_mm_rt_weights(const int8_t *w, int KH, int KW, int C, int KS /* >= (KH KW + 3) / 4: padded with zero steps */) {
    const int NQ = C / 16, T = KH * KW;
    std::vector<int8_t> out((size_t)NQ * KS * 1024, 0);
    for (int q = 0; q < NQ; ++q)
        for (int ks = 0; ks < KS; ++ks)
            for (int lane = 0; lane < 64; ++lane) {
                const int r = lane & 15, t = 4 * ks + (lane >> 4);
                if (t < T) out[(((size_t)q * KS + ks) * 64 + lane) * 16 + r] = w[(size_t)t * C + 16 * q + r];
            }
    return out;
}

// The same operand for FEWER than 16 channels (chain_rt, k_chain.hip): P = 16 / C horizontally adjacent pixels are one 16-channel
// "superpixel", stride S in superpixels.  Row r = (output pixel p = r / C of superpixel X, channel r % C); block g = input superpixel
// S X - 1 + g, whose byte (pp, c') is input pixel P (S X - 1 + g) + pp: non-zero for c' == channel and the tap column
// tx = P (g - 1) + pp - S p + 1 in 0..2 (output pixel P X + p reads input pixels S (P X + p) + tx - 1).
std::vector<int8_t> build_dw_mm_weights_sp(const int8_t *w /*[3][3][C]*/, int C, int S) {
    const int P = 16 / C;
    std::vector<int8_t> out((size_t)3 * 64 * 16, 0);
    for (int ty = 0; ty < 3; ++ty)
        for (int lane = 0; lane < 64; ++lane) {
            const int r = lane & 15, g = lane >> 4, p = r / C, c = r % C;
            int8_t *dst = &out[((size_t)ty * 64 + lane) * 16];
            for (int pp = 0; pp < P; ++pp) {
                const int tx = P * (g - 1) + pp - S * p + 1;
                if (g < 3 && tx >= 0 && tx <= 2) dst[pp * C + c] = w[(ty * 3 + tx) * C + c];
            }
        }
    return out;
}

// Pointwise weights [N][K] as operands A of v_mfma_i32_16x16x32_i8 for dwpw_rr (k_fused_mm.hip), whose B operand
// is the depthwise result as it sits in registers: [16-row tile m][lane][8 bytes].  Lane (r = lane & 15,
// g = lane >> 4) holds K-bytes 8g .. 8g+7 of MFMA row r.
//   rows : row 4g' + i of tile m is output channel (N/4) g' + 4m + i, so that lane g' of the result owns N/4
//          consecutive output bytes.  K = 8: rows are (pixel parity g' >> 1, channel 8 (g' & 1) + 4m + i).
//   K    : byte b < 4 is input channel 4g + b (K = 8: channel 4 (g & 1) + b of the pixel with parity g >> 1, used
//          only by the rows of that pixel); byte b >= 4 is channel 16 + 4g + b - 4 when K = 32, else unused.
std::vector<int8_t> build_pw_rr_weights(const int8_t *w /*[N][K]*/, int K, int N) {
    const bool pair = K == 8;
    const int NT = (pair ? 2 * N : N) / 16;
    std::vector<int8_t> out((size_t)NT * 64 * 8, 0);
    for (int m = 0; m < NT; ++m)
        for (int lane = 0; lane < 64; ++lane) {
            const int r = lane & 15, g = lane >> 4;
            const int gr = r >> 2, i = r & 3;
            const int n = pair ? 8 * (gr & 1) + 4 * m + i : (N / 4) * gr + 4 * m + i;
            int8_t *dst = &out[((size_t)m * 64 + lane) * 8];
            for (int b = 0; b < 8; ++b) {
                int k = -1;
                if (pair) {
                    if (b < 4 && (g >> 1) == (gr >> 1)) k = 4 * (g & 1) + b;
                } else if (b < 4) {
                    k = 4 * g + b;
                } else if (K == 32) {
                    k = 16 + 4 * g + (b - 4);
                }
                dst[b] = k >= 0 ? w[(size_t)n * K + k] : (int8_t)0;
            }
        }
    return out;
}

// Operand A of v_mfma_i32_16x16x64_i8 for pw_rt (k_rt.hip): [16-row tile nt][k step ks][lane][16 bytes]; lane (r, g) holds
// K-bytes 64 ks + 16 g .. + 15 of row 16 nt + r.  `group` pixels form one row of the product: row gi * N + n multiplies
// only the K-bytes gj * K .. of its own pixel (block diagonal).  With `ones`, KS more KiB follow: a tile whose every row
// is 1 on the real K-bytes (the row sum a weight zero point needs).
std::vector<int8_t> build_pw_rt_weights(const int8_t *w /*[N][K]*/, int K, int N, int group, bool ones) {
    const int Kg = K * group, Ng = N * group, KS = (Kg + 63) / 64, NT = (Ng + 15) / 16;
    std::vector<int8_t> out(((size_t)NT * KS + (ones ? KS : 0)) * 1024, 0);
    for (int nt = 0; nt < NT; ++nt)
        for (int ks = 0; ks < KS; ++ks)
            for (int lane = 0; lane < 64; ++lane) {
                const int row = 16 * nt + (lane & 15), g = lane >> 4;
                if (row >= Ng) continue;
                const int gi = row / N, n = row % N;
                int8_t *dst = &out[(((size_t)nt * KS + ks) * 64 + lane) * 16];
                for (int i = 0; i < 16; ++i) {
                    const int kk = ks * 64 + g * 16 + i;
                    if (kk < Kg && kk / K == gi) dst[i] = w[(size_t)n * K + kk % K];
                }
            }
    if (ones)
        for (int ks = 0; ks < KS; ++ks)
            for (int lane = 0; lane < 64; ++lane)
                for (int i = 0; i < 16; ++i)
                    if (ks * 64 + (lane >> 4) * 16 + i < Kg) out[(((size_t)NT * KS + ks) * 64 + lane) * 16 + i] = 1;
    return out;
}

// The same product for pw_rt with the weights in registers: [block][tile t][k step][lane][16 bytes], TB tiles per block, and
// row 4 gr + i of tile t = channel 16 TB blk + 4 TB gr + 4 t + i (so that a lane ends with 4 TB consecutive output bytes).
std::vector<int8_t> build_pw_rt_reg_weights(const int8_t *w /*[N][K]*/, int K, int N, int group, int TB, int NBLK) {
    const int Kg = K * group, Ng = N * group, KS = (Kg + 63) / 64;
    std::vector<int8_t> out((size_t)NBLK * TB * KS * 1024, 0);
    for (int blk = 0; blk < NBLK; ++blk)
        for (int t = 0; t < TB; ++t)
            for (int ks = 0; ks < KS; ++ks)
                for (int lane = 0; lane < 64; ++lane) {
                    const int r = lane & 15, g = lane >> 4;
                    const int row = 16 * TB * blk + 4 * TB * (r >> 2) + 4 * t + (r & 3);
                    if (row >= Ng) continue;
                    const int gi = row / N, n = row % N;
                    int8_t *dst = &out[((((size_t)blk * TB + t) * KS + ks) * 64 + lane) * 16];
                    for (int i = 0; i < 16; ++i) {
                        const int kk = ks * 64 + g * 16 + i;
                        if (kk < Kg && kk / K == gi) dst[i] = w[(size_t)n * K + kk % K];
                    }
                }
    return out;
}

} // namespace

// layer-wise DepthwiseConv2D 3x3: taps on the matrix pipe (dwpw_mm's depthwise phase, k_fused_mm.hip) where that form is the
// faster one (the three large early layers), the v_dot4 kernel dw3x3_nhwc elsewhere: k::launch_dw_mm decides per shape
static bool dw_taps_on_matrix_pipe() { return true; }

OpImpl *op_create(int device, const OpSpec &spec) {
    dev_require(device);
    std::unique_ptr<OpImpl> op(new OpImpl);
    op->device = device;
    OpSpec s = spec;
    // T = u8: move the integer side to the i8 domain (every value and zero point minus 128);
    // ozp, A, the clamp and the saturation stay in the u8 domain (kernels.hpp)
    const int beta = s.u8 ? 128 : 0;
    const int xr = s.u8 ? 0x80 : 0;
    std::vector<int8_t> w_i8;
    std::vector<int> wzp_i8;
    auto to_i8_domain = [&](size_t wbytes) {
        if (!s.u8) return;
        w_i8.resize(wbytes);
        for (size_t i = 0; i < wbytes; ++i) w_i8[i] = (int8_t)(s.weights[i] ^ (int8_t)0x80);
        wzp_i8.resize((size_t)s.nq);
        for (int i = 0; i < s.nq; ++i) wzp_i8[(size_t)i] = s.wzp[i] - beta;
        s.weights = w_i8.data(), s.wzp = wzp_i8.data();
        s.izp -= beta;
    };
    int lo, hi;
    act_bounds(s.act, s.oscale, s.ozp, s.u8, lo, hi);

    switch (s.kind) {
    case MF_OP_CONV_2D:
    case MF_OP_DEPTHWISE_CONV_2D: {
        const bool dw = s.kind == MF_OP_DEPTHWISE_CONV_2D;
        if (s.H <= 0 || s.W <= 0 || s.C <= 0 || s.N <= 0 || s.KH <= 0 || s.KW <= 0 || s.OH <= 0 ||
            s.OW <= 0 || s.sh <= 0 || s.sw <= 0 || s.nq <= 0 || s.nc1 <= 0 || !s.weights || !s.wzp ||
            !s.c0 || !s.c1)
            fail(MF_ERR_INVALID_ARG, "conv: bad arguments");
        if (s.pad == MF_PAD_VALID &&
            ((s.OH - 1) * s.sh + s.KH > s.H || (s.OW - 1) * s.sw + s.KW > s.W))
            fail(MF_ERR_INVALID_ARG, "conv: VALID view leaves the input (the reference would panic, src/tensor.rs:223)");
        op->in_elems = (size_t)s.H * s.W * s.C;
        op->out_elems = (size_t)s.OH * s.OW * s.N;
        to_i8_domain(dw ? (size_t)s.KH * s.KW * s.N : (size_t)s.N * s.KH * s.KW * s.C);
        std::vector<float> A, S;
        std::vector<int32_t> Kc, wzp;
        // the fast kernels may convert the accumulator to f32 by bit pattern when it provably
        // stays below 2^22 in magnitude (requant_t<true> in k_common.hpp)
        const bool no_magic = switches().no_magic; // tests: force the convert form
        std::vector<int64_t> acc_bound_c; // per output channel: max |v - izp| * sum_taps |w - wzp|
        std::vector<std::pair<int64_t, int64_t>> acc_range_c; // per output channel: the exact interval of acc over all inputs
        const int64_t acc_bound = fold_conv_constants(*op, s, dw, A, S, Kc, wzp, &acc_bound_c, &acc_range_c);
        int magic = !no_magic && acc_bound < (1 << 22) ? 1 : 0;
        // mode 2 (k_common.hpp): the clamp is the element type's whole range (so a saturating pack can do it) and
        // |x| = |A + S * acc| stays below 2^15 for every input (so x + 128 fits the i16 the pack saturates from)
        const bool no_sat = switches().no_sat_pack; // tests: force the v_med3 form
        if (magic && !no_sat && lo == (s.u8 ? 0 : -128) && hi == (s.u8 ? 255 : 127) && all_finite(A) && all_finite(S)) {
            double xmax = 0.0;
            for (int c = 0; c < s.N; ++c)
                xmax = std::max(xmax, std::fabs((double)A[(size_t)c]) + std::fabs((double)S[(size_t)c]) * (double)acc_bound_c[(size_t)c]);
            if (xmax < 30000.0) magic = 2;
        }
        const bool epi_dbg = switches().debug_epi; // which epilogue mode each operator gets, and why
        if (epi_dbg)
            fprintf(stderr, "[epi] %s %dx%dx%d -> %d: |acc| < %lld, clamp [%d, %d] -> mode %d\n", dw ? "depthwise" : "conv", s.H, s.W, s.C, s.N,
                    (long long)acc_bound, lo, hi, magic);
        op->magic_mode = magic;
        const size_t wbytes = dw ? (size_t)s.KH * s.KW * s.N : (size_t)s.N * s.KH * s.KW * s.C;
        op->d_w.upload(s.weights, wbytes);
        op->d_wzp.upload(wzp.data(), wzp.size() * 4);
        op->d_A.upload(A.data(), A.size() * 4);
        op->d_S.upload(S.data(), S.size() * 4);
        op->d_Kc.upload(Kc.data(), Kc.size() * 4);
        {
            const std::vector<int> zeros((size_t)k::DYNQ_INTS * k::DYNQ_RING, 0);
            op->d_queue.upload(zeros.data(), zeros.size() * sizeof(int));
        }
        // Epilogue mode 3: y = v_cvt_pk_u8_f32(v_fma_f32(S', bits(acc + pivot), C')).  Conditions: bit-pattern accumulators (mode >= 1),
        // the clamp is the element type's whole range (the conversion's saturation IS the clamp), finite constants; then a solution
        // for EVERY channel (host, exact: epi_fma.cpp) that the device confirms on every reachable accumulator.
        const bool no_fma = switches().no_fma_epi; // tests / A-B: keep the two-rounding forms
        if (magic >= 1 && !no_fma && lo == (s.u8 ? 0 : -128) && hi == (s.u8 ? 255 : 127) && all_finite(A) && all_finite(S)) {
            std::vector<float> A3((size_t)s.N), S3((size_t)s.N);
            std::vector<int32_t> K3((size_t)s.N), piv((size_t)s.N), amn((size_t)s.N), amx((size_t)s.N), pP((size_t)s.N, 0), pR((size_t)s.N, 0);
            k::EpiPatch patch{};
            bool ok = true;
            int fail_c = -1;
            for (int c = 0; c < s.N && ok; ++c) {
                FmaForm f;
                const auto &r = acc_range_c[(size_t)c];
                ok = fma_form_search(A[(size_t)c], S[(size_t)c], s.u8 ? 0 : 128, lo, hi, r.first, r.second, f, nullptr, true);
                if (ok && f.patch_delta != 0) { // this channel's line needs one accumulator replaced by its neighbour
                    if (patch.n == k::EPI_PATCH_MAX) ok = false; // (more than the kernels' patch list holds: the operator keeps the two-rounding form)
                    else {
                        const int32_t P = wrap_add(wrap_add(k::MF_MAGIC_I, (int32_t)f.patch_acc), f.d);
                        patch.ch[patch.n] = c, patch.P[patch.n] = P, patch.R[patch.n] = P + f.patch_delta, ++patch.n;
                        pP[(size_t)c] = P, pR[(size_t)c] = P + f.patch_delta;
                    }
                }
                if (!ok) fail_c = c;
                A3[(size_t)c] = f.C, S3[(size_t)c] = f.S, piv[(size_t)c] = f.d, K3[(size_t)c] = wrap_add(Kc[(size_t)c], f.d);
                amn[(size_t)c] = (int32_t)r.first, amx[(size_t)c] = (int32_t)r.second;
            }
            unsigned long long nbad = 0;
            if (ok) {
                op->d_A3.upload(A3.data(), A3.size() * 4), op->d_S3.upload(S3.data(), S3.size() * 4), op->d_Kc3.upload(K3.data(), K3.size() * 4);
                DevBuf d_piv, d_amn, d_amx, d_bad, d_pP, d_pR;
                const std::vector<unsigned long long> zero((size_t)s.N, 0ull);
                d_piv.upload(piv.data(), piv.size() * 4), d_amn.upload(amn.data(), amn.size() * 4), d_amx.upload(amx.data(), amx.size() * 4);
                d_pP.upload(pP.data(), pP.size() * 4), d_pR.upload(pR.data(), pR.size() * 4);
                d_bad.upload(zero.data(), zero.size() * 8);
                std::vector<unsigned long long> bad((size_t)s.N, ~0ull);
                if (k::verify_fma_form(op->d_A.as<float>(), op->d_S.as<float>(), op->d_A3.as<float>(), op->d_S3.as<float>(), d_piv.as<int>(),
                                       d_amn.as<int>(), d_amx.as<int>(), d_pP.as<int>(), d_pR.as<int>(), s.N, (float)lo, (float)hi, s.u8,
                                       (unsigned long long *)d_bad.p, nullptr))
                    MF_HIP(hipMemcpy(bad.data(), d_bad.p, bad.size() * 8, hipMemcpyDeviceToHost));
                for (int c = 0; c < s.N; ++c)
                    if (bad[(size_t)c]) nbad += bad[(size_t)c], fail_c = c;
                ok = nbad == 0;
                if (!ok) // the host's exact arithmetic and the device disagree: that is a bug in one of them, say so -- and do not use the form
                    fprintf(stderr, "[microflow_amd] single-fma epilogue REJECTED by the device check (%llu accumulators differ, channel %d)\n", nbad, fail_c);
            }
            if (ok) op->fma_patch = patch;
            if (ok) op->fma_ok = true, op->h_A3 = A3, op->h_S3 = S3, op->h_Kc3 = K3;
            if (epi_dbg) {
                fprintf(stderr, "[epi] single-fma form: %s%s\n", ok ? ("all channels, " + std::to_string(patch.n) + " patched").c_str() : "no: channel ",
                        ok ? "" : std::to_string(fail_c).c_str());
                if (ok && patch.n) {
                    fprintf(stderr, "[epi]   patched channels:");
                    for (int e = 0; e < patch.n; ++e) fprintf(stderr, " %d", patch.ch[e]);
                    fprintf(stderr, "\n");
                }
            }
        }
        k::ConvArgs &a = op->conv;
        a.H = s.H, a.W = s.W, a.C = s.C, a.N = s.N, a.KH = s.KH, a.KW = s.KW, a.sh = s.sh, a.sw = s.sw;
        a.OH = s.OH, a.OW = s.OW, a.pad_same = s.pad == MF_PAD_SAME, a.izp = s.izp;
        a.lo_f = (float)lo, a.hi_f = (float)hi;
        a.w = op->d_w.as<int8_t>(), a.wzp = op->d_wzp.as<int>(), a.A = op->d_A.as<float>();
        a.S = op->d_S.as<float>(), a.Kc = op->d_Kc.as<int>();
        a.xr = xr;
        op->generic_name = dw ? "dwconv_generic" : "conv2d_generic";

        // (for u8 these are the shifted zero points: the fast kernels need wzp_u8 == 128)
        // ... and finite constants (their epilogue has no NaN test)
        op->finite_consts = all_finite(A) && all_finite(S);
        const bool zero_wzp = all_zero(wzp) && op->finite_consts;
        const bool same3x3 = s.KH == 3 && s.KW == 3 && s.pad == MF_PAD_SAME && s.sh == s.sw &&
                             s.OH == (s.H + s.sh - 1) / s.sh && s.OW == (s.W + s.sw - 1) / s.sw;
        const bool no_table = switches().no_table; // A-B: the run-time-geometry kernels on table shapes
        if (!no_table && dw && zero_wzp && same3x3 && s.C == s.N && k::dw_fast_name(s.H, s.W, s.C, s.sh)) {
            op->fast = OpImpl::DW_NHWC;
            op->fast_name = k::dw_fast_name(s.H, s.W, s.C, s.sh);
            k::DwFastArgs &f = op->dwf;
            f.w = a.w, f.A = a.A, f.S = a.S, f.Kc = a.Kc;
            f.izp4 = 0x01010101u * (uint32_t)(uint8_t)(int8_t)s.izp;
            f.lo_f = a.lo_f, f.hi_f = a.hi_f, f.magic = magic, f.xr = xr;
            f.wmm = nullptr, f.wsp = nullptr;
            f.queue = (int *)op->d_queue.p, f.qlaunch = &op->q_launches;
            if (s.C == 8 || s.C % 16 == 0) { // matrix-pipe form of the taps for the fused pair kernels
                const std::vector<int8_t> prep = build_dw_mm_weights(s.weights, s.C);
                op->d_wprep.upload(prep.data(), prep.size());
                f.wmm = op->d_wprep.p;
                const std::vector<int8_t> sp = build_dw_sp_weights(prep, s.C == 8 ? 1 : s.C / 16); // the sparse form of the same taps (k_quad.hip)
                if (!sp.empty()) {
                    op->d_wsp.upload(sp.data(), sp.size());
                    f.wsp = op->d_wsp.p;
                }
                if (dw_taps_on_matrix_pipe() && k::dw_mm_name(s.H, s.W, s.C, s.sh)) op->fast_name = k::dw_mm_name(s.H, s.W, s.C, s.sh);
            }
        } else if (!no_table && dw && zero_wzp && same3x3 && s.C == 1 && k::dw_stem_name(s.H, s.W, s.N, s.sh)) {
            op->fast = OpImpl::DW_STEM;
            op->fast_name = k::dw_stem_name(s.H, s.W, s.N, s.sh);
            k::DwStemArgs &f = op->stem;
            f.queue = (int *)op->d_queue.p, f.qlaunch = &op->q_launches;
            for (int ky = 0; ky < 3; ++ky)
                for (int c = 0; c < 8; ++c) {
                    uint32_t d = 0;
                    for (int kx = 0; kx < 3; ++kx)
                        d |= (uint32_t)(uint8_t)s.weights[((size_t)ky * 3 + kx) * s.N + c] << (8 * kx);
                    f.wrow[ky][c] = d;
                }
            // matrix-pipe form: accumulator row r = (p, c) = pixel 2j + p of a pixel pair, channel c; lane group g
            // = filter row; the lane's 8 K-bytes are input columns 4j-4 .. 4j+3 of that row, of which pixel 2j uses
            // bytes 3..5 and pixel 2j+1 bytes 5..7
            for (int lane = 0; lane < 64; ++lane) {
                const int r = lane & 15, g = lane >> 4, pp = r >> 3, c = r & 7;
                uint8_t b[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                if (g < 3)
                    for (int kx = 0; kx < 3; ++kx) b[3 + 2 * pp + kx] = (uint8_t)s.weights[((size_t)g * 3 + kx) * s.N + c];
                f.wmm[lane][0] = (uint32_t)b[0] | (uint32_t)b[1] << 8 | (uint32_t)b[2] << 16 | (uint32_t)b[3] << 24;
                f.wmm[lane][1] = (uint32_t)b[4] | (uint32_t)b[5] << 8 | (uint32_t)b[6] << 16 | (uint32_t)b[7] << 24;
            }
            for (int c = 0; c < 8; ++c) f.A[c] = A[c], f.S[c] = S[c], f.Kc[c] = Kc[c];
            f.izp4 = 0x01010101u * (uint32_t)(uint8_t)(int8_t)s.izp;
            f.lo_f = a.lo_f, f.hi_f = a.hi_f, f.magic = magic, f.xr = xr;
        } else if (!switches().no_rt && !switches().no_stem_rt && dw && zero_wzp && same3x3 && s.C == 1 && s.sh == 2 &&
                   k::dw_stem_rt_plan(op->stemrt, s.H, s.W, s.N, s.OH, s.OW)) {
            // a one-channel 3x3 stride-2 stem at any resolution: the taps as one MFMA per 256 output bytes (k_rt.hip)
            op->fast = OpImpl::DW_STEM_RT;
            op->fast_name = "dw3x3_stem_rt<" + std::to_string(s.N) + ">";
            k::DwStemRtArgs &f = op->stemrt;
            f.queue = (int *)op->d_queue.p, f.qlaunch = &op->q_launches;
            // operand A: accumulator row r = (p, c) = pixel p of a 16-byte output group, channel c; lane group g = filter row; the
            // lane's K-bytes are input columns XS j - 4 .. of that row, of which pixel p uses bytes 3 + 2 p .. 5 + 2 p
            const int KB = s.N == 8 ? 8 : 16;
            for (int lane = 0; lane < 64; ++lane) {
                const int r = lane & 15, g = lane >> 4, pp = r / s.N, c = r % s.N;
                uint8_t b[16] = {0};
                if (g < 3)
                    for (int kx = 0; kx < 3; ++kx) b[3 + 2 * pp + kx] = (uint8_t)s.weights[((size_t)g * 3 + kx) * s.N + c];
                for (int d = 0; d < 4; ++d)
                    f.wmm[lane][d] = d * 4 < KB ? ((uint32_t)b[4 * d] | (uint32_t)b[4 * d + 1] << 8 | (uint32_t)b[4 * d + 2] << 16 | (uint32_t)b[4 * d + 3] << 24) : 0u;
            }
            for (int c = 0; c < 8; ++c) f.A[c] = A[(size_t)(c % s.N)], f.S[c] = S[(size_t)(c % s.N)], f.Kc[c] = Kc[(size_t)(c % s.N)];
            f.izp4 = 0x01010101u * (uint32_t)(uint8_t)(int8_t)s.izp;
            f.lo_f = a.lo_f, f.hi_f = a.hi_f, f.magic = magic, f.xr = xr;
        } else if (dw && zero_wzp && s.C == 1 && s.N <= 8) {
            // one input channel, few output channels, any filter: LDS-staged direct kernel
            k::DwC1Args &f = op->dwc1;
            f.H = s.H, f.W = s.W, f.N = s.N, f.KH = s.KH, f.KW = s.KW, f.sh = s.sh, f.sw = s.sw;
            f.OH = s.OH, f.OW = s.OW, f.pad_same = s.pad == MF_PAD_SAME, f.izp = s.izp;
            f.lo_f = a.lo_f, f.hi_f = a.hi_f, f.A = a.A, f.S = a.S, f.Kc = a.Kc, f.magic = magic, f.xr = xr;
            f.KG = (s.KW + 3) / 4;
            // a window row is read as KG + 1 aligned dwords starting at (row start & ~3)
            f.TWP = ((((s.OW - 1) * s.sw + 3) & ~3) + 4 * (f.KG + 1) + 3) & ~3;
            if (k::dw_c1_supported(f)) {
                std::vector<uint32_t> wp((size_t)s.KH * f.KG * 8, 0);
                for (int ky = 0; ky < s.KH; ++ky)
                    for (int kx = 0; kx < s.KW; ++kx)
                        for (int c = 0; c < s.N; ++c)
                            wp[((size_t)ky * f.KG + kx / 4) * 8 + c] |=
                                (uint32_t)(uint8_t)s.weights[((size_t)ky * s.KW + kx) * s.N + c] << (8 * (kx & 3));
                op->d_wprep.upload(wp.data(), wp.size() * 4);
                f.wpack = op->d_wprep.as<uint32_t>();
                op->fast = OpImpl::DW_C1;
                op->fast_name = "dw_c1_lds";
            }
        } else if (!no_table && !dw && zero_wzp && s.KH == 1 && s.KW == 1 && s.sh == 1 && s.sw == 1 &&
                   s.OH == s.H && s.OW == s.W && k::pw_name(s.C, s.N) &&
                   (s.C != 8 || ((s.H * s.W) % 2 == 0))) {
            op->fast = OpImpl::PW_MFMA;
            op->fast_name = k::pw_name(s.C, s.N);
            const std::vector<int8_t> prep = build_pw_weights(s.weights, s.C, s.N);
            op->d_wprep.upload(prep.data(), prep.size());
            k::PwArgs &f = op->pw;
            f.wprep = op->d_wprep.p, f.A = a.A, f.S = a.S, f.Kc = a.Kc;
            f.lo_f = a.lo_f, f.hi_f = a.hi_f, f.magic = magic, f.xr = xr;
            f.wrr = nullptr;
            if ((s.C == 8 || s.C == 16 || s.C == 32) && (s.C == 8 ? 2 * s.N : s.N) % 16 == 0) {
                const std::vector<int8_t> rr = build_pw_rr_weights(s.weights, s.C, s.N);
                op->d_wrr.upload(rr.data(), rr.size());
                f.wrr = op->d_wrr.p;
            }
        }
        // shapes outside the tables: the run-time-geometry kernels (k_rt.hip), with or without weight zero points
        const bool no_rt = switches().no_rt; // tests / A-B: shape-generic kernels instead
        if (op->fast == OpImpl::NONE && !no_rt && op->finite_consts && dw && same3x3 && s.C == s.N &&
            k::dw_rt_plan(op->dwrt, s.H, s.W, s.C, s.sh, s.OH, s.OW)) {
            op->fast = OpImpl::DW_RT;
            op->rt_wz = !all_zero(wzp);
            op->fast_name = std::string("dw3x3_rt<") + std::to_string(s.sh) + (op->rt_wz ? ",wzp>" : ">");
            k::DwFastArgs &f = op->dwrt.dw;
            f.w = a.w, f.A = a.A, f.S = a.S, f.Kc = a.Kc, f.wmm = nullptr;
            f.izp4 = 0x01010101u * (uint32_t)(uint8_t)(int8_t)s.izp;
            f.lo_f = a.lo_f, f.hi_f = a.hi_f, f.magic = magic, f.xr = xr, f.queue = (int *)op->d_queue.p, f.qlaunch = &op->q_launches;
            op->dwrt.wzp = a.wzp;
            if (!op->rt_wz && (s.C % 16 == 0 || s.C == 8)) { // matrix-pipe form of the taps: what the fused chain kernel (k_chain.hip) multiplies
                const std::vector<int8_t> prep = build_dw_mm_weights(s.weights, s.C);
                op->d_wprep.upload(prep.data(), prep.size());
                f.wmm = op->d_wprep.p;
            }
        }
        if (op->fast == OpImpl::NONE && !no_rt && op->finite_consts && !dw && s.KH == 1 && s.KW == 1 && s.sh == 1 && s.sw == 1 &&
            s.OH == s.H && s.OW == s.W) {
            const bool wz = !all_zero(wzp);
            // K not a multiple of 16: `group` consecutive pixels form one row of the product (K = 8, 24, 40 ...: 2; K = 4, 12, 20 ...: 4)
            int group = s.C % 16 == 0 ? 1 : (s.C % 8 == 0 ? 2 : (s.C % 4 == 0 ? 4 : 0));
            // weights in registers (zero weight zero points, N * group <= 256): more pixels per row while the 64-deep k
            // step has room, so that the MFMA's k span is used and a row's output is a long contiguous run
            if (s.N % 4 != 0) group = 0; // packed dword results: N in whole fours (a 2-output head runs conv1x1_rowwave)
            bool reg = group >= 1 && !wz && s.N * group <= 256 && s.C * group <= 512;
            // (... but not past 64 output bytes per row: one wave then stores whole rows, 1 KiB contiguous per store
            // instruction, instead of two waves storing half lines -- MF_PW_RT_NCAP: tuning)
            const int ncap = switches().pw_rt_ncap;
            if (reg)
                while (s.C * group * 2 <= 64 && s.N * group * 2 <= std::max(ncap, s.N)) group *= 2;
            if (group >= 1 && !(wz && group > 1) && (reg || k::pw_rt_supported(s.C * group, s.N * group, wz))) {
                op->fast = OpImpl::PW_RT;
                op->rt_wz = wz, op->pw_group = group;
                op->fast_name = "pw_rt<" + std::to_string(s.C) + "," + std::to_string(s.N) + (wz ? ",wzp>" : ">");
                const int NTg = (s.N * group + 15) / 16;
                const int TB = reg ? (NTg < 4 ? NTg : 4) : 0;
                const int NBLK = reg ? (NTg + TB - 1) / TB : 0;            // <= 4 because N * group <= 256
                const int NSPLIT = NBLK <= 1 ? 1 : (NBLK == 2 ? 2 : 4);
                const std::vector<int8_t> prep = reg ? build_pw_rt_reg_weights(s.weights, s.C, s.N, group, TB, NSPLIT)
                                                     : build_pw_rt_weights(s.weights, s.C, s.N, group, wz);
                op->d_wprep.upload(prep.data(), prep.size());
                k::PwRtArgs &f = op->pwrt;
                f.wprep = op->d_wprep.p;
                f.K = s.C * group, f.N = s.N * group, f.KS = (f.K + 63) / 64, f.NT = (f.N + 15) / 16;
                f.patch_pitch = (f.N + 15) & ~15;
                f.TB = TB, f.NSPLIT = NSPLIT;
                f.lo_f = a.lo_f, f.hi_f = a.hi_f, f.magic = magic, f.xr = xr;
                if (group == 1) {
                    f.A = a.A, f.S = a.S, f.Kc = a.Kc, f.wzp = a.wzp;
                } else { // the constants of row gi * N + n are channel n's
                    std::vector<float> gA((size_t)f.N), gS((size_t)f.N);
                    std::vector<int32_t> gK((size_t)f.N), gZ((size_t)f.N);
                    for (int i = 0; i < f.N; ++i) gA[(size_t)i] = A[(size_t)(i % s.N)], gS[(size_t)i] = S[(size_t)(i % s.N)], gK[(size_t)i] = Kc[(size_t)(i % s.N)], gZ[(size_t)i] = wzp[(size_t)(i % s.N)];
                    op->d_rtA.upload(gA.data(), gA.size() * 4), op->d_rtS.upload(gS.data(), gS.size() * 4);
                    op->d_rtKc.upload(gK.data(), gK.size() * 4), op->d_rtwzp.upload(gZ.data(), gZ.size() * 4);
                    f.A = op->d_rtA.as<float>(), f.S = op->d_rtS.as<float>(), f.Kc = op->d_rtKc.as<int>(), f.wzp = op->d_rtwzp.as<int>();
                }
            }
        }
        // few input channels (a first convolution; a one-channel depthwise with more than 8 outputs): window rows as dwords
        // (a one-channel depthwise keeps dw_c1_lds only where the one-launch speech kernel builds on it, k_dwfc.hip;
        // measured on the 96x96 stem: dw_c1_lds 1.22 ms, conv_rows_lds 0.70 ms; MF_DW_C1=lds forces the old kernel)
        const bool c1_lds = switches().dw_c1_lds;
        const bool rows_for_c1 = !c1_lds && !k::dwfc_supported(s.H, s.W, s.KH, s.KW, s.sh, s.sw, s.OH, s.OW, s.N, 4);
        if ((op->fast == OpImpl::NONE || (op->fast == OpImpl::DW_C1 && rows_for_c1)) && !no_rt && op->finite_consts &&
            (!dw || s.C == 1) && !(s.KH == 1 && s.KW == 1 && !dw && k::conv1x1_rowwave_supported(a)) &&
            k::conv_rows_plan(op->crows, s.H, s.W, s.C, s.N, s.KH, s.KW, s.sh, s.sw, s.OH, s.OW, s.pad == MF_PAD_SAME)) {
            k::ConvRowsArgs &f = op->crows;
            std::vector<uint32_t> wp((size_t)s.KH * f.KG * f.NP, 0), mk((size_t)f.KG, 0);
            const int RWB = s.KW * s.C;
            for (int n = 0; n < s.N; ++n)
                for (int ky = 0; ky < s.KH; ++ky)
                    for (int b = 0; b < RWB; ++b) {
                        // conv filters [N][KH][KW][C]: byte b of row ky = (kx, c) in memory order; depthwise [KH][KW][N], C == 1
                        const int8_t wv = dw ? s.weights[((size_t)ky * s.KW + b) * s.N + n] : s.weights[((size_t)n * s.KH + ky) * RWB + b];
                        wp[((size_t)ky * f.KG + b / 4) * f.NP + n] |= (uint32_t)(uint8_t)wv << (8 * (b & 3));
                    }
            for (int b = 0; b < RWB; ++b) mk[(size_t)(b / 4)] |= 1u << (8 * (b & 3));
            op->d_crw.upload(wp.data(), wp.size() * 4), op->d_crm.upload(mk.data(), mk.size() * 4);
            f.wpack = op->d_crw.as<uint32_t>(), f.mask = op->d_crm.as<uint32_t>();
            f.A = a.A, f.S = a.S, f.Kc = a.Kc, f.wzp = a.wzp;
            f.izp4 = 0x01010101u * (uint32_t)(uint8_t)(int8_t)s.izp;
            f.lo_f = a.lo_f, f.hi_f = a.hi_f, f.magic = magic, f.xr = xr;
            op->fast = OpImpl::CONV_ROWS;
            op->rt_wz = !all_zero(wzp);
            op->fast_name = std::string(dw ? "dw_rows_lds" : "conv_rows_lds") + (op->rt_wz ? "<wzp>" : "");
        }
        // any other DepthwiseConv2D with C % 16 == 0 and one output per channel -- a filter other than 3x3, VALID padding, unequal
        // strides -- : the same kernel in its depthwise mode, taps of a 16-channel group on the matrix pipe against block-diagonal weights
        if (op->fast == OpImpl::NONE && !no_rt && op->finite_consts && dw && s.C == s.N && s.C % 16 == 0 && all_zero(wzp)) {
            std::vector<int> tap;
            if (k::conv_mm_plan(op->cmm, tap, s.H, s.W, s.C, s.N, s.KH, s.KW, s.sh, s.sw, s.OH, s.OW, s.pad == MF_PAD_SAME, false, true)) {
                k::ConvMmArgs &f = op->cmm;
                const std::vector<int8_t> prep = build_dw_mm_rt_weights(s.weights, s.KH, s.KW, s.C, f.KS);
                op->d_wprep.upload(prep.data(), prep.size());
                op->d_tap.upload(tap.data(), tap.size() * sizeof(int));
                f.wprep = op->d_wprep.p, f.tap_off = op->d_tap.as<int>();
                f.A = a.A, f.S = a.S, f.Kc = a.Kc, f.wzp = a.wzp;
                f.izp4 = 0x01010101u * (uint32_t)(uint8_t)(int8_t)s.izp;
                f.lo_f = a.lo_f, f.hi_f = a.hi_f, f.magic = magic, f.xr = xr;
                op->fast = OpImpl::CONV_MM;
                op->rt_wz = false;
                op->fast_name = "dw_mm_rt<" + std::to_string(s.KH) + "x" + std::to_string(s.KW) + ">";
            }
        }
        // any other Conv2D with C % 16 == 0: MFMA product over K = KH KW C with the image staged in LDS
        if (op->fast == OpImpl::NONE && !no_rt && op->finite_consts && !dw && !(s.KH == 1 && s.KW == 1 && k::conv1x1_rowwave_supported(a))) {
            const bool wz = !all_zero(wzp);
            std::vector<int> tap;
            if (k::conv_mm_plan(op->cmm, tap, s.H, s.W, s.C, s.N, s.KH, s.KW, s.sh, s.sw, s.OH, s.OW, s.pad == MF_PAD_SAME, wz)) {
                k::ConvMmArgs &f = op->cmm;
                const int Ktot = s.KH * s.KW * s.C;
                std::vector<int8_t> prep = build_pw_rt_reg_weights(s.weights, Ktot, s.N, 1, f.TB, f.NBLK); // [N][KH][KW][C] IS [N][K]
                if (wz) { // + a tile of ones over the real k-bytes
                    const size_t base_sz = prep.size();
                    prep.resize(base_sz + (size_t)f.KS * 1024, 0);
                    for (int ks = 0; ks < f.KS; ++ks)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int i = 0; i < 16; ++i)
                                if (ks * 64 + (lane >> 4) * 16 + i < Ktot) prep[base_sz + ((size_t)ks * 64 + lane) * 16 + i] = 1;
                }
                op->d_wprep.upload(prep.data(), prep.size());
                op->d_tap.upload(tap.data(), tap.size() * sizeof(int));
                f.wprep = op->d_wprep.p, f.tap_off = op->d_tap.as<int>();
                f.A = a.A, f.S = a.S, f.Kc = a.Kc, f.wzp = a.wzp;
                f.izp4 = 0x01010101u * (uint32_t)(uint8_t)(int8_t)s.izp;
                f.lo_f = a.lo_f, f.hi_f = a.hi_f, f.magic = magic, f.xr = xr;
                op->fast = OpImpl::CONV_MM;
                op->rt_wz = wz;
                op->fast_name = std::string("conv_mm_rt") + (wz ? "<wzp>" : "");
            }
        }
        if (op->fast == OpImpl::NONE && !dw && k::conv1x1_rowwave_supported(a)) // few outputs: one wavefront per pixel
            op->fast = OpImpl::CONV1X1_ROW, op->fast_name = "conv1x1_rowwave";
        if (op->fma_ok) { // the single-fma constants beside the two-rounding ones, in the blocks of the kernels that have the form
            op->dwf.A3 = op->pw.A3 = op->d_A3.as<float>(), op->dwf.S3 = op->pw.S3 = op->d_S3.as<float>();
            op->dwf.Kc3 = op->pw.Kc3 = op->d_Kc3.as<int>();
            op->dwf.npatch3 = op->pw.npatch3 = op->fma_patch.n;
            if (op->fast == OpImpl::DW_STEM && op->fma_strict()) {
                for (int c = 0; c < 8; ++c) op->stem.A3[c] = op->h_A3[(size_t)c], op->stem.S3[c] = op->h_S3[(size_t)c], op->stem.Kc3[c] = op->h_Kc3[(size_t)c];
                op->stem.fma_ok = 1;
            }
        }
        if (switches().verbose)
            fprintf(stderr, "[microflow_amd] %s %dx%dx%d -> %d: kernel %s, worst-case |acc| %lld%s\n",
                    dw ? "depthwise_conv_2d" : "conv_2d", s.H, s.W, s.C, s.N,
                    op->fast != OpImpl::NONE ? op->fast_name.c_str() : op->generic_name.c_str(),
                    (long long)acc_bound, op->fast == OpImpl::NONE || !magic ? "" : magic == 2 ? " (bit-pattern int->f32, saturating pack)" : " (bit-pattern int->f32)");
        break;
    }
    case MF_OP_AVERAGE_POOL_2D: {
        if (s.H <= 0 || s.W <= 0 || s.C <= 0 || s.KH <= 0 || s.KW <= 0 || s.OH <= 0 || s.OW <= 0 ||
            s.sh <= 0 || s.sw <= 0)
            fail(MF_ERR_INVALID_ARG, "average_pool_2d: bad arguments");
        if (s.pad == MF_PAD_VALID &&
            ((s.OH - 1) * s.sh + s.KH > s.H || (s.OW - 1) * s.sw + s.KW > s.W))
            fail(MF_ERR_INVALID_ARG, "average_pool_2d: VALID view leaves the input");
        op->in_elems = (size_t)s.H * s.W * s.C;
        op->out_elems = (size_t)s.OH * s.OW * s.C;
        k::PoolArgs &a = op->pool;
        a.H = s.H, a.W = s.W, a.C = s.C, a.KH = s.KH, a.KW = s.KW, a.sh = s.sh, a.sw = s.sw;
        a.OH = s.OH, a.OW = s.OW, a.pad_same = s.pad == MF_PAD_SAME;
        a.c0 = s.pool_c0, a.c1 = s.pool_c1, a.lo = lo, a.hi = hi;
        a.bias = beta, a.xr = xr;
        a.sat_lo = s.u8 ? 0.0f : -128.0f, a.sat_hi = s.u8 ? 255.0f : 127.0f;
        op->generic_name = "avgpool_generic";
        if (s.C % 4 == 0) op->fast = OpImpl::POOL_C4, op->fast_name = "avgpool_c4"; // 4 channels per thread, dword loads
        break;
    }
    case MF_OP_FULLY_CONNECTED: {
        if (s.M <= 0 || s.K <= 0 || s.N <= 0 || !s.weights || !s.c0 || !s.c2)
            fail(MF_ERR_INVALID_ARG, "fully_connected: bad arguments");
        op->in_elems = (size_t)s.M * s.K;
        op->out_elems = (size_t)s.M * s.N;
        const int wzp_t = s.wzp ? s.wzp[0] : 0; // in T's domain
        if (!s.wzp) s.nq = 0;
        to_i8_domain((size_t)s.N * s.K);
        std::vector<float> A(s.N);
        std::vector<int32_t> Kc(s.N);
        for (int j = 0; j < s.N; ++j) {
            volatile float a = (float)s.ozp + s.c0[j];
            A[j] = a;
            Kc[j] = wrap_sub(s.c3, s.c2[j]); // acc = x0 - x1 - c2[j] + c3
            if (s.u8) {
                // with x = x' + 128, w = w' + 128:  x0 - x1 = sum x'w' - (wzp - 128) sum x'
                //                                   + 128 sum_k w'[j][k] + K 128 (128 - wzp)
                int32_t ws = 0;
                for (int k = 0; k < s.K; ++k) ws = wrap_add(ws, s.weights[(size_t)j * s.K + k]);
                Kc[j] = wrap_add(Kc[j], wrap_add(wrap_mul(beta, ws),
                                                 wrap_mul(wrap_mul(s.K, beta), beta - wzp_t)));
            }
        }
        op->d_w.upload(s.weights, (size_t)s.N * s.K);
        op->d_A.upload(A.data(), A.size() * 4);
        op->d_Kc.upload(Kc.data(), Kc.size() * 4);
        k::FcArgs &a = op->fc;
        a.K = s.K, a.N = s.N, a.wzp = wzp_t - beta, a.S = s.c1[0];
        a.lo_f = (float)lo, a.hi_f = (float)hi, a.xr = xr;
        a.w = op->d_w.as<int8_t>(), a.A = op->d_A.as<float>(), a.Kc = op->d_Kc.as<int>();
        op->generic_name = "fc_generic";
        const bool finite = all_finite(A) && std::isfinite(a.S);
        if (!finite) {
            // degenerate constants: the generic kernel reproduces Rust's NaN -> 0 cast
        } else if (s.K % 16 == 0 && s.K >= 256 && (s.N == 1 || s.N == 2 || s.N == 4 || s.N == 8)) {
            op->fast = OpImpl::FC_ROWWAVE;
            op->fast_name = "fc_rowwave<" + std::to_string(s.N) + ">";
        } else if (s.N % 128 == 0 && s.K % 128 == 0) {
            // dense contraction: int8 MFMA GEMM whenever the batch supplies whole 128-row tiles
            op->fast = OpImpl::FC_MFMA;
            op->fast_name = "fc_mfma";
        }
        break;
    }
    case MF_OP_SOFTMAX: {
        if (s.M <= 0 || s.N <= 0) fail(MF_ERR_INVALID_ARG, "softmax: bad arguments");
        op->in_elems = op->out_elems = (size_t)s.M * s.N;
        // exp table over the 256 possible int8 inputs: expf(f32(q) * input.scale[0])
        // (src/ops/softmax.rs:20-21), libm's algorithm evaluated on the host
        // (entry = stored byte + 128, which is q + 128 for i8 and q itself for u8)
        std::vector<float> table(256);
        h_softmax_table(s.in_scale, s.u8, table.data());
        op->d_table.upload(table.data(), 256 * 4);
        k::SoftmaxArgs &a = op->sm;
        a.rows = s.M, a.cols = s.N, a.oscale = s.oscale, a.ozp_f = (float)s.ozp;
        a.exp_table = op->d_table.as<float>();
        a.sat_lo = s.u8 ? 0.0f : -128.0f, a.sat_hi = s.u8 ? 255.0f : 127.0f, a.xr = xr;
        op->generic_name = "softmax_table";
        break;
    }
    default:
        fail(MF_ERR_UNSUPPORTED, "unsupported operator kind " + std::to_string(s.kind));
    }
    op->s = s;
    // the spec's host pointers die with the caller
    op->s.weights = nullptr, op->s.wzp = nullptr, op->s.c0 = op->s.c1 = nullptr, op->s.c2 = nullptr;
    return op.release();
}

void op_destroy(OpImpl *op) {
    if (!op) return;
    (void)hipSetDevice(op->device);
    if (op->scratch_ev) {
        (void)hipEventSynchronize(op->scratch_ev);
        (void)hipEventDestroy(op->scratch_ev);
    }
    if (op->d_rowsum) (void)hipFree(op->d_rowsum);
    if (op->d_ext) (void)hipFree(op->d_ext);
    delete op;
}

size_t op_in_elems(const OpImpl *op) { return op->in_elems; }
size_t op_out_elems(const OpImpl *op) { return op->out_elems; }
const char *op_kernel_name(const OpImpl *op) {
    return (op->fast != OpImpl::NONE && !op->force_generic) ? op->fast_name.c_str()
                                                             : op->generic_name.c_str();
}
void op_set_generic(OpImpl *op, bool g) { op->force_generic = g; }
// the layer-wise kernels that have the single-fma epilogue (k_common.hpp mode 3): the matrix-pipe depthwise, the MFMA pointwise, the stem
static bool op_runs_fma(const OpImpl *op) {
    if (!op->fma_strict()) return false; // (the layer-wise kernels have no patch support)
    const OpSpec &sp = op->s;
    const bool stem_valu = switches().stem_valu;
    return (op->fast == OpImpl::DW_NHWC && dw_taps_on_matrix_pipe() && op->dwf.wmm && k::dw_mm_name(sp.H, sp.W, sp.C, sp.sh)) ||
           op->fast == OpImpl::PW_MFMA || (op->fast == OpImpl::DW_STEM && !stem_valu);
}
int op_epilogue_mode(const OpImpl *op) { // of the operator's own (layer-wise) launch; -1: not a convolution-like operator
    if (op->s.kind != MF_OP_CONV_2D && op->s.kind != MF_OP_DEPTHWISE_CONV_2D) return -1;
    return op_runs_fma(op) ? 3 : op->magic_mode;
}
bool op_has_fma_epilogue(const OpImpl *op) { return op->fma_ok; }
int op_fma_patches(const OpImpl *op) { return op->fma_ok ? op->fma_patch.n : -1; }

void op_run(OpImpl *op, const int8_t *d_in, size_t batch, int8_t *d_out, void *stream) {
    if (!batch) return;
    if (!d_in || !d_out) fail(MF_ERR_INVALID_ARG, "op_run: null device pointer");
    hipStream_t s = (hipStream_t)stream;
    const OpSpec &sp = op->s;
    // the fast kernels move 4- and 16-byte words (vector loads, LDS-DMA, packed stores): buffers that are not 16-byte
    // aligned -- an offset into a larger allocation handed to mf_op_run -- take the byte-wise shape-generic kernels
    const bool aligned = (((uintptr_t)d_in | (uintptr_t)d_out) & 15) == 0;
    const bool fast = op->fast != OpImpl::NONE && !op->force_generic && aligned;
    bool done = false;
    if (fast) {
        if (batch > 0x7fffffffull / 4) fail(MF_ERR_INVALID_ARG, "batch too large for one launch");
        switch (op->fast) {
        case OpImpl::DW_NHWC: {
            k::DwFastArgs f3 = op->dwf;
            const bool fma = op_runs_fma(op) && f3.use_fma();
            done = (dw_taps_on_matrix_pipe() && k::launch_dw_mm(sp.H, sp.W, sp.C, sp.sh, d_in, d_out, fma ? f3 : op->dwf, (int)batch, s)) ||
                   k::launch_dw_fast(sp.H, sp.W, sp.C, sp.sh, d_in, d_out, op->dwf, (int)batch, s);
            break;
        }
        case OpImpl::DW_STEM: {
            k::DwStemArgs f3 = op->stem;
            const bool fma = op_runs_fma(op) && f3.use_fma();
            done = k::launch_dw_stem(sp.H, sp.W, sp.N, sp.sh, d_in, d_out, fma ? f3 : op->stem, (int)batch, s);
            break;
        }
        case OpImpl::DW_STEM_RT:
            k::launch_dw_stem_rt(d_in, d_out, op->stemrt, (int)batch, s);
            done = true;
            break;
        case OpImpl::CONV1X1_ROW:
            k::launch_conv1x1_rowwave(d_in, d_out, op->conv, batch, s);
            done = true;
            break;
        case OpImpl::POOL_C4:
            k::launch_avgpool_c4(d_in, d_out, op->pool, batch, s);
            done = true;
            break;
        case OpImpl::DW_C1:
            k::launch_dw_c1(d_in, d_out, op->dwc1, batch, s);
            done = true;
            break;
        case OpImpl::PW_MFMA: {
            k::PwArgs f3 = op->pw;
            const bool fma = op_runs_fma(op) && f3.use_fma();
            done = k::launch_pw(sp.C, sp.N, d_in, d_out, fma ? f3 : op->pw, (long long)batch * sp.H * sp.W, s);
            break;
        }
        case OpImpl::CONV_MM:
            if (op->cmm.dwise) k::launch_dw_mm(d_in, d_out, op->cmm, (int)batch, s);
            else k::launch_conv_mm(d_in, d_out, op->cmm, op->rt_wz, (int)batch, s);
            done = true;
            break;
        case OpImpl::CONV_ROWS:
            k::launch_conv_rows(d_in, d_out, op->crows, op->rt_wz, (int)batch, s);
            done = true;
            break;
        case OpImpl::DW_RT:
            k::launch_dw_rt(d_in, d_out, op->dwrt, sp.sh, op->rt_wz, (int)batch, s);
            done = true;
            break;
        case OpImpl::PW_RT: {
            // `pw_group` pixels are one row of the product; the few pixels left over when the pixel count is not a
            // multiple of it go through the shape-generic kernel as one short 1 x rem image
            const long long npix = (long long)batch * sp.H * sp.W, full = npix / op->pw_group * op->pw_group;
            if (full) k::launch_pw_rt(d_in, d_out, op->pwrt, op->rt_wz, full / op->pw_group, s);
            if (npix > full) {
                k::ConvArgs t = op->conv;
                t.H = t.OH = 1, t.W = t.OW = (int)(npix - full);
                k::launch_conv2d_generic(d_in + full * sp.C, d_out + full * sp.N, t, 1, s);
            }
            done = true;
            break;
        }
        case OpImpl::FC_ROWWAVE:
            done = k::launch_fc_rowwave(d_in, d_out, op->fc, batch * sp.M, s);
            break;
        case OpImpl::FC_MFMA: {
            const size_t rows = batch * sp.M;
            if (!k::fc_mfma_supported(rows, sp.N, sp.K)) break; // fewer than 64 rows: generic kernel
            k::FcGemmArgs g{};
            g.w = op->fc.w, g.A = op->fc.A, g.Kc = op->fc.Kc, g.wzp = op->fc.wzp, g.S = op->fc.S;
            g.lo_f = op->fc.lo_f, g.hi_f = op->fc.hi_f, g.M = (int)rows, g.N = sp.N, g.K = sp.K;
            g.xr4 = 0x01010101u * (uint32_t)op->fc.xr;
            g.rowsum = nullptr, g.rs_sums = nullptr, g.rs_sync = nullptr;
            // the weight-zero-point term needs sum_k x[m][k]: formed by the GEMM launch itself (its prologue) where the shape
            // allows, else by a pre-pass launch; either way in the operator's one row-sum scratch (+ the row tiles' counter pairs
            // behind it), whose uses the event handshake serialises across streams
            const bool inlaunch = op->fc.wzp != 0 && k::fc_mfma_rowsum_prologue(rows, sp.N);
            const bool prepass = op->fc.wzp != 0 && (inlaunch || k::fc_mfma_rowsum_prepass());
            if (prepass) {
                const size_t ntm = (rows + 255) / 256, need = rows + 2 * ntm;
                op->scratch_acquire(s, op->rowsum_cap < need);
                if (op->rowsum_cap < need) {
                    if (op->d_rowsum) (void)hipFree(op->d_rowsum);
                    op->d_rowsum = nullptr, op->rowsum_cap = 0;
                    MF_HIP(hipMalloc((void **)&op->d_rowsum, need * sizeof(int)));
                    MF_HIP(hipMemset(op->d_rowsum, 0, need * sizeof(int))); // (the counter pairs start at zero and every launch leaves them so)
                    op->rowsum_cap = need;
                    op->rowsum_rows = 0;
                }
                if (inlaunch && op->rowsum_rows != rows) { // the counters sit behind the sums of THIS row count
                    if (op->rowsum_rows) MF_HIP(hipMemsetAsync(op->d_rowsum, 0, op->rowsum_cap * sizeof(int), s));
                    op->rowsum_rows = rows;
                }
                if (inlaunch) g.rs_sums = op->d_rowsum, g.rs_sync = op->d_rowsum + rows;
                else k::launch_fc_rowsum(d_in, op->d_rowsum, rows, sp.K, s), g.rowsum = op->d_rowsum;
            }
            k::launch_fc_mfma(d_in, d_out, g, s);
            if (prepass) op->scratch_release(s);
            done = true;
            break;
        }
        default: break;
        }
    }
    if (!done) {
        switch (sp.kind) {
        case MF_OP_CONV_2D: k::launch_conv2d_generic(d_in, d_out, op->conv, batch, s); break;
        case MF_OP_DEPTHWISE_CONV_2D: k::launch_dwconv_generic(d_in, d_out, op->conv, batch, s); break;
        case MF_OP_AVERAGE_POOL_2D: k::launch_avgpool_generic(d_in, d_out, op->pool, batch, s); break;
        case MF_OP_FULLY_CONNECTED: k::launch_fc_generic(d_in, d_out, op->fc, batch * sp.M, s); break;
        case MF_OP_SOFTMAX: k::launch_softmax(d_in, d_out, op->sm, batch, s); break;
        default: fail(MF_ERR_UNSUPPORTED, "op_run: bad kind");
        }
    }
    MF_HIP(hipGetLastError());
}

// The C ABI's mf_op_run: a u8 operator takes and returns real u8 bytes.
void op_run_external(OpImpl *op, const int8_t *d_in, size_t batch, int8_t *d_out, void *stream) {
    MF_HIP(hipSetDevice(op->device)); // the operator's buffers and kernels live on its device
    if (!op->s.u8) return op_run(op, d_in, batch, d_out, stream);
    if (!batch) return;
    if (!d_in || !d_out) fail(MF_ERR_INVALID_ARG, "op_run: null device pointer");
    hipStream_t s = (hipStream_t)stream;
    const size_t n_in = batch * op->in_elems;
    // (two scratch buffers may be in play -- d_ext here, d_rowsum inside op_run for an FC with a weight zero point -- behind one
    // event: op_run's own acquire then waits for nothing new, its release is superseded by the one below)
    op->scratch_acquire(s, op->ext_cap < n_in);
    if (op->ext_cap < n_in) {
        if (op->d_ext) (void)hipFree(op->d_ext);
        op->d_ext = nullptr, op->ext_cap = 0;
        MF_HIP(hipMalloc((void **)&op->d_ext, n_in + 256));
        op->ext_cap = n_in;
    }
    k::launch_xor80(d_in, op->d_ext, n_in, s);
    op_run(op, op->d_ext, batch, d_out, stream);
    op->scratch_release(s); // op_run's kernels were the last readers of d_ext
    k::launch_xor80(d_out, d_out, batch * op->out_elems, s);
    MF_HIP(hipGetLastError());
}

// Boundary quantisation fused into the first operator (M::predict on f32 input): only the stem
// kernel has an f32-input variant.  `zp` is a value of T.
// quant_div (k_common.hpp): true when the fast form gives the same byte as the true division for EVERY float input
static bool quant_div_verified(int device, float scale, float rcp, float zp_f, float sat_lo, float sat_hi) {
    const bool off = switches().no_fast_quant_div;
    if (off || !std::isfinite(scale) || !std::isfinite(rcp) || scale == 0.0f) return false;
    static std::mutex mu;
    static std::map<std::array<uint32_t, 4>, bool> cache;
    uint32_t kb[4];
    memcpy(&kb[0], &scale, 4), memcpy(&kb[1], &zp_f, 4), memcpy(&kb[2], &sat_lo, 4), memcpy(&kb[3], &sat_hi, 4);
    const std::array<uint32_t, 4> key{kb[0], kb[1], kb[2], kb[3]};
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    MF_HIP(hipSetDevice(device));
    const unsigned long long bad = k::verify_quant_div(scale, rcp, zp_f, sat_lo, sat_hi, nullptr);
    if (bad == ~0ull) return false; // the check itself could not run: keep the true division now, try again next time
    const bool ok = bad == 0;
    if (switches().verbose) fprintf(stderr, "[microflow_amd] boundary quantisation: 3-instruction division %s for scale %g\n", ok ? "verified" : "REJECTED", (double)scale);
    cache[key] = ok;
    return ok;
}
bool op_set_input_quant(OpImpl *op, float scale, int zp, bool u8) {
    if (op->fast != OpImpl::DW_STEM || !(scale == scale)) return false;
    k::DwStemArgs &f = op->stem;
    f.in_scale = scale, f.in_zp_f = (float)zp;
    f.in_sat_lo = u8 ? 0.0f : -128.0f, f.in_sat_hi = u8 ? 255.0f : 127.0f;
    f.in_xr4 = u8 ? 0x80808080u : 0u;
    // the 3-instruction division of the boundary quantisation, if it is exact for these parameters (checked over all
    // 2^32 inputs on the device, a few ms, once per parameter set and process)
    f.in_rcp = (float)(1.0 / (double)scale);
    f.in_fast = quant_div_verified(op->device, scale, f.in_rcp, f.in_zp_f, f.in_sat_lo, f.in_sat_hi) ? 1 : 0;
    op->accepts_f32 = true;
    return true;
}
bool op_accepts_f32(const OpImpl *op) { return op->accepts_f32 && !op->force_generic; }
void op_run_f32(OpImpl *op, const float *d_in, size_t batch, int8_t *d_out, void *stream) {
    if (!batch) return;
    if (!op_accepts_f32(op)) fail(MF_ERR_UNSUPPORTED, "operator has no f32-input kernel");
    if (!d_in || !d_out) fail(MF_ERR_INVALID_ARG, "op_run_f32: null device pointer");
    if (batch > 0x7fffffffull / 4) fail(MF_ERR_INVALID_ARG, "batch too large for one launch");
    const OpSpec &sp = op->s;
    // (never the single-fma epilogue here: that form runs its kernel in round-toward-zero, and this kernel's boundary quantisation
    // needs round-to-nearest)
    if (!k::launch_dw_stem(sp.H, sp.W, sp.N, sp.sh, (const int8_t *)d_in, d_out, op->stem, (int)batch,
                           (hipStream_t)stream, true))
        fail(MF_ERR_UNSUPPORTED, "f32 stem kernel missing");
    MF_HIP(hipGetLastError());
}

struct FusedImpl {
    enum Kind { DWPW, TAIL, FCSM, STAGE, DWFC, PAIRTAIL, QUAD, CHAIN } kind;
    OpImpl *a, *b, *c;
    k::DwPwArgs dwpw;
    k::TailArgs tail;
    std::string name;
    // STAGE: the whole late stage in one kernel
    k::StageArgs stage{};
    int stage_pairs = 0;
    std::vector<std::unique_ptr<DevBuf>> stage_w; // the stage kernel's own operand arrays and its pair table
    // DWFC: one-input-channel depthwise -> FullyConnected -> Softmax in one kernel (operand tables in stage_w)
    k::DwFcArgs dwfc{};
    // PAIRTAIL: the last pair + the tail in one kernel (operand arrays in stage_w)
    k::PairTailArgs pairtail{};
    k::PairFrontArgs pairfront{}; // ... with the pair in front of it in the same launch (has_front; k_tail3.hip FRONT)
    bool has_front = false;
    // QUAD: two consecutive pairs in one kernel (k_quad.hip); a = the first pair's depthwise, b = the second pair's conv
    k::QuadArgs quad{};
    bool quad_mm = false; // the C = 64 quad (k_quad_mm.hip): intermediate tensors through LDS, the pairs' dwpw_mm argument blocks
    int quad_shape[10] = {0};
    OpImpl *quad_ops[4] = {nullptr, nullptr, nullptr, nullptr}; // the two pairs' operators (the stem variant rebuilds the blocks from them)
    // CHAIN: 1 .. CHAIN_MAX consecutive pairs of any geometry in one launch (k_chain.hip); table and weights in stage_w
    k::ChainArgs chain{};
    std::vector<std::pair<OpImpl *, OpImpl *>> chain_members;
    int epi_mode = -1; // epilogue mode (k_common.hpp) of the launch's requantising operators; -1: not recorded (the operators' minimum)
};
// the pair's argument blocks as the operators hold them (two-rounding constants), and switched to the single-fma form when
// both operators have it
// fma: 0 = the two-rounding constants; 1 = the single-fma form if both operators have it WITHOUT patched accumulators (quads,
// register-resident pairs); 2 = ... with or without (dwpw_mm, the stage: kernels that apply the patch list)
static k::DwPwArgs pair_args(const OpImpl *dw, const OpImpl *pw, int fma) {
    k::DwPwArgs a;
    a.dw = dw->dwf, a.pw = pw->pw;
    if ((fma == 1 && dw->fma_strict() && pw->fma_strict()) || (fma == 2 && dw->fma_ok && pw->fma_ok)) a.dw.use_fma(true), a.pw.use_fma(true);
    return a;
}
static int pair_mode(const k::DwPwArgs &a) { return std::min(a.dw.magic, a.pw.magic); }
// An operator's patch list as the table a kernel reads: one EpiPatchRec per tile, `tile_of(channel, reg, lane_group)` = the tile
// index of the channel in that kernel (and which of a lane's four accumulators / which 16-lane group hold it).  false: two patched
// channels share a tile -- the kernel's record holds one -- so this launch cannot use the single-fma form.
template <typename F> static bool patch_table(const k::EpiPatch &pl, std::vector<k::EpiPatchRec> &tab, size_t base, F tile_of) {
    for (int e = 0; e < pl.n; ++e) {
        int reg = 0, grp = 0;
        const int t = tile_of(pl.ch[e], reg, grp);
        if (t < 0 || base + (size_t)t >= tab.size() || tab[base + (size_t)t].P != 0) return false;
        tab[base + (size_t)t] = k::epi_patch_rec(pl.P[e], pl.R[e], reg, grp);
    }
    return true;
}

// ---- run-time-geometry chains (k_chain.hip) ----
static bool chain_enabled() {
    const bool off = switches().no_chain;
    return !off;
}
// a DepthwiseConv2D 3x3 + Conv2D 1x1 pair the chain kernel can run: any H / W, C % 16 == 0, N % 16 == 0
static int chain_superpixel(int C) {
    const bool no_sp = switches().chain_no_sp; // A/B: only C == 8 stride 1 (round 4's first form) below 16 channels
    if (C % 16 == 0) return 1;
    if (C == 8 || (!no_sp && (C == 4 || C == 2))) return 16 / C;
    return 0;
}
static bool chain_pair_ok(const OpImpl *dw, const OpImpl *pw) {
    if (!dw || !pw || dw->device != pw->device || dw->force_generic || pw->force_generic) return false;
    const OpSpec &d = dw->s, &q = pw->s;
    if (d.kind != MF_OP_DEPTHWISE_CONV_2D || q.kind != MF_OP_CONV_2D || d.u8 != q.u8) return false;
    if (dw->fast != OpImpl::DW_RT && dw->fast != OpImpl::DW_NHWC) return false;
    const k::DwFastArgs &f = dw->fast == OpImpl::DW_NHWC ? dw->dwf : dw->dwrt.dw;
    const int P = chain_superpixel(d.C);  // pixels per 16-channel "superpixel" (1: C % 16 == 0; 0: no such form)
    if ((P == 1 && !f.wmm) || (dw->fast == OpImpl::DW_RT && dw->rt_wz)) return false;
    if (d.KH != 3 || d.KW != 3 || d.pad != MF_PAD_SAME || d.sh != d.sw || (d.sh != 1 && d.sh != 2) || d.C != d.N) return false;
    if (P == 0 || d.W % (P * d.sh) != 0) return false; // (C < 16: whole superpixels in and out, see chain_geom)
    if (P > 1 && d.sh == 2 && switches().chain_no_sp) return false;
    if (q.KH != 1 || q.KW != 1 || q.sh != 1 || q.sw != 1 || q.OH != q.H || q.OW != q.W || (P * q.N) % 16 != 0) return false;
    if (q.H != d.OH || q.W != d.OW || q.C != d.N) return false;
    if (pw->fast != OpImpl::PW_RT && pw->fast != OpImpl::PW_MFMA) return false;
    if (pw->fast == OpImpl::PW_RT && pw->rt_wz) return false;
    if (!dw->finite_consts || !pw->finite_consts) return false;
    if ((dw->magic_mode < 1 || pw->magic_mode < 1) && q.C < 256) return false; // (the v_cvt epilogue exists for four k steps only: launch_chain)
    return true;
}
// The geometry the planner sees.  C < 16 (the first pairs of a MobileNet-v1-shaped network: C = 8, or 4 and 8 at width 0.5): P = 16 / C
// adjacent pixels form one 16-channel "superpixel" -- the depthwise taps are build_dw_mm_weights_sp (MFMA rows = (pixel of the
// superpixel, channel), filter-row blocks = neighbouring superpixels, either stride), the 1x1 convolution is block diagonal over the P
// pixels -- so the pair IS a 16-channel pair of 1/P the width with P times the outputs.  Such a pair only ever runs alone (its output
// tensor is not in the next pair's units).
static k::ChainGeom chain_geom(const OpImpl *dw, const OpImpl *pw) {
    const OpSpec &d = dw->s;
    const k::DwFastArgs &f = dw->fast == OpImpl::DW_NHWC ? dw->dwf : dw->dwrt.dw;
    const int P = chain_superpixel(d.C);
    if (P > 1) return k::ChainGeom{d.H, d.W / P, 16, d.sh, d.OH, d.OW / P, P * pw->s.N, f.izp4};
    return k::ChainGeom{d.H, d.W, d.C, d.sh, d.OH, d.OW, pw->s.N, f.izp4};
}
static FusedImpl *chain_create(const std::pair<OpImpl *, OpImpl *> *mem, int n, int force_G = 0, int force_dbuf = -1) {
    if (!chain_enabled() || n < 1 || n > k::CHAIN_MAX) return nullptr;
    std::vector<k::ChainGeom> geo((size_t)n);
    for (int i = 0; i < n; ++i) {
        if (!chain_pair_ok(mem[i].first, mem[i].second)) {
            if (n == 1 && switches().chain_verbose)
                fprintf(stderr, "[microflow_amd] not a chain pair: %dx%dx%d s%d (kernels %s, %s; epilogue modes %d, %d)\n", mem[i].first->s.H, mem[i].first->s.W,
                        mem[i].first->s.C, mem[i].first->s.sh, mem[i].first->fast_name.c_str(), mem[i].second->fast_name.c_str(), mem[i].first->magic_mode, mem[i].second->magic_mode);
            return nullptr;
        }
        if (mem[i].first->device != mem[0].first->device || mem[i].first->s.u8 != mem[0].first->s.u8) return nullptr;
        if (mem[i].first->s.C < 16 && n != 1) return nullptr;
        geo[(size_t)i] = chain_geom(mem[i].first, mem[i].second);
    }
    std::vector<k::ChainPair> tab((size_t)n);
    std::unique_ptr<FusedImpl> c(new FusedImpl{FusedImpl::CHAIN, mem[0].first, mem[n - 1].second, nullptr, {}, {}, ""});
    if (!k::chain_plan(geo.data(), n, tab.data(), c->chain, 150 * 1024, force_G, force_dbuf)) {
        if (n == 1 && force_G == 0 && switches().chain_verbose)
            fprintf(stderr, "[microflow_amd] no chain plan for %dx%dx%d s%d -> %d\n", geo[0].H, geo[0].W, geo[0].C, geo[0].S, geo[0].N);
        return nullptr;
    }
    int magic = 2;
    std::string name = "chain_rt<";
    for (int i = 0; i < n; ++i) {
        OpImpl *dw = mem[i].first, *pw = mem[i].second;
        const k::DwFastArgs &f = dw->fast == OpImpl::DW_NHWC ? dw->dwf : dw->dwrt.dw;
        k::ChainPair &t = tab[(size_t)i];
        t.dw_wmm = f.wmm, t.dwA = f.A, t.dwS = f.S, t.dwK = f.Kc, t.dw_lo = f.lo_f, t.dw_hi = f.hi_f;
        const OpSpec &q = pw->s;
        const int group = chain_superpixel(dw->s.C); // pixels per MFMA column / product row
        std::vector<int8_t> host((size_t)q.N * q.C);
        MF_HIP(hipMemcpy(host.data(), pw->conv.w, host.size(), hipMemcpyDeviceToHost)); // [N][1][1][C], i8 domain, as uploaded
        const std::vector<int8_t> prep = build_pw_rt_reg_weights(host.data(), q.C, q.N, group, t.TB, t.NBLK);
        c->stage_w.emplace_back(new DevBuf);
        c->stage_w.back()->upload(prep.data(), prep.size());
        t.pw_w = c->stage_w.back()->p;
        t.pwA = pw->conv.A, t.pwS = pw->conv.S, t.pwK = pw->conv.Kc, t.pw_lo = pw->conv.lo_f, t.pw_hi = pw->conv.hi_f;
        if (group > 1) {
            // the taps in superpixel form (from the depthwise weights as uploaded: [3][3][C], i8 domain)
            std::vector<int8_t> hw((size_t)9 * dw->s.C);
            MF_HIP(hipMemcpy(hw.data(), dw->conv.w, hw.size(), hipMemcpyDeviceToHost));
            const std::vector<int8_t> sp = build_dw_mm_weights_sp(hw.data(), dw->s.C, dw->s.sh);
            c->stage_w.emplace_back(new DevBuf);
            c->stage_w.back()->upload(sp.data(), sp.size());
            t.dw_wmm = c->stage_w.back()->p;
            // the constants of MFMA row (pixel of the superpixel, channel) are the channel's: `group` copies of the arrays
            auto copies = [&](const void *d_src, int count) {
                std::vector<int32_t> h((size_t)group * count);
                MF_HIP(hipMemcpy(h.data(), d_src, (size_t)count * 4, hipMemcpyDeviceToHost));
                for (int e = count; e < group * count; ++e) h[(size_t)e] = h[(size_t)(e % count)];
                c->stage_w.emplace_back(new DevBuf);
                c->stage_w.back()->upload(h.data(), h.size() * 4);
                return c->stage_w.back()->p;
            };
            const int C = dw->s.C;
            t.dwA = (const float *)copies(f.A, C), t.dwS = (const float *)copies(f.S, C), t.dwK = (const int *)copies(f.Kc, C);
            t.pwA = (const float *)copies(pw->conv.A, q.N), t.pwS = (const float *)copies(pw->conv.S, q.N), t.pwK = (const int *)copies(pw->conv.Kc, q.N);
        }
        {
            std::vector<int> rt;
            k::chain_rtab(t, rt);
            c->stage_w.emplace_back(new DevBuf);
            c->stage_w.back()->upload(rt.data(), rt.size() * sizeof(int));
            t.rtab = c->stage_w.back()->as<int>();
        }
        magic = std::min(magic, std::min(dw->magic_mode, pw->magic_mode));
        name += (i ? "|" : "") + std::to_string(dw->s.H) + "x" + std::to_string(dw->s.W) + "x" + std::to_string(dw->s.C) +
                (dw->s.sh == 2 ? "s2" : "") + "-" + std::to_string(q.N);
        c->chain_members.push_back(mem[i]);
    }
    name += ";G" + std::to_string(c->chain.G) + ">";
    c->name = name;
    if (switches().chain_verbose) fprintf(stderr, "[microflow_amd] %s est %.4f us/image/CU lds %d nwave %d dbuf %d\n", name.c_str(), c->chain.est_us_per_image, c->chain.lds_bytes, c->chain.nwave, c->chain.dbuf);
    c->stage_w.emplace_back(new DevBuf);
    c->stage_w.back()->upload(tab.data(), tab.size() * sizeof(k::ChainPair));
    c->chain.pairs = (const k::ChainPair *)c->stage_w.back()->p;
    if (magic == 0 && c->chain.KSC != 4) return nullptr;
    c->chain.magic = magic, c->chain.xr = mem[0].first->s.u8 ? 0x80 : 0;
    c->epi_mode = magic;
    c->chain.queue = (int *)mem[0].first->d_queue.p, c->chain.qlaunch = &mem[0].first->q_launches;
    return c.release();
}
// second level: `n` consecutive single-pair chain groups as ONE chain (nullptr: no plan fits)
FusedImpl *fused_chain_create(FusedImpl *const *groups, int n, int force_G) {
    std::vector<std::pair<OpImpl *, OpImpl *>> mem;
    for (int i = 0; i < n; ++i) {
        if (!groups[i] || groups[i]->kind != FusedImpl::CHAIN || groups[i]->chain_members.size() != 1) return nullptr;
        mem.push_back(groups[i]->chain_members[0]);
    }
    return chain_create(mem.data(), n, force_G);
}
bool fused_is_chain_single(const FusedImpl *f) { return f && f->kind == FusedImpl::CHAIN && f->chain_members.size() == 1; }
// How to run `n` consecutive single-pair chain groups: seg_len[i] = number of pairs of the chain that starts at pair i (0: pair i is
// inside a chain that started earlier); unfused[i] = pair i is cheapest as two separate operator launches.  Dynamic programme over the
// planner's cost estimates (k_chain.hip: chain_plan / chain_unfused_us_per_image).
void fused_run(FusedImpl *f, const int8_t *d_in, size_t batch, int8_t *d_out, void *stream);
namespace {
// The partition by MEASUREMENT (the default when a device is there, i.e. always: operators are created on one): every plannable
// candidate "pairs i .. i + len - 1 as one chain_rt launch", and every pair as its two separate operators, is run on scratch tensors
// at a batch that fills the chip for dozens of steps, and the dynamic programme takes the times.  The cost model's errors were
// 0.7 - 1.5x on single pairs and 1.0 - 1.3x on chains (profiles/r04/chain_calib.txt) -- larger than the differences it decides
// between -- and every kernel change moved them.  ~0.1 - 0.3 s per model at creation; MF_CHAIN_AUTOTUNE=0 goes back to the estimates.
struct ChainTimer {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipStream_t st = nullptr; // a private NON-BLOCKING stream: the timing neither waits for nor stalls the caller's other streams
    DevBuf a, b, c;
    size_t cap = 0;
    bool ok = false;
    explicit ChainTimer(size_t bytes) {
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); st = nullptr; return; }
        if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return;
        for (DevBuf *d : {&a, &b, &c}) {
            if (hipMalloc(&d->p, bytes) != hipSuccess) { (void)hipGetLastError(); return; }
            (void)hipMemsetAsync(d->p, 0, bytes, st);
        }
        cap = bytes, ok = hipStreamSynchronize(st) == hipSuccess;
    }
    ~ChainTimer() {
        if (st) (void)hipStreamSynchronize(st);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        if (st) (void)hipStreamDestroy(st);
    }
    template <typename F> double us(F &&launch) { // best of five after a warm-up; < 0: failed (a throwing launch counts as failed)
        try {
            launch();
            double best = -1;
            for (int r = 0; r < 5; ++r) {
                if (hipEventRecord(e0, st) != hipSuccess) return -1;
                launch();
                if (hipEventRecord(e1, st) != hipSuccess || hipEventSynchronize(e1) != hipSuccess) return -1;
                float ms = 0;
                if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess) return -1;
                best = best < 0 || ms * 1e3 < best ? ms * 1e3 : best;
            }
            return best;
        } catch (const Error &) {
            (void)hipGetLastError();
            failed = true;
            return -1;
        }
    }
    bool failed = false; // a launch threw: the caller drops every measurement and plans from the estimates
};
} // namespace

void fused_chain_partition(FusedImpl *const *groups, int n, int *seg_len, bool *unfused, int *seg_G, bool autotune_opt) {
    std::vector<k::ChainGeom> geo((size_t)n);
    for (int i = 0; i < n; ++i) {
        const std::pair<OpImpl *, OpImpl *> &m = groups[i]->chain_members[0];
        geo[(size_t)i] = chain_geom(m.first, m.second);
    }
    const bool force_fuse = switches().chain_force; // tests: never prefer the unfused operators
    const double INF = 1e30;
    std::vector<double> best((size_t)n + 1, INF);
    std::vector<int> choice((size_t)n + 1, 1);
    std::vector<char> choice_unf((size_t)n + 1, 0);
    best[(size_t)n] = 0;
    std::vector<k::ChainPair> tab((size_t)k::CHAIN_MAX);
    // measuring is the CALLER's choice (mf_model_set_autotune; off by default: model creation is then deterministic, allocates no
    // scratch and launches nothing); MF_CHAIN_AUTOTUNE=1 / =0 overrides it for scripts
    const int env_tune = switches().chain_autotune;
    bool autotune = env_tune < 0 ? autotune_opt : env_tune != 0;
    const bool verbose_t = switches().chain_verbose;
    const bool tune_g = switches().chain_tune_g;
    // measured[i][len]: microseconds per image of the candidate (< 0: not measured); measured_unf[i]: of the pair's two operators
    std::vector<std::vector<double>> measured((size_t)n, std::vector<double>((size_t)k::CHAIN_MAX + 1, -1.0));
    std::vector<double> measured_unf((size_t)n, -1.0);
    if (autotune && n >= 1) {
        const size_t CAP = (size_t)512 << 20; // upper limit of a scratch tensor: dozens of steps per workgroup also for 2 KB images
        auto tensor_bytes = [&](int i, int len) { // the largest tensor any operator of pairs i .. i + len - 1 touches, per image
            size_t m = 0;
            for (int j = i; j < i + len; ++j) {
                const OpSpec &d = groups[j]->chain_members[0].first->s, &q = groups[j]->chain_members[0].second->s;
                m = std::max(m, std::max((size_t)d.H * d.W * d.C, std::max((size_t)d.OH * d.OW * d.N, (size_t)q.OH * q.OW * q.N)));
            }
            return m;
        };
        auto batch_of = [&](size_t tb) { return std::min<size_t>(CAP / std::max<size_t>(tb, 1), 262144) & ~(size_t)63; };
        size_t need = 0; // the scratch the largest candidate needs (not a fixed 512 MB)
        for (int i = 0; i < n; ++i)
            for (int len = 1; len <= n - i && len <= k::CHAIN_MAX; ++len) need = std::max(need, batch_of(tensor_bytes(i, len)) * tensor_bytes(i, len));
        ChainTimer tm(need + 256);
        for (int i = 0; i < n && tm.ok; ++i) {
            for (int len = 1; len <= n - i && len <= k::CHAIN_MAX; ++len) {
                bool ok = true;
                for (int j = i; j < i + len && ok; ++j)
                    ok = groups[j]->chain_members[0].first->s.u8 == groups[i]->chain_members[0].first->s.u8 && (len == 1 || groups[j]->chain_members[0].first->s.C >= 16);
                if (!ok) break;
                const size_t B = batch_of(tensor_bytes(i, len));
                if (B < 256) continue;
                std::unique_ptr<FusedImpl> owned(len == 1 ? nullptr : fused_chain_create(groups + i, len)); // (freed on every path)
                FusedImpl *f = len == 1 ? groups[i] : owned.get();
                if (!f) continue; // (no plan: longer candidates from i may still exist -- a later pair can be smaller)
                const double t = tm.us([&] { fused_run(f, (const int8_t *)tm.a.p, B, (int8_t *)tm.b.p, tm.st); });
                if (t > 0) measured[(size_t)i][(size_t)len] = t / (double)B;
                if (len == 1 && tune_g && t > 0) {
                    // the single pair's images per step and double buffering, measured: every multiple of the column grids' images up to
                    // 4x / down to 1/4 of the planner's choice, with and without the second input buffer.  The winner's plan replaces
                    // the group's in place.
                    const int G0 = groups[i]->chain.G, cg = std::max(1, groups[i]->chain.max_cg), db0 = groups[i]->chain.dbuf;
                    double best_t = t;
                    std::unique_ptr<FusedImpl> best_f;
                    const int cands[8] = {G0 / 4, G0 / 2, 3 * G0 / 4, G0, 3 * G0 / 2, 2 * G0, 3 * G0, 4 * G0};
                    for (int ci = 0; ci < 8; ++ci) {
                        const int G = cands[ci];
                        if (G < cg || G > 128 || G % cg != 0 || (ci > 0 && G == cands[ci - 1])) continue;
                        for (int db = 0; db < 2; ++db) {
                            if (G == G0 && db == db0) continue;
                            std::unique_ptr<FusedImpl> cand(chain_create(&groups[i]->chain_members[0], 1, G, db));
                            if (!cand || cand->chain.dbuf != db) continue;
                            const double tc = tm.us([&] { fused_run(cand.get(), (const int8_t *)tm.a.p, B, (int8_t *)tm.b.p, tm.st); });
                            if (verbose_t) fprintf(stderr, "[microflow_amd] chain autotune: pair %d G %d dbuf %d: %.4f us/image (planner's G %d dbuf %d: %.4f)\n", i, G, db, tc / (double)B, G0, db0, t / (double)B);
                            if (tc > 0 && tc < (best_f ? best_t : t * 0.96)) best_t = tc, best_f = std::move(cand); // (a clear win over the planner's: 4 %)
                        }
                    }
                    if (best_f) {
                        std::swap(*groups[i], *best_f);
                        measured[(size_t)i][1] = best_t / (double)B;
                    }
                }
                if (len == 1) {
                    OpImpl *dw = groups[i]->chain_members[0].first, *pw = groups[i]->chain_members[0].second;
                    const double u = tm.us([&] {
                        op_run(dw, (const int8_t *)tm.a.p, B, (int8_t *)tm.c.p, tm.st);
                        op_run(pw, (const int8_t *)tm.c.p, B, (int8_t *)tm.b.p, tm.st);
                    });
                    if (u > 0) measured_unf[(size_t)i] = u / (double)B;
                }
                if (verbose_t)
                    fprintf(stderr, "[microflow_amd] chain autotune: pairs %d..%d batch %zu: %.4f us/image%s\n", i, i + len - 1, B, measured[(size_t)i][(size_t)len],
                            len == 1 ? (" (unfused " + std::to_string(measured_unf[(size_t)i]) + ")").c_str() : "");
            }
        }
        if (tm.st) (void)hipStreamSynchronize(tm.st);
        (void)hipGetLastError();
        // The measurements are used only as a whole: every pair must have its own time (measured us/image of the chip and estimated
        // us/image per CU are different units and must never meet in one sum); a launch that threw, a pair too large for the scratch
        // or a failed timer send the whole run back to the estimates.
        bool complete = tm.ok && !tm.failed;
        for (int i = 0; i < n && complete; ++i) complete = measured[(size_t)i][1] > 0;
        if (!complete) {
            if (verbose_t || switches().verbose) fprintf(stderr, "[microflow_amd] chain autotune incomplete: planning %d pairs from the cost model\n", n);
            autotune = false;
        }
    }
    for (int i = n - 1; i >= 0; --i) {
        for (int len = 1; len <= n - i && len <= k::CHAIN_MAX; ++len) {
            if (autotune) { // measured costs (a candidate that was not measured does not exist)
                double c = measured[(size_t)i][(size_t)len];
                if (c <= 0) continue;
                if (len > 1) c *= 1.05; // (a chain has to win clearly: isolated timings of this size repeat to 2 - 3 %)
                char unf = 0;
                if (len == 1 && !force_fuse && measured_unf[(size_t)i] > 0 && measured_unf[(size_t)i] < c) c = measured_unf[(size_t)i], unf = 1;
                if (c + best[(size_t)i + len] < best[(size_t)i]) best[(size_t)i] = c + best[(size_t)i + len], choice[(size_t)i] = len, choice_unf[(size_t)i] = unf;
                continue;
            }
            k::ChainArgs a{};
            bool ok = true;
            for (int j = i; j < i + len && ok; ++j)
                ok = groups[j]->chain_members[0].first->s.u8 == groups[i]->chain_members[0].first->s.u8 && (len == 1 || groups[j]->chain_members[0].first->s.C >= 16);
            if (!ok || !k::chain_plan(geo.data() + i, len, tab.data(), a, 150 * 1024)) {
                if (len == 1) { // (cannot happen for a group that exists; keep the programme total)
                    if (best[(size_t)i + 1] < best[(size_t)i]) best[(size_t)i] = best[(size_t)i + 1], choice[(size_t)i] = 1, choice_unf[(size_t)i] = 1;
                }
                continue;
            }
            double c = a.est_us_per_image;
            char unf = 0;
            if (len == 1 && !force_fuse) {
                const double u = k::chain_unfused_us_per_image(geo.data() + i, 1);
                if (u < c) c = u, unf = 1;
            }
            if (c + best[(size_t)i + len] < best[(size_t)i]) best[(size_t)i] = c + best[(size_t)i + len], choice[(size_t)i] = len, choice_unf[(size_t)i] = unf;
        }
    }
    for (int i = 0; i < n; ++i) seg_len[i] = 0, unfused[i] = false;
    for (int i = 0; i < n; i += choice[(size_t)i]) seg_len[i] = choice[(size_t)i], unfused[i] = choice_unf[(size_t)i] != 0;
    // the images per step of the chosen multi-pair chains, measured like the single pairs' (1/2 ... 2x the planner's)
    if (seg_G) {
        for (int i = 0; i < n; ++i) seg_G[i] = 0;
        if (autotune && tune_g) {
            const size_t CAP = (size_t)512 << 20;
            std::unique_ptr<ChainTimer> tm;
            for (int i = 0; i < n; ++i) {
                const int len = seg_len[i];
                if (len < 2) continue;
                if (!tm) {
                    size_t need2 = 0;
                    for (int i2 = 0; i2 < n; ++i2) {
                        if (seg_len[i2] < 2) continue;
                        size_t mx2 = 1;
                        for (int j = i2; j < i2 + seg_len[i2]; ++j) {
                            const OpSpec &d = groups[j]->chain_members[0].first->s, &q = groups[j]->chain_members[0].second->s;
                            mx2 = std::max(mx2, std::max((size_t)d.H * d.W * d.C, std::max((size_t)d.OH * d.OW * d.N, (size_t)q.OH * q.OW * q.N)));
                        }
                        need2 = std::max(need2, (std::min<size_t>(CAP / mx2, 262144) & ~(size_t)63) * mx2);
                    }
                    tm.reset(new ChainTimer(need2 + 256));
                }
                if (!tm->ok || tm->failed) break;
                size_t mx = 1;
                for (int j = i; j < i + len; ++j) {
                    const OpSpec &d = groups[j]->chain_members[0].first->s, &q = groups[j]->chain_members[0].second->s;
                    mx = std::max(mx, std::max((size_t)d.H * d.W * d.C, std::max((size_t)d.OH * d.OW * d.N, (size_t)q.OH * q.OW * q.N)));
                }
                const size_t B = std::min<size_t>(CAP / mx, 262144) & ~(size_t)63;
                std::unique_ptr<FusedImpl> base(fused_chain_create(groups + i, len, 0));
                if (!base || B < 256) continue;
                const double t0 = tm->us([&] { fused_run(base.get(), (const int8_t *)tm->a.p, B, (int8_t *)tm->b.p, tm->st); });
                const int G0 = base->chain.G, cg = std::max(1, base->chain.max_cg);
                double best_t = t0;
                const int cands[4] = {G0 / 2, 3 * G0 / 4, 3 * G0 / 2, 2 * G0};
                for (int ci = 0; ci < 4 && t0 > 0; ++ci) {
                    const int G = cands[ci];
                    if (G < cg || G > 128 || G % cg != 0 || G == G0) continue;
                    std::unique_ptr<FusedImpl> cand(fused_chain_create(groups + i, len, G));
                    if (!cand) continue;
                    const double tc = tm->us([&] { fused_run(cand.get(), (const int8_t *)tm->a.p, B, (int8_t *)tm->b.p, tm->st); });
                    if (verbose_t) fprintf(stderr, "[microflow_amd] chain autotune: chain %d..%d G %d: %.4f us/image (planner's G %d: %.4f)\n", i, i + len - 1, G, tc / (double)B, G0, t0 / (double)B);
                    if (tc > 0 && tc < (seg_G[i] ? best_t : t0 * 0.96)) best_t = tc, seg_G[i] = G;
                }
            }
            if (tm && tm->st) (void)hipStreamSynchronize(tm->st);
            (void)hipGetLastError();
        }
    }
    const bool verbose = switches().chain_verbose || switches().verbose; // the plan, so that a run can be reproduced
    if (verbose) {
        fprintf(stderr, "[microflow_amd] chain partition of %d pairs:", n);
        for (int i = 0; i < n; ++i)
            if (seg_len[i]) fprintf(stderr, " [%d..%d%s]", i, i + seg_len[i] - 1, unfused[i] ? " unfused" : "");
        fprintf(stderr, " %s %.4f us/image%s\n", autotune ? "measured" : "est", best[0], autotune ? "" : "/CU");
    }
}

FusedImpl *fused_create(OpImpl *dw, OpImpl *pw) {
    const bool chain_all = switches().chain_all; // tests / A-B: the chain kernel on table shapes too
    if (dw && pw && (chain_all || dw->fast != OpImpl::DW_NHWC || pw->fast != OpImpl::PW_MFMA ||
                     !k::dwpw_name(dw->s.H, dw->s.W, dw->s.C, dw->s.sh, pw->s.N))) {
        const std::pair<OpImpl *, OpImpl *> one(dw, pw);
        if (FusedImpl *c = chain_create(&one, 1)) return c;
    }
    if (!dw || !pw || dw->fast != OpImpl::DW_NHWC || pw->fast != OpImpl::PW_MFMA) return nullptr;
    const OpSpec &d = dw->s, &q = pw->s;
    // the pointwise conv must consume exactly the depthwise output tensor
    if (q.H != d.OH || q.W != d.OW || q.C != d.N || dw->device != pw->device) return nullptr;
    const char *nm = k::dwpw_name(d.H, d.W, d.C, d.sh, q.N);
    if (!nm) return nullptr;
    FusedImpl *f = new FusedImpl{FusedImpl::DWPW, dw, pw, nullptr, {}, {}, nm};
    f->dwpw = pair_args(dw, pw, 2);
    if (pair_mode(f->dwpw) == 3 && (dw->fma_patch.n || pw->fma_patch.n)) {
        // patched accumulators: dwpw_mm applies them (launch_dwpw routes there), from one record per tile.  Its depthwise tiles are the
        // aligned 16-channel groups; pointwise tile (blk, tt) holds channels blk NB + pg NB / 4 + 4 tt + i in lane group pg (k_fused_mm.hip)
        const int NB = q.N < 64 ? q.N : 64, TB = NB / 16, NQ = d.C / 16;
        std::vector<k::EpiPatchRec> tab((size_t)NQ + (size_t)(q.N / NB) * TB, k::EpiPatchRec{0, 0});
        bool ok = d.C >= 16 && patch_table(dw->fma_patch, tab, 0, [&](int ch, int &reg, int &grp) { return reg = ch & 3, grp = (ch >> 2) & 3, ch >> 4; });
        ok = ok && patch_table(pw->fma_patch, tab, (size_t)NQ, [&](int ch, int &reg, int &grp) {
                 const int rel = ch % NB, within = rel % (NB / 4);
                 return reg = within & 3, grp = rel / (NB / 4), (ch / NB) * TB + within / 4;
             });
        if (ok) {
            f->stage_w.emplace_back(new DevBuf);
            f->stage_w.back()->upload(tab.data(), tab.size() * sizeof(k::EpiPatchRec));
            const k::EpiPatchRec *t = f->stage_w.back()->as<k::EpiPatchRec>();
            if (dw->fma_patch.n) f->dwpw.dw.patch = t;
            if (pw->fma_patch.n) f->dwpw.pw.patch = t + NQ;
            f->name = k::dwpw_mm_name(d.H, d.W, d.C, d.sh, q.N);
        } else {
            f->dwpw = pair_args(dw, pw, 1); // (both strict, or the two-rounding forms)
        }
    }
    f->epi_mode = pair_mode(f->dwpw);
    return f;
}

FusedImpl *fused_tail_create(OpImpl *pool, OpImpl *conv, OpImpl *sm) {
    if (!pool || !conv || !sm) return nullptr;
    const OpSpec &p = pool->s, &c = conv->s, &m = sm->s;
    if (p.u8 != c.u8 || c.u8 != m.u8) return nullptr;
    if (!conv->finite_consts || !std::isfinite(p.pool_c0) || !std::isfinite(p.pool_c1)) return nullptr;
    if (p.kind != MF_OP_AVERAGE_POOL_2D || c.kind != MF_OP_CONV_2D || m.kind != MF_OP_SOFTMAX) return nullptr;
    if (p.OH != 1 || p.OW != 1) return nullptr;                       // one pooling window
    if (c.KH != 1 || c.KW != 1 || c.H != 1 || c.W != 1 || c.OH != 1 || c.OW != 1 || c.C != p.C) return nullptr;
    if (m.M != 1 || m.N != c.N) return nullptr;                       // softmax over the head's N values
    // the in-range taps of the single window (focus (0,0); src/tensor.rs:180-228)
    const int shy = p.pad == MF_PAD_SAME ? (p.KH - 1) / 2 : 0, shx = p.pad == MF_PAD_SAME ? (p.KW - 1) / 2 : 0;
    std::vector<int> taps;
    for (int ky = 0; ky < p.KH; ++ky)
        for (int kx = 0; kx < p.KW; ++kx) {
            const int iy = ky - shy, ix = kx - shx;
            if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) taps.push_back((iy * p.W + ix) * p.C);
        }
    if (!k::tail_supported(p.C, c.N, (int)taps.size())) return nullptr;
    FusedImpl *f = new FusedImpl{FusedImpl::TAIL, pool, conv, sm, {}, {}, "tail_pool_head_softmax<" + std::to_string(c.N) + ">"};
    k::TailArgs &t = f->tail;
    t.H = p.H, t.W = p.W, t.C = p.C, t.N = c.N;
    t.ntaps = (int)taps.size();
    for (int i = 0; i < t.ntaps; ++i) t.tap_off[i] = taps[(size_t)i];
    volatile float inv = 1.0f / (float)t.ntaps; // 1. / view.len as f32 (average_pool_2d.rs:52)
    t.inv_len = inv;
    t.pool_c0 = pool->pool.c0, t.pool_c1 = pool->pool.c1, t.pool_lo = pool->pool.lo, t.pool_hi = pool->pool.hi;
    t.w = conv->conv.w, t.wzp = conv->conv.wzp, t.A = conv->conv.A, t.S = conv->conv.S, t.Kc = conv->conv.Kc;
    t.lo_f = conv->conv.lo_f, t.hi_f = conv->conv.hi_f;
    t.exp_table = sm->sm.exp_table, t.sm_oscale = sm->sm.oscale, t.sm_ozp_f = sm->sm.ozp_f;
    t.pool_bias = pool->pool.bias, t.pool_sat_lo = pool->pool.sat_lo, t.pool_sat_hi = pool->pool.sat_hi;
    t.sm_sat_lo = sm->sm.sat_lo, t.sm_sat_hi = sm->sm.sat_hi, t.xr = pool->pool.xr;
    return f;
}

// FullyConnected (one row per inference, the few-outputs row-wave kernel) -> [Reshape] -> Softmax
// over exactly those outputs
FusedImpl *fused_fc_softmax_create(OpImpl *fc, OpImpl *sm) {
    if (!fc || !sm || fc->fast != OpImpl::FC_ROWWAVE || sm->s.kind != MF_OP_SOFTMAX) return nullptr;
    if (fc->s.M != 1 || sm->s.M != 1 || sm->s.N != fc->s.N || fc->s.N < 2 || fc->device != sm->device) return nullptr;
    if (fc->s.u8 != sm->s.u8) return nullptr;
    return new FusedImpl{FusedImpl::FCSM, fc, sm, nullptr, {}, {}, "fc_rowwave_softmax<" + std::to_string(fc->s.N) + ">"};
}

// Pointwise weights [N][K] as operands A of v_mfma_i32_16x16x64_i8 for the stage kernel: [tile][k-step][lane][16 B],
// row r = lane & 15 of tile tt is output channel 16 tt + r, K-bytes 64 ks + 16 (lane >> 4) .. + 15
static std::vector<int8_t> build_pw_plain_weights(const int8_t *w, int K, int N) {
    const int KS = K / 64, NT = N / 16;
    std::vector<int8_t> out((size_t)NT * KS * 64 * 16);
    for (int tt = 0; tt < NT; ++tt)
        for (int ks = 0; ks < KS; ++ks)
            for (int lane = 0; lane < 64; ++lane) {
                const int r = lane & 15, g = lane >> 4;
                std::memcpy(&out[(((size_t)tt * KS + ks) * 64 + lane) * 16], w + (size_t)(16 * tt + r) * K + 64 * ks + 16 * g, 16);
            }
    return out;
}

// A run of `npairs` identical DepthwiseConv2D 3x3 (stride 1) + Conv2D 1x1 pairs on one small tensor as one persistent
// kernel (k_stage.hip: five pairs on 6x6x128 = person_detect ops 13..22).  `pairs` are the already created pair
// groups; returns nullptr when no stage kernel exists for them.
FusedImpl *fused_stage_create(FusedImpl *const *pairs, int npairs) {
    const bool off = switches().no_stage;
    if (off || npairs < 2 || !pairs[0] || pairs[0]->kind != FusedImpl::DWPW) return nullptr;
    const OpSpec &d0 = pairs[0]->a->s;
    const char *nm = k::stage_name(d0.H, d0.W, d0.C, npairs);
    if (!nm) return nullptr;
    for (int i = 0; i < npairs; ++i) {
        const FusedImpl *f = pairs[i];
        if (!f || f->kind != FusedImpl::DWPW) return nullptr;
        const OpSpec &d = f->a->s, &q = f->b->s;
        if (d.H != d0.H || d.W != d0.W || d.C != d0.C || d.sh != 1 || q.N != d0.C) return nullptr; // same tensor in and out
        if (d.u8 != d0.u8 || q.u8 != d0.u8 || !f->dwpw.dw.magic || !f->dwpw.pw.magic || !f->dwpw.dw.wmm) return nullptr; // bit-pattern epilogues
        if (f->dwpw.dw.izp4 != pairs[0]->dwpw.dw.izp4 || f->a->device != pairs[0]->a->device) return nullptr;
    }
    bool all_fma = true; // the single-fma form needs it of every operator of the run
    for (int i = 0; i < npairs; ++i) all_fma = all_fma && pairs[i]->a->fma_ok && pairs[i]->b->fma_ok;
    // ... and their patched accumulators as the kernel's table: [pair][depthwise, pointwise][wave = aligned 16-channel group][2]
    std::vector<k::EpiPatchRec> ptab((size_t)npairs * 32, k::EpiPatchRec{0, 0});
    for (int i = 0; i < npairs && all_fma; ++i)
        for (int ph = 0; ph < 2 && all_fma; ++ph) {
            const k::EpiPatch &pl = (ph ? pairs[i]->b : pairs[i]->a)->fma_patch;
            for (int e = 0; e < pl.n && all_fma; ++e) {
                const int ch = pl.ch[e];
                if (ch < 0 || ch >= 128) { // (the table has the eight 16-channel groups of this kernel's 128 channels)
                    all_fma = false;
                    break;
                }
                k::EpiPatchRec *slot = &ptab[(((size_t)i * 2 + ph) * 8 + (size_t)(ch >> 4)) * 2];
                if (slot[0].P != 0) ++slot;
                if (slot->P != 0) all_fma = false; // (three in one group: the kernel's table holds two)
                else *slot = k::epi_patch_rec(pl.P[e], pl.R[e], ch & 3, (ch >> 2) & 3);
                if (all_fma && slot != &ptab[(((size_t)i * 2 + ph) * 8 + (size_t)(ch >> 4)) * 2]) slot[-1].meta |= 32; // "a second record follows"
            }
        }
    std::unique_ptr<FusedImpl> s(new FusedImpl{FusedImpl::STAGE, pairs[0]->a, pairs[npairs - 1]->b, nullptr, {}, {}, nm});
    s->stage_pairs = npairs;
    std::vector<k::StagePair> table((size_t)npairs);
    for (int i = 0; i < npairs; ++i) {
        const FusedImpl *fp = pairs[i];
        FusedImpl tmp{FusedImpl::DWPW, fp->a, fp->b, nullptr, {}, {}, ""};
        tmp.dwpw = pair_args(fp->a, fp->b, all_fma ? 2 : 0);
        const FusedImpl *f = &tmp;
        k::StagePair &sp = table[(size_t)i];
        // Kc + the bit-pattern offset of requant_t<true> (k_common.hpp), as separate arrays for this kernel
        auto with_magic = [&](const int *d_kc, int n) {
            std::vector<int32_t> h((size_t)n);
            MF_HIP(hipMemcpy(h.data(), d_kc, h.size() * 4, hipMemcpyDeviceToHost));
            for (int32_t &v : h) v = wrap_add(v, 0x4B400000);
            s->stage_w.emplace_back(new DevBuf);
            s->stage_w.back()->upload(h.data(), h.size() * 4);
            return (const int *)s->stage_w.back()->p;
        };
        sp.dw_wmm = f->dwpw.dw.wmm, sp.dwA = f->dwpw.dw.A, sp.dwS = f->dwpw.dw.S, sp.dwK = with_magic(f->dwpw.dw.Kc, f->a->s.N);
        sp.dw_lo = f->dwpw.dw.lo_f, sp.dw_hi = f->dwpw.dw.hi_f;
        const OpSpec &q = f->b->s;
        std::vector<int8_t> host((size_t)q.N * q.C);
        MF_HIP(hipMemcpy(host.data(), f->b->conv.w, host.size(), hipMemcpyDeviceToHost)); // [N][1][1][C] as uploaded
        const std::vector<int8_t> prep = build_pw_plain_weights(host.data(), q.C, q.N);
        s->stage_w.emplace_back(new DevBuf);
        s->stage_w.back()->upload(prep.data(), prep.size());
        sp.pw_w = s->stage_w.back()->p;
        sp.pwA = f->dwpw.pw.A, sp.pwS = f->dwpw.pw.S, sp.pwK = with_magic(f->dwpw.pw.Kc, q.N);
        sp.pw_lo = f->dwpw.pw.lo_f, sp.pw_hi = f->dwpw.pw.hi_f;
    }
    s->stage.patch_tab = nullptr;
    if (all_fma) { // (also when nothing is patched: the kernel fetches its records with the operands, unconditionally)
        s->stage_w.emplace_back(new DevBuf);
        s->stage_w.back()->upload(ptab.data(), ptab.size() * sizeof(k::EpiPatchRec));
        s->stage.patch_tab = s->stage_w.back()->as<k::EpiPatchRec>();
    }
    s->stage_w.emplace_back(new DevBuf);
    s->stage_w.back()->upload(table.data(), table.size() * sizeof(k::StagePair));
    s->stage.pairs = (const k::StagePair *)s->stage_w.back()->p;
    s->stage.izp4 = pairs[0]->dwpw.dw.izp4;
    s->stage.xr4 = d0.u8 ? 0x80808080u : 0u;
    s->stage.queue = pairs[0]->dwpw.dw.queue, s->stage.qlaunch = pairs[0]->dwpw.dw.qlaunch;
    s->stage.mode = all_fma ? 3 : 2; // the saturating-pack epilogue needs it of every operator of the run
    for (int i = 0; i < npairs && !all_fma; ++i)
        if (pairs[i]->a->magic_mode != 2 || pairs[i]->b->magic_mode != 2) s->stage.mode = 1;
    s->epi_mode = s->stage.mode;
    return s.release();
}

// DepthwiseConv2D with one input channel (the dw_c1_lds operator) -> [Reshape] -> FullyConnected + Softmax group, as
// one kernel (k_dwfc.hip; speech.tflite ops 1..3).  Second level like the stage: the operator and the group inside
// stay available for mf_model_run_until.  nullptr when the shapes are not the compiled instance.
FusedImpl *fused_dwfc_create(OpImpl *dw, FusedImpl *fcsm) {
    const bool off = switches().no_dwfc;
    if (off || !dw || !fcsm || dw->fast != OpImpl::DW_C1 || fcsm->kind != FusedImpl::FCSM) return nullptr;
    OpImpl *fc = fcsm->a, *sm = fcsm->b;
    const OpSpec &d = dw->s, &q = fc->s;
    using Gm = k::DwFcGeom;
    if (!k::dwfc_supported(d.H, d.W, d.KH, d.KW, d.sh, d.sw, d.OH, d.OW, d.N, q.N)) return nullptr;
    if (d.pad != MF_PAD_SAME || d.C != 1 || q.M != 1 || q.K != d.OH * d.OW * d.N || dw->device != fc->device) return nullptr;
    if (d.u8 != q.u8) return nullptr;
    std::unique_ptr<FusedImpl> f(new FusedImpl{FusedImpl::DWFC, dw, fc, sm, {}, {}, k::dwfc_name()});
    // both operators' weights as they were uploaded (i8 domain): the depthwise taps from dw_c1_lds's packed form
    // [ky][4-tap group][8 channels] dwords, the FullyConnected matrix [N][K]
    const k::DwC1Args &c1 = dw->dwc1;
    std::vector<uint32_t> wp((size_t)Gm::KH * c1.KG * 8);
    MF_HIP(hipMemcpy(wp.data(), c1.wpack, wp.size() * 4, hipMemcpyDeviceToHost));
    auto dw_w = [&](int ky, int kx, int c) { return (int8_t)(wp[((size_t)ky * c1.KG + kx / 4) * 8 + c] >> (8 * (kx & 3))); };
    std::vector<int8_t> fc_w((size_t)q.N * q.K);
    MF_HIP(hipMemcpy(fc_w.data(), fc->fc.w, fc_w.size(), hipMemcpyDeviceToHost));
    // operand A: taps of filter row ky = (4k + g) - 2p at byte b = kx + 2s + E0 of the 16-byte window, for row
    // r = (p, c) of the accumulator tile; zero elsewhere
    std::vector<int8_t> wa((size_t)4 * 3 * 64 * 16, 0);
    for (int sft = 0; sft < 4; ++sft)
        for (int kk = 0; kk < 3; ++kk)
            for (int lane = 0; lane < 64; ++lane) {
                const int r = lane & 15, g = lane >> 4, p = r >> 3, c = r & 7;
                const int ky = 4 * kk + g - Gm::S * p;
                if (ky < 0 || ky >= Gm::KH) continue;
                for (int kx = 0; kx < Gm::KW; ++kx)
                    wa[(((size_t)sft * 3 + kk) * 64 + lane) * 16 + (size_t)(kx + Gm::S * sft + Gm::E0)] =
                        dw_w(ky, kx, c);
            }
    // FullyConnected as operand A of one more MFMA per unit (t, m): lane group g = (p, channel half) holds, for row
    // n < 4, the weights of the 16 activations it packs -- pixels (2t + p, 4m + s), s = 0..3, channels 4 (g & 1) .. + 3
    // of the NHWC flattening -- and for row 4 ones (the row sum); nothing for pixel rows beyond the image
    std::vector<int8_t> wf((size_t)Gm::FCW_BYTES, 0);
    for (int u = 0; u < Gm::NU; ++u)
        for (int g = 0; g < 4; ++g) {
            const int t = u / Gm::NM, m = u % Gm::NM, oy = 2 * t + (g >> 1);
            if (oy >= Gm::OH) continue;
            for (int sft = 0; sft < 4; ++sft) {
                const size_t k0 = ((size_t)oy * Gm::OW + 4 * m + sft) * 8 + 4 * (size_t)(g & 1);
                for (int b = 0; b < 4; ++b) {
                    for (int n = 0; n < 4; ++n) wf[(((size_t)u * 4 + g) * 5 + n) * 16 + 4 * sft + b] = fc_w[(size_t)n * q.K + k0 + b];
                    wf[(((size_t)u * 4 + g) * 5 + 4) * 16 + 4 * sft + b] = 1;
                }
            }
        }
    f->stage_w.emplace_back(new DevBuf);
    f->stage_w.back()->upload(wa.data(), wa.size());
    f->dwfc.wA = f->stage_w.back()->p;
    f->stage_w.emplace_back(new DevBuf);
    f->stage_w.back()->upload(wf.data(), wf.size());
    f->dwfc.wfc = f->stage_w.back()->p;
    f->dwfc.dwA = c1.A, f->dwfc.dwS = c1.S, f->dwfc.dwKc = c1.Kc, f->dwfc.dw_lo = c1.lo_f, f->dwfc.dw_hi = c1.hi_f;
    f->dwfc.izp4 = 0x01010101u * (uint32_t)(uint8_t)(int8_t)c1.izp;
    f->dwfc.magic = c1.magic, f->dwfc.xr = c1.xr;
    f->epi_mode = c1.magic;
    f->dwfc.fc = fc->fc, f->dwfc.sm = sm->sm;
    return f.release();
}

// The last pair group (DepthwiseConv2D 3x3 stride 1 + Conv2D 1x1 on 3x3x256) followed by the tail group
// (AveragePool2D over the whole tensor -> head Conv2D -> Softmax) as one kernel (k_tail3.hip).  Second level like the
// stage; nullptr when the shapes are not the compiled instance.
FusedImpl *fused_pair_tail_create(FusedImpl *pair, FusedImpl *tail) {
    const bool off = switches().no_pairtail;
    if (off || !pair || !tail || tail->kind != FusedImpl::TAIL) return nullptr;
    // the pair: a table group (dwpw_mm) or a single-pair run-time-geometry chain group
    OpImpl *dw = nullptr, *pw = nullptr;
    if (pair->kind == FusedImpl::DWPW) dw = pair->a, pw = pair->b;
    else if (pair->kind == FusedImpl::CHAIN && pair->chain_members.size() == 1) dw = pair->chain_members[0].first, pw = pair->chain_members[0].second;
    if (!dw || !pw || (dw->fast != OpImpl::DW_NHWC && dw->fast != OpImpl::DW_RT) || (dw->fast == OpImpl::DW_RT && dw->rt_wz)) return nullptr;
    if ((pw->fast != OpImpl::PW_MFMA && pw->fast != OpImpl::PW_RT) || (pw->fast == OpImpl::PW_RT && pw->rt_wz)) return nullptr;
    const OpSpec &d = dw->s, &q = pw->s;
    const k::TailArgs &t = tail->tail;
    if (!k::pair_tail_supported(d.H, d.W, d.C, q.N, t.N, t.ntaps) || d.sh != 1 || d.sw != 1 || d.u8 != q.u8) return nullptr;
    if (d.KH != 3 || d.KW != 3 || d.pad != MF_PAD_SAME || d.C != d.N || q.KH != 1 || q.KW != 1 || q.C != d.N) return nullptr;
    if ((d.u8 ? 0x80 : 0) != t.xr) return nullptr;
    if (t.H != d.OH || t.W != d.OW || t.C != q.N || dw->device != tail->a->device) return nullptr;
    const k::DwFastArgs &df = dw->fast == OpImpl::DW_NHWC ? dw->dwf : dw->dwrt.dw;
    if (!df.wmm || !dw->finite_consts || !pw->finite_consts) return nullptr;
    const int magic = (dw->magic_mode >= 1 && pw->magic_mode >= 1) ? 1 : 0; // bit-pattern epilogues, or the v_cvt form for both
    std::unique_ptr<FusedImpl> f(new FusedImpl{FusedImpl::PAIRTAIL, dw, tail->b, tail->c, {}, {}, k::pair_tail_name(d.H, d.C)});
    auto with_magic = [&](const int *d_kc, int n) { // Kc + the bit-pattern offset of requant_t<true> (k_common.hpp)
        if (!magic) return d_kc;
        std::vector<int32_t> h((size_t)n);
        MF_HIP(hipMemcpy(h.data(), d_kc, h.size() * 4, hipMemcpyDeviceToHost));
        for (int32_t &v : h) v = wrap_add(v, 0x4B400000);
        f->stage_w.emplace_back(new DevBuf);
        f->stage_w.back()->upload(h.data(), h.size() * 4);
        return (const int *)f->stage_w.back()->p;
    };
    k::PairTailArgs &a = f->pairtail;
    a.H = d.H, a.C = d.C, a.magic = magic;
    f->epi_mode = magic;
    a.dw_wmm = df.wmm, a.dwA = df.A, a.dwS = df.S, a.dwK = with_magic(df.Kc, d.N);
    a.dw_lo = df.lo_f, a.dw_hi = df.hi_f, a.izp4 = df.izp4;
    std::vector<int8_t> host((size_t)q.N * q.C);
    MF_HIP(hipMemcpy(host.data(), pw->conv.w, host.size(), hipMemcpyDeviceToHost)); // [N][1][1][C] as uploaded
    const std::vector<int8_t> prep = build_pw_plain_weights(host.data(), q.C, q.N);
    f->stage_w.emplace_back(new DevBuf);
    f->stage_w.back()->upload(prep.data(), prep.size());
    a.pw_w = f->stage_w.back()->p;
    a.pwA = pw->conv.A, a.pwS = pw->conv.S, a.pwK = with_magic(pw->conv.Kc, q.N);
    a.pw_lo = pw->conv.lo_f, a.pw_hi = pw->conv.hi_f;
    a.tail = t;
    return f.release();
}

// The pair group in front of a pair + tail launch joins it (person_detect ops 23..30 in one launch; k_tail3.hip FRONT): second
// level like the others -- the pair group and the pair + tail stage stay for mf_model_run_until.  Borrows the pair + tail stage's
// device arrays (destroy it first); the front pair's own arrays are in stage_w.
FusedImpl *fused_front_pair_tail_create(FusedImpl *front, FusedImpl *pairtail) {
    if (switches().no_pair_front || !front || !pairtail || pairtail->kind != FusedImpl::PAIRTAIL || pairtail->has_front) return nullptr;
    if (front->kind != FusedImpl::DWPW) return nullptr;
    OpImpl *dw = front->a, *pw = front->b;
    if (!dw || !pw || dw->fast != OpImpl::DW_NHWC || pw->fast != OpImpl::PW_MFMA) return nullptr;
    const OpSpec &d = dw->s, &q = pw->s;
    const k::PairTailArgs &t = pairtail->pairtail;
    if (d.sh != d.sw || !k::pair_front_supported(d.H, d.W, d.C, d.sh, q.N, t.H, t.C)) return nullptr;
    if (d.KH != 3 || d.KW != 3 || d.pad != MF_PAD_SAME || d.C != d.N || q.KH != 1 || q.KW != 1 || q.C != d.N) return nullptr;
    if (q.OH != t.H || q.OW != t.H || d.u8 != q.u8 || (d.u8 ? 0x80u : 0u) != (uint32_t)t.tail.xr || dw->device != pairtail->a->device) return nullptr;
    const k::DwFastArgs &df = dw->dwf;
    if (!df.wmm || !dw->finite_consts || !pw->finite_consts) return nullptr;
    // one epilogue form for the launch's four convolutions: the bit-pattern one needs it of all four
    const int magic = (dw->magic_mode >= 1 && pw->magic_mode >= 1) ? 1 : 0;
    if (magic != t.magic) return nullptr;
    std::unique_ptr<FusedImpl> f(new FusedImpl{FusedImpl::PAIRTAIL, dw, pairtail->b, pairtail->c, {}, {}, "pair_front_tail<6,6,128,2,256|3,3,256,2>"});
    auto with_magic = [&](const int *d_kc, int n) {
        if (!magic) return d_kc;
        std::vector<int32_t> h((size_t)n);
        MF_HIP(hipMemcpy(h.data(), d_kc, h.size() * 4, hipMemcpyDeviceToHost));
        for (int32_t &v : h) v = wrap_add(v, 0x4B400000);
        f->stage_w.emplace_back(new DevBuf);
        f->stage_w.back()->upload(h.data(), h.size() * 4);
        return (const int *)f->stage_w.back()->p;
    };
    f->pairtail = t, f->has_front = true, f->epi_mode = magic;
    k::PairFrontArgs &a = f->pairfront;
    a.dw_wmm = df.wmm, a.dwA = df.A, a.dwS = df.S, a.dwK = with_magic(df.Kc, d.N);
    a.dw_lo = df.lo_f, a.dw_hi = df.hi_f, a.izp4 = df.izp4;
    std::vector<int8_t> host((size_t)q.N * q.C);
    MF_HIP(hipMemcpy(host.data(), pw->conv.w, host.size(), hipMemcpyDeviceToHost)); // [N][1][1][C] as uploaded
    const std::vector<int8_t> prep = build_pw_plain_weights(host.data(), q.C, q.N);
    f->stage_w.emplace_back(new DevBuf);
    f->stage_w.back()->upload(prep.data(), prep.size());
    a.pw_w = f->stage_w.back()->p;
    a.pwA = pw->conv.A, a.pwS = pw->conv.S, a.pwK = with_magic(pw->conv.Kc, q.N);
    a.pw_lo = pw->conv.lo_f, a.pw_hi = pw->conv.hi_f;
    return f.release();
}

// Two consecutive DepthwiseConv2D 3x3 + Conv2D 1x1 pair groups as one kernel (k_quad.hip), when a quad kernel exists for the two
// shapes.  Second level like the stage: the pairs inside stay available for mf_model_run_until.
FusedImpl *fused_quad_create(FusedImpl *p1, FusedImpl *p2) {
    const bool off = switches().no_quad;
    if (off || !p1 || !p2 || p1->kind != FusedImpl::DWPW || p2->kind != FusedImpl::DWPW) return nullptr;
    const OpSpec &d1 = p1->a->s, &q1 = p1->b->s, &d2 = p2->a->s, &q2 = p2->b->s;
    if (p1->a->device != p2->a->device || d1.u8 != d2.u8) return nullptr;
    if (d2.H != q1.H || d2.W != q1.W || d2.C != q1.N) return nullptr; // the second pair consumes the first pair's output
    if (k::quad_mm_shape(d1.H, d1.W, d1.C, d1.sh, q1.N, d2.H, d2.W, d2.C, d2.sh, q2.N)) {
        if (switches().no_quad_mm) return nullptr;
        // the pairs' own blocks (dwpw_mm's: matrix-pipe depthwise weights, pw_mfma-layout pointwise weights, patch tables); one
        // epilogue mode for the launch: the single-fma form if both pairs run it, else the two-rounding forms for both
        k::DwPwArgs a = p1->dwpw, b = p2->dwpw;
        if (pair_mode(a) != 3 || pair_mode(b) != 3) a = pair_args(p1->a, p1->b, 0), b = pair_args(p2->a, p2->b, 0);
        if (!a.dw.wmm || !a.pw.wprep || !b.dw.wmm || !b.pw.wprep) return nullptr;
        if (!a.dw.magic || !a.pw.magic || !b.dw.magic || !b.pw.magic) return nullptr; // bit-pattern epilogues
        FusedImpl *f = new FusedImpl{FusedImpl::QUAD, p1->a, p2->b, nullptr, {}, {}, "quad_mm<12,12,64,1,64|12,12,64,2,128>"};
        f->quad.a = a, f->quad.b = b, f->quad_mm = true;
        f->epi_mode = std::min(pair_mode(a), pair_mode(b));
        f->quad_ops[0] = p1->a, f->quad_ops[1] = p1->b, f->quad_ops[2] = p2->a, f->quad_ops[3] = p2->b;
        const int shp[10] = {d1.H, d1.W, d1.C, d1.sh, q1.N, d2.H, d2.W, d2.C, d2.sh, q2.N};
        for (int i = 0; i < 10; ++i) f->quad_shape[i] = shp[i];
        return f;
    }
    const char *nm = k::quad_name(d1.H, d1.W, d1.C, d1.sh, q1.N, d2.H, d2.W, d2.C, d2.sh, q2.N);
    if (!nm) return nullptr;
    // (the single-fma form needs it of all four operators)
    const bool fma = p1->a->fma_strict() && p1->b->fma_strict() && p2->a->fma_strict() && p2->b->fma_strict();
    const k::DwPwArgs a = pair_args(p1->a, p1->b, fma ? 1 : 0), b = pair_args(p2->a, p2->b, fma ? 1 : 0);
    if (!a.dw.wmm || !a.pw.wrr || !b.dw.wmm || !b.pw.wrr) return nullptr;
    if (!a.dw.magic || !a.pw.magic || !b.dw.magic || !b.pw.magic) return nullptr; // bit-pattern epilogues
    FusedImpl *f = new FusedImpl{FusedImpl::QUAD, p1->a, p2->b, nullptr, {}, {}, nm};
    f->quad.a = a, f->quad.b = b;
    f->epi_mode = std::min(pair_mode(a), pair_mode(b));
    f->quad_ops[0] = p1->a, f->quad_ops[1] = p1->b, f->quad_ops[2] = p2->a, f->quad_ops[3] = p2->b;
    const int shp[10] = {d1.H, d1.W, d1.C, d1.sh, q1.N, d2.H, d2.W, d2.C, d2.sh, q2.N};
    for (int i = 0; i < 10; ++i) f->quad_shape[i] = shp[i];
    return f;
}

// The network's one-input-channel stem in front of a quad: five operators in one launch (k_quad.hip, STEM instance).  The quad
// itself stays (mf_model_run_until, and the f32 entry point, whose boundary quantisation is fused into the stem kernel).
FusedImpl *fused_quad_stem_create(OpImpl *stem, FusedImpl *quad) {
    const bool off = switches().no_penta;
    if (off || !stem || !quad || quad->kind != FusedImpl::QUAD || quad->quad.stem || stem->fast != OpImpl::DW_STEM) return nullptr;
    const OpSpec &t = stem->s, &d1 = quad->a->s;
    if (stem->device != quad->a->device || t.u8 != d1.u8 || stem->force_generic) return nullptr;
    if (t.OH != d1.H || t.OW != d1.W || t.N != d1.C || t.sh != 2 || t.sw != 2 || t.C != 1) return nullptr; // pair A consumes the stem's output
    const int *q = quad->quad_shape;
    const char *nm = k::quad_stem_name(t.H, t.W, q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7], q[8], q[9]);
    if (!nm || !stem->stem.magic) return nullptr;
    std::unique_ptr<FusedImpl> f(new FusedImpl{FusedImpl::QUAD, stem, quad->b, nullptr, {}, {}, nm});
    f->quad = quad->quad;
    for (int i = 0; i < 10; ++i) f->quad_shape[i] = q[i];
    // five operators, one epilogue mode: the single-fma form if all five have it, else everyone's two-rounding constants
    OpImpl *const *qo = quad->quad_ops;
    const bool fma = stem->fma_strict() && qo[0]->fma_strict() && qo[1]->fma_strict() && qo[2]->fma_strict() && qo[3]->fma_strict();
    f->quad.a = pair_args(qo[0], qo[1], fma ? 1 : 0), f->quad.b = pair_args(qo[2], qo[3], fma ? 1 : 0);
    k::DwStemArgs sa = stem->stem;
    if (fma) sa.use_fma();
    std::vector<uint32_t> tab(152);
    for (int l = 0; l < 64; ++l) tab[(size_t)2 * l] = sa.wmm[l][0], tab[(size_t)2 * l + 1] = sa.wmm[l][1];
    for (int c = 0; c < 8; ++c) {
        memcpy(&tab[(size_t)128 + c], &sa.A[c], 4);
        memcpy(&tab[(size_t)136 + c], &sa.S[c], 4);
        memcpy(&tab[(size_t)144 + c], &sa.Kc[c], 4);
    }
    f->stage_w.emplace_back(new DevBuf);
    f->stage_w.back()->upload(tab.data(), tab.size() * 4);
    f->quad.stem = f->stage_w.back()->as<uint32_t>();
    f->quad.stem_izp4 = sa.izp4, f->quad.stem_lo = sa.lo_f, f->quad.stem_hi = sa.hi_f, f->quad.stem_magic = sa.magic;
    // the f32 entry (model.cpp sets the stem's input quantisation before the groups are built): same launch, f32 image in
    f->quad.in_scale = sa.in_scale, f->quad.in_zp_f = sa.in_zp_f, f->quad.in_sat_lo = sa.in_sat_lo, f->quad.in_sat_hi = sa.in_sat_hi;
    f->quad.in_rcp = sa.in_rcp, f->quad.in_xr4 = sa.in_xr4, f->quad.in_fast = sa.in_fast, f->quad.f32_ok = stem->accepts_f32 ? 1 : 0;
    f->epi_mode = std::min(std::min(pair_mode(f->quad.a), pair_mode(f->quad.b)), sa.magic);
    return f.release();
}

void fused_destroy(FusedImpl *f) { delete f; }
const char *fused_kernel_name(const FusedImpl *f) { return f->name.c_str(); }
int fused_epilogue_mode(const FusedImpl *f) {
    if (f->epi_mode >= 0) return f->epi_mode;
    int mode = -1; // not recorded by the builder: the minimum over the group's conv-like operators
    for (const OpImpl *o : {f->a, f->b, f->c})
        if (o && (o->s.kind == MF_OP_CONV_2D || o->s.kind == MF_OP_DEPTHWISE_CONV_2D)) mode = mode < 0 ? o->magic_mode : std::min(mode, o->magic_mode);
    return mode;
}
// the f32 entry of a group that starts with the network's first operator (M::predict: the boundary quantisation inside the launch)
bool fused_accepts_f32(const FusedImpl *f) {
    return f && f->kind == FusedImpl::QUAD && f->quad.stem && f->quad.f32_ok && !switches().no_f32_group;
}
void fused_run_f32(FusedImpl *f, const float *d_in, size_t batch, int8_t *d_out, void *stream) {
    if (!batch) return;
    if (!fused_accepts_f32(f)) fail(MF_ERR_UNSUPPORTED, "group has no f32-input kernel");
    if (!d_in || !d_out || ((uintptr_t)d_in & 15)) fail(MF_ERR_INVALID_ARG, "fused_run_f32: null or unaligned device pointer");
    if (batch > 0x7fffffffull / 4) fail(MF_ERR_INVALID_ARG, "batch too large for one launch");
    const int *q = f->quad_shape;
    if (!k::launch_quad_f32(q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7], q[8], q[9], d_in, d_out, f->quad, (int)batch, (hipStream_t)stream))
        fail(MF_ERR_UNSUPPORTED, "f32 quad kernel missing");
    MF_HIP(hipGetLastError());
}
void fused_run(FusedImpl *f, const int8_t *d_in, size_t batch, int8_t *d_out, void *stream) {
    if (!batch) return;
    if (f->kind == FusedImpl::STAGE) {
        if (batch > 0x7fffffffull / 4) fail(MF_ERR_INVALID_ARG, "batch too large for one launch");
        const OpSpec &d = f->a->s;
        if (!k::launch_stage(d.H, d.W, d.C, f->stage_pairs, d_in, d_out, f->stage, (int)batch, (hipStream_t)stream))
            fail(MF_ERR_UNSUPPORTED, "stage kernel missing");
        MF_HIP(hipGetLastError());
        return;
    }
    if (f->kind == FusedImpl::CHAIN) {
        if (batch > 0x7fffffffull / 4) fail(MF_ERR_INVALID_ARG, "batch too large for one launch");
        k::launch_chain(d_in, d_out, f->chain, (int)batch, (hipStream_t)stream);
        MF_HIP(hipGetLastError());
        return;
    }
    if (f->kind == FusedImpl::QUAD) {
        if (batch > 0x7fffffffull / 4) fail(MF_ERR_INVALID_ARG, "batch too large for one launch");
        const int *q = f->quad_shape;
        if (f->quad_mm) {
            k::launch_quad_mm(d_in, d_out, f->quad, (int)batch, (hipStream_t)stream);
            MF_HIP(hipGetLastError());
            return;
        }
        if (!k::launch_quad(q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7], q[8], q[9], d_in, d_out, f->quad, (int)batch, (hipStream_t)stream))
            fail(MF_ERR_UNSUPPORTED, "quad kernel missing");
        MF_HIP(hipGetLastError());
        return;
    }
    if (f->kind == FusedImpl::TAIL) {
        k::launch_tail(d_in, d_out, f->tail, batch, (hipStream_t)stream);
        MF_HIP(hipGetLastError());
        return;
    }
    if (f->kind == FusedImpl::PAIRTAIL) {
        if (batch && f->has_front) k::launch_pair_front_tail(d_in, d_out, f->pairtail, f->pairfront, batch, (hipStream_t)stream);
        else if (batch) k::launch_pair_tail(d_in, d_out, f->pairtail, batch, (hipStream_t)stream);
        MF_HIP(hipGetLastError());
        return;
    }
    if (f->kind == FusedImpl::DWFC) {
        if (batch) k::launch_dwfc(d_in, d_out, f->dwfc, batch, (hipStream_t)stream);
        MF_HIP(hipGetLastError());
        return;
    }
    if (f->kind == FusedImpl::FCSM) {
        if (!k::launch_fc_rowwave_softmax(d_in, d_out, f->a->fc, f->b->sm, batch, (hipStream_t)stream))
            fail(MF_ERR_UNSUPPORTED, "fused kernel missing");
        MF_HIP(hipGetLastError());
        return;
    }
    if (batch > 0x7fffffffull / 4) fail(MF_ERR_INVALID_ARG, "batch too large for one launch");
    const OpSpec &d = f->a->s;
    if (!k::launch_dwpw(d.H, d.W, d.C, d.sh, f->b->s.N, d_in, d_out, f->dwpw, (int)batch, (hipStream_t)stream))
        fail(MF_ERR_UNSUPPORTED, "fused kernel missing");
    MF_HIP(hipGetLastError());
}

void dev_quantize(int device, const float *d_in, size_t n, float scale, int zp, bool u8, int8_t *d_out,
                  void *stream) {
    dev_require(device);
    if (!n) return;
    k::launch_quantize(d_in, d_out, n, scale, (float)zp, u8, (hipStream_t)stream);
    MF_HIP(hipGetLastError());
}
void dev_dequantize(int device, const int8_t *d_in, size_t n, float scale, int zp, bool u8, float *d_out,
                    void *stream) {
    dev_require(device);
    if (!n) return;
    // f32(q) - f32(zp) with q = stored + 128: both small integers, so the shift moves to zp exactly
    k::launch_dequantize(d_in, d_out, n, scale, (float)(zp - (u8 ? 128 : 0)), false, (hipStream_t)stream);
    MF_HIP(hipGetLastError());
}
void dev_dequantize_u8_raw(int device, const uint8_t *d_in, size_t n, float scale, int zp, float *d_out,
                           void *stream) {
    dev_require(device);
    if (!n) return;
    k::launch_dequantize((const int8_t *)d_in, d_out, n, scale, (float)zp, true, (hipStream_t)stream);
    MF_HIP(hipGetLastError());
}
void dev_xor80(int device, const int8_t *d_in, size_t n, int8_t *d_out, void *stream) {
    dev_require(device);
    if (!n) return;
    k::launch_xor80(d_in, d_out, n, (hipStream_t)stream);
    MF_HIP(hipGetLastError());
}
void dev_synth_i8(int device, uint64_t seed, uint64_t first, size_t n, int8_t *d_out, void *stream) {
    dev_require(device);
    if (!n) return;
    k::launch_synth(d_out, n, seed, first, (hipStream_t)stream);
    MF_HIP(hipGetLastError());
}
uint64_t dev_selftest_epilogue(int device, int mode, bool u8, bool have_as, float A, float S, int lo, int hi) {
    dev_require(device);
    if (mode == 3 && !have_as && !u8) { // negative control of the rounding check: plain round-to-nearest-even (must mismatch)
        const unsigned long long r3 = k::selftest_rounding(3, false, (float)lo, (float)hi, nullptr);
        if (r3 == ~0ull) fail(MF_ERR_HIP, "selftest kernel could not run");
        return r3;
    }
    if ((mode != 1 && mode != 2) || lo > hi || lo < (u8 ? 0 : -128) || hi > (u8 ? 255 : 127))
        fail(MF_ERR_INVALID_ARG, "selftest: mode 1 or 2 and a clamp inside the element type's range");
    if (mode == 2 && (lo != (u8 ? 0 : -128) || hi != (u8 ? 255 : 127)))
        fail(MF_ERR_INVALID_ARG, "selftest: the saturating pack (mode 2) is only ever used with the whole range as clamp");
    const unsigned long long r = have_as ? k::selftest_requant(mode, u8, A, S, (float)lo, (float)hi, nullptr)
                                         : k::selftest_rounding(mode, u8, (float)lo, (float)hi, nullptr);
    if (r == ~0ull) fail(MF_ERR_HIP, "selftest kernel could not run");
    return r;
}
uint64_t dev_selftest_fma_epilogue(int device, float A, float S, bool u8, int64_t amin, int64_t amax, float S3, float C3, int pivot,
                                   int64_t patch_acc, int patch_delta) {
    dev_require(device);
    MF_HIP(hipSetDevice(device));
    DevBuf dA, dS, dC3, dS3, dpiv, dmn, dmx, dbad, dpP, dpR;
    const int32_t mn = (int32_t)amin, mx = (int32_t)amax, pP = patch_delta ? k::MF_MAGIC_I + (int32_t)patch_acc + pivot : 0, pR = pP + patch_delta;
    const unsigned long long zero = 0;
    dpP.upload(&pP, 4), dpR.upload(&pR, 4);
    dA.upload(&A, 4), dS.upload(&S, 4), dC3.upload(&C3, 4), dS3.upload(&S3, 4), dpiv.upload(&pivot, 4), dmn.upload(&mn, 4), dmx.upload(&mx, 4);
    dbad.upload(&zero, 8);
    unsigned long long bad = ~0ull;
    if (!k::verify_fma_form(dA.as<float>(), dS.as<float>(), dC3.as<float>(), dS3.as<float>(), dpiv.as<int>(), dmn.as<int>(), dmx.as<int>(), dpP.as<int>(), dpR.as<int>(), 1,
                            u8 ? 0.0f : -128.0f, u8 ? 255.0f : 127.0f, u8, (unsigned long long *)dbad.p, nullptr))
        fail(MF_ERR_HIP, "selftest kernel could not run");
    MF_HIP(hipMemcpy(&bad, dbad.p, 8, hipMemcpyDeviceToHost));
    return (uint64_t)bad;
}
uint64_t dev_selftest_cvt_pk(int device) {
    dev_require(device);
    MF_HIP(hipSetDevice(device));
    const unsigned long long r = k::selftest_cvt_pk(nullptr);
    if (r == ~0ull) fail(MF_ERR_HIP, "selftest kernel could not run");
    return (uint64_t)r;
}
uint64_t dev_verify_quant_div(int device, float scale, float rcp, int zp, bool u8) {
    dev_require(device);
    MF_HIP(hipSetDevice(device));
    return k::verify_quant_div(scale, rcp, (float)zp, u8 ? 0.0f : -128.0f, u8 ? 255.0f : 127.0f, nullptr);
}
uint64_t dev_checksum_i8(int device, const int8_t *d_in, size_t n, void *stream) {
    dev_require(device);
    unsigned long long *d_res = nullptr, h = 0;
    MF_HIP(hipMalloc((void **)&d_res, 8));
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(d_res, 0, 8, s);
    if (e == hipSuccess && n) {
        k::launch_checksum(d_in, n, d_res, s);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&h, d_res, 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d_res);
    if (e != hipSuccess) fail(MF_ERR_HIP, std::string("checksum: ") + hipGetErrorString(e));
    return (uint64_t)h;
}

} // namespace mf
