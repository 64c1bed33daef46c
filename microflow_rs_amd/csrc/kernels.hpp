// kernels.hpp -- argument blocks and launchers of the HIP kernels (k_*.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <vector>

namespace mf {
namespace k {

// Per-channel folded epilogue constants live in HBM (tiny, L2-resident):
//   A[c]  = fl32(f32(ozp) + c0[c])      S[c] = c1[c or 0]
//   Kc[c] = -izp * sum_taps w[c] + T * izp * wzp[c]   (FC: c3 - c2[c])
//   wzp[c] expanded to one int per channel
// lo_f / hi_f: activation clamp merged with the int8 saturation, as floats.
//
// Element type u8 (the reference's `T = u8`): activations and weights live in HBM in the i8
// domain (byte ^ 0x80, i.e. value - 128) and every zero point is shifted by -128, which leaves
// each (v - zp) factor -- hence the i32 accumulator -- unchanged.  The f32 epilogue runs in the
// u8 domain exactly as the reference's (A = f32(ozp_u8) + c0, clamp inside [0, 255]); `xr` =
// 0x80 then moves the stored byte back to the i8 domain.  xr = 0 for i8.
struct ConvArgs {
    int H, W, C;        // input (per inference); for depthwise C = input channels
    int N;              // output channels
    int KH, KW, sh, sw, OH, OW;
    int pad_same;
    int izp;
    float lo_f, hi_f;
    const int8_t *w;    // conv [N][KH][KW][C]; depthwise [KH][KW][N]
    const int *wzp;     // [N]
    const float *A;     // [N]
    const float *S;     // [N]
    const int *Kc;      // [N]
    int xr;             // 0 (i8) or 0x80 (u8)
};
struct PoolArgs {
    int H, W, C, KH, KW, sh, sw, OH, OW, pad_same;
    float c0, c1;
    int lo, hi;
    int bias;           // added per in-bounds tap: 0 (i8) or 128 (u8: stored byte + 128 = value)
    float sat_lo, sat_hi; // `as T` saturation: [-128,127] or [0,255]
    int xr;
};
struct FcArgs {
    int K, N;
    int wzp;
    float S;            // c1 (per-tensor weight scale only)
    float lo_f, hi_f;
    const int8_t *w;    // [N][K]
    const float *A;     // [N]
    const int *Kc;      // [N] = c3 - c2[j]
    int xr;
};
struct FcGemmArgs {
    const int8_t *w;    // [N][K]
    const float *A;     // [N]
    const int *Kc;      // [N]
    const int *rowsum;  // [rows] sum_k x[row][k] from the fc_rowsum pre-pass, or nullptr: wzp == 0, or the GEMM forms the sums itself
    // Row sums formed by the GEMM launch itself, in its prologue (k_gemm.hip fc_mfma<..., RSP>): every workgroup sums a slice of its
    // tile row's rows into rs_sums, counts itself in rs_sync[2 tm] and reads the finished sums in its epilogue; rs_sync[2 tm + 1]
    // counts the readers, and the last one zeroes the pair (no memset between launches).  nullptr: not this form.
    int *rs_sums;       // [rows]
    int *rs_sync;       // [2 x row tiles], zero between launches
    int wzp;
    float S;
    float lo_f, hi_f;
    int M, N, K;        // M = total rows (batch * rows per inference)
    uint32_t xr4;       // 0 or 0x80808080
};
struct SoftmaxArgs {
    int rows, cols;
    float oscale, ozp_f;
    const float *exp_table; // [256]: expf(f32(q) * input_scale), index = stored byte + 128
    float sat_lo, sat_hi;
    int xr;
};
// Epilogue mode 3 with patched accumulators (epi_fma.cpp): for up to EPI_PATCH_MAX channels of an operator ONE accumulator bit
// pattern P is replaced by R = P +- 1 before the fma.  EpiPatch is the operator's list (host side); the kernels that support it
// (dwpw_mm, stage_6x6x128) get it as a table of EpiPatchRec indexed by THEIR tiles -- one record per 16-channel tile of a wave, at
// most one patched channel per tile (an operator with two in one tile keeps the two-rounding form in that kernel) -- which a wave
// reads with one scalar load per tile; the replacement is two VALU instructions behind a scalar branch on P != 0.
constexpr int MF_MAGIC_I = 0x4B400000; // the bit-pattern accumulators' offset: the f32 1.5 * 2^23 (k_common.hpp requant_t)
constexpr int EPI_PATCH_MAX = 4;
struct EpiPatch {
    int n;                      // entries in use (0: none -- the only value the other mode-3 kernels accept)
    int ch[EPI_PATCH_MAX];      // output channel
    int P[EPI_PATCH_MAX];       // 0x4B400000 + accumulator + pivot: the bit pattern to replace ...
    int R[EPI_PATCH_MAX];       // ... and its replacement (P + 1 or P - 1)
};
struct EpiPatchRec {            // one tile's patched channel, as the kernel sees it
    int P;                      // the bit pattern to replace; 0: nothing in this tile
    int meta;                   // bits 0..1: which of a lane's four accumulators; bits 2..3: which 16-lane group holds the channel;
                                // bit 4: the replacement is P - 1 (else P + 1); bit 5 (stage kernel, first record of a wave): a
                                // second record follows
};
static inline EpiPatchRec epi_patch_rec(int P, int R, int reg, int lane_group) {
    return EpiPatchRec{P, (reg & 3) | (lane_group & 3) << 2 | (R < P ? 16 : 0)};
}
constexpr int DYNQ_INTS = 8 * 32 + 32; // = DynSteps::INTS (k_common.hpp): the counters of one dynamic step queue
// An operator owns DYNQ_RING counter sets and every launch takes the next one (dq_slot, k_common.hpp): two launches of one
// handle that are in flight together (two streams) then draw from different counters.  A launch leaves its set zeroed, so a
// set is reusable as soon as the launch that used it has finished; a handle may therefore have up to DYNQ_RING launches
// in flight (include/microflow_amd.h states the limit).
constexpr int DYNQ_RING = 32;
struct DwFastArgs {
    const int8_t *w;    // [3][3][C]
    const void *wmm;    // matrix-pipe form of w (k_fused_mm.hip): [C/16 or 1][3 filter rows][64 lanes] x 16 bytes
    const void *wsp;    // the same taps for v_smfmac_i32_16x16x128_i8 + one v_mfma_i32_16x16x32_i8 (k_quad.hip; ops.hip build_dw_sp_weights):
                        // [C/16 or 1][64 lanes] x 32 bytes = {sparse A (16 B), index dword, ninth tap's dense A (8 B), pad}
    const float *A;
    const float *S;
    const int *Kc;
    uint32_t izp4;      // izp replicated in 4 bytes
    float lo_f, hi_f;
    int magic;          // epilogue mode (k_common.hpp requant_t): 1: worst-case |acc| < 2^22 -> bit-pattern int->float
                        // conversion; 2: also clamp == the element type's range and |x| < 2^15 -> saturating pack
    int xr;             // 0 (i8) or 0x80 (u8): see ConvArgs
    int *queue;         // dynamic step queue of the persistent kernels (k_common.hpp DynSteps): DYNQ_INTS zeroed device ints
    unsigned long *qlaunch; // HOST memory: launches of this ring so far (next to the ring in its owner; k_common.hpp dq_slot)
    int qcfg;           // its configuration for this launch (set by the launcher: dq_config)
    // the single-fma form of this operator's epilogue (mode 3, k_common.hpp): C', S', Kc + pivot -- or nullptr when the host
    // search (epi_fma.cpp) or the device check failed for a channel.  use_fma() switches a COPY of the block to it.
    const float *A3, *S3;
    const int *Kc3;
    int npatch3;            // patched accumulators that come with A3 / S3 / Kc3 (0 for most operators)
    const EpiPatchRec *patch; // (device) this launch's patch table, indexed by the kernel's depthwise tiles (16-channel groups), or nullptr
    // with_patches: the launch's kernel applies a patch table, which the launch builder provides (ops.hip); without, an operator
    // that needs patches has no single-fma form
    bool use_fma(bool with_patches = false) {
        if (!A3 || (npatch3 != 0 && !with_patches)) return false;
        return A = A3, S = S3, Kc = Kc3, magic = 3, true;
    }
};
// depthwise with ONE input channel and up to 8 output channels, any filter / stride (speech op 1)
struct DwC1Args {
    int H, W, N, KH, KW, sh, sw, OH, OW, pad_same;
    int izp;
    float lo_f, hi_f;
    int KG, TWP;        // 4-tap groups per filter row = ceil(KW/4); LDS tile row pitch (bytes, multiple of 4)
    const uint32_t *wpack; // [KH][KG][8]: bytes (w[ky][4g..4g+3][c]), zero beyond KW and beyond N
    const float *A;
    const float *S;
    const int *Kc;
    int magic, xr;
};
struct DwStemArgs {
    uint32_t wrow[3][8]; // [ky][c] = bytes (w[ky][0][c], w[ky][1][c], w[ky][2][c], 0)
    uint32_t wmm[64][2]; // the same taps as operand A of v_mfma_i32_16x16x32_i8 (dw3x3_stem8_mm): lane -> 8 K-bytes
    float A[8], S[8];
    int Kc[8];
    uint32_t izp4;
    float lo_f, hi_f;
    int magic, xr;
    float A3[8], S3[8]; // the single-fma form (mode 3), valid when fma_ok
    int Kc3[8];
    int fma_ok;
    bool use_fma() {
        if (!fma_ok) return false;
        for (int c = 0; c < 8; ++c) A[c] = A3[c], S[c] = S3[c], Kc[c] = Kc3[c];
        return magic = 3, true;
    }
    // f32-input variant (boundary quantisation fused into the staging): q = sat(roundf(x / in_scale + in_zp_f))
    float in_scale, in_zp_f, in_sat_lo, in_sat_hi;
    uint32_t in_xr4;
    float in_rcp;        // 1 / in_scale (rounded) for quant_div
    int in_fast;         // 1: the 3-instruction division was verified for these parameters (k_common.hpp: quant_div)
    int *queue;          // dynamic step queue (k_common.hpp DynSteps)
    unsigned long *qlaunch; // HOST memory: launches of this ring so far (next to the ring in its owner; k_common.hpp dq_slot)
    int qcfg;            // set by the launcher (dq_config)
};
struct DwPwArgs;
struct PwArgs {
    const void *wprep;  // [blk][q][tile][kstep][lane] x 16 bytes, MFMA operand-A layout
    const void *wrr;    // register-resident pairs (dwpw_rr): [16-row tile][lane] x 8 bytes, or nullptr
    const float *A;
    const float *S;
    const int *Kc;
    float lo_f, hi_f;
    int magic, xr;
    const float *A3, *S3; // the single-fma form (see DwFastArgs)
    const int *Kc3;
    int npatch3;
    const EpiPatchRec *patch; // (device) indexed by the kernel's pointwise tiles (dwpw_mm: blk * TB + tt), or nullptr
    bool use_fma(bool with_patches = false) {
        if (!A3 || (npatch3 != 0 && !with_patches)) return false;
        return A = A3, S = S3, Kc = Kc3, magic = 3, true;
    }
};

// run-time-geometry kernels (k_rt.hip): any H, W, C
struct DwRtArgs {
    DwFastArgs dw;      // weights [3][3][C], folded constants, clamp, epilogue mode, step queue
    const int *wzp;     // [C] weight zero points (the WZ instance only)
    int H, W, C, OH, OW;
    int G;              // whole images per step (1 in band mode)
    int R;              // output rows per task (2 or 3)
    int BH, NBANDS;     // output rows per band (a multiple of R) and bands per image (1 = whole images)
    int RB;             // tile rows staged per image / band
    int ROW, LP, TILE, BUF; // tile row pitch, side pad, bytes per staged image, bytes per staging buffer
    int NTHR;           // threads per workgroup (256 or 512)
};
struct PwRtArgs {
    const void *wprep;  // [16-channel tile][64-deep k step][lane] x 16 bytes (+ a tile of ones per k step when wzp != 0)
    const float *A;
    const float *S;
    const int *Kc;
    const int *wzp;     // [N] (the WZ instance only)
    float lo_f, hi_f;
    int K, N, KS, NT;   // KS = ceil(K / 64), NT = ceil(N / 16)
    int patch_pitch;    // N rounded up to 16 (pw_rt_lds)
    int TB, NSPLIT;     // pw_rt (weights in registers): tiles per wave block (1..4; 0 = use pw_rt_lds) and waves per chunk (1, 2, 4)
    int magic, xr;
};
// Conv2D with few input channels / DepthwiseConv2D with one input channel, any filter (k_rt.hip: conv_rows_lds)
struct ConvRowsArgs {
    int H, W, C, N, KH, KW, sh, sw, OH, OW;
    int ROWB;            // W * C: image row bytes (a multiple of 4)
    int KG, NP;          // dword groups per window row = ceil(KW C / 4); N rounded up to 8
    int shy, XO, X0;     // halo rows above the image; tile byte of image column 0 / of output column 0's window
    int TWP, TILE, G;    // tile row pitch, bytes per image tile, images per step
    uint32_t izp4;
    float lo_f, hi_f;
    const uint32_t *wpack; // [KH][KG][NP] dwords: byte b = weight (n, ky, 4 kg + b) (zero beyond KW C and beyond N)
    const uint32_t *mask;  // [KG] dwords: 1 in every byte that is a real tap
    const float *A;
    const float *S;
    const int *Kc;
    const int *wzp;
    int magic, xr;
};
// DepthwiseConv2D 3x3 stride 2 SAME with ONE input channel and 4 or 8 outputs, any H x W with W % 16 == 0 (a MobileNet stem at any
// resolution / width 0.5; k_rt.hip: dw3x3_stem_rt)
struct DwStemRtArgs {
    uint32_t wmm[64][4]; // the taps as operand A: lane -> 8 K-bytes (DM = 8, v_mfma_i32_16x16x32_i8) or 16 (DM = 4, 16x16x64)
    float A[8], S[8];
    int Kc[8];
    int H, W, OH, OW, DM;
    int G, TILE;         // images per step; bytes of one image tile [guard 16][izp row][H rows][izp rows]
    int QR, QTOT, NQUAD; // 16-byte output groups per output row / per image; groups of 4 x 16 of them per image
    float inv_qr;
    uint32_t izp4;
    float lo_f, hi_f;
    int magic, xr;
    int *queue;
    unsigned long *qlaunch; // HOST memory: launches of this ring so far (next to the ring in its owner; k_common.hpp dq_slot)
    int qcfg;
};
bool dw_stem_rt_plan(DwStemRtArgs &a, int H, int W, int DM, int OH, int OW);
void launch_dw_stem_rt(const int8_t *in, int8_t *out, const DwStemRtArgs &a, int batch, hipStream_t s);
bool conv_rows_plan(ConvRowsArgs &a, int H, int W, int C, int N, int KH, int KW, int sh, int sw, int OH, int OW, bool pad_same);
void launch_conv_rows(const int8_t *in, int8_t *out, const ConvRowsArgs &a, bool wz, int batch, hipStream_t s);
// Conv2D with any filter, C % 16 == 0, as an MFMA product over K = KH KW C (k_rt.hip: conv_mm_rt)
struct ConvMmArgs {
    int H, W, C, N, KH, KW, sh, sw, OH, OW;
    int padl, padt;      // SAME: (KW - 1) / 2, (KH - 1) / 2 (src/tensor.rs:193); VALID: 0
    int LP, ROW, RB, TILE, G, BH, NBANDS; // tile geometry as DwRtArgs
    int KS, TB, NBLK;    // 64-deep k steps over K = KH KW C; tiles per block (<= 4); blocks
    int dwise;           // 1: DepthwiseConv2D with C % 16 == 0 (src/ops/depthwise_conv_2d.rs:28-105), kernel dw_mm_rt: block b is the 16-channel
                         // group b (TB = 1, NBLK = C / 16), its k steps run over the TAPS only -- lane group g of step ks supplies the
                         // group's 16 bytes of tap 4 ks + g -- against block-diagonal weights (ops.hip build_dw_mm_rt_weights)
    int NTHR;            // threads per workgroup: 256, or 1024 when the weights leave room for one workgroup per CU only
    uint32_t izp4;
    float lo_f, hi_f;
    const void *wprep;   // [block][tile][k step][lane] x 16 bytes (+ a tile of ones per k step when filter zero points != 0)
    const int *tap_off;  // [KS][4]: byte offset, from a pixel's window start, of the 16 tile bytes lane group g supplies in k step ks
    const float *A;
    const float *S;
    const int *Kc;
    const int *wzp;
    int magic, xr;
};
bool conv_mm_plan(ConvMmArgs &a, std::vector<int> &tap, int H, int W, int C, int N, int KH, int KW, int sh, int sw, int OH, int OW,
                  bool pad_same, bool wz, bool dwise = false);
void launch_conv_mm(const int8_t *in, int8_t *out, const ConvMmArgs &a, bool wz, int batch, hipStream_t s);
void launch_dw_mm(const int8_t *in, int8_t *out, const ConvMmArgs &a, int batch, hipStream_t s); // a.dwise: dw_mm_rt (k_rt.hip)
bool dw_rt_plan(DwRtArgs &a, int H, int W, int C, int S, int OH, int OW); // fills the geometry; false: not supported
void launch_dw_rt(const int8_t *in, int8_t *out, const DwRtArgs &a, int S, bool wz, int batch, hipStream_t s);
bool pw_rt_supported(int K, int N, bool wz);
void launch_pw_rt(const int8_t *in, int8_t *out, const PwRtArgs &a, bool wz, long long npix, hipStream_t s);

// fused DepthwiseConv2D 3x3 -> Conv2D 1x1: both argument blocks
struct DwPwArgs {
    DwFastArgs dw;
    PwArgs pw;
};

// ---- run-time-geometry fused chains (k_chain.hip) ----
// A chain = 1 .. CHAIN_MAX consecutive DepthwiseConv2D 3x3 (SAME, stride 1 or 2) + Conv2D 1x1 pairs of ANY height / width with
// C % 16 == 0 input and N % 16 == 0 output channels, run in ONE launch on G whole images per workgroup step; every tensor between
// the chain's first input and last output stays in LDS.  Geometry, work split and LDS plan are made by the host (chain_plan) and
// handed over in a device table the kernel reads with scalar loads.
constexpr int CHAIN_MAX = 16;
struct ChainGeom {           // one pair, as the model states it
    int H, W, C, S, OH, OW, N;
    uint32_t izp4;           // input zero point of the depthwise, in every byte
};
struct ChainPair {           // one pair, planned (device table entry)
    int H, W, C, S, OH, OW, N;
    int NQ, lgNQ, KS, TB, NBLK;    // 16-channel groups (lg = -1: not a power of two); 64-deep k steps; output tiles per wave block; blocks
    int ROW, TILE, tile_off;       // input tile: row pitch, image pitch, LDS offset
    int swz_sh, swz_mask;          // 16-byte group index of tile column x (halo column = 0) is XOR-ed with (x >> sh) & mask
    int lgCX, lgCY;                // depthwise unit = CG images x CY rows x CX columns (CG CY CX = 16)
    int UG, UY, UX, NU;            // units per channel group along images / rows / columns, and their product
    int P, NCH, PLANE, plimit_img; // output pixels per step, 16-pixel chunks, MID plane pitch, output pixels per image
    int dst_off, dROW, dTILE, dC, dswz_sh, dswz_mask, otab_off; // next pair's tile (dst_off < 0: the chain's output, HBM)
    float dw_lo, dw_hi, pw_lo, pw_hi;
    int ustart[16][4];             // first depthwise unit of wave w: (q, ux, j = ug * UY + uy, -)
    int ucount[16];
    int single_q;                  // every wave's unit range lies inside one channel group
    int pad_;
    const int *rtab;               // [UG * UY] x {tile offset, MID offset} of unit (ug, uy) relative to unit (0, 0) (chain_rtab)
    const void *dw_wmm;            // DwFastArgs::wmm
    const float *dwA, *dwS;
    const int *dwK;
    const void *pw_w;              // build_pw_rt_reg_weights(K, N, group 1, TB, NBLK)
    const float *pwA, *pwS;
    const int *pwK;
};
struct ChainArgs {
    const ChainPair *pairs;
    int npairs, G;
    int lds_bytes, mid_off, q_off;
    int dbuf, dbuf_stride;         // pair 0's input tile is double buffered (the next step's images are staged a whole step ahead)
    int stage_after;               // the next step's images are DMA-staged after the depthwise phase of this pair (the last reader of pair 0's tile region)
    int nfill;
    int fill_off[CHAIN_MAX], fill_bytes[CHAIN_MAX];
    uint32_t fill_izp4[CHAIN_MAX];
    int *queue;
    unsigned long *qlaunch; // HOST memory: launches of this ring so far (next to the ring in its owner; k_common.hpp dq_slot)
    int qcfg;
    int KSC;                       // template selector: 1, 2 or 4 k steps (max over the pairs)
    int max_cg;                    // images per depthwise column grid (G is a multiple of it)
    int resident;                  // single pair whose waves each stay inside one channel group: the reload-free kernel instance
    int nwave;                     // waves per workgroup: 8, or 16 when the LDS plan admits one workgroup per CU only (KSC == 1)
    double est_us_per_image;       // the planner's cost estimate (per CU), for choosing between chainings
    int magic, xr;                 // epilogue mode of the whole chain (min over its operators), element type
    double hbm_bytes, requant_bytes; // per image (the step queue's duration estimate)
};
// plans `n` pairs as one chain: fills `pairs` (everything but the operand pointers and clamps) and the LDS part of `a`.
// false: no plan (a channel count, the LDS budget, ...).  `lds_budget`: bytes a workgroup may use.
bool chain_plan(const ChainGeom *g, int n, ChainPair *pairs, ChainArgs &a, int lds_budget, int force_G = 0, int force_dbuf = -1); // force_*: the caller's G / double buffering
// estimated microseconds per image and CU of the same pairs run one operator at a time (the run-time-geometry kernels of k_rt.hip)
double chain_unfused_us_per_image(const ChainGeom *g, int n);
void chain_rtab(const ChainPair &c, std::vector<int> &out); // the unit offset table ChainPair::rtab points to
void launch_chain(const int8_t *in, int8_t *out, const ChainArgs &a, int batch, hipStream_t s);

// two consecutive pairs in one launch (k_quad.hip)
struct QuadArgs {
    DwPwArgs a, b;
    // the one-input-channel stem in front of pair a (k_quad.hip, STEM instance: person_detect ops 0..4 in one launch), or nullptr.
    // Device dwords: [0, 128) operand A of v_mfma_i32_16x16x32_i8 per lane (DwStemArgs::wmm), [128, 136) A, [136, 144) S,
    // [144, 152) Kc of the 8 output channels
    const uint32_t *stem;
    uint32_t stem_izp4;
    float stem_lo, stem_hi;
    int stem_magic;
    // the f32 entry of the stem instance (launch_quad_f32): the model-boundary quantisation q = sat(roundf(x / in_scale + in_zp_f)),
    // DwStemArgs' fields (the f32-input stem kernel's); f32_ok: they are set
    float in_scale, in_zp_f, in_sat_lo, in_sat_hi, in_rcp;
    uint32_t in_xr4;
    int in_fast, f32_ok;
};
const char *quad_name(int H, int W, int C, int S, int N, int H2, int W2, int C2, int S2, int N2);
const char *quad_stem_name(int SH, int SW, int H, int W, int C, int S, int N, int H2, int W2, int C2, int S2, int N2); // with a.stem set
bool launch_quad(int H, int W, int C, int S, int N, int H2, int W2, int C2, int S2, int N2, const int8_t *in, int8_t *out, const QuadArgs &a,
                 int batch, hipStream_t s);
// person_detect ops 9..12 -- the two C = 64 pairs, intermediate tensors through LDS (k_quad_mm.hip): QuadArgs' a / b blocks as dwpw_mm
// takes them (dw.wmm, pw.wprep, the pairs' own patch tables)
bool quad_mm_shape(int H, int W, int C, int S, int N, int H2, int W2, int C2, int S2, int N2);
void launch_quad_mm(const int8_t *in, int8_t *out, const QuadArgs &a, int batch, hipStream_t s);
bool launch_quad_f32(int H, int W, int C, int S, int N, int H2, int W2, int C2, int S2, int N2, const float *in, int8_t *out, const QuadArgs &a,
                     int batch, hipStream_t s);

// fused network tail: AveragePool2D (to 1x1) -> Conv2D 1x1 (N <= 8) -> [Reshape] -> Softmax
struct TailArgs {
    int H, W, C, N;          // pool input; head outputs
    int ntaps;               // in-range taps of the single pooling window
    int tap_off[64];         // their element offsets (iy * W + ix) * C
    float inv_len;           // 1.0f / f32(ntaps), correctly rounded on the host
    float pool_c0, pool_c1;
    int pool_lo, pool_hi;
    const int8_t *w;         // head filters [N][C]
    const int *wzp;          // [N]
    const float *A;          // [N]
    const float *S;          // [N]
    const int *Kc;           // [N]
    float lo_f, hi_f;        // head activation clamp
    const float *exp_table;  // softmax table [256]
    float sm_oscale, sm_ozp_f;
    // element type: 0 / 128 added per pooled tap (stored byte -> value), the `as T` saturation ranges, and the byte
    // flip between the value domain and the stored i8 domain (0 for i8, 0x80 for u8: see ConvArgs)
    int pool_bias;
    float pool_sat_lo, pool_sat_hi, sm_sat_lo, sm_sat_hi;
    int xr;
};

// DepthwiseConv2D (one input channel) -> FullyConnected -> Softmax in one launch (k_dwfc.hip): speech.tflite ops 1..3.
// Geometry of the one compiled instance; the kernel's comment explains the layout.
struct DwFcGeom {
    static constexpr int H = 49, W = 40, KH = 10, KW = 8, S = 2, OH = 25, OW = 20;
    static constexpr int IMGS = 16;                 // images per workgroup step = the 16 MFMA columns
    static constexpr int PT = (KH - 1) / 2, PL = (KW - 1) / 2; // SAME padding: rows above / columns left of the image
    static constexpr int XO = 4;                    // tile byte of image column 0 (dword aligned staging)
    static constexpr int E0 = XO - PL;              // window of pixel ox starts at tile byte S * ox + E0
    static constexpr int NM = OW / 4, NT = (OH + 1) / 2, NU = NT * NM; // 8-byte groups per row, row pairs, units per shift
    static constexpr int ROWS = 4 * (NT - 1) + 12;  // tile rows read by the last row pair
    static constexpr int RP = 56;                   // row pitch, bytes (>= 8 (NM - 1) + 16; 14 words: see bank note)
    static constexpr int TILE = ((ROWS * RP + 15) / 32) * 32 + 16; // image pitch = 16 mod 32 bytes (4 mod 8 words)
    static constexpr int FCW_BYTES = NU * 4 * 5 * 16; // FullyConnected operand A, rows 0..4 only: [unit][lane group][row][16 B]
    static_assert(S == 2 && KW == 8 && OW % 4 == 0 && KH + S <= 12, "dwc1_fc_softmax: window / row-pair scheme");
    static_assert(8 * (NM - 1) + 16 <= RP && XO + W <= RP && TILE >= ROWS * RP && PT + H <= ROWS, "dwc1_fc_softmax: tile");
};
struct DwFcArgs {
    const void *wA;          // [4 shifts][3 k-steps][64 lanes] x 16 B: depthwise taps as MFMA operand A
    const void *wfc;         // [NU][4 lane groups][5 rows][16 B]: FullyConnected weights (rows 0..3), ones (row 4)
    const float *dwA, *dwS;
    const int *dwKc;
    float dw_lo, dw_hi;
    uint32_t izp4;
    int magic, xr;           // of the depthwise operator
    FcArgs fc;
    SoftmaxArgs sm;
};
#ifndef MF_DWFC_THREADS
#define MF_DWFC_THREADS 512
#endif
bool dwfc_supported(int H, int W, int KH, int KW, int sh, int sw, int OH, int OW, int DM, int NFC);
const char *dwfc_name();
void launch_dwfc(const int8_t *in, int8_t *out, const DwFcArgs &a, size_t batch, hipStream_t s);

// DepthwiseConv2D 3x3 + Conv2D 1x1 on a 3x3x256 tensor + AveragePool2D + head Conv2D + Softmax in one launch
// (k_tail3.hip: person_detect ops 25..30)
struct PairTailArgs {
    const void *dw_wmm;      // depthwise taps, matrix-pipe form (DwFastArgs::wmm)
    const float *dwA, *dwS;
    const int *dwK;          // folded constants (+ 0x4B400000, the bit-pattern int->float offset, added on the host when magic != 0)
    float dw_lo, dw_hi;
    uint32_t izp4;
    int H, C;                // the tensor is H x H x C (H = 2, 3 or 4; C = 256 or 128)
    int magic;               // 1: bit-pattern epilogue (|acc| < 2^22 for both convolutions), 0: v_cvt
    const void *pw_w;        // pointwise weights [N/16][K/64][64 lanes] x 16 bytes (row r of tile tt = channel 16 tt + r)
    const float *pwA, *pwS;
    const int *pwK;          // + 0x4B400000
    float pw_lo, pw_hi;
    TailArgs tail;
};
// ... with the pair IN FRONT of that pair in the same launch (person_detect ops 23..30): DepthwiseConv2D 3x3 stride 2 on 2H x 2H x C/2
// + Conv2D 1x1 C/2 -> C, whose H x H x C output is pair3_tail's input tensor and never leaves LDS
struct PairFrontArgs {
    const void *dw_wmm;      // depthwise taps, matrix-pipe form
    const float *dwA, *dwS;
    const int *dwK;          // (+ 0x4B400000 when PairTailArgs::magic)
    float dw_lo, dw_hi;
    uint32_t izp4;           // zero point of the front depthwise's input, in every byte
    const void *pw_w;        // pointwise weights [C/16][C/2/64][64 lanes] x 16 bytes
    const float *pwA, *pwS;
    const int *pwK;
    float pw_lo, pw_hi;
};
bool pair_front_supported(int H, int W, int C, int S, int N, int tailH, int tailC);
void launch_pair_front_tail(const int8_t *in, int8_t *out, const PairTailArgs &a, const PairFrontArgs &fr, size_t batch, hipStream_t s);
bool pair_tail_supported(int H, int W, int C, int N_pw, int N_head, int ntaps);
const char *pair_tail_name(int H, int C);
void launch_pair_tail(const int8_t *in, int8_t *out, const PairTailArgs &a, size_t batch, hipStream_t s);

// a run of identical depthwise + pointwise pairs on a small tensor as one persistent kernel (k_stage.hip)
struct StagePair {
    const void *dw_wmm;      // depthwise taps, matrix-pipe form (DwFastArgs::wmm)
    const float *dwA, *dwS;
    const int *dwK;          // folded constants + 0x4B400000 (the bit-pattern int->float offset, added on the host so
                             // that an operand fetch has no dependent instruction and can stay in flight)
    float dw_lo, dw_hi;
    const void *pw_w;        // pointwise weights [N/16][K/64][64 lanes] x 16 bytes: row r of tile tt = channel 16 tt + r
    const float *pwA, *pwS;
    const int *pwK;          // + 0x4B400000, like dwK
    float pw_lo, pw_hi;
};
struct StageArgs {
    const StagePair *pairs;  // [number of pairs], in device memory
    uint32_t izp4;           // zero point of every depthwise input of the run (they must agree: one halo fill)
    uint32_t xr4;            // 0 (i8) or 0x80808080 (u8): XOR of every stored dword
    int *queue;              // dynamic step queue (k_common.hpp DynSteps)
    unsigned long *qlaunch; // HOST memory: launches of this ring so far (next to the ring in its owner; k_common.hpp dq_slot)
    int qcfg;                // set by the launcher (dq_config)
    int nrep;                // number of pairs in the run (set by the launcher)
    int mode;                // epilogue mode of the whole run (k_common.hpp): 1, or 2 when every clamp is the type's range; 3: single fma
    const EpiPatchRec *patch_tab; // mode 3: [pair][depthwise, pointwise][wave = 16-channel group][2] (device), or nullptr: no patched channel
};

// shapes with a compiled fast depthwise kernel: H, W, C, stride, images per step, threads per
// workgroup.  LDS (two staging buffers) decides how many workgroups fit a CU; the thread count
// is chosen so that 4-6 waves per SIMD are resident (the kernels are VALU-bound).
#define MF_DW_SHAPES(X)      \
    X(48, 48, 8, 1, 1, 512)  \
    X(48, 48, 16, 2, 1, 512) \
    X(24, 24, 32, 1, 1, 512) \
    X(24, 24, 32, 2, 1, 512) \
    X(12, 12, 64, 1, 2, 512) \
    X(12, 12, 64, 2, 2, 256) \
    X(6, 6, 128, 1, 4, 512)  \
    X(6, 6, 128, 2, 4, 256)  \
    X(3, 3, 256, 1, 4, 256)

// tuning candidates for the table above (MF_DW_ALT=<i>, scripts/tune_dw.sh); empty in the product
// build.  Last sweep (r01, 25 candidates): 12x12x64 s1 -> (G=2, 512 thr) -11 %, 6x6x128 s1 ->
// (G=4, 512 thr) -16 %, 24x24x32 s2 -> 512 thr -4 %; the rest already at their best.
#define MF_DW_ALT_SHAPES(X)

// Fused DepthwiseConv2D 3x3 + Conv2D 1x1 pairs with the depthwise taps on the matrix pipe (dwpw_mm, k_fused_mm.hip): H, W, C, stride, N,
// images per step, threads, double-buffered staging, then the column grid of a depthwise unit -- CG images x
// CY rows x (16 / CG / CY) x-positions, ORD = which of them varies fastest over the 16 MFMA columns
// (0 gyx, 1 gxy, 2 ygx, 3 yxg, 4 xgy, 5 xyg) -- the row pitch padding (bytes) and the tile swizzle TS
// (nibble i = 1 + the bit of x that flips bit i of the 16-byte group index inside a pixel; 0 = none), and
// WPE = waves per SIMD the register allocation leaves room for (workgroups the LDS admits x waves / 4).
// Grid / padding / swizzle come from scripts/model/dwmm_search.py: every ds_read_b128 of the tap loads is
// bank-conflict free except on the 6x6x128 pair (2-way).
#define MF_DWMM_SHAPES(X)                                 \
    X(48, 48, 8, 1, 16, 1, 512, 1, 1, 4, 0, 32, 0x000, 4)    \
    X(48, 48, 16, 2, 32, 1, 768, 1, 1, 2, 0, 32, 0x000, 3)   \
    X(24, 24, 32, 1, 32, 1, 512, 1, 1, 2, 0, 0, 0x002, 4)    \
    X(24, 24, 32, 2, 64, 2, 256, 0, 1, 4, 0, 32, 0x002, 2)   \
    X(12, 12, 64, 1, 64, 2, 512, 0, 1, 4, 0, 16, 0x000, 4)   \
    X(12, 12, 64, 2, 128, 4, 256, 0, 4, 2, 0, 32, 0x021, 2)  \
    X(6, 6, 128, 1, 128, 8, 512, 0, 4, 2, 2, 16, 0x101, 2)   \
    X(6, 6, 128, 2, 256, 8, 512, 0, 4, 1, 3, 16, 0x321, 2)   \
    X(3, 3, 256, 1, 256, 8, 512, 0, 4, 1, 0, 64, 0x021, 2)
// tuning candidates (MF_DWMM_ALT=<i>).  Round 3 re-sweep (new epilogue + dynamic step queue, scripts/r03_f.sh): 12x12x64 s1
// (G = 4, 512 thr, 2 waves per SIMD: one workgroup per CU) -> (G = 2, 512 thr, 4 waves per SIMD: two workgroups per CU)
// 0.31 -> 0.26 ms; nothing else moved.
#define MF_DWMM_ALT_SHAPES(X)                             \
    X(12, 12, 64, 1, 64, 2, 256, 0, 1, 4, 0, 16, 0x000, 3)   \
    X(12, 12, 64, 1, 64, 4, 512, 0, 1, 4, 0, 16, 0x000, 2)   \
    X(12, 12, 64, 1, 64, 1, 256, 1, 1, 4, 0, 16, 0x000, 3)   \
    X(6, 6, 128, 1, 128, 4, 256, 0, 4, 2, 2, 16, 0x001, 3)   \
    X(6, 6, 128, 1, 128, 4, 512, 0, 4, 2, 2, 16, 0x101, 3)   \
    X(6, 6, 128, 1, 128, 8, 1024, 0, 4, 2, 2, 16, 0x101, 4)  \
    X(6, 6, 128, 2, 256, 4, 256, 0, 4, 1, 3, 16, 0x021, 3)   \
    X(3, 3, 256, 1, 256, 4, 256, 0, 4, 1, 0, 64, 0x021, 2)   \
    X(3, 3, 256, 1, 256, 4, 512, 0, 4, 1, 0, 64, 0x021, 2)

// Pairs with C <= 32 whose intermediate tensor stays in registers (dwpw_rr, k_fused_mm.hip); same columns.
#define MF_DWRR_SHAPES(X)                                   \
    X(48, 48, 8, 1, 16, 1, 512, 1, 1, 4, 0, 32, 0x000, 4)   \
    X(48, 48, 16, 2, 32, 1, 768, 1, 1, 2, 0, 32, 0x000, 3)  \
    X(24, 24, 32, 1, 32, 1, 256, 1, 1, 2, 0, 0, 0x002, 3)   \
    X(24, 24, 32, 2, 64, 1, 192, 1, 1, 4, 0, 32, 0x002, 2)
// tuning candidates (MF_DWRR_ALT=<index>, scripts/tune_fused.sh MF_DWRR_ALT 2).  Last sweep (r02): 192 / 384 / 768 threads, one
// staging buffer, 2 / 4 waves per SIMD for 24x24x32 and 256 / 384 / 768 threads, 3 waves per SIMD for 48x48x8 were
// all 3 .. 50 % slower than the shipped rows; two of them are kept here as the template for the next sweep.
#define MF_DWRR_ALT_SHAPES(X)                               \
    X(24, 24, 32, 1, 32, 1, 256, 1, 1, 2, 0, 0, 0x002, 2)   \
    X(48, 48, 8, 1, 16, 1, 512, 1, 1, 4, 0, 32, 0x000, 3)

// (K = input channels, N = output channels) with a compiled pointwise MFMA kernel
#define MF_PW_SHAPES(X) \
    X(8, 16)            \
    X(16, 32)           \
    X(32, 32)           \
    X(32, 64)           \
    X(64, 64)           \
    X(64, 128)          \
    X(128, 128)         \
    X(128, 256)         \
    X(256, 256)

void launch_conv2d_generic(const int8_t *in, int8_t *out, const ConvArgs &a, size_t batch, hipStream_t s);
void launch_dwconv_generic(const int8_t *in, int8_t *out, const ConvArgs &a, size_t batch, hipStream_t s);
bool conv1x1_rowwave_supported(const ConvArgs &a); // 1x1 filter, stride 1, N <= 8, C % 4 == 0
void launch_conv1x1_rowwave(const int8_t *in, int8_t *out, const ConvArgs &a, size_t batch, hipStream_t s);
void launch_avgpool_generic(const int8_t *in, int8_t *out, const PoolArgs &a, size_t batch, hipStream_t s);
void launch_avgpool_c4(const int8_t *in, int8_t *out, const PoolArgs &a, size_t batch, hipStream_t s); // C % 4 == 0
void launch_fc_generic(const int8_t *in, int8_t *out, const FcArgs &a, size_t rows, hipStream_t s);
bool launch_fc_rowwave(const int8_t *in, int8_t *out, const FcArgs &a, size_t rows, hipStream_t s);
bool launch_fc_rowwave_softmax(const int8_t *in, int8_t *out, const FcArgs &a, const SoftmaxArgs &sm, size_t rows,
                               hipStream_t s);
// int8 MFMA GEMM (M >= 64 rows -- a ragged last row tile is masked --, N % 128 == 0, K % 128 == 0)
bool fc_mfma_supported(size_t rows, int N, int K);
void launch_fc_rowsum(const int8_t *in, int *rowsum, size_t rows, int K, hipStream_t s);
void launch_fc_mfma(const int8_t *in, int8_t *out, const FcGemmArgs &a, hipStream_t s);
bool fc_mfma_rowsum_prepass();
bool fc_mfma_rowsum_prologue(size_t rows, int N); // the in-launch row sums (fc_mfma<..., RSP>) take this shape
bool fc_mfma_rowsum_prologue(size_t rows, int N); // the RSP instance takes this shape (256 x 256 tiles, tile columns dividing 256) // true (default): fc_rowsum runs in front of the GEMM; MF_FC_ROWSUM_FOLD=1: the GEMM forms the sums itself
void launch_softmax(const int8_t *in, int8_t *out, const SoftmaxArgs &a, size_t batch, hipStream_t s);
// number of float bit patterns (of all 2^32) whose quantised byte differs between quant_div's fast form and the true
// division, for these parameters; synchronises the stream
unsigned long long verify_quant_div(float scale, float rcp, float zp_f, float sat_lo, float sat_hi, hipStream_t s);
// self-tests of the epilogue forms (k_common.hpp): number of bytes that differ from round 2's form (a) over every float bit
// pattern taken as the pre-rounding value x, (b) over every accumulator in (-2^22, 2^22) at one (A, S); ~0 = could not run
unsigned long long selftest_rounding(int mode, bool u8, float lo, float hi, hipStream_t s);
unsigned long long selftest_requant(int mode, bool u8, float A, float S, float lo, float hi, hipStream_t s);
unsigned long long selftest_cvt_pk(hipStream_t s); // v_cvt_pk_u8_f32 over all 2^32 inputs against the model epi_fma.cpp uses
// exhaustive device check of the single-fma epilogue of one operator (k_generic.hip); device arrays of n channels
bool verify_fma_form(const float *A, const float *S, const float *C3, const float *S3, const int *piv, const int *amin, const int *amax,
                     const int *patchP, const int *patchR, int n, float lo, float hi, bool u8, unsigned long long *bad, hipStream_t s);
void launch_quantize(const float *in, int8_t *out, size_t n, float scale, float zp_f, bool u8, hipStream_t s);
void launch_xor80(const int8_t *in, int8_t *out, size_t n, hipStream_t s);
void launch_dequantize(const int8_t *in, float *out, size_t n, float scale, float zp_f, bool raw_u8, hipStream_t s);
void launch_synth(int8_t *out, size_t n, uint64_t seed, uint64_t first, hipStream_t s);
void launch_checksum(const int8_t *in, size_t n, unsigned long long *result, hipStream_t s);

const char *dw_fast_name(int H, int W, int C, int S);
bool launch_dw_fast(int H, int W, int C, int S, const int8_t *in, int8_t *out, const DwFastArgs &a,
                    int batch, hipStream_t s);
bool dw_c1_supported(const DwC1Args &a);
void launch_dw_c1(const int8_t *in, int8_t *out, const DwC1Args &a, size_t batch, hipStream_t s);
const char *dw_stem_name(int H, int W, int DM, int S);
bool launch_dw_stem(int H, int W, int DM, int S, const int8_t *in, int8_t *out, const DwStemArgs &a,
                    int batch, hipStream_t s, bool f32_input = false);
const char *dwpw_name(int H, int W, int C, int S, int N);
bool launch_dwpw(int H, int W, int C, int S, int N, const int8_t *in, int8_t *out, const DwPwArgs &a,
                 int batch, hipStream_t s);
const char *dwpw_mm_name(int H, int W, int C, int S, int N);
// the depthwise operator alone with its taps on the matrix pipe (layer-wise execution; same shapes as dwpw_mm)
const char *dw_mm_name(int H, int W, int C, int S);
bool launch_dw_mm(int H, int W, int C, int S, const int8_t *in, int8_t *out, const DwFastArgs &dw, int batch, hipStream_t s);
bool launch_dwpw_mm(int H, int W, int C, int S, int N, const int8_t *in, int8_t *out, const DwPwArgs &a,
                    int batch, hipStream_t s);
const char *dwpw_rr_name(int H, int W, int C, int S, int N);
bool launch_dwpw_rr(int H, int W, int C, int S, int N, const int8_t *in, int8_t *out, const DwPwArgs &a,
                    int batch, hipStream_t s);
const char *stage_name(int H, int W, int C, int npairs);
bool launch_stage(int H, int W, int C, int npairs, const int8_t *in, int8_t *out, const StageArgs &a, int batch, hipStream_t s);
bool tail_supported(int C, int N, int ntaps);
void launch_tail(const int8_t *in, int8_t *out, const TailArgs &a, size_t batch, hipStream_t s);
const char *pw_name(int K, int N);
bool launch_pw(int K, int N, const int8_t *in, int8_t *out, const PwArgs &a, long long npix, hipStream_t s);

} // namespace k
} // namespace mf
