// mf_internal.hpp -- internal declarations of libmicroflow_amd.so (not part of the ABI).
//
// Layering:
//   capi.cpp      extern "C" boundary (include/microflow_amd.h), error translation
//   model.cpp     .tflite -> ParsedModel -> prepared device ops -> launch sequence
//   tflite.cpp    minimal FlatBuffers reader for the ~10 TFLite tables the path needs
//   hostmath.cpp  f32 constant preparation in the reference's evaluation order
//   ops.hip       prepared operators: constant folding, kernel routing, launches
//   k_*.hip       the HIP kernels (gfx950), one file per family; k_common.hpp shared helpers
#pragma once
#include "mf_switches.hpp"
#include <cstddef>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "../../include/microflow_amd.h"

namespace mf {

struct Error {
    int code;
    std::string msg;
};
[[noreturn]] void fail(int code, const std::string &msg);
void set_last_error(const std::string &msg);

// ---- host numerics (hostmath.cpp) --------------------------------------
// All individually rounded f32 operations; this translation unit is compiled
// with -ffp-contract=off.
float h_roundf(float x);                              // round half away from zero
float h_expf(float x);                                // libm 0.2 (musl-derived) expf
void h_softmax_table(float in_scale, bool u8, float *table256); // expf over the 256 possible Softmax inputs
int8_t h_sat_i8(float x);                             // Rust `as i8`
int8_t h_quantize(float x, float scale, int8_t zp);   // src/quantize.rs:16-18
int h_quantize_t(float x, float scale, int zp, bool u8); // either element type
void h_preprocess_fc(float iscale, int izp, int in_shape1, const int8_t *w, bool u8, int K, int N,
                     float wscale, int wzp, const int32_t *bias, float bscale, int32_t bzp,
                     float oscale, float *c0, float *c1, int32_t *c2, int32_t *c3);
void h_preprocess_conv(float iscale, int n, const int32_t *bias, const float *bscale, int nbs,
                       const int32_t *bzp, int nbz, const float *fscale, int nfq, float oscale,
                       float *c0, float *c1);
void h_preprocess_pool(float iscale, int izp, float oscale, int ozp, float *c0, float *c1);

// ---- single-fma requantisation: host search (epi_fma.cpp) -------------------
// y_u8(acc) = v_cvt_pk_u8_f32(v_fma_f32(S, bits_as_f32(0x4B400000 + acc + d), C))  -- k_common.hpp, epilogue mode 3
struct FmaForm {
    float S = 0, C = 0;
    int32_t d = 0;
    // at most ONE accumulator per channel is replaced by its neighbour before the fma (patch_delta = +-1; 0: none): the reference's
    // own roundings can put a single step one accumulator off every line (epi_fma.cpp)
    int64_t patch_acc = 0;
    int32_t patch_delta = 0;
};
struct FmaSearchStats {
    int steps = 0;            // steps of the reference staircase inside the reachable accumulator range
    int s_candidates = 0;     // f32 neighbours of S with a non-empty real-arithmetic window
    int s_rank = -1;          // which of them (by window width) gave the solution
    double best_width = 0;    // widest window
    long long pivots_tried = 0, exact_checks = 0;
};
// off: 128 (i8: the form works in the u8 domain, results are XOR-ed back) or 0 (u8); [lo, hi]: the clamp in T's domain;
// [amin, amax]: the accumulators this channel can produce (inside (-2^22, 2^22)).  false: no (S', C', d) reproduces the reference on it.
bool fma_form_search(float A, float S, int off, int lo, int hi, int64_t amin, int64_t amax, FmaForm &out, FmaSearchStats *stats = nullptr,
                     bool allow_patch = false);
int ref_form_eval(float A, float S, int off, int lo, int hi, int64_t acc); // the reference's tail, u8 domain
int fma_form_eval(const FmaForm &f, int64_t acc);                          // the device's instructions, emulated exactly
uint64_t fma_form_mismatches(float A, float S, int off, int lo, int hi, int64_t amin, int64_t amax, const FmaForm &f); // exhaustive, host

// ---- parsed model (tflite.cpp) -------------------------------------------
struct ParsedOp {
    int kind = 0;
    int in_rank = 0, in_shape[4] = {0, 0, 0, 0};
    int out_rank = 0, out_shape[4] = {0, 0, 0, 0};
    int KH = 0, KW = 0, sh = 0, sw = 0, pad = 0, act = 0;
    float in_scale = 0, out_scale = 0;
    int in_zp = 0, out_zp = 0;
    size_t in_elems = 0, out_elems = 0;
    // geometry
    int H = 0, W = 0, C = 0; // conv/dw/pool input
    int N = 0;               // conv filters / dw channels / fc outputs / softmax cols
    int M = 0, K = 0;        // fc rows / depth; softmax rows = M
    int OH = 0, OW = 0;
    std::vector<int8_t> weights; // fc [N][K]; conv [N][KH][KW][C]; dw [KH][KW][N] (raw bytes of T)
    std::vector<int> wzp;        // weight zero points (1 or per channel), values of T
    std::vector<float> c0, c1;
    std::vector<int32_t> c2;
    int32_t c3 = 0;
};
struct ParsedModel {
    bool u8 = false; // element type T of every quantized tensor: INT8 (false) or UINT8 (true)
    int in_rank = 0, in_shape[4] = {0, 0, 0, 0};
    int out_rank = 0, out_shape[4] = {0, 0, 0, 0};
    float in_scale = 0, out_scale = 0;
    int in_zp = 0, out_zp = 0;
    size_t in_elems = 0, out_elems = 0, max_elems = 0;
    std::vector<ParsedOp> ops;
};
ParsedModel parse_tflite(const uint8_t *buf, size_t len);

// ---- prepared device operator (ops.hip) ------------------------------------
struct OpImpl; // device buffers + kernel routing
struct OpSpec {
    int kind = 0;
    // Element type T.  With u8 = true the zero points (izp, ozp, wzp) are u8 values, `weights`
    // are raw u8 bytes, c2/c3 come from the u8 preprocess; activations are exchanged with the
    // operator in the INTERNAL i8 domain (byte ^ 0x80) -- see kernels.hpp.
    bool u8 = false;
    int M = 0, K = 0, N = 0;
    int H = 0, W = 0, C = 0, KH = 0, KW = 0, sh = 1, sw = 1, pad = 0, OH = 0, OW = 0;
    int act = 0;
    int izp = 0;             // zero point of the running input tensor
    float in_scale = 0;      // softmax only
    float oscale = 0;
    int ozp = 0;
    const int8_t *weights = nullptr;
    const int *wzp = nullptr;
    int nq = 0;
    const float *c0 = nullptr;
    const float *c1 = nullptr;
    int nc1 = 0;
    const int32_t *c2 = nullptr;
    int32_t c3 = 0;
    float pool_c0 = 0, pool_c1 = 0;
};
OpImpl *op_create(int device, const OpSpec &spec);
void op_destroy(OpImpl *op);
void op_run(OpImpl *op, const int8_t *d_in, size_t batch, int8_t *d_out, void *stream);
// the ABI's view: a u8 operator exchanges real u8 bytes (two extra byte passes)
void op_run_external(OpImpl *op, const int8_t *d_in, size_t batch, int8_t *d_out, void *stream);
size_t op_in_elems(const OpImpl *op);
size_t op_out_elems(const OpImpl *op);
const char *op_kernel_name(const OpImpl *op);
int op_epilogue_mode(const OpImpl *op); // k_common.hpp epilogue mode (0 .. 3) of the operator's own launch; -1: no requantising epilogue of that kind
bool op_has_fma_epilogue(const OpImpl *op);
int op_fma_patches(const OpImpl *op); // patched accumulators of the operator's single-fma form (0 for most); -1: no such form // the single-fma form (mode 3) was found for every channel and confirmed on the device
void op_set_generic(OpImpl *op, bool generic);
// fuse the model-boundary quantize (f32 -> T) into this operator if it has an f32-input kernel
bool op_set_input_quant(OpImpl *op, float scale, int zp, bool u8);
bool op_accepts_f32(const OpImpl *op);
void op_run_f32(OpImpl *op, const float *d_in, size_t batch, int8_t *d_out, void *stream);
// fused DepthwiseConv2D 3x3 -> Conv2D 1x1 (borrows both operators' device buffers; nullptr when
// the pair has no fused kernel)
struct FusedImpl;
FusedImpl *fused_create(OpImpl *dw, OpImpl *pw);
// fused tail: AveragePool2D (1x1 output) -> Conv2D 1x1 (N <= 8) -> Softmax (one row of N)
FusedImpl *fused_tail_create(OpImpl *pool, OpImpl *conv, OpImpl *softmax);
// fused FullyConnected (row-wave kernel, one row per inference) -> Softmax over its outputs
FusedImpl *fused_fc_softmax_create(OpImpl *fc, OpImpl *softmax);
// a run of identical depthwise + pointwise pair groups as one persistent kernel (borrows their device buffers:
// destroy it before them); nullptr when no stage kernel exists for the shape / count
FusedImpl *fused_stage_create(FusedImpl *const *pairs, int npairs);
// one-input-channel DepthwiseConv2D + the FullyConnected/Softmax group as one kernel (second level, like the stage)
FusedImpl *fused_dwfc_create(OpImpl *dw, FusedImpl *fc_softmax);
// the last depthwise + pointwise pair group (3x3x256) + the pool/head/softmax tail group as one kernel (second level)
FusedImpl *fused_pair_tail_create(FusedImpl *pair, FusedImpl *tail);
// ... and the pair group in front of that pair in the same launch (6x6x128 stride 2 -> 3x3x256 + the above), or nullptr
FusedImpl *fused_front_pair_tail_create(FusedImpl *front_pair, FusedImpl *pair_tail);
FusedImpl *fused_chain_create(FusedImpl *const *single_pair_chains, int n, int force_G = 0); // consecutive run-time-geometry pairs as one launch (k_chain.hip); force_G: images per step (0: the planner's)
bool fused_is_chain_single(const FusedImpl *f);
void fused_chain_partition(FusedImpl *const *single_pair_chains, int n, int *seg_len, bool *unfused, int *seg_G, bool autotune); // seg_G[i]: measured images per step of the chain starting at i (0: the planner's); autotune: time the candidates on the device instead of using the cost model
FusedImpl *fused_quad_create(FusedImpl *pair1, FusedImpl *pair2); // two consecutive pairs in one launch (k_quad.hip)
FusedImpl *fused_quad_stem_create(OpImpl *stem, FusedImpl *quad); // the one-input-channel stem + a quad in one launch, or nullptr
void fused_destroy(FusedImpl *f);
void fused_run(FusedImpl *f, const int8_t *d_in, size_t batch, int8_t *d_out, void *stream);
bool fused_accepts_f32(const FusedImpl *f);
void fused_run_f32(FusedImpl *f, const float *d_in, size_t batch, int8_t *d_out, void *stream);
const char *fused_kernel_name(const FusedImpl *f);
int fused_epilogue_mode(const FusedImpl *f); // epilogue mode of the launch (one for all its requantising operators); -1: none

// zp is a value of T; with u8 the int8 buffer is in the internal domain (byte ^ 0x80)
void dev_quantize(int device, const float *d_in, size_t n, float scale, int zp, bool u8, int8_t *d_out,
                  void *stream);
void dev_dequantize(int device, const int8_t *d_in, size_t n, float scale, int zp, bool u8, float *d_out,
                    void *stream);
void dev_dequantize_u8_raw(int device, const uint8_t *d_in, size_t n, float scale, int zp, float *d_out,
                           void *stream); // real u8 bytes in
void dev_xor80(int device, const int8_t *d_in, size_t n, int8_t *d_out, void *stream);
void dev_synth_i8(int device, uint64_t seed, uint64_t first, size_t n, int8_t *d_out, void *stream);
uint64_t dev_checksum_i8(int device, const int8_t *d_in, size_t n, void *stream);
// exhaustive check of the 3-instruction boundary-quantisation division (k_common.hpp: quant_div): mismatching bytes
uint64_t dev_verify_quant_div(int device, float scale, float rcp, int zp, bool u8);
uint64_t dev_selftest_epilogue(int device, int mode, bool u8, bool have_as, float A, float S, int lo, int hi);
// one channel of the single-fma epilogue on the device, every accumulator of [amin, amax] (k_generic.hip verify_fma_form): mismatches
uint64_t dev_selftest_fma_epilogue(int device, float A, float S, bool u8, int64_t amin, int64_t amax, float S3, float C3, int pivot,
                                   int64_t patch_acc = 0, int patch_delta = 0);
uint64_t dev_selftest_cvt_pk(int device);
int dev_count();
void dev_require(int device); // throws MF_ERR_NO_DEVICE

// ---- model runtime (model.cpp) ----------------------------------------------
struct ModelImpl;
ModelImpl *model_create(const uint8_t *buf, size_t len);
void model_destroy(ModelImpl *m);
const ParsedModel &model_parsed(const ModelImpl *m);
const char *model_op_kernel(const ModelImpl *m, int i);
int model_op_epilogue_mode(const ModelImpl *m, int i); // of the launch that starts at operator i: the minimum over its operators
void model_prepare(ModelImpl *m, int device, size_t max_batch);
void model_set_stream(ModelImpl *m, void *stream);
void model_sync(ModelImpl *m);
void model_set_generic(ModelImpl *m, bool generic);
void model_set_fusion(ModelImpl *m, bool enabled);
// before model_prepare: measure the run-time-geometry chain candidates at creation (ops.hip fused_chain_partition) instead of planning from the cost model
void model_set_autotune(ModelImpl *m, bool enabled);
// replay the device-resident launch sequence as a hipGraph (captured on the 2nd identical call)
void model_set_graph(ModelImpl *m, bool enabled);
uint64_t model_graph_launches(const ModelImpl *m);
// in_f32 or in_i8 (exactly one non-null); out_f32 or out_i8 (exactly one non-null)
void model_run(ModelImpl *m, const float *in_f32, const int8_t *in_i8, size_t batch,
               float *out_f32, int8_t *out_i8, int mem, int last_op);
void model_time_device(ModelImpl *m, const int8_t *d_in, size_t batch, int8_t *d_out, int warmup,
                       int iters, float *avg_ms, float *per_op_ms);

} // namespace mf
