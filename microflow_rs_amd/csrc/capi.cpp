// capi.cpp -- the extern "C" boundary declared in include/microflow_amd.h.
// Nothing crosses it but plain pointers, sizes and status codes; C++ exceptions are
// translated to mf_status + a thread-local message.
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "mf_internal.hpp"

namespace mf {
static thread_local std::string g_last_error;
void set_last_error(const std::string &msg) { g_last_error = msg; }
void fail(int code, const std::string &msg) { throw Error{code, msg}; }
} // namespace mf

struct mf_op {
    mf::OpImpl *impl;
};
struct mf_model {
    mf::ModelImpl *impl;
};

#define MF_TRY(...)                                    \
    try {                                              \
        __VA_ARGS__;                                   \
        return MF_OK;                                  \
    } catch (const mf::Error &e) {                     \
        mf::set_last_error(e.msg);                     \
        return e.code;                                 \
    } catch (const std::bad_alloc &) {                 \
        mf::set_last_error("out of host memory");      \
        return MF_ERR_OOM;                             \
    } catch (const std::exception &e) {                \
        mf::set_last_error(e.what());                  \
        return MF_ERR_INVALID_ARG;                     \
    }

#define MF_NEED(cond)                                                         \
    if (!(cond)) mf::fail(MF_ERR_INVALID_ARG, "invalid argument: " #cond)

extern "C" {

const char *mf_last_error(void) { return mf::g_last_error.c_str(); }
int mf_abi_version(void) { return MF_ABI_VERSION; }
#ifndef MF_BUILD_EXTRA
#define MF_BUILD_EXTRA ""
#endif
const char *mf_build_info(void) { return MF_BUILD_EXTRA; }
int mf_device_count(void) { return mf::dev_count(); }

// ---- 1. constant preparation ------------------------------------------------
int mf_preprocess_fully_connected(float input_scale, int8_t input_zero_point, int in_shape1,
                                  const int8_t *weights, int K, int N, float weights_scale,
                                  int8_t weights_zero_point, const int32_t *bias, float bias_scale,
                                  int32_t bias_zero_point, float output_scale, float *c0,
                                  float *c1, int32_t *c2, int32_t *c3) {
    MF_TRY({
        MF_NEED(weights && bias && c0 && c1 && c2 && c3 && K > 0 && N > 0);
        mf::h_preprocess_fc(input_scale, input_zero_point, in_shape1, weights, false, K, N, weights_scale,
                            weights_zero_point, bias, bias_scale, bias_zero_point, output_scale, c0,
                            c1, c2, c3);
    })
}
int mf_preprocess_fully_connected_u8(float input_scale, uint8_t input_zero_point, int in_shape1,
                                     const uint8_t *weights, int K, int N, float weights_scale,
                                     uint8_t weights_zero_point, const int32_t *bias, float bias_scale,
                                     int32_t bias_zero_point, float output_scale, float *c0,
                                     float *c1, int32_t *c2, int32_t *c3) {
    MF_TRY({
        MF_NEED(weights && bias && c0 && c1 && c2 && c3 && K > 0 && N > 0);
        mf::h_preprocess_fc(input_scale, input_zero_point, in_shape1, (const int8_t *)weights, true, K, N,
                            weights_scale, weights_zero_point, bias, bias_scale, bias_zero_point,
                            output_scale, c0, c1, c2, c3);
    })
}

int mf_preprocess_conv_2d(float input_scale, int n, const int32_t *bias, const float *bias_scale,
                          const int32_t *bias_zero_point, int nbq, const float *filter_scale,
                          int nfq, float output_scale, float *c0, float *c1) {
    MF_TRY({
        MF_NEED(bias && bias_scale && bias_zero_point && filter_scale && c0 && c1 && n > 0 &&
                nbq > 0 && nfq > 0);
        mf::h_preprocess_conv(input_scale, n, bias, bias_scale, nbq, bias_zero_point, nbq, filter_scale,
                              nfq, output_scale, c0, c1);
    })
}

int mf_preprocess_depthwise_conv_2d(float input_scale, int n, const int32_t *bias,
                                    const float *bias_scale, const int32_t *bias_zero_point,
                                    int nbq, const float *weights_scale, int nfq,
                                    float output_scale, float *c0, float *c1) {
    // identical formulas (depthwise_conv_2d.rs:106-119 vs conv_2d.rs:100-113)
    return mf_preprocess_conv_2d(input_scale, n, bias, bias_scale, bias_zero_point, nbq,
                                 weights_scale, nfq, output_scale, c0, c1);
}

int mf_preprocess_average_pool_2d(float input_scale, int8_t input_zero_point, float output_scale,
                                  int8_t output_zero_point, float *c0, float *c1) {
    MF_TRY({
        MF_NEED(c0 && c1);
        mf::h_preprocess_pool(input_scale, input_zero_point, output_scale, output_zero_point, c0, c1);
    })
}

int mf_preprocess_average_pool_2d_u8(float input_scale, uint8_t input_zero_point, float output_scale,
                                     uint8_t output_zero_point, float *c0, float *c1) {
    MF_TRY({
        MF_NEED(c0 && c1);
        mf::h_preprocess_pool(input_scale, input_zero_point, output_scale, output_zero_point, c0, c1);
    })
}

int mf_preprocess_softmax(float input_scale, int is_u8, float *exp_table) {
    MF_TRY({
        MF_NEED(exp_table);
        mf::h_softmax_table(input_scale, is_u8 != 0, exp_table);
    })
}

// ---- 2. prepared operators ----------------------------------------------------
static int check_act_arg(int a) {
    if (a != MF_ACT_NONE && a != MF_ACT_RELU && a != MF_ACT_RELU6)
        mf::fail(MF_ERR_UNSUPPORTED, "unsupported fused activation: " + std::to_string(a) +
                                         ". Supported activations are NONE, RELU, and RELU6");
    return a;
}
static int check_pad_arg(int p) {
    if (p != MF_PAD_SAME && p != MF_PAD_VALID) mf::fail(MF_ERR_INVALID_ARG, "bad view_padding");
    return p;
}
static void wrap_op(mf::OpImpl *impl, mf_op **out) {
    mf_op *h = new (std::nothrow) mf_op{impl};
    if (!h) {
        mf::op_destroy(impl);
        mf::fail(MF_ERR_OOM, "out of host memory");
    }
    *out = h;
}

// One implementation per operator for both element types: zero points arrive widened to int,
// weights as raw bytes of T.
static void create_fc(int device, bool u8, int M, int K, int N, const void *weights, int wzp,
                      float output_scale, int ozp, int act, const float *c0, float c1,
                      const int32_t *c2, int32_t c3, mf_op **op) {
    MF_NEED(op && weights && c0 && c2);
    mf::OpSpec s;
    s.kind = MF_OP_FULLY_CONNECTED, s.u8 = u8;
    s.M = M, s.K = K, s.N = N;
    s.weights = (const int8_t *)weights, s.wzp = &wzp, s.nq = 1;
    s.oscale = output_scale, s.ozp = ozp, s.act = check_act_arg(act);
    s.c0 = c0, s.c1 = &c1, s.nc1 = 1, s.c2 = c2, s.c3 = c3;
    wrap_op(mf::op_create(device, s), op);
}
extern "C++" {
template <typename T>
static std::vector<int> widen(const T *zp, int nq) {
    MF_NEED(zp && nq > 0);
    return std::vector<int>(zp, zp + nq);
}
}
static void create_conv(int device, bool u8, bool dw, int H, int W, int Cin, int N, int KH, int KW,
                        const void *weights, const std::vector<int> &wzp, int izp, float output_scale,
                        int ozp, int act, int pad, int sh, int sw, int OH, int OW, const float *c0,
                        const float *c1, int nc1, mf_op **op) {
    MF_NEED(op && weights && c0 && c1);
    mf::OpSpec s;
    s.kind = dw ? MF_OP_DEPTHWISE_CONV_2D : MF_OP_CONV_2D, s.u8 = u8;
    s.H = H, s.W = W, s.C = Cin, s.N = N, s.KH = KH, s.KW = KW;
    s.weights = (const int8_t *)weights, s.wzp = wzp.data(), s.nq = (int)wzp.size(), s.izp = izp;
    s.oscale = output_scale, s.ozp = ozp, s.act = check_act_arg(act);
    s.pad = check_pad_arg(pad), s.sh = sh, s.sw = sw, s.OH = OH, s.OW = OW;
    s.c0 = c0, s.c1 = c1, s.nc1 = nc1;
    wrap_op(mf::op_create(device, s), op);
}
static void create_pool(int device, bool u8, int H, int W, int C, int FH, int FW, float output_scale,
                        int ozp, int act, int pad, int sh, int sw, int OH, int OW, float c0, float c1,
                        mf_op **op) {
    MF_NEED(op);
    mf::OpSpec s;
    s.kind = MF_OP_AVERAGE_POOL_2D, s.u8 = u8;
    s.H = H, s.W = W, s.C = C, s.N = C, s.KH = FH, s.KW = FW;
    s.oscale = output_scale, s.ozp = ozp, s.act = check_act_arg(act);
    s.pad = check_pad_arg(pad), s.sh = sh, s.sw = sw, s.OH = OH, s.OW = OW;
    s.pool_c0 = c0, s.pool_c1 = c1;
    wrap_op(mf::op_create(device, s), op);
}
static void create_softmax(int device, bool u8, int rows, int cols, float input_scale,
                           float output_scale, int ozp, mf_op **op) {
    MF_NEED(op);
    mf::OpSpec s;
    s.kind = MF_OP_SOFTMAX, s.u8 = u8;
    s.M = rows, s.N = cols, s.in_scale = input_scale;
    s.oscale = output_scale, s.ozp = ozp;
    wrap_op(mf::op_create(device, s), op);
}

int mf_fully_connected_create(int device, int M, int K, int N, const int8_t *weights,
                              int8_t weights_zero_point, float output_scale,
                              int8_t output_zero_point, int fused_activation, const float *c0,
                              float c1, const int32_t *c2, int32_t c3, mf_op **op) {
    MF_TRY({
        create_fc(device, false, M, K, N, weights, weights_zero_point, output_scale, output_zero_point,
                  fused_activation, c0, c1, c2, c3, op);
    })
}
int mf_fully_connected_create_u8(int device, int M, int K, int N, const uint8_t *weights,
                                 uint8_t weights_zero_point, float output_scale,
                                 uint8_t output_zero_point, int fused_activation, const float *c0,
                                 float c1, const int32_t *c2, int32_t c3, mf_op **op) {
    MF_TRY({
        create_fc(device, true, M, K, N, weights, weights_zero_point, output_scale, output_zero_point,
                  fused_activation, c0, c1, c2, c3, op);
    })
}

int mf_conv_2d_create(int device, int H, int W, int C, int N, int KH, int KW, const int8_t *filters,
                      const int8_t *filters_zero_point, int nq, int8_t input_zero_point,
                      float output_scale, int8_t output_zero_point, int fused_activation,
                      int view_padding, int stride_h, int stride_w, int OH, int OW,
                      const float *c0, const float *c1, int nc1, mf_op **op) {
    MF_TRY({
        create_conv(device, false, false, H, W, C, N, KH, KW, filters, widen(filters_zero_point, nq),
                    input_zero_point, output_scale, output_zero_point, fused_activation, view_padding,
                    stride_h, stride_w, OH, OW, c0, c1, nc1, op);
    })
}
int mf_conv_2d_create_u8(int device, int H, int W, int C, int N, int KH, int KW, const uint8_t *filters,
                         const uint8_t *filters_zero_point, int nq, uint8_t input_zero_point,
                         float output_scale, uint8_t output_zero_point, int fused_activation,
                         int view_padding, int stride_h, int stride_w, int OH, int OW,
                         const float *c0, const float *c1, int nc1, mf_op **op) {
    MF_TRY({
        create_conv(device, true, false, H, W, C, N, KH, KW, filters, widen(filters_zero_point, nq),
                    input_zero_point, output_scale, output_zero_point, fused_activation, view_padding,
                    stride_h, stride_w, OH, OW, c0, c1, nc1, op);
    })
}

int mf_depthwise_conv_2d_create(int device, int H, int W, int Cin, int KH, int KW, int C,
                                const int8_t *weights, const int8_t *weights_zero_point, int nq,
                                int8_t input_zero_point, float output_scale,
                                int8_t output_zero_point, int fused_activation, int view_padding,
                                int stride_h, int stride_w, int OH, int OW, const float *c0,
                                const float *c1, int nc1, mf_op **op) {
    MF_TRY({
        create_conv(device, false, true, H, W, Cin, C, KH, KW, weights, widen(weights_zero_point, nq),
                    input_zero_point, output_scale, output_zero_point, fused_activation, view_padding,
                    stride_h, stride_w, OH, OW, c0, c1, nc1, op);
    })
}
int mf_depthwise_conv_2d_create_u8(int device, int H, int W, int Cin, int KH, int KW, int C,
                                   const uint8_t *weights, const uint8_t *weights_zero_point, int nq,
                                   uint8_t input_zero_point, float output_scale,
                                   uint8_t output_zero_point, int fused_activation, int view_padding,
                                   int stride_h, int stride_w, int OH, int OW, const float *c0,
                                   const float *c1, int nc1, mf_op **op) {
    MF_TRY({
        create_conv(device, true, true, H, W, Cin, C, KH, KW, weights, widen(weights_zero_point, nq),
                    input_zero_point, output_scale, output_zero_point, fused_activation, view_padding,
                    stride_h, stride_w, OH, OW, c0, c1, nc1, op);
    })
}

int mf_average_pool_2d_create(int device, int H, int W, int C, int FH, int FW, float output_scale,
                              int8_t output_zero_point, int fused_activation, int view_padding,
                              int stride_h, int stride_w, int OH, int OW, float c0, float c1,
                              mf_op **op) {
    MF_TRY({
        create_pool(device, false, H, W, C, FH, FW, output_scale, output_zero_point, fused_activation,
                    view_padding, stride_h, stride_w, OH, OW, c0, c1, op);
    })
}
int mf_average_pool_2d_create_u8(int device, int H, int W, int C, int FH, int FW, float output_scale,
                                 uint8_t output_zero_point, int fused_activation, int view_padding,
                                 int stride_h, int stride_w, int OH, int OW, float c0, float c1,
                                 mf_op **op) {
    MF_TRY({
        create_pool(device, true, H, W, C, FH, FW, output_scale, output_zero_point, fused_activation,
                    view_padding, stride_h, stride_w, OH, OW, c0, c1, op);
    })
}

int mf_softmax_create(int device, int rows, int cols, float input_scale, float output_scale,
                      int8_t output_zero_point, mf_op **op) {
    MF_TRY({ create_softmax(device, false, rows, cols, input_scale, output_scale, output_zero_point, op); })
}
int mf_softmax_create_u8(int device, int rows, int cols, float input_scale, float output_scale,
                         uint8_t output_zero_point, mf_op **op) {
    MF_TRY({ create_softmax(device, true, rows, cols, input_scale, output_scale, output_zero_point, op); })
}

int mf_op_run(mf_op *op, const int8_t *d_input, size_t batch, int8_t *d_output, void *stream) {
    MF_TRY({
        MF_NEED(op && op->impl);
        mf::op_run_external(op->impl, d_input, batch, d_output, stream);
    })
}
size_t mf_op_input_elems(const mf_op *op) { return op && op->impl ? mf::op_in_elems(op->impl) : 0; }
size_t mf_op_output_elems(const mf_op *op) { return op && op->impl ? mf::op_out_elems(op->impl) : 0; }
const char *mf_op_kernel_name(const mf_op *op) { return op && op->impl ? mf::op_kernel_name(op->impl) : ""; }
int mf_op_set_generic(mf_op *op, int generic) {
    MF_TRY({
        MF_NEED(op && op->impl);
        mf::op_set_generic(op->impl, generic != 0);
    })
}
void mf_op_destroy(mf_op *op) {
    if (!op) return;
    mf::op_destroy(op->impl);
    delete op;
}

int mf_quantize(int device, const float *d_input, size_t n, float scale, int8_t zero_point,
                int8_t *d_output, void *stream) {
    MF_TRY({
        MF_NEED(n == 0 || (d_input && d_output));
        mf::dev_quantize(device, d_input, n, scale, zero_point, false, d_output, stream);
    })
}
int mf_quantize_u8(int device, const float *d_input, size_t n, float scale, uint8_t zero_point,
                   uint8_t *d_output, void *stream) {
    MF_TRY({
        MF_NEED(n == 0 || (d_input && d_output));
        // quantize into the internal domain, then back to real u8 bytes in place
        mf::dev_quantize(device, d_input, n, scale, zero_point, true, (int8_t *)d_output, stream);
        mf::dev_xor80(device, (const int8_t *)d_output, n, (int8_t *)d_output, stream);
    })
}
int mf_dequantize(int device, const int8_t *d_input, size_t n, float scale, int8_t zero_point,
                  float *d_output, void *stream) {
    MF_TRY({
        MF_NEED(n == 0 || (d_input && d_output));
        mf::dev_dequantize(device, d_input, n, scale, zero_point, false, d_output, stream);
    })
}
int mf_dequantize_u8(int device, const uint8_t *d_input, size_t n, float scale, uint8_t zero_point,
                     float *d_output, void *stream) {
    MF_TRY({
        MF_NEED(n == 0 || (d_input && d_output));
        mf::dev_dequantize_u8_raw(device, d_input, n, scale, zero_point, d_output, stream);
    })
}

// ---- 3. whole model -------------------------------------------------------------
int mf_model_create(const uint8_t *tflite, size_t len, mf_model **model) {
    MF_TRY({
        MF_NEED(model);
        if (!tflite) mf::fail(MF_ERR_INVALID_MODEL, "invalid model, please provide a valid TensorFlow Lite model");
        mf::ModelImpl *impl = mf::model_create(tflite, len);
        mf_model *h = new (std::nothrow) mf_model{impl};
        if (!h) {
            mf::model_destroy(impl);
            mf::fail(MF_ERR_OOM, "out of host memory");
        }
        *model = h;
    })
}
void mf_model_destroy(mf_model *model) {
    if (!model) return;
    mf::model_destroy(model->impl);
    delete model;
}

int mf_model_get_info(const mf_model *model, mf_model_info *info) {
    MF_TRY({
        MF_NEED(model && model->impl && info);
        const mf::ParsedModel &pm = mf::model_parsed(model->impl);
        std::memset(info, 0, sizeof(*info));
        info->input_rank = pm.in_rank, info->output_rank = pm.out_rank;
        for (int i = 0; i < 4; ++i) info->input_shape[i] = pm.in_shape[i], info->output_shape[i] = pm.out_shape[i];
        info->input_scale = pm.in_scale, info->output_scale = pm.out_scale;
        info->input_zero_point = pm.in_zp, info->output_zero_point = pm.out_zp;
        info->input_elems = pm.in_elems, info->output_elems = pm.out_elems;
        info->num_ops = (int)pm.ops.size();
        info->element_type = pm.u8 ? MF_ELEM_U8 : MF_ELEM_I8;
    })
}

int mf_model_get_op(const mf_model *model, int index, mf_op_desc *d) {
    MF_TRY({
        MF_NEED(model && model->impl && d);
        const mf::ParsedModel &pm = mf::model_parsed(model->impl);
        MF_NEED(index >= 0 && index < (int)pm.ops.size());
        const mf::ParsedOp &o = pm.ops[(size_t)index];
        std::memset(d, 0, sizeof(*d));
        d->kind = o.kind, d->in_rank = o.in_rank, d->out_rank = o.out_rank;
        for (int i = 0; i < 4; ++i) d->in_shape[i] = o.in_shape[i], d->out_shape[i] = o.out_shape[i];
        d->KH = o.KH, d->KW = o.KW, d->stride_h = o.sh, d->stride_w = o.sw;
        d->padding = o.pad, d->activation = o.act;
        d->n_c0 = (int)o.c0.size(), d->n_c1 = (int)o.c1.size();
        d->in_scale = o.in_scale, d->out_scale = o.out_scale;
        d->in_zero_point = o.in_zp, d->out_zero_point = o.out_zp;
        d->out_elems = o.out_elems;
        d->kernel = mf::model_op_kernel(model->impl, index);
    })
}

int mf_model_get_op_epilogue_mode(const mf_model *model, int index, int *mode) {
    MF_TRY({
        MF_NEED(model && model->impl && mode);
        *mode = mf::model_op_epilogue_mode(model->impl, index);
    })
}

int mf_model_get_op_constants(const mf_model *model, int index, float *c0, float *c1, int32_t *c2,
                              int32_t *c3) {
    MF_TRY({
        MF_NEED(model && model->impl);
        const mf::ParsedModel &pm = mf::model_parsed(model->impl);
        MF_NEED(index >= 0 && index < (int)pm.ops.size());
        const mf::ParsedOp &o = pm.ops[(size_t)index];
        if (c0 && !o.c0.empty()) std::memcpy(c0, o.c0.data(), o.c0.size() * sizeof(float));
        if (c1 && !o.c1.empty()) std::memcpy(c1, o.c1.data(), o.c1.size() * sizeof(float));
        if (c2 && !o.c2.empty()) std::memcpy(c2, o.c2.data(), o.c2.size() * sizeof(int32_t));
        if (c3) *c3 = o.c3;
    })
}

int mf_model_prepare(mf_model *model, int device, size_t max_batch) {
    MF_TRY({
        MF_NEED(model && model->impl);
        mf::model_prepare(model->impl, device, max_batch);
    })
}
int mf_model_set_stream(mf_model *model, void *stream) {
    MF_TRY({
        MF_NEED(model && model->impl);
        mf::model_set_stream(model->impl, stream);
    })
}
int mf_model_sync(mf_model *model) {
    MF_TRY({
        MF_NEED(model && model->impl);
        mf::model_sync(model->impl);
    })
}
int mf_model_set_generic(mf_model *model, int generic) {
    MF_TRY({
        MF_NEED(model && model->impl);
        mf::model_set_generic(model->impl, generic != 0);
    })
}

int mf_model_set_autotune(mf_model *model, int enabled) {
    MF_TRY({
        MF_NEED(model && model->impl);
        mf::model_set_autotune(model->impl, enabled != 0);
    })
}
int mf_model_set_fusion(mf_model *model, int enabled) {
    MF_TRY({
        MF_NEED(model && model->impl);
        mf::model_set_fusion(model->impl, enabled != 0);
    })
}

int mf_model_set_graph(mf_model *model, int enabled) {
    MF_TRY({
        MF_NEED(model && model->impl);
        mf::model_set_graph(model->impl, enabled != 0);
    })
}
unsigned long long mf_model_graph_launches(const mf_model *model) {
    return model && model->impl ? mf::model_graph_launches(model->impl) : 0;
}

int mf_model_predict(mf_model *model, const float *input, size_t batch, float *output, int mem) {
    MF_TRY({
        MF_NEED(model && model->impl && (batch == 0 || (input && output)));
        mf::model_run(model->impl, input, nullptr, batch, output, nullptr, mem, -1);
    })
}
int mf_model_predict_quantized(mf_model *model, const int8_t *input, size_t batch, float *output,
                               int mem) {
    MF_TRY({
        MF_NEED(model && model->impl && (batch == 0 || (input && output)));
        mf::model_run(model->impl, nullptr, input, batch, output, nullptr, mem, -1);
    })
}
int mf_model_run_quantized(mf_model *model, const int8_t *input, size_t batch, int8_t *output,
                           int mem) {
    MF_TRY({
        MF_NEED(model && model->impl && (batch == 0 || (input && output)));
        mf::model_run(model->impl, nullptr, input, batch, nullptr, output, mem, -1);
    })
}
// One call over several prepared handles (normally one per GPU of the node).  The batch is cut
// into contiguous shards -- handle i gets images [first_i, first_i + count_i), sizes differing by
// at most one -- and every shard runs on its own host thread through that handle's device and
// stream: no collective, no peer access, the model is replicated.  Host buffers only.
static void run_sharded(mf_model *const *models, int n, const float *in_f32, const int8_t *in_i8, size_t batch,
                        float *out_f32, int8_t *out_i8) {
    MF_NEED(models && n >= 1 && (batch == 0 || ((in_f32 || in_i8) && (out_f32 || out_i8))));
    for (int i = 0; i < n; ++i) MF_NEED(models[i] && models[i]->impl);
    const mf::ParsedModel &pm = mf::model_parsed(models[0]->impl);
    for (int i = 1; i < n; ++i) {
        const mf::ParsedModel &q = mf::model_parsed(models[i]->impl);
        if (q.in_elems != pm.in_elems || q.out_elems != pm.out_elems || q.u8 != pm.u8 || q.ops.size() != pm.ops.size())
            mf::fail(MF_ERR_INVALID_ARG, "the handles are not replicas of one model");
    }
    std::vector<mf::Error> errs((size_t)n, mf::Error{MF_OK, ""});
    std::vector<std::thread> th;
    const size_t base = batch / (size_t)n, rem = batch % (size_t)n;
    size_t first = 0;
    for (int i = 0; i < n; ++i) {
        const size_t count = base + ((size_t)i < rem ? 1 : 0);
        if (count) {
            th.emplace_back([=, &errs] {
                try {
                    mf::model_run(models[i]->impl, in_f32 ? in_f32 + first * pm.in_elems : nullptr,
                                  in_i8 ? in_i8 + first * pm.in_elems : nullptr, count,
                                  out_f32 ? out_f32 + first * pm.out_elems : nullptr,
                                  out_i8 ? out_i8 + first * pm.out_elems : nullptr, MF_MEM_HOST, -1);
                } catch (const mf::Error &e) {
                    errs[(size_t)i] = e;
                } catch (const std::exception &e) {
                    errs[(size_t)i] = mf::Error{MF_ERR_HIP, e.what()};
                }
            });
        }
        first += count;
    }
    for (std::thread &t : th) t.join();
    for (int i = 0; i < n; ++i)
        if (errs[(size_t)i].code != MF_OK)
            mf::fail(errs[(size_t)i].code, "shard " + std::to_string(i) + ": " + errs[(size_t)i].msg);
}
int mf_models_run_quantized(mf_model *const *models, int n_models, const int8_t *input, size_t batch,
                            int8_t *output) {
    MF_TRY({ run_sharded(models, n_models, nullptr, input, batch, nullptr, output); })
}
int mf_models_predict(mf_model *const *models, int n_models, const float *input, size_t batch, float *output) {
    MF_TRY({ run_sharded(models, n_models, input, nullptr, batch, output, nullptr); })
}
int mf_models_predict_quantized(mf_model *const *models, int n_models, const int8_t *input, size_t batch,
                                float *output) {
    MF_TRY({ run_sharded(models, n_models, nullptr, input, batch, output, nullptr); })
}

int mf_model_run_until(mf_model *model, const int8_t *input, size_t batch, int last_op,
                       int8_t *output, int mem) {
    MF_TRY({
        MF_NEED(model && model->impl && (batch == 0 || (input && output)) && last_op >= 0);
        mf::model_run(model->impl, nullptr, input, batch, nullptr, output, mem, last_op);
    })
}

// ---- 4. measurement helpers -------------------------------------------------------
int mf_synth_i8(int device, uint64_t seed, uint64_t first_byte, size_t n, int8_t *d_output,
                void *stream) {
    MF_TRY({
        MF_NEED(n == 0 || d_output);
        mf::dev_synth_i8(device, seed, first_byte, n, d_output, stream);
    })
}
int mf_verify_quant_div(int device, float scale, float reciprocal, int zero_point, int is_u8, uint64_t *mismatches) {
    MF_TRY({
        MF_NEED(mismatches);
        *mismatches = mf::dev_verify_quant_div(device, scale, reciprocal, zero_point, is_u8 != 0);
    })
}
int mf_selftest_rounding(int device, int mode, int is_u8, int lo, int hi, uint64_t *mismatches) {
    MF_TRY({
        MF_NEED(mismatches);
        *mismatches = mf::dev_selftest_epilogue(device, mode, is_u8 != 0, false, 0.0f, 0.0f, lo, hi);
    })
}
int mf_selftest_requant(int device, int mode, int is_u8, float A, float S, int lo, int hi, uint64_t *mismatches) {
    MF_TRY({
        MF_NEED(mismatches);
        *mismatches = mf::dev_selftest_epilogue(device, mode, is_u8 != 0, true, A, S, lo, hi);
    })
}
int mf_fma_epilogue_search(float A, float S, int is_u8, long long acc_min, long long acc_max, int allow_patch, float *S_out,
                           float *C_out, int *pivot_out, long long *patch_acc_out, int *patch_delta_out, int *found) {
    MF_TRY({
        MF_NEED(S_out && C_out && pivot_out && patch_acc_out && patch_delta_out && found);
        mf::FmaForm f;
        *found = mf::fma_form_search(A, S, is_u8 ? 0 : 128, is_u8 ? 0 : -128, is_u8 ? 255 : 127, acc_min, acc_max, f, nullptr, allow_patch != 0) ? 1 : 0;
        *S_out = f.S, *C_out = f.C, *pivot_out = f.d, *patch_acc_out = f.patch_acc, *patch_delta_out = f.patch_delta;
    })
}
static bool fma_args_ok(long long acc_min, long long acc_max, int pivot, long long patch_acc, int patch_delta) {
    if (!(acc_min <= acc_max && acc_min > -(1ll << 22) && acc_max < (1ll << 22))) return false;
    if (!(acc_min + pivot >= -(1ll << 22) && acc_max + pivot < (1ll << 22))) return false;
    if (patch_delta != 0 && patch_delta != 1 && patch_delta != -1) return false;
    return patch_delta == 0 || (patch_acc + patch_delta + pivot >= -(1ll << 22) && patch_acc + patch_delta + pivot < (1ll << 22));
}
int mf_fma_epilogue_check_host(float A, float S, int is_u8, long long acc_min, long long acc_max, float S_fma, float C_fma,
                               int pivot, long long patch_acc, int patch_delta, uint64_t *mismatches) {
    MF_TRY({
        MF_NEED(mismatches && fma_args_ok(acc_min, acc_max, pivot, patch_acc, patch_delta));
        mf::FmaForm f;
        f.S = S_fma, f.C = C_fma, f.d = pivot, f.patch_acc = patch_acc, f.patch_delta = patch_delta;
        *mismatches = mf::fma_form_mismatches(A, S, is_u8 ? 0 : 128, is_u8 ? 0 : -128, is_u8 ? 255 : 127, acc_min, acc_max, f);
    })
}
int mf_selftest_fma_epilogue(int device, float A, float S, int is_u8, long long acc_min, long long acc_max, float S_fma,
                             float C_fma, int pivot, long long patch_acc, int patch_delta, uint64_t *mismatches) {
    MF_TRY({
        MF_NEED(mismatches && fma_args_ok(acc_min, acc_max, pivot, patch_acc, patch_delta));
        *mismatches = mf::dev_selftest_fma_epilogue(device, A, S, is_u8 != 0, acc_min, acc_max, S_fma, C_fma, pivot, patch_acc, patch_delta);
    })
}
int mf_selftest_cvt_pk(int device, uint64_t *mismatches) {
    MF_TRY({
        MF_NEED(mismatches);
        *mismatches = mf::dev_selftest_cvt_pk(device);
    })
}
int mf_checksum_i8(int device, const int8_t *d_input, size_t n, uint64_t *checksum, void *stream) {
    MF_TRY({
        MF_NEED(checksum && (n == 0 || d_input));
        *checksum = mf::dev_checksum_i8(device, d_input, n, stream);
    })
}
int mf_model_time_device(mf_model *model, const int8_t *d_input, size_t batch, int8_t *d_output,
                         int warmup, int iters, float *avg_ms, float *per_op_ms) {
    MF_TRY({
        MF_NEED(model && model->impl && d_input && d_output);
        mf::model_time_device(model->impl, d_input, batch, d_output, warmup, iters, avg_ms, per_op_ms);
    })
}

} // extern "C"
