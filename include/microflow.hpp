// microflow.hpp -- C++17 mirror of MicroFlow's user surface over the C ABI
// (include/microflow_amd.h).  Header only; link with -lmicroflow_amd.
//
// The reference is Rust; this is the same surface for a C++ host, with the reference's
// names, argument order and meaning so that tests read like the reference's own
// (src/ops/*.rs `mod tests`, tests/*.rs):
//
//   microflow::Tensor2D / Tensor4D                      src/tensor.rs:27-47
//   microflow::FusedActivation, TensorViewPadding       src/activation.rs:6-13, src/tensor.rs:9-15
//   microflow::ops::fully_connected / conv_2d / depthwise_conv_2d / average_pool_2d /
//                   softmax / reshape                    src/ops/*.rs
//   microflow::Model  (what #[model("x.tflite")] generates: predict, predict_quantized)
//                                                        microflow-macros/src/lib.rs:185-203
//
// Buffers are row-major / NHWC std::vector<int8_t> on the HOST; each call stages through HBM
// (hipMalloc/hipMemcpy from libamdhip64, declared below to keep this header free of HIP
// includes).  Errors: the reference has compile-time errors only; here every failure throws
// microflow::Error carrying the mf_status and the library's message.
#pragma once
#include <array>
#include <cstdint>
#include <fstream>
#include <iterator>
#include <stdexcept>
#include <string>
#include <vector>

#include "microflow_amd.h"

extern "C" {
int hipMalloc(void **ptr, size_t size);
int hipFree(void *ptr);
int hipMemcpy(void *dst, const void *src, size_t size, int kind);
int hipDeviceSynchronize(void);
}

namespace microflow {

struct Error : std::runtime_error {
    int status;
    Error(int s, const std::string &m) : std::runtime_error(m), status(s) {}
};
inline void check(int status) {
    if (status != MF_OK) throw Error(status, mf_last_error());
}

enum class FusedActivation { None = MF_ACT_NONE, Relu = MF_ACT_RELU, Relu6 = MF_ACT_RELU6 };
enum class TensorViewPadding { Same = MF_PAD_SAME, Valid = MF_PAD_VALID };

// src/tensor.rs:27-31 -- [rows][cols] row-major
struct Tensor2D {
    std::vector<int8_t> buffer;
    int rows = 0, cols = 0;
    std::vector<float> scale{1.0f};
    std::vector<int8_t> zero_point{0};
};
// src/tensor.rs:37-47 -- [batches][rows][cols][chans] (NHWC)
struct Tensor4D {
    std::vector<int8_t> buffer;
    int batches = 1, rows = 0, cols = 0, chans = 0;
    std::vector<float> scale{1.0f};
    std::vector<int8_t> zero_point{0};
};

namespace detail {
struct DeviceBuffer {
    void *p = nullptr;
    explicit DeviceBuffer(size_t n) {
        if (hipMalloc(&p, n ? n : 1) != 0) throw Error(MF_ERR_OOM, "hipMalloc failed");
    }
    ~DeviceBuffer() { hipFree(p); }
    DeviceBuffer(const DeviceBuffer &) = delete;
    DeviceBuffer &operator=(const DeviceBuffer &) = delete;
};
// run a prepared operator over host buffers (H2D, launch, D2H), then destroy it
inline std::vector<int8_t> run(mf_op *op, const std::vector<int8_t> &in, size_t batch) {
    struct Guard {
        mf_op *o;
        ~Guard() { mf_op_destroy(o); }
    } g{op};
    const size_t in_n = mf_op_input_elems(op) * batch, out_n = mf_op_output_elems(op) * batch;
    if (in.size() != in_n) throw Error(MF_ERR_INVALID_ARG, "input size does not match the operator");
    DeviceBuffer din(in_n), dout(out_n);
    if (hipMemcpy(din.p, in.data(), in_n, 1 /*H2D*/) != 0) throw Error(MF_ERR_HIP, "hipMemcpy H2D failed");
    check(mf_op_run(op, (const int8_t *)din.p, batch, (int8_t *)dout.p, nullptr));
    std::vector<int8_t> out(out_n);
    if (hipMemcpy(out.data(), dout.p, out_n, 2 /*D2H*/) != 0) throw Error(MF_ERR_HIP, "hipMemcpy D2H failed");
    return out;
}
} // namespace detail

namespace ops {

struct FullyConnectedOptions { // src/ops/fully_connected.rs:9-11
    FusedActivation fused_activation = FusedActivation::None;
};
struct Conv2DOptions { // src/ops/conv_2d.rs:11-15
    FusedActivation fused_activation = FusedActivation::None;
    TensorViewPadding view_padding = TensorViewPadding::Same;
    std::array<int, 2> strides{1, 1};
};
using DepthwiseConv2DOptions = Conv2DOptions; // src/ops/depthwise_conv_2d.rs:11-15
using AveragePool2DOptions = Conv2DOptions;   // src/ops/average_pool_2d.rs:12-16

struct FullyConnectedConstants { // (Buffer2D<f32,N,1>, f32, Buffer2D<i32,1,N>, i32)
    std::vector<float> c0;
    float c1 = 0;
    std::vector<int32_t> c2;
    int32_t c3 = 0;
};
struct ConvConstants { // (Buffer2D<f32,N,1>, Buffer2D<f32,Q,1>)
    std::vector<float> c0, c1;
};

// src/ops/fully_connected.rs:24-41.  `weights` is K x N like the reference's
// Tensor2D<T, INPUT_COLS, WEIGHTS_COLS> (rows = input index).
inline Tensor2D fully_connected(const Tensor2D &input, const Tensor2D &weights, std::array<float, 1> output_scale,
                                std::array<int8_t, 1> output_zero_point, FullyConnectedOptions options,
                                const FullyConnectedConstants &constants, int device = 0) {
    const int K = weights.rows, N = weights.cols;
    std::vector<int8_t> w_nk((size_t)N * K);
    for (int k = 0; k < K; ++k)
        for (int j = 0; j < N; ++j) w_nk[(size_t)j * K + k] = weights.buffer[(size_t)k * N + j];
    mf_op *op = nullptr;
    check(mf_fully_connected_create(device, input.rows, K, N, w_nk.data(), weights.zero_point[0], output_scale[0],
                                    output_zero_point[0], (int)options.fused_activation, constants.c0.data(),
                                    constants.c1, constants.c2.data(), constants.c3, &op));
    Tensor2D out;
    out.buffer = detail::run(op, input.buffer, 1);
    out.rows = input.rows, out.cols = N;
    out.scale = {output_scale[0]}, out.zero_point = {output_zero_point[0]};
    return out;
}

// src/ops/conv_2d.rs:28-49.  filters: batches = N, rows = KH, cols = KW, chans = C.
inline Tensor4D conv_2d(const Tensor4D &input, const Tensor4D &filters, std::array<float, 1> output_scale,
                        std::array<int8_t, 1> output_zero_point, Conv2DOptions options, const ConvConstants &constants,
                        std::array<int, 2> output_shape, int device = 0) {
    mf_op *op = nullptr;
    check(mf_conv_2d_create(device, input.rows, input.cols, input.chans, filters.batches, filters.rows, filters.cols,
                            filters.buffer.data(), filters.zero_point.data(), (int)filters.zero_point.size(),
                            input.zero_point[0], output_scale[0], output_zero_point[0], (int)options.fused_activation,
                            (int)options.view_padding, options.strides[0], options.strides[1], output_shape[0],
                            output_shape[1], constants.c0.data(), constants.c1.data(), (int)constants.c1.size(), &op));
    Tensor4D out;
    out.buffer = detail::run(op, input.buffer, (size_t)input.batches);
    out.batches = input.batches, out.rows = output_shape[0], out.cols = output_shape[1], out.chans = filters.batches;
    out.scale = {output_scale[0]}, out.zero_point = {output_zero_point[0]};
    return out;
}

// src/ops/depthwise_conv_2d.rs:28-49.  weights: batches = 1, rows = KH, cols = KW, chans = C.
inline Tensor4D depthwise_conv_2d(const Tensor4D &input, const Tensor4D &weights, std::array<float, 1> output_scale,
                                  std::array<int8_t, 1> output_zero_point, DepthwiseConv2DOptions options,
                                  const ConvConstants &constants, std::array<int, 2> output_shape, int device = 0) {
    mf_op *op = nullptr;
    check(mf_depthwise_conv_2d_create(device, input.rows, input.cols, input.chans, weights.rows, weights.cols,
                                      weights.chans, weights.buffer.data(), weights.zero_point.data(),
                                      (int)weights.zero_point.size(), input.zero_point[0], output_scale[0],
                                      output_zero_point[0], (int)options.fused_activation, (int)options.view_padding,
                                      options.strides[0], options.strides[1], output_shape[0], output_shape[1],
                                      constants.c0.data(), constants.c1.data(), (int)constants.c1.size(), &op));
    Tensor4D out;
    out.buffer = detail::run(op, input.buffer, (size_t)input.batches);
    out.batches = input.batches, out.rows = output_shape[0], out.cols = output_shape[1], out.chans = weights.chans;
    out.scale = {output_scale[0]}, out.zero_point = {output_zero_point[0]};
    return out;
}

// src/ops/average_pool_2d.rs:29-45.  filter_shape = (rows, cols), constants = (c0, c1).
inline Tensor4D average_pool_2d(const Tensor4D &input, std::array<int, 2> filter_shape, std::array<float, 1> output_scale,
                                std::array<int8_t, 1> output_zero_point, AveragePool2DOptions options,
                                std::array<float, 2> constants, std::array<int, 2> output_shape, int device = 0) {
    mf_op *op = nullptr;
    check(mf_average_pool_2d_create(device, input.rows, input.cols, input.chans, filter_shape[0], filter_shape[1],
                                    output_scale[0], output_zero_point[0], (int)options.fused_activation,
                                    (int)options.view_padding, options.strides[0], options.strides[1], output_shape[0],
                                    output_shape[1], constants[0], constants[1], &op));
    Tensor4D out;
    out.buffer = detail::run(op, input.buffer, (size_t)input.batches);
    out.batches = input.batches, out.rows = output_shape[0], out.cols = output_shape[1], out.chans = input.chans;
    out.scale = {output_scale[0]}, out.zero_point = {output_zero_point[0]};
    return out;
}

// src/ops/softmax.rs:15-19
inline Tensor2D softmax(const Tensor2D &input, std::array<float, 1> output_scale,
                        std::array<int8_t, 1> output_zero_point, int device = 0) {
    mf_op *op = nullptr;
    check(mf_softmax_create(device, input.rows, input.cols, input.scale[0], output_scale[0], output_zero_point[0], &op));
    Tensor2D out;
    out.buffer = detail::run(op, input.buffer, 1);
    out.rows = input.rows, out.cols = input.cols;
    out.scale = {output_scale[0]}, out.zero_point = {output_zero_point[0]};
    return out;
}

// src/ops/reshape.rs:3-8 with the From impls of src/tensor.rs:103-141 (logical NHWC order kept)
inline Tensor4D reshape(const Tensor2D &input, std::array<int, 4> shape) {
    Tensor4D out;
    out.buffer = input.buffer;
    out.batches = shape[0], out.rows = shape[1], out.cols = shape[2], out.chans = shape[3];
    out.scale = input.scale, out.zero_point = input.zero_point;
    return out;
}
inline Tensor2D reshape(const Tensor4D &input, std::array<int, 2> shape) {
    Tensor2D out;
    out.buffer = input.buffer;
    out.rows = shape[0], out.cols = shape[1];
    out.scale = input.scale, out.zero_point = input.zero_point;
    return out;
}

} // namespace ops

// What `#[model("path.tflite")] struct M;` generates (microflow-macros/src/lib.rs:185-203).
class Model {
  public:
    explicit Model(const std::string &path, int device = 0) {
        std::ifstream f(path, std::ios::binary);
        if (!f) throw Error(MF_ERR_INVALID_ARG, "couldn't find '" + path + "', please provide a valid path");
        std::vector<uint8_t> bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        check(mf_model_create(bytes.data(), bytes.size(), &m_));
        check(mf_model_get_info(m_, &info_));
        check(mf_model_prepare(m_, device, 1));
    }
    ~Model() { mf_model_destroy(m_); }
    Model(const Model &) = delete;
    Model &operator=(const Model &) = delete;
    const mf_model_info &info() const { return info_; }
    // predict: f32 in -> f32 out; `input` holds batch x input_elems values
    std::vector<float> predict(const std::vector<float> &input) {
        const size_t batch = input.size() / info_.input_elems;
        std::vector<float> out(batch * info_.output_elems);
        check(mf_model_predict(m_, input.data(), batch, out.data(), MF_MEM_HOST));
        return out;
    }
    std::vector<float> predict_quantized(const std::vector<int8_t> &input) {
        const size_t batch = input.size() / info_.input_elems;
        std::vector<float> out(batch * info_.output_elems);
        check(mf_model_predict_quantized(m_, input.data(), batch, out.data(), MF_MEM_HOST));
        return out;
    }

  private:
    mf_model *m_ = nullptr;
    mf_model_info info_{};
};

} // namespace microflow
