// The reference's operator unit tests and whole-model tests, written against the C++ mirror
// (include/microflow.hpp) the way the Rust tests are written against microflow::ops.
// Values are the reference's test constants (src/ops/*.rs `mod tests`, tests/*.rs).
// Built by tests/test_cpp_mirror.py with g++; exits non-zero on the first mismatch.
#include <cstdio>
#include <cstdlib>

#include "microflow.hpp"

using namespace microflow;

#define EXPECT_EQ_VEC(got, ...)                                                        \
    do {                                                                               \
        const std::vector<int8_t> want{__VA_ARGS__};                                   \
        if ((got) != want) {                                                           \
            std::fprintf(stderr, "%s:%d mismatch in %s\n", __FILE__, __LINE__, #got);  \
            return 1;                                                                  \
        }                                                                              \
    } while (0)

static Tensor4D input_2x3x2() { // the INPUT of the conv/depthwise/pool tests
    Tensor4D t;
    t.buffer = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12};
    t.rows = 2, t.cols = 3, t.chans = 2;
    t.scale = {0.13f}, t.zero_point = {14};
    return t;
}

int main(int argc, char **argv) {
    // fully_connected_layer  (src/ops/fully_connected.rs:90-147)
    {
        Tensor2D input{{1, 2, 3, 4, 5, 6}, 2, 3, {0.7f}, {8}};
        Tensor2D weights{{9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20}, 3, 4, {0.21f}, {22}};
        ops::FullyConnectedConstants c{{-4.6551723f, -3.724138f, -2.7931035f, -1.862069f}, 0.50689656f,
                                       {312, 336, 360, 384}, 528};
        Tensor2D out = ops::fully_connected(input, weights, {0.29f}, {30}, {FusedActivation::Relu}, c);
        EXPECT_EQ_VEC(out.buffer, 112, 103, 95, 87, 70, 67, 63, 60);
    }
    // conv_2d_layer  (src/ops/conv_2d.rs:118-181)
    {
        Tensor4D filters;
        filters.buffer = {15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 36, 37, 38};
        filters.batches = 2, filters.rows = 2, filters.cols = 3, filters.chans = 2;
        filters.scale = {0.39f, 0.40f}, filters.zero_point = {41, 42};
        ops::ConvConstants c{{-3.6734694f, -3.755102f}, {0.10346939f, 0.10612245f}};
        Tensor4D out = ops::conv_2d(input_2x3x2(), filters, {0.49f}, {50},
                                    {FusedActivation::None, TensorViewPadding::Same, {1, 1}}, c, {2, 3});
        EXPECT_EQ_VEC(out.buffer, 127, 116, 127, 127, 127, 113, 98, 74, 114, 84, 82, 67);
    }
    // depthwise_conv_2d_layer  (src/ops/depthwise_conv_2d.rs:115-172)
    {
        Tensor4D weights;
        weights.buffer = {15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26};
        weights.rows = 2, weights.cols = 3, weights.chans = 2;
        weights.scale = {0.27f, 0.28f}, weights.zero_point = {29, 30};
        ops::ConvConstants c{{-3.5675676f, -3.6756757f}, {0.09486486f, 0.098378378f}};
        Tensor4D out = ops::depthwise_conv_2d(input_2x3x2(), weights, {0.37f}, {38},
                                              {FusedActivation::None, TensorViewPadding::Same, {1, 1}}, c, {2, 3});
        EXPECT_EQ_VEC(out.buffer, 66, 63, 82, 78, 65, 62, 47, 45, 52, 49, 44, 42);
    }
    // average_pool_2d_layer  (src/ops/average_pool_2d.rs:74-113)
    {
        Tensor4D out = ops::average_pool_2d(input_2x3x2(), {2, 3}, {0.15f}, {16},
                                            {FusedActivation::None, TensorViewPadding::Same, {1, 1}},
                                            {0.8666667f, 3.8666666f}, {2, 3});
        EXPECT_EQ_VEC(out.buffer, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13, 13);
    }
    // softmax_layer  (src/ops/softmax.rs:29-56)
    {
        Tensor2D input{{1, 2, 3, 4, 5, 6}, 2, 3, {0.7f}, {8}};
        Tensor2D out = ops::softmax(input, {0.9f}, {10});
        EXPECT_EQ_VEC(out.buffer, 10, 10, 10, 10, 10, 11);
    }
    // reshape_layer  (src/ops/reshape.rs:10-34)
    {
        Tensor2D input{{1, 2, 3, 4, 5, 6}, 2, 3, {0.7f}, {8}};
        Tensor4D out = ops::reshape(input, {2, 1, 3, 1});
        EXPECT_EQ_VEC(out.buffer, 1, 2, 3, 4, 5, 6);
        if (out.batches != 2 || out.rows != 1 || out.cols != 3 || out.chans != 1) return 1;
    }
    // whole models  (tests/sine.rs, tests/speech.rs, tests/person_detect.rs): constant 0.5 input
    if (argc > 1) {
        const std::string dir = argv[1];
        struct Case {
            const char *file;
            std::vector<float> want;
        } cases[] = {{"sine.tflite", {0.41348344f}},
                     {"speech.tflite", {0.15625f, 0.2734375f, 0.2734375f, 0.296875f}},
                     {"person_detect.tflite", {0.8046875f, 0.1953125f}}};
        for (const Case &c : cases) {
            Model m(dir + "/" + c.file);
            const std::vector<float> out = m.predict(std::vector<float>(m.info().input_elems, 0.5f));
            if (out != c.want) {
                std::fprintf(stderr, "model %s: output mismatch\n", c.file);
                return 1;
            }
        }
        try {
            Model missing(dir + "/nope.tflite");
            return 1;
        } catch (const Error &e) {
            if (std::string(e.what()).find("couldn't find") == std::string::npos) return 1;
        }
    }
    std::puts("reference KATs through microflow.hpp: ok");
    return 0;
}
