// The Rust shim's call sequence, statement by statement, in C++ (rust/ cannot be compiled in this image):
//
//   #[model("models/x.tflite")] struct M;            rust/microflow-amd-macros/src/lib.rs
//     M::predict(Buffer)            -> ModelSet::new -> Model::try_new (mf_model_create, mf_model_get_info,
//                                      mf_model_prepare(.., 1)) per device; replica 0: flatten -> mf_model_predict
//                                      (MF_MEM_HOST) -> unflatten
//     M::predict_quantized(Buffer)  -> the same through mf_model_predict_quantized
//     M::predict_batch(&[Buffer])   -> flatten each, mf_models_predict over ALL replicas, chunk, unflatten
//     M::predict_quantized_batch(&[Buffer<i8|u8>]) -> the same through mf_models_predict_quantized
//
// The buffers on the Rust side are nalgebra SMatrix (column-major) / [SMatrix<[T; CH], R, C>; B]
// (memory order [b][col][row][ch], src/buffer.rs:5-16); `Buffer2D` / `Buffer4D` below have exactly those
// memory orders, and flatten_* / unflatten_* are rust/microflow-amd/src/lib.rs `mod layout`.
// Checked against the reference's whole-model vectors (tests/{sine,speech,person_detect}.rs).
// Built and run by tests/test_cpp_mirror.py; exits non-zero on the first mismatch.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "microflow_amd.h"

#define CHECK(call)                                                                       \
    do {                                                                                  \
        const int st_ = (call);                                                           \
        if (st_ != MF_OK) {                                                               \
            std::fprintf(stderr, "%s:%d %s -> %d: %s\n", __FILE__, __LINE__, #call, st_, mf_last_error()); \
            return 1;                                                                     \
        }                                                                                 \
    } while (0)

// column-major R x C matrix (nalgebra SMatrix)
template <typename T> struct Buffer2D {
    int R, C;
    std::vector<T> data; // data[j * R + i] = element (i, j)
    Buffer2D(int r, int c, T fill) : R(r), C(c), data((size_t)r * c, fill) {}
    T &at(int i, int j) { return data[(size_t)j * R + i]; }
    const T &at(int i, int j) const { return data[(size_t)j * R + i]; }
};
// [B] matrices whose elements are [T; CH]
template <typename T> struct Buffer4D {
    int B, R, C, CH;
    std::vector<T> data; // data[((b * C + j) * R + i) * CH + c]
    Buffer4D(int b, int r, int c, int ch, T fill) : B(b), R(r), C(c), CH(ch), data((size_t)b * r * c * ch, fill) {}
    T &at(int b, int i, int j, int c) { return data[(((size_t)b * C + j) * R + i) * CH + c]; }
    const T &at(int b, int i, int j, int c) const { return data[(((size_t)b * C + j) * R + i) * CH + c]; }
};
template <typename T> std::vector<T> flatten_2d(const Buffer2D<T> &b) { // layout::flatten_2d: row-major out
    std::vector<T> v;
    for (int i = 0; i < b.R; ++i)
        for (int j = 0; j < b.C; ++j) v.push_back(b.at(i, j));
    return v;
}
template <typename T> std::vector<T> flatten_4d(const Buffer4D<T> &b) { // layout::flatten_4d: NHWC out
    std::vector<T> v;
    for (int n = 0; n < b.B; ++n)
        for (int i = 0; i < b.R; ++i)
            for (int j = 0; j < b.C; ++j)
                for (int c = 0; c < b.CH; ++c) v.push_back(b.at(n, i, j, c));
    return v;
}
static Buffer2D<float> unflatten_2d(const float *v, int R, int C) { // layout::unflatten_2d
    Buffer2D<float> b(R, C, 0.0f);
    for (int i = 0; i < R; ++i)
        for (int j = 0; j < C; ++j) b.at(i, j) = v[(size_t)i * C + j];
    return b;
}

static std::vector<uint8_t> read_file(const std::string &p) {
    std::vector<uint8_t> v;
    FILE *f = std::fopen(p.c_str(), "rb");
    if (!f) return v;
    std::fseek(f, 0, SEEK_END);
    v.resize((size_t)std::ftell(f));
    std::fseek(f, 0, SEEK_SET);
    if (std::fread(v.data(), 1, v.size(), f) != v.size()) v.clear();
    std::fclose(f);
    return v;
}

struct ModelSet { // rust: ModelSet { replicas: Vec<Model> }
    std::vector<mf_model *> replicas;
    mf_model_info info{};
    ~ModelSet() {
        for (mf_model *m : replicas) mf_model_destroy(m);
    }
};

static int run_model(const std::string &dir, const char *file, const std::vector<float> &want) {
    const std::vector<uint8_t> bytes = read_file(dir + "/" + file); // include_bytes!(<absolute path>)
    if (bytes.empty()) {
        std::fprintf(stderr, "cannot read %s/%s\n", dir.c_str(), file);
        return 1;
    }
    // ---- ModelSet::new: one replica per device that prepares ----
    ModelSet set;
    int ndev = mf_device_count();
    if (ndev < 1) ndev = 1;
    for (int d = 0; d < ndev; ++d) {
        mf_model *raw = nullptr;
        CHECK(mf_model_create(bytes.data(), bytes.size(), &raw)); // Model::try_new
        mf_model_info info{};
        CHECK(mf_model_get_info(raw, &info));
        if (mf_model_prepare(raw, d, 1) != MF_OK) { // a device that cannot be prepared is left out
            mf_model_destroy(raw);
            continue;
        }
        set.replicas.push_back(raw);
        set.info = info;
    }
    if (set.replicas.empty()) {
        std::fprintf(stderr, "no usable GPU: %s\n", mf_last_error());
        return 1;
    }
    const mf_model_info &info = set.info;
    if (info.element_type != MF_ELEM_I8) return 1;

    // ---- M::predict(input filled with 0.5): flatten -> mf_model_predict on replica 0 -> unflatten ----
    std::vector<float> flat;
    if (info.input_rank == 2) {
        flat = flatten_2d(Buffer2D<float>(info.input_shape[0], info.input_shape[1], 0.5f));
    } else {
        flat = flatten_4d(Buffer4D<float>(info.input_shape[0], info.input_shape[1], info.input_shape[2], info.input_shape[3], 0.5f));
    }
    if (flat.size() != info.input_elems) return 1;
    std::vector<float> out(info.output_elems);
    CHECK(mf_model_predict(set.replicas[0], flat.data(), 1, out.data(), MF_MEM_HOST));
    const Buffer2D<float> got = unflatten_2d(out.data(), info.output_shape[0], info.output_shape[1]);
    for (size_t k = 0; k < want.size(); ++k)
        if (got.at(0, (int)k) != want[k]) { // the reference's assert_eq on f32 values (tests/*.rs)
            std::fprintf(stderr, "%s predict: output %zu = %.9g, want %.9g\n", file, k, got.at(0, (int)k), want[k]);
            return 1;
        }

    // ---- M::predict_quantized: the caller quantizes (src/quantize.rs:16-18), same result ----
    {
        const float q = std::round(0.5f / info.input_scale + (float)info.input_zero_point);
        const int8_t qi = (int8_t)(q < -128.f ? -128.f : (q > 127.f ? 127.f : q));
        std::vector<int8_t> xq(info.input_elems, qi);
        std::vector<float> oq(info.output_elems);
        CHECK(mf_model_predict_quantized(set.replicas[0], xq.data(), 1, oq.data(), MF_MEM_HOST));
        for (size_t k = 0; k < want.size(); ++k)
            if (oq[k] != want[k]) {
                std::fprintf(stderr, "%s predict_quantized: output %zu = %.9g, want %.9g\n", file, k, oq[k], want[k]);
                return 1;
            }
    }

    // ---- M::predict_batch(&[a, b, c]): mf_models_predict over every replica, chunked ----
    {
        const size_t B = 3;
        std::vector<float> v;
        for (size_t b = 0; b < B; ++b) v.insert(v.end(), flat.begin(), flat.end());
        std::vector<float> ob(B * info.output_elems);
        CHECK(mf_models_predict(set.replicas.data(), (int)set.replicas.size(), v.data(), B, ob.data()));
        const size_t n = ob.size() / B;
        for (size_t b = 0; b < B; ++b)
            for (size_t k = 0; k < want.size(); ++k)
                if (ob[b * n + k] != want[k]) {
                    std::fprintf(stderr, "%s predict_batch[%zu]: output %zu = %.9g, want %.9g\n", file, b, k, ob[b * n + k], want[k]);
                    return 1;
                }
        // predict_batch(&[]) returns before touching the library; the ABI accepts an empty batch too
        CHECK(mf_models_predict(set.replicas.data(), (int)set.replicas.size(), nullptr, 0, nullptr));
    }

    // ---- M::predict_quantized_batch(&[a, b, c, d, e]): ModelSet::predict_quantized -> mf_models_predict_quantized ----
    {
        const size_t B = 5;
        const float q = std::round(0.5f / info.input_scale + (float)info.input_zero_point);
        const int8_t qi = (int8_t)(q < -128.f ? -128.f : (q > 127.f ? 127.f : q));
        std::vector<int8_t> v(B * info.input_elems, qi); // B flattened Buffer<i8> inputs back to back
        std::vector<float> ob(B * info.output_elems);
        CHECK(mf_models_predict_quantized(set.replicas.data(), (int)set.replicas.size(), v.data(), B, ob.data()));
        const size_t n = ob.size() / B;
        for (size_t b = 0; b < B; ++b)
            for (size_t k = 0; k < want.size(); ++k)
                if (ob[b * n + k] != want[k]) {
                    std::fprintf(stderr, "%s predict_quantized_batch[%zu]: output %zu = %.9g, want %.9g\n", file, b, k, ob[b * n + k], want[k]);
                    return 1;
                }
        CHECK(mf_models_predict_quantized(set.replicas.data(), (int)set.replicas.size(), nullptr, 0, nullptr));
    }
    return 0;
}

int main(int argc, char **argv) {
    const std::string dir = argc > 1 ? argv[1] : "models";
    if (run_model(dir, "sine.tflite", {0.41348344f})) return 1;                                            // tests/sine.rs:7-12
    if (run_model(dir, "speech.tflite", {0.15625f, 0.2734375f, 0.2734375f, 0.296875f})) return 1;          // tests/speech.rs:8-13
    if (run_model(dir, "person_detect.tflite", {0.8046875f, 0.1953125f})) return 1;                        // tests/person_detect.rs:8-13
    std::printf("shim flow ok\n");
    return 0;
}
