// hostmath.cpp -- host-side f32 numerics of the product: constant preparation
// ("preprocess") and the scalar helpers the model builder needs.
//
// Bit-exactness contract: every f32 operation is
// individually rounded and evaluated in the reference's order.  This file must be
// compiled with -ffp-contract=off (build.py does); the volatile temporaries below
// additionally keep each intermediate a real f32 value.
#include <cmath>
#include <cstring>

#include "mf_internal.hpp"

namespace mf {

namespace {
inline uint32_t bits(float f) {
    uint32_t u;
    std::memcpy(&u, &f, sizeof u);
    return u;
}
inline float from_bits(uint32_t u) {
    float f;
    std::memcpy(&f, &u, sizeof f);
    return f;
}
inline int32_t wrap_sub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
inline int32_t wrap_mul(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }
inline int32_t wrap_add(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
} // namespace

// libm::roundf -- half away from zero.  Same formulation as the device epilogue:
// trunc(x + copysign(pred(0.5), x)), exact for every finite x (the only input for
// which x + 0.5 would round up wrongly is pred(0.5), which pred(0.5) + pred(0.5) < 1
// handles).
float h_roundf(float x) {
    if (!(std::fabs(x) < 8388608.0f)) return x;
    volatile float s = x + std::copysign(from_bits(0x3effffffu), x);
    return std::trunc(s);
}

// Rust `as i8` on f32: NaN -> 0, saturating, truncating.
int8_t h_sat_i8(float x) {
    if (std::isnan(x)) return 0;
    if (x >= 127.0f) return 127;
    if (x <= -128.0f) return -128;
    return (int8_t)(int32_t)x;
}

// Rust `as u8` on f32
static int h_sat_u8(float x) {
    if (std::isnan(x)) return 0;
    if (x >= 255.0f) return 255;
    if (x <= 0.0f) return 0;
    return (int)(int32_t)x;
}

// src/quantize.rs:16-18
int8_t h_quantize(float x, float scale, int8_t zp) {
    volatile float q = x / scale;
    volatile float s = q + (float)zp;
    return h_sat_i8(h_roundf(s));
}
// the same for either element type; zp and the result are plain ints in T's range
int h_quantize_t(float x, float scale, int zp, bool u8) {
    volatile float q = x / scale;
    volatile float s = q + (float)zp;
    return u8 ? h_sat_u8(h_roundf(s)) : (int)h_sat_i8(h_roundf(s));
}

// expf of the `libm` 0.2 crate (musl expf.c lineage): used on the HOST to build the
// 256-entry exp table each softmax op needs (softmax inputs are int8, so only 256
// arguments exist per op; src/ops/softmax.rs:20-21).  f32 throughout.
// the 256 values expf(f32(q) * input_scale) a Softmax over int8 (or u8) inputs can meet (src/ops/softmax.rs:20-21):
// entry = stored byte + 128 for i8 (q = entry - 128), the value itself for u8
void h_softmax_table(float in_scale, bool u8, float *table256) {
    for (int i = 0; i < 256; ++i) {
        volatile float e = (float)(u8 ? i : i - 128) * in_scale;
        table256[i] = h_expf(e);
    }
}

float h_expf(float x) {
    const float LN2_HI = from_bits(0x3f317200u), LN2_LO = from_bits(0x35bfbe8eu);
    const float INV_LN2 = from_bits(0x3fb8aa3bu);
    const float P1 = from_bits(0x3e2aaa8fu), P2 = from_bits(0xbb355215u);
    uint32_t ux = bits(x);
    const bool neg = (ux >> 31) != 0;
    const uint32_t ax = ux & 0x7fffffffu;
    if (ax >= 0x42aeac50u) { // |x| >= 87.33655f or NaN
        if (ax > 0x7f800000u) return x;
        if (!neg && ax >= 0x42b17218u) return x * from_bits(0x7f000000u); // overflow -> inf
        if (neg && ax >= 0x42cff1b5u) return 0.0f;                         // underflow
    }
    int k = 0;
    volatile float hi = x, lo = 0.0f;
    if (ax > 0x3eb17218u) { // |x| > 0.5 ln2
        if (ax > 0x3f851592u) {
            volatile float t = INV_LN2 * x;
            volatile float u = t + (neg ? -0.5f : 0.5f);
            k = (int)u;
        } else {
            k = neg ? -1 : 1;
        }
        volatile float khi = (float)k * LN2_HI;
        hi = x - khi;
        lo = (float)k * LN2_LO;
    } else if (ax <= 0x39000000u) { // |x| <= 2^-14
        return 1.0f + x;
    }
    volatile float r = hi - lo;
    volatile float rr = r * r;
    volatile float p = rr * P2;
    p = P1 + p;
    p = rr * p;
    volatile float c = r - p;
    volatile float num = r * c;
    volatile float den = 2.0f - c;
    volatile float quo = num / den;
    volatile float y = quo - lo;
    y = y + hi;
    y = 1.0f + y;
    return k == 0 ? (float)y : std::scalbn((float)y, k);
}

// microflow-macros/src/ops/fully_connected.rs:100-123
// izp / wzp are values of T; `u8` says how the weight bytes read
void h_preprocess_fc(float iscale, int izp, int in_shape1, const int8_t *w, bool u8, int K, int N,
                     float wscale, int wzp, const int32_t *bias, float bscale, int32_t bzp,
                     float oscale, float *c0, float *c1, int32_t *c2, int32_t *c3) {
    volatile float ratio = bscale / oscale; // (bias_scale / output_scale) * f32(bias - zp)
    for (int j = 0; j < N; ++j) {
        volatile float d = (float)wrap_sub(bias[j], bzp);
        c0[j] = ratio * d;
    }
    volatile float prod = iscale * wscale; // (input_scale * weights_scale) / output_scale
    *c1 = prod / oscale;
    for (int j = 0; j < N; ++j) { // column sums of the K x N matrix, times the input zero point
        int32_t s = 0;
        const int8_t *col = w + (size_t)j * K;
        for (int k = 0; k < K; ++k) s = wrap_add(s, u8 ? (int32_t)(uint8_t)col[k] : (int32_t)col[k]);
        c2[j] = wrap_mul(s, izp);
    }
    *c3 = wrap_mul(wrap_mul(in_shape1, izp), wzp);
}

// microflow-macros/src/ops/conv_2d.rs:100-113, depthwise_conv_2d.rs:106-119
void h_preprocess_conv(float iscale, int n, const int32_t *bias, const float *bscale, int nbs,
                       const int32_t *bzp, int nbz, const float *fscale, int nfq, float oscale,
                       float *c0, float *c1) {
    for (int b = 0; b < n; ++b) {
        // biases.scale.get(b).unwrap_or(scale[0]) and zero_point.get(b).unwrap_or(zp[0]): each array
        // falls back on its own (conv_2d.rs:100-108)
        volatile float ratio = bscale[b < nbs ? b : 0] / oscale;
        volatile float d = (float)wrap_sub(bias[b], bzp[b < nbz ? b : 0]);
        c0[b] = ratio * d;
    }
    for (int b = 0; b < nfq; ++b) {
        volatile float prod = iscale * fscale[b];
        c1[b] = prod / oscale;
    }
}

// microflow-macros/src/ops/average_pool_2d.rs:77-83
void h_preprocess_pool(float iscale, int izp, float oscale, int ozp, float *c0, float *c1) {
    *c0 = iscale / oscale;
    volatile float prod = iscale * (float)izp;
    volatile float q = prod / oscale;
    *c1 = (float)ozp - q;
}

} // namespace mf
