/*
 * mf_oracle.c -- CPU ORACLE (test infrastructure only; see mf_oracle.h).
 *
 * Plain-C restatement of the MicroFlow reference algorithm for the quantized
 * operator hot path.  Citations are file:line in the upstream repository
 * (matteocarnelos/microflow-rs, microflow 0.1.3).
 *
 * Structure is deliberately the reference's: one view extraction per output
 * pixel, then per output channel three integer passes (dot product, view sum,
 * masked filter sum) and the f32 epilogue.  Do not "optimise" this file: it is
 * also the reference-faithful CPU baseline that bench.py times.
 */
#include "mf_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ======================================================================= */
/* scalar primitives                                                        */
/* ======================================================================= */

static inline uint32_t f2u(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}
static inline float u2f(uint32_t u) {
    float f;
    memcpy(&f, &u, 4);
    return f;
}

/* libm::roundf (Cargo.toml:27, call sites src/quantize.rs:17 and every op
 * epilogue): round half away from zero, exact. */
float orc_roundf(float x) {
    if (!(fabsf(x) < 8388608.0f)) return x; /* |x| >= 2^23, inf, NaN: already integral */
    float t = truncf(x);
    float d = x - t; /* exact */
    if (fabsf(d) >= 0.5f) t += copysignf(1.0f, x);
    return t;
}

/* libm::expf as shipped by the `libm` 0.2 crate (a port of musl's expf.c,
 * itself FreeBSD msun e_expf.c).  The crate source is not in the reference
 * tree; this restates the published algorithm: argument reduction
 * x = k*ln2 + r, |r| <= 0.5 ln2, then exp(r) = 1 + r + r*c/(2-c) with
 * c = r - r^2*(P1 + r^2*P2); all arithmetic in f32, no fusing.
 * Call sites: src/activation.rs:45, src/ops/softmax.rs:21. */
float orc_expf(float x) {
    static const float half[2] = {0.5f, -0.5f};
    const float ln2hi = 6.9314575195e-1f;  /* 0x3f317200 */
    const float ln2lo = 1.4286067653e-6f;  /* 0x35bfbe8e */
    const float invln2 = 1.4426950216e+0f; /* 0x3fb8aa3b */
    const float P1 = 1.6666625440e-1f;     /* 0x3e2aaa8f */
    const float P2 = -2.7667332906e-3f;    /* 0xbb355215 */
    const float x1p127 = u2f(0x7f000000u);
    uint32_t hx = f2u(x);
    int sign = (int)(hx >> 31);
    int k;
    float hi, lo, c, xx, y;
    hx &= 0x7fffffffu;
    if (hx >= 0x42aeac50u) { /* |x| >= 87.33655 or NaN */
        if (hx > 0x7f800000u) return x; /* NaN */
        if (hx >= 0x42b17218u && !sign) { /* x >= 88.722839 */
            x *= x1p127;
            return x;
        }
        if (sign) {
            if (hx >= 0x42cff1b5u) return 0.0f; /* x <= -103.972084 */
        }
    }
    if (hx > 0x3eb17218u) {     /* |x| > 0.5 ln2 */
        if (hx > 0x3f851592u) { /* |x| > 1.5 ln2 */
            k = (int)(invln2 * x + half[sign]);
        } else {
            k = 1 - sign - sign;
        }
        hi = x - (float)k * ln2hi; /* k*ln2hi is exact here */
        lo = (float)k * ln2lo;
        x = hi - lo;
    } else if (hx > 0x39000000u) { /* |x| > 2**-14 */
        k = 0;
        hi = x;
        lo = 0.0f;
    } else {
        return 1.0f + x;
    }
    xx = x * x;
    c = x - xx * (P1 + xx * P2);
    y = 1.0f + (x * c / (2.0f - c) - lo + hi);
    if (k == 0) return y;
    return scalbnf(y, k);
}

/* wrapping i32 arithmetic (Rust release semantics) */
static inline int32_t wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
static inline int32_t wsub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
static inline int32_t wmul(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }


/* element-type generic scalar primitives, view extraction and operators */
#define ELEM int8_t
#define ELEM_MIN (-128)
#define ELEM_MAX 127
#define FN(n) n
#include "mf_oracle_ops.inc"
#undef ELEM
#undef ELEM_MIN
#undef ELEM_MAX
#undef FN
int8_t orc_sat_i8(float x) { return orc_sat(x); }
#define ELEM uint8_t
#define ELEM_MIN 0
#define ELEM_MAX 255
#define FN(n) n##_u8
#include "mf_oracle_ops.inc"
#undef ELEM
#undef ELEM_MIN
#undef ELEM_MAX
#undef FN

/* ======================================================================= */
/* constant preparation                                                     */
/* ======================================================================= */

/* microflow-macros/src/ops/conv_2d.rs:100-113 and depthwise_conv_2d.rs:106-119 */
/* biases.scale.get(b).unwrap_or(scale[0]) and biases.zero_point.get(b).unwrap_or(zero_point[0]) fall back
 * independently (conv_2d.rs:100-108): nbs scales, nbz zero points */
static void preprocess_conv_int(float iscale, int n, const int32_t *bias, const float *bscale, int nbs,
                                const int32_t *bzp, int nbz, const float *fscale, int nfq, float oscale,
                                float *c0, float *c1) {
    for (int b = 0; b < n; ++b) {
        float bs = b < nbs ? bscale[b] : bscale[0];
        int32_t bz = b < nbz ? bzp[b] : bzp[0];
        float r = bs / oscale;
        c0[b] = r * (float)wsub(bias[b], bz);
    }
    for (int b = 0; b < nfq; ++b) {
        float p = iscale * fscale[b];
        c1[b] = p / oscale;
    }
}
void orc_preprocess_conv(float iscale, int n, const int32_t *bias, const float *bscale,
                         const int32_t *bzp, int nbq, const float *fscale, int nfq, float oscale,
                         float *c0, float *c1) {
    preprocess_conv_int(iscale, n, bias, bscale, nbq, bzp, nbq, fscale, nfq, oscale, c0, c1);
}

/* microflow-macros/src/ops/average_pool_2d.rs:77-83 */
static void preprocess_pool_int(float iscale, int izp, float oscale, int ozp, float *c0, float *c1) {
    *c0 = iscale / oscale;
    float p = iscale * (float)izp;
    float q = p / oscale;
    *c1 = (float)ozp - q;
}
void orc_preprocess_average_pool_2d(float iscale, int8_t izp, float oscale, int8_t ozp, float *c0,
                                    float *c1) {
    preprocess_pool_int(iscale, izp, oscale, ozp, c0, c1);
}

/* ======================================================================= */
/* minimal FlatBuffers reader for the TFLite schema                         */
/* (microflow-macros/flatbuffers/tflite.fbs; field ids in comments)         */
/* ======================================================================= */
typedef struct {
    const uint8_t *p;
    size_t n;
    int bad;
} fb_t;

static uint32_t rd32(fb_t *b, size_t o) {
    if (o + 4 > b->n) {
        b->bad = 1;
        return 0;
    }
    return (uint32_t)b->p[o] | (uint32_t)b->p[o + 1] << 8 | (uint32_t)b->p[o + 2] << 16 |
           (uint32_t)b->p[o + 3] << 24;
}
static uint16_t rd16(fb_t *b, size_t o) {
    if (o + 2 > b->n) {
        b->bad = 1;
        return 0;
    }
    return (uint16_t)(b->p[o] | b->p[o + 1] << 8);
}
static uint8_t rd8(fb_t *b, size_t o) {
    if (o + 1 > b->n) {
        b->bad = 1;
        return 0;
    }
    return b->p[o];
}
/* absolute offset of field `id` inside table `t`, 0 when absent */
static size_t fld(fb_t *b, size_t t, int id) {
    if (!t) return 0;
    int32_t so = (int32_t)rd32(b, t);
    size_t vt = (size_t)((int64_t)t - so);
    uint16_t vsz = rd16(b, vt);
    size_t e = 4 + 2 * (size_t)id;
    if (e + 2 > vsz) return 0;
    uint16_t off = rd16(b, vt + e);
    return off ? t + off : 0;
}
/* follow a uoffset stored at o (0 when o == 0) */
static size_t ind(fb_t *b, size_t o) { return o ? o + rd32(b, o) : 0; }
static size_t tbl(fb_t *b, size_t t, int id) { return ind(b, fld(b, t, id)); }
static uint32_t veclen(fb_t *b, size_t v) { return v ? rd32(b, v) : 0; }
static size_t vec_tbl(fb_t *b, size_t v, uint32_t i) { return ind(b, v + 4 + 4 * (size_t)i); }
static int32_t fld_i32(fb_t *b, size_t t, int id, int32_t d) {
    size_t o = fld(b, t, id);
    return o ? (int32_t)rd32(b, o) : d;
}
static int8_t fld_i8(fb_t *b, size_t t, int id, int8_t d) {
    size_t o = fld(b, t, id);
    return o ? (int8_t)rd8(b, o) : d;
}

/* ======================================================================= */
/* model                                                                    */
/* ======================================================================= */
typedef struct {
    int shape[4], rank;
    int type; /* TensorType: INT32=2, UINT8=3, INT8=9 */
    uint32_t buffer;
    int nscale, nzp;
    float *scale;
    int64_t *zp;
    const uint8_t *data;
    size_t data_len;
} tens_t;

typedef struct {
    orc_op_info info;
    /* geometry in oracle terms */
    int H, W, C, N, WC, M, K, OH, OW;
    int8_t *weights;
    int8_t *wzp; /* nq entries */
    int nq;
    float *c0, *c1;
    int32_t *c2, c3;
    float pool_c0, pool_c1;
} op_t;

struct orc_model {
    int is_u8; /* element type of every activation/weight tensor: 0 = INT8, 1 = UINT8 */
    int nops;
    op_t *ops;
    int in_shape[4], in_rank, out_shape[4], out_rank;
    float in_scale, out_scale;
    int in_zp, out_zp;
    size_t in_elems, out_elems, max_elems, layers_elems;
};

/* zero points are i64 in the file, cast to T (microflow-macros/src/tensor.rs:81-88) */
static int zp_of(int64_t z, int is_u8) { return is_u8 ? (int)(uint8_t)z : (int)(int8_t)z; }

static size_t prod(const int *s, int r) {
    size_t p = 1;
    for (int i = 0; i < r; ++i) p *= (size_t)s[i];
    return p;
}

static void tens_free(tens_t *t) {
    free(t->scale);
    free(t->zp);
}

/* Tensor { shape:0, type:1, buffer:2, name:3, quantization:4 }
 * QuantizationParameters { min:0, max:1, scale:2, zero_point:3 }
 * Buffer { data:0 } */
static int read_tensor(fb_t *b, size_t tensors, size_t buffers, int idx, tens_t *t) {
    memset(t, 0, sizeof(*t));
    if (idx < 0 || (uint32_t)idx >= veclen(b, tensors)) return -1;
    size_t tt = vec_tbl(b, tensors, (uint32_t)idx);
    size_t sh = tbl(b, tt, 0);
    uint32_t r = veclen(b, sh);
    if (r > 4) return -1;
    t->rank = (int)r;
    for (uint32_t i = 0; i < r; ++i) t->shape[i] = (int32_t)rd32(b, sh + 4 + 4 * i);
    t->type = fld_i8(b, tt, 1, 0);
    size_t bo = fld(b, tt, 2);
    t->buffer = bo ? rd32(b, bo) : 0;
    size_t q = tbl(b, tt, 4);
    if (q) {
        size_t sv = tbl(b, q, 2), zv = tbl(b, q, 3);
        t->nscale = (int)veclen(b, sv);
        t->nzp = (int)veclen(b, zv);
        t->scale = (float *)malloc(sizeof(float) * (size_t)(t->nscale ? t->nscale : 1));
        t->zp = (int64_t *)malloc(sizeof(int64_t) * (size_t)(t->nzp ? t->nzp : 1));
        for (int i = 0; i < t->nscale; ++i) t->scale[i] = u2f(rd32(b, sv + 4 + 4 * (size_t)i));
        for (int i = 0; i < t->nzp; ++i) {
            uint64_t lo = rd32(b, zv + 4 + 8 * (size_t)i), hi = rd32(b, zv + 8 + 8 * (size_t)i);
            t->zp[i] = (int64_t)(lo | hi << 32);
        }
    }
    if (t->buffer < veclen(b, buffers)) {
        size_t bt = vec_tbl(b, buffers, t->buffer);
        size_t dv = tbl(b, bt, 0);
        t->data_len = veclen(b, dv);
        t->data = dv ? b->p + dv + 4 : NULL;
        if (dv && dv + 4 + t->data_len > b->n) return -1;
    }
    return b->bad ? -1 : 0;
}

/* the macro's rank fix for 2-D token tensors: microflow-macros/src/tensor.rs:67-70 */
static void rank1_fix(tens_t *t) {
    if (t->rank == 1) {
        t->shape[1] = t->shape[0];
        t->shape[0] = 1;
        t->rank = 2;
    }
}

static void op_free(op_t *o) {
    free(o->weights);
    free(o->wzp);
    free(o->c0);
    free(o->c1);
    free(o->c2);
}

void orc_model_free(orc_model *m) {
    if (!m) return;
    for (int i = 0; i < m->nops; ++i) op_free(&m->ops[i]);
    free(m->ops);
    free(m);
}

#define FAIL(msg)            \
    do {                     \
        if (err) *err = msg; \
        goto fail;           \
    } while (0)

static void fill_info_shapes(op_t *o, const tens_t *in, const tens_t *out, int is_u8) {
    o->info.in_rank = in->rank;
    o->info.out_rank = out->rank;
    for (int i = 0; i < 4; ++i) {
        o->info.in_shape[i] = i < in->rank ? in->shape[i] : 0;
        o->info.out_shape[i] = i < out->rank ? out->shape[i] : 0;
    }
    o->info.in_scale = in->nscale ? in->scale[0] : 0.0f;
    o->info.in_zp = in->nzp ? zp_of(in->zp[0], is_u8) : 0;
    o->info.out_scale = out->nscale ? out->scale[0] : 0.0f;
    o->info.out_zp = out->nzp ? zp_of(out->zp[0], is_u8) : 0;
    o->info.out_elems = prod(out->shape, out->rank);
}

orc_model *orc_model_load(const uint8_t *buf, size_t len, const char **err) {
    fb_t fb = {buf, len, 0};
    fb_t *b = &fb;
    orc_model *m = (orc_model *)calloc(1, sizeof(*m));
    tens_t tin = {0}, tw = {0}, tb = {0}, tout = {0};
    if (!m) return NULL;
    if (len < 8) FAIL("invalid model");
    /* Model { version:0, operator_codes:1, subgraphs:2, description:3, buffers:4 } */
    size_t model = rd32(b, 0); /* root uoffset */
    size_t opcodes = tbl(b, model, 1), subgraphs = tbl(b, model, 2), buffers = tbl(b, model, 4);
    if (b->bad || !opcodes || !subgraphs || !buffers || veclen(b, subgraphs) < 1)
        FAIL("invalid model");
    /* SubGraph { tensors:0, inputs:1, outputs:2, operators:3 }; subgraph 0 only (lib.rs:62) */
    size_t sg = vec_tbl(b, subgraphs, 0);
    size_t tensors = tbl(b, sg, 0), sin = tbl(b, sg, 1), sout = tbl(b, sg, 2),
           ops = tbl(b, sg, 3);
    if (b->bad || !tensors || !sin || !sout || !ops || !veclen(b, sin) || !veclen(b, sout))
        FAIL("invalid model");

    /* model input: lib.rs:66-126 */
    if (read_tensor(b, tensors, buffers, (int32_t)rd32(b, sin + 4), &tin)) FAIL("invalid model");
    rank1_fix(&tin);
    if (tin.type != 9 && tin.type != 3) FAIL("unsupported input tensor type");
    m->is_u8 = tin.type == 3;
    const int ET = tin.type; /* every quantized tensor of the model must have this type */
    if (tin.rank != 2 && tin.rank != 4) FAIL("unsupported input tensor rank");
    if (!tin.nscale || !tin.nzp) FAIL("invalid model");
    m->in_rank = tin.rank;
    memcpy(m->in_shape, tin.shape, sizeof(m->in_shape));
    m->in_scale = tin.scale[0];
    m->in_zp = zp_of(tin.zp[0], m->is_u8);
    m->in_elems = prod(tin.shape, tin.rank);
    tens_free(&tin);
    memset(&tin, 0, sizeof(tin));

    /* model output: lib.rs:153-183 */
    if (read_tensor(b, tensors, buffers, (int32_t)rd32(b, sout + 4), &tout)) FAIL("invalid model");
    rank1_fix(&tout);
    if (tout.type != ET) FAIL("unsupported output tensor type");
    if (tout.rank != 2 && tout.rank != 4) FAIL("unsupported output tensor rank");
    if (!tout.nscale || !tout.nzp) FAIL("invalid model");
    m->out_rank = tout.rank;
    memcpy(m->out_shape, tout.shape, sizeof(m->out_shape));
    m->out_scale = tout.scale[0];
    m->out_zp = zp_of(tout.zp[0], m->is_u8);
    m->out_elems = prod(tout.shape, tout.rank);
    tens_free(&tout);
    memset(&tout, 0, sizeof(tout));

    m->nops = (int)veclen(b, ops);
    m->ops = (op_t *)calloc((size_t)(m->nops ? m->nops : 1), sizeof(op_t));
    if (!m->ops) FAIL("out of memory");
    m->max_elems = m->in_elems;

    for (int oi = 0; oi < m->nops; ++oi) { /* lib.rs:130-151 */
        op_t *o = &m->ops[oi];
        /* Operator { opcode_index:0, inputs:1, outputs:2, builtin_options_type:3, builtin_options:4 } */
        size_t op = vec_tbl(b, ops, (uint32_t)oi);
        size_t oc_o = fld(b, op, 0);
        uint32_t opcode_index = oc_o ? rd32(b, oc_o) : 0;
        if (opcode_index >= veclen(b, opcodes)) FAIL("invalid model");
        /* OperatorCode { deprecated_builtin_code:0 } -- the reference reads only this (lib.rs:131-137) */
        int code = fld_i8(b, vec_tbl(b, opcodes, opcode_index), 0, 0);
        size_t oin = tbl(b, op, 1), oout = tbl(b, op, 2), opt = tbl(b, op, 4);
        if (!oin || !oout || !veclen(b, oin) || !veclen(b, oout)) FAIL("invalid model");
        int i0 = (int32_t)rd32(b, oin + 4);
        int o0 = (int32_t)rd32(b, oout + 4);
        o->info.kind = code;
        if (read_tensor(b, tensors, buffers, i0, &tin) || read_tensor(b, tensors, buffers, o0, &tout))
            FAIL("invalid model");
        if (code != ORC_OP_RESHAPE && (tin.type != ET || tout.type != ET || !tin.nscale || !tin.nzp ||
                                       !tout.nscale || !tout.nzp))
            FAIL("operator tensors must all have the model's element type (INT8 or UINT8)");

        if (code == ORC_OP_FULLY_CONNECTED) {
            /* microflow-macros/src/ops/fully_connected.rs:66-98 */
            if (veclen(b, oin) < 3) FAIL("invalid model");
            if (read_tensor(b, tensors, buffers, (int32_t)rd32(b, oin + 8), &tw) ||
                read_tensor(b, tensors, buffers, (int32_t)rd32(b, oin + 12), &tb))
                FAIL("invalid model");
            rank1_fix(&tin);
            rank1_fix(&tout);
            rank1_fix(&tw);
            rank1_fix(&tb);
            if (tw.rank != 2 || tw.type != ET || tb.type != 2 || !tw.nscale || !tw.nzp ||
                !tb.nscale || !tb.nzp)
                FAIL("invalid fully_connected tensors");
            int N = tw.shape[0], K = tw.shape[1];
            if (tw.data_len < (size_t)N * K || tb.data_len < (size_t)N * 4) FAIL("invalid model");
            fill_info_shapes(o, &tin, &tout, m->is_u8);
            o->M = tin.shape[0];
            o->K = K;
            o->N = N;
            if (prod(tin.shape, tin.rank) != (size_t)o->M * K) FAIL("fully_connected shape mismatch");
            o->info.out_elems = (size_t)o->M * N;
            o->info.act = fld_i8(b, opt, 0, 0); /* FullyConnectedOptions { act:0 } */
            o->weights = (int8_t *)malloc((size_t)N * K);
            memcpy(o->weights, tw.data, (size_t)N * K);
            o->nq = 1;
            o->wzp = (int8_t *)malloc(1);
            o->wzp[0] = (int8_t)tw.zp[0];
            int32_t *bias = (int32_t *)malloc(sizeof(int32_t) * (size_t)N);
            memcpy(bias, tb.data, sizeof(int32_t) * (size_t)N);
            o->c0 = (float *)malloc(sizeof(float) * (size_t)N);
            o->c1 = (float *)malloc(sizeof(float));
            o->c2 = (int32_t *)malloc(sizeof(int32_t) * (size_t)N);
            o->info.n_c0 = N;
            o->info.n_c1 = 1;
            if (m->is_u8)
                orc_preprocess_fully_connected_u8(tin.scale[0], (uint8_t)tin.zp[0], tin.shape[1],
                                                  (const uint8_t *)o->weights, K, N, tw.scale[0],
                                                  (uint8_t)tw.zp[0], bias, tb.scale[0], (int32_t)tb.zp[0],
                                                  tout.scale[0], o->c0, o->c1, o->c2, &o->c3);
            else
                orc_preprocess_fully_connected(tin.scale[0], (int8_t)tin.zp[0], tin.shape[1], o->weights,
                                               K, N, tw.scale[0], (int8_t)tw.zp[0], bias, tb.scale[0],
                                               (int32_t)tb.zp[0], tout.scale[0], o->c0, o->c1, o->c2,
                                               &o->c3);
            free(bias);
        } else if (code == ORC_OP_CONV_2D || code == ORC_OP_DEPTHWISE_CONV_2D) {
            /* microflow-macros/src/ops/conv_2d.rs:58-83, depthwise_conv_2d.rs:62-89 */
            if (veclen(b, oin) < 3) FAIL("invalid model");
            if (read_tensor(b, tensors, buffers, (int32_t)rd32(b, oin + 8), &tw) ||
                read_tensor(b, tensors, buffers, (int32_t)rd32(b, oin + 12), &tb))
                FAIL("invalid model");
            rank1_fix(&tb);
            if (tin.rank != 4 || tout.rank != 4 || tw.rank != 4 || tw.type != ET || tb.type != 2 ||
                !tw.nscale || !tw.nzp || !tb.nscale || !tb.nzp)
                FAIL("invalid conv tensors");
            if (tin.shape[0] != 1) FAIL("conv path has batch 1 only (src/ops/conv_2d.rs:40)");
            fill_info_shapes(o, &tin, &tout, m->is_u8);
            o->H = tin.shape[1];
            o->W = tin.shape[2];
            o->C = tin.shape[3];
            o->info.KH = tw.shape[1];
            o->info.KW = tw.shape[2];
            o->OH = tout.shape[1];
            o->OW = tout.shape[2];
            /* Conv2DOptions { padding:0, stride_w:1, stride_h:2, act:3 }
             * DepthwiseConv2DOptions { padding:0, stride_w:1, stride_h:2, depth_multiplier:3, act:4 } */
            o->info.pad = fld_i8(b, opt, 0, 0);
            o->info.sw = fld_i32(b, opt, 1, 0);
            o->info.sh = fld_i32(b, opt, 2, 0);
            o->info.act = fld_i8(b, opt, code == ORC_OP_CONV_2D ? 3 : 4, 0);
            size_t wn = prod(tw.shape, 4);
            if (tw.data_len < wn) FAIL("invalid model");
            int n0;
            if (code == ORC_OP_CONV_2D) {
                o->N = tw.shape[0];
                if (tw.shape[3] != o->C) FAIL("conv_2d channel mismatch");
                n0 = o->N;
            } else {
                o->WC = tw.shape[3];
                if (tw.shape[0] != 1) FAIL("depthwise weights batch must be 1");
                n0 = o->WC;
            }
            if (tb.data_len < (size_t)n0 * 4) FAIL("invalid model");
            o->weights = (int8_t *)malloc(wn);
            memcpy(o->weights, tw.data, wn);
            o->nq = tw.nzp;
            o->wzp = (int8_t *)malloc((size_t)tw.nzp);
            for (int i = 0; i < tw.nzp; ++i) o->wzp[i] = (int8_t)tw.zp[i];
            int32_t *bias = (int32_t *)malloc(sizeof(int32_t) * (size_t)n0);
            memcpy(bias, tb.data, sizeof(int32_t) * (size_t)n0);
            int32_t *bz = (int32_t *)malloc(sizeof(int32_t) * (size_t)tb.nzp);
            for (int i = 0; i < tb.nzp; ++i) bz[i] = (int32_t)tb.zp[i];
            o->c0 = (float *)malloc(sizeof(float) * (size_t)n0);
            o->c1 = (float *)malloc(sizeof(float) * (size_t)tw.nscale);
            o->info.n_c0 = n0;
            o->info.n_c1 = tw.nscale;
            preprocess_conv_int(tin.scale[0], n0, bias, tb.scale, tb.nscale, bz, tb.nzp, tw.scale, tw.nscale,
                                tout.scale[0], o->c0, o->c1);
            free(bias);
            free(bz);
        } else if (code == ORC_OP_AVERAGE_POOL_2D) {
            /* microflow-macros/src/ops/average_pool_2d.rs:47-66 */
            if (tin.rank != 4 || tout.rank != 4) FAIL("invalid pool tensors");
            if (tin.shape[0] != 1) FAIL("pool path has batch 1 only");
            fill_info_shapes(o, &tin, &tout, m->is_u8);
            o->H = tin.shape[1];
            o->W = tin.shape[2];
            o->C = tin.shape[3];
            o->OH = tout.shape[1];
            o->OW = tout.shape[2];
            /* Pool2DOptions { padding:0, stride_w:1, stride_h:2, filter_width:3, filter_height:4, act:5 } */
            o->info.pad = fld_i8(b, opt, 0, 0);
            o->info.sw = fld_i32(b, opt, 1, 0);
            o->info.sh = fld_i32(b, opt, 2, 0);
            o->info.KW = fld_i32(b, opt, 3, 0);
            o->info.KH = fld_i32(b, opt, 4, 0);
            o->info.act = fld_i8(b, opt, 5, 0);
            preprocess_pool_int(tin.scale[0], zp_of(tin.zp[0], m->is_u8), tout.scale[0],
                                zp_of(tout.zp[0], m->is_u8), &o->pool_c0, &o->pool_c1);
            o->info.n_c0 = 1;
            o->info.n_c1 = 1;
        } else if (code == ORC_OP_SOFTMAX) {
            /* microflow-macros/src/ops/softmax.rs:44-49 */
            rank1_fix(&tin);
            rank1_fix(&tout);
            if (tout.rank != 2) FAIL("softmax output must be rank 2");
            fill_info_shapes(o, &tin, &tout, m->is_u8);
            o->M = tout.shape[0];
            o->N = tout.shape[1];
        } else if (code == ORC_OP_RESHAPE) {
            /* microflow-macros/src/ops/reshape.rs:33-42 */
            if (tout.rank != 2 && tout.rank != 4) FAIL("Reshape supports only output ranks 2 and 4");
            fill_info_shapes(o, &tin, &tout, m->is_u8);
        } else {
            FAIL("unsupported operator"); /* lib.rs:148 */
        }
        if (b->bad) FAIL("invalid model");
        if (o->info.out_elems > m->max_elems) m->max_elems = o->info.out_elems;
        m->layers_elems += o->info.out_elems;
        tens_free(&tin);
        tens_free(&tw);
        tens_free(&tb);
        tens_free(&tout);
        memset(&tin, 0, sizeof(tin));
        memset(&tw, 0, sizeof(tw));
        memset(&tb, 0, sizeof(tb));
        memset(&tout, 0, sizeof(tout));
    }
    return m;
fail:
    tens_free(&tin);
    tens_free(&tw);
    tens_free(&tb);
    tens_free(&tout);
    orc_model_free(m);
    return NULL;
}

int orc_model_num_ops(const orc_model *m) { return m->nops; }
int orc_model_op_info(const orc_model *m, int i, orc_op_info *info) {
    if (i < 0 || i >= m->nops) return -1;
    *info = m->ops[i].info;
    return 0;
}
int orc_model_op_constants(const orc_model *m, int i, float *c0, float *c1, int32_t *c2,
                           int32_t *c3) {
    if (i < 0 || i >= m->nops) return -1;
    const op_t *o = &m->ops[i];
    if (o->info.kind == ORC_OP_AVERAGE_POOL_2D) {
        if (c0) c0[0] = o->pool_c0;
        if (c1) c1[0] = o->pool_c1;
        return 0;
    }
    if (c0 && o->c0) memcpy(c0, o->c0, sizeof(float) * (size_t)o->info.n_c0);
    if (c1 && o->c1) memcpy(c1, o->c1, sizeof(float) * (size_t)o->info.n_c1);
    if (c2 && o->c2) memcpy(c2, o->c2, sizeof(int32_t) * (size_t)o->info.n_c0);
    if (c3) *c3 = o->c3;
    return 0;
}
int orc_model_is_u8(const orc_model *m) { return m->is_u8; }
size_t orc_model_input_elems(const orc_model *m) { return m->in_elems; }
size_t orc_model_output_elems(const orc_model *m) { return m->out_elems; }
size_t orc_model_layers_elems(const orc_model *m) { return m->layers_elems; }
void orc_model_io_quant(const orc_model *m, float *iscale, int *izp, float *oscale, int *ozp) {
    if (iscale) *iscale = m->in_scale;
    if (izp) *izp = m->in_zp;
    if (oscale) *oscale = m->out_scale;
    if (ozp) *ozp = m->out_zp;
}
void orc_model_io_shape(const orc_model *m, int *in_shape, int *in_rank, int *out_shape,
                        int *out_rank) {
    if (in_shape) memcpy(in_shape, m->in_shape, sizeof(m->in_shape));
    if (in_rank) *in_rank = m->in_rank;
    if (out_shape) memcpy(out_shape, m->out_shape, sizeof(m->out_shape));
    if (out_rank) *out_rank = m->out_rank;
}

/* predict_inner: microflow-macros/src/lib.rs:198-201 -- the ops in file order,
 * each consuming the running tensor (value + scale + zero point). */
int orc_model_run_quantized(const orc_model *m, const int8_t *in_q, int8_t *out_q,
                            int8_t *layers) {
    int8_t *a = (int8_t *)malloc(m->max_elems), *bb = (int8_t *)malloc(m->max_elems);
    if (!a || !bb) {
        free(a);
        free(bb);
        return -1;
    }
    memcpy(a, in_q, m->in_elems);
    size_t cur_elems = m->in_elems;
    float cur_scale = m->in_scale; /* running tensor's scale[0] */
    int cur_zp = m->in_zp;
    int rc = 0;
    size_t loff = 0;
    const int U = m->is_u8;
    const uint8_t *ua;
    uint8_t *ub;
    for (int i = 0; i < m->nops && rc == 0; ++i) {
        const op_t *o = &m->ops[i];
        const orc_op_info *f = &o->info;
        ua = (const uint8_t *)a;
        ub = (uint8_t *)bb;
        const uint8_t *uw = (const uint8_t *)o->weights, *uz = (const uint8_t *)o->wzp;
        switch (f->kind) {
            case ORC_OP_FULLY_CONNECTED:
                if (U)
                    orc_fully_connected_u8(ua, o->M, o->K, uw, o->N, uz[0], f->out_scale,
                                           (uint8_t)f->out_zp, f->act, o->c0, o->c1[0], o->c2, o->c3,
                                           ub);
                else
                    orc_fully_connected(a, o->M, o->K, o->weights, o->N, o->wzp[0], f->out_scale,
                                        (int8_t)f->out_zp, f->act, o->c0, o->c1[0], o->c2, o->c3, bb);
                break;
            case ORC_OP_CONV_2D:
                if (U)
                    rc = orc_conv_2d_u8(ua, o->H, o->W, o->C, uw, o->N, f->KH, f->KW, uz, o->nq,
                                        (uint8_t)cur_zp, f->out_scale, (uint8_t)f->out_zp, f->act,
                                        f->pad, f->sh, f->sw, o->OH, o->OW, o->c0, o->c1, f->n_c1, ub);
                else
                    rc = orc_conv_2d(a, o->H, o->W, o->C, o->weights, o->N, f->KH, f->KW, o->wzp,
                                     o->nq, (int8_t)cur_zp, f->out_scale, (int8_t)f->out_zp, f->act,
                                     f->pad, f->sh, f->sw, o->OH, o->OW, o->c0, o->c1, f->n_c1, bb);
                break;
            case ORC_OP_DEPTHWISE_CONV_2D:
                if (U)
                    rc = orc_depthwise_conv_2d_u8(ua, o->H, o->W, o->C, uw, f->KH, f->KW, o->WC, uz,
                                                  o->nq, (uint8_t)cur_zp, f->out_scale,
                                                  (uint8_t)f->out_zp, f->act, f->pad, f->sh, f->sw,
                                                  o->OH, o->OW, o->c0, o->c1, f->n_c1, ub);
                else
                    rc = orc_depthwise_conv_2d(a, o->H, o->W, o->C, o->weights, f->KH, f->KW, o->WC,
                                               o->wzp, o->nq, (int8_t)cur_zp, f->out_scale,
                                               (int8_t)f->out_zp, f->act, f->pad, f->sh, f->sw, o->OH,
                                               o->OW, o->c0, o->c1, f->n_c1, bb);
                break;
            case ORC_OP_AVERAGE_POOL_2D:
                if (U)
                    rc = orc_average_pool_2d_u8(ua, o->H, o->W, o->C, f->KH, f->KW, f->out_scale,
                                                (uint8_t)f->out_zp, f->act, f->pad, f->sh, f->sw,
                                                o->OH, o->OW, o->pool_c0, o->pool_c1, ub);
                else
                    rc = orc_average_pool_2d(a, o->H, o->W, o->C, f->KH, f->KW, f->out_scale,
                                             (int8_t)f->out_zp, f->act, f->pad, f->sh, f->sw, o->OH,
                                             o->OW, o->pool_c0, o->pool_c1, bb);
                break;
            case ORC_OP_SOFTMAX:
                if (U)
                    orc_softmax_u8(ua, o->M, o->N, cur_scale, f->out_scale, (uint8_t)f->out_zp, ub);
                else
                    orc_softmax(a, o->M, o->N, cur_scale, f->out_scale, (int8_t)f->out_zp, bb);
                break;
            case ORC_OP_RESHAPE: /* src/ops/reshape.rs:3-8 + src/tensor.rs:103-141: logical
                                    NHWC order is preserved, so in row-major memory it is a copy */
                memcpy(bb, a, cur_elems);
                break;
            default: rc = -1;
        }
        if (rc) break;
        if (f->kind != ORC_OP_RESHAPE) { /* Tensor::new(output, output_scale, output_zero_point) */
            cur_scale = f->out_scale;
            cur_zp = f->out_zp;
        }
        cur_elems = f->out_elems;
        if (layers) {
            memcpy(layers + loff, bb, cur_elems);
            loff += cur_elems;
        }
        int8_t *t = a;
        a = bb;
        bb = t;
    }
    if (rc == 0) memcpy(out_q, a, m->out_elems);
    free(a);
    free(bb);
    return rc;
}

/* lib.rs:193-196 ; dequantize uses the output tensor's parameters, which is
 * what the last op stamped on the running tensor */
int orc_model_predict_quantized(const orc_model *m, const int8_t *in_q, float *out) {
    int8_t *q = (int8_t *)malloc(m->out_elems ? m->out_elems : 1);
    if (!q) return -1;
    int rc = orc_model_run_quantized(m, in_q, q, NULL);
    if (rc == 0)
        for (size_t i = 0; i < m->out_elems; ++i)
            out[i] = m->is_u8 ? orc_dequantize_u8((uint8_t)q[i], m->out_scale, (uint8_t)m->out_zp)
                              : orc_dequantize(q[i], m->out_scale, (int8_t)m->out_zp);
    free(q);
    return rc;
}

/* lib.rs:188-191 */
int orc_model_predict(const orc_model *m, const float *in, float *out) {
    int8_t *q = (int8_t *)malloc(m->in_elems ? m->in_elems : 1);
    if (!q) return -1;
    for (size_t i = 0; i < m->in_elems; ++i)
        q[i] = m->is_u8 ? (int8_t)orc_quantize_u8(in[i], m->in_scale, (uint8_t)m->in_zp)
                        : orc_quantize(in[i], m->in_scale, (int8_t)m->in_zp);
    int rc = orc_model_predict_quantized(m, q, out);
    free(q);
    return rc;
}

int orc_model_run_quantized_batch(const orc_model *m, const int8_t *in_q, size_t n,
                                  int8_t *out_q) {
    for (size_t i = 0; i < n; ++i) {
        int rc = orc_model_run_quantized(m, in_q + i * m->in_elems, out_q + i * m->out_elems, NULL);
        if (rc) return rc;
    }
    return 0;
}
