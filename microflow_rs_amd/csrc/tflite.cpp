// tflite.cpp -- reads a .tflite FlatBuffer into a ParsedModel and runs the
// constant preparation for every operator.  This is the host-side replacement of
// the proc-macro's front half (microflow-macros/src/lib.rs:46-151 and the
// Token*::new() constructors in microflow-macros/src/ops/*.rs); the generated
// flatbuffers bindings (23k lines upstream) are replaced by a bounds-checked
// cursor over the handful of tables the path reads.  Field ids are from
// microflow-macros/flatbuffers/tflite.fbs.
#include <cstring>

#include "mf_internal.hpp"

namespace mf {
namespace {

class Buf {
  public:
    Buf(const uint8_t *p, size_t n) : p_(p), n_(n) {}
    template <typename T> T at(size_t off) const {
        if (off + sizeof(T) > n_ || off + sizeof(T) < off)
            fail(MF_ERR_INVALID_MODEL, "invalid model, please provide a valid TensorFlow Lite model");
        T v;
        std::memcpy(&v, p_ + off, sizeof(T));
        return v;
    }
    const uint8_t *ptr(size_t off, size_t len) const {
        if (off + len > n_ || off + len < off)
            fail(MF_ERR_INVALID_MODEL, "invalid model, please provide a valid TensorFlow Lite model");
        return p_ + off;
    }

  private:
    const uint8_t *p_;
    size_t n_;
};

// a FlatBuffers table: position + vtable lookup
struct Table {
    const Buf *b = nullptr;
    size_t pos = 0;
    explicit operator bool() const { return b != nullptr; }
    size_t field(int id) const { // absolute offset of the field, 0 if absent
        if (!b) return 0;
        const int32_t soff = b->at<int32_t>(pos);
        const size_t vt = (size_t)((int64_t)pos - soff);
        const uint16_t vsize = b->at<uint16_t>(vt);
        const size_t slot = 4 + 2 * (size_t)id;
        if (slot + 2 > vsize) return 0;
        const uint16_t o = b->at<uint16_t>(vt + slot);
        return o ? pos + o : 0;
    }
    template <typename T> T scalar(int id, T dflt) const {
        const size_t f = field(id);
        return f ? b->at<T>(f) : dflt;
    }
    Table table(int id) const {
        const size_t f = field(id);
        if (!f) return {};
        return {b, f + b->at<uint32_t>(f)};
    }
};
struct Vec {
    const Buf *b = nullptr;
    size_t pos = 0; // points at the length word
    uint32_t size() const { return b ? b->at<uint32_t>(pos) : 0; }
    template <typename T> T get(uint32_t i) const { return b->at<T>(pos + 4 + sizeof(T) * (size_t)i); }
    Table table(uint32_t i) const {
        const size_t e = pos + 4 + 4 * (size_t)i;
        return {b, e + b->at<uint32_t>(e)};
    }
    const uint8_t *bytes(size_t len) const { return b->ptr(pos + 4, len); }
};
Vec vec(const Table &t, int id) {
    const size_t f = t.field(id);
    if (!f) return {};
    return {t.b, f + t.b->at<uint32_t>(f)};
}

enum { TT_INT32 = 2, TT_UINT8 = 3, TT_INT8 = 9 };

struct TensorInfo {
    std::vector<int> shape;
    int type = 0;
    std::vector<float> scale;
    std::vector<int64_t> zp;
    const uint8_t *data = nullptr;
    size_t data_len = 0;
    size_t elems() const { // dimensions are <= 2^24 each (read_tensor): cap the product well below size_t overflow
        size_t n = 1;
        for (int s : shape) {
            n *= (size_t)s;
            if (n > ((size_t)1 << 40)) fail(MF_ERR_INVALID_MODEL, "invalid model: tensor too large");
        }
        return n;
    }
};

// Tensor { shape:0 type:1 buffer:2 name:3 quantization:4 }
// QuantizationParameters { min:0 max:1 scale:2 zero_point:3 }   Buffer { data:0 }
TensorInfo read_tensor(const Vec &tensors, const Vec &buffers, int32_t index) {
    if (index < 0 || (uint32_t)index >= tensors.size())
        fail(MF_ERR_INVALID_MODEL, "invalid model: tensor index out of range");
    const Table t = tensors.table((uint32_t)index);
    TensorInfo ti;
    const Vec sh = vec(t, 0);
    if (sh.size() > 4) fail(MF_ERR_UNSUPPORTED, "unsupported tensor rank: " + std::to_string(sh.size()));
    for (uint32_t i = 0; i < sh.size(); ++i) {
        const int32_t d = sh.get<int32_t>(i);
        if (d <= 0 || d > (1 << 24)) fail(MF_ERR_INVALID_MODEL, "invalid model: bad tensor dimension");
        ti.shape.push_back(d);
    }
    ti.type = t.scalar<int8_t>(1, 0);
    const Table q = t.table(4);
    if (q) {
        const Vec sc = vec(q, 2), zp = vec(q, 3);
        for (uint32_t i = 0; i < sc.size(); ++i) ti.scale.push_back(sc.get<float>(i));
        for (uint32_t i = 0; i < zp.size(); ++i) ti.zp.push_back(zp.get<int64_t>(i));
    }
    const uint32_t bi = t.scalar<uint32_t>(2, 0);
    if (bi < buffers.size()) {
        const Vec d = vec(buffers.table(bi), 0);
        if (d.b) {
            ti.data_len = d.size();
            ti.data = d.bytes(ti.data_len);
        }
    }
    return ti;
}

// rank-1 shapes get a leading 1 (lib.rs:67-70, microflow-macros/src/tensor.rs:67-70)
void fix_rank1(TensorInfo &t) {
    if (t.shape.size() == 1) t.shape.insert(t.shape.begin(), 1);
}

void need_quant(const TensorInfo &t, const char *what) {
    if (t.scale.empty() || t.zp.empty())
        fail(MF_ERR_INVALID_MODEL, std::string("invalid model: ") + what + " has no quantization");
}

// The reference monomorphises every operator on its input tensor's type (i8 or u8, e.g.
// microflow-macros/src/ops/conv_2d.rs:58-83); a model whose tensors mix the two does not
// type-check there.  Here: every quantized tensor must have the model input's type.
void require_elem(const TensorInfo &t, const char *op, bool u8) {
    if (t.type != TT_INT8 && t.type != TT_UINT8)
        fail(MF_ERR_UNSUPPORTED, std::string(op) + " supports only INT8/UINT8 input tensors, got type " +
                                     std::to_string(t.type));
    if ((t.type == TT_UINT8) != u8)
        fail(MF_ERR_UNSUPPORTED, std::string(op) + ": tensor element type differs from the model input's (mixed INT8/UINT8)");
}
// zero points are i64 in the file and cast to T (microflow-macros/src/tensor.rs:81-88)
int zp_of(int64_t z, bool u8) { return u8 ? (int)(uint8_t)z : (int)(int8_t)z; }

int check_act(int a) { // microflow-macros/src/activation.rs:26-38
    if (a != MF_ACT_NONE && a != MF_ACT_RELU && a != MF_ACT_RELU6)
        fail(MF_ERR_UNSUPPORTED, "unsupported fused activation: " + std::to_string(a) +
                                     ". Supported activations are NONE, RELU, and RELU6");
    return a;
}
int check_pad(int p) {
    if (p != MF_PAD_SAME && p != MF_PAD_VALID) fail(MF_ERR_INVALID_MODEL, "invalid model: bad padding");
    return p;
}

void set_shapes(ParsedOp &op, const TensorInfo &in, const TensorInfo &out, bool u8) {
    op.in_rank = (int)in.shape.size();
    op.out_rank = (int)out.shape.size();
    for (int i = 0; i < op.in_rank; ++i) op.in_shape[i] = in.shape[i];
    for (int i = 0; i < op.out_rank; ++i) op.out_shape[i] = out.shape[i];
    op.in_elems = in.elems();
    op.out_elems = out.elems();
    if (!in.scale.empty()) op.in_scale = in.scale[0];
    if (!in.zp.empty()) op.in_zp = zp_of(in.zp[0], u8);
    if (!out.scale.empty()) op.out_scale = out.scale[0];
    if (!out.zp.empty()) op.out_zp = zp_of(out.zp[0], u8);
}

} // namespace

ParsedModel parse_tflite(const uint8_t *data, size_t len) {
    if (!data || len < 8) fail(MF_ERR_INVALID_MODEL, "invalid model, please provide a valid TensorFlow Lite model");
    const Buf buf(data, len);
    // Model { version:0 operator_codes:1 subgraphs:2 description:3 buffers:4 }
    const Table model{&buf, buf.at<uint32_t>(0)};
    const Vec opcodes = vec(model, 1), subgraphs = vec(model, 2), buffers = vec(model, 4);
    if (!opcodes.b || !subgraphs.b || !buffers.b || subgraphs.size() < 1)
        fail(MF_ERR_INVALID_MODEL, "invalid model, please provide a valid TensorFlow Lite model");
    // SubGraph { tensors:0 inputs:1 outputs:2 operators:3 } -- subgraph 0 only (lib.rs:62)
    const Table sg = subgraphs.table(0);
    const Vec tensors = vec(sg, 0), g_in = vec(sg, 1), g_out = vec(sg, 2), operators = vec(sg, 3);
    if (!tensors.b || !g_in.size() || !g_out.size() || !operators.b)
        fail(MF_ERR_INVALID_MODEL, "invalid model, please provide a valid TensorFlow Lite model");

    ParsedModel pm;
    { // model input (lib.rs:66-126)
        TensorInfo t = read_tensor(tensors, buffers, g_in.get<int32_t>(0));
        fix_rank1(t);
        if (t.type != TT_INT8 && t.type != TT_UINT8)
            fail(MF_ERR_UNSUPPORTED, "unsupported input tensor type: " + std::to_string(t.type) +
                                         ". Supported input types are INT8 and UINT8");
        if (t.shape.size() != 2 && t.shape.size() != 4)
            fail(MF_ERR_UNSUPPORTED, "unsupported input tensor rank: " + std::to_string(t.shape.size()) +
                                         ". Supported ranks are 2 and 4");
        need_quant(t, "model input");
        pm.u8 = t.type == TT_UINT8;
        pm.in_rank = (int)t.shape.size();
        for (int i = 0; i < pm.in_rank; ++i) pm.in_shape[i] = t.shape[i];
        pm.in_scale = t.scale[0];
        pm.in_zp = zp_of(t.zp[0], pm.u8);
        pm.in_elems = t.elems();
    }
    { // model output (lib.rs:153-183)
        TensorInfo t = read_tensor(tensors, buffers, g_out.get<int32_t>(0));
        fix_rank1(t);
        if (t.type != TT_INT8 && t.type != TT_UINT8)
            fail(MF_ERR_UNSUPPORTED, "unsupported output tensor type: " + std::to_string(t.type) +
                                         ". Supported output types are INT8 and UINT8");
        if ((t.type == TT_UINT8) != pm.u8)
            fail(MF_ERR_UNSUPPORTED, "model input and output element types differ (mixed INT8/UINT8)");
        if (t.shape.size() != 2 && t.shape.size() != 4)
            fail(MF_ERR_UNSUPPORTED, "unsupported output tensor rank: " + std::to_string(t.shape.size()) +
                                         ". Supported ranks are 2 and 4");
        need_quant(t, "model output");
        pm.out_rank = (int)t.shape.size();
        for (int i = 0; i < pm.out_rank; ++i) pm.out_shape[i] = t.shape[i];
        pm.out_scale = t.scale[0];
        pm.out_zp = zp_of(t.zp[0], pm.u8);
        pm.out_elems = t.elems();
    }
    pm.max_elems = pm.in_elems;
    // The macro threads ONE running value through the operators (`let input = op(input, ...)`,
    // lib.rs:130-151), so operator i consumes operator i-1's result whatever tensor indices the file
    // names; a shape disagreement is a Rust type error there.  Here it would make a kernel read its
    // input with the wrong per-inference stride, so the chain is checked: tensor index, or at least the
    // same shape, from the graph input through every operator to the graph output.
    int32_t prev_index = g_in.get<int32_t>(0);
    std::vector<int> prev_shape(pm.in_shape, pm.in_shape + pm.in_rank);
    auto same_tensor = [](std::vector<int> a, std::vector<int> b) {
        if (a.size() == 1) a.insert(a.begin(), 1);
        if (b.size() == 1) b.insert(b.begin(), 1);
        return a == b;
    };

    for (uint32_t oi = 0; oi < operators.size(); ++oi) { // lib.rs:130-151
        // Operator { opcode_index:0 inputs:1 outputs:2 builtin_options_type:3 builtin_options:4 }
        const Table op = operators.table(oi);
        const uint32_t oc = op.scalar<uint32_t>(0, 0);
        if (oc >= opcodes.size()) fail(MF_ERR_INVALID_MODEL, "invalid model: opcode index out of range");
        // OperatorCode { deprecated_builtin_code:0 } is what the reference dispatches on
        const int code = opcodes.table(oc).scalar<int8_t>(0, 0);
        const Vec ins = vec(op, 1), outs = vec(op, 2);
        const Table opt = op.table(4);
        if (!ins.size() || !outs.size()) fail(MF_ERR_INVALID_MODEL, "invalid model: operator without tensors");
        TensorInfo in = read_tensor(tensors, buffers, ins.get<int32_t>(0));
        TensorInfo out = read_tensor(tensors, buffers, outs.get<int32_t>(0));
        if (ins.get<int32_t>(0) != prev_index && !same_tensor(in.shape, prev_shape))
            fail(MF_ERR_UNSUPPORTED, "operator " + std::to_string(oi) + " does not consume the previous operator's "
                                     "output (or the model input): only linear operator chains are supported");
        prev_index = outs.get<int32_t>(0);
        prev_shape = out.shape;
        ParsedOp po;
        po.kind = code;

        switch (code) {
        case MF_OP_FULLY_CONNECTED: { // microflow-macros/src/ops/fully_connected.rs:66-98
            require_elem(in, "FullyConnected", pm.u8), require_elem(out, "FullyConnected", pm.u8);
            if (ins.size() < 3) fail(MF_ERR_INVALID_MODEL, "invalid model: FullyConnected needs 3 inputs");
            TensorInfo w = read_tensor(tensors, buffers, ins.get<int32_t>(1));
            TensorInfo b = read_tensor(tensors, buffers, ins.get<int32_t>(2));
            fix_rank1(in), fix_rank1(out), fix_rank1(w), fix_rank1(b);
            need_quant(in, "FullyConnected input"), need_quant(out, "FullyConnected output");
            need_quant(w, "FullyConnected weights"), need_quant(b, "FullyConnected bias");
            if (w.shape.size() != 2 || w.type != (pm.u8 ? TT_UINT8 : TT_INT8) || b.type != TT_INT32)
                fail(MF_ERR_UNSUPPORTED, "FullyConnected: unsupported weights/bias tensors");
            po.N = w.shape[0];
            po.K = w.shape[1];
            po.M = in.shape[0];
            if (in.elems() != (size_t)po.M * po.K || w.data_len < (size_t)po.N * po.K ||
                b.data_len < (size_t)po.N * 4)
                fail(MF_ERR_INVALID_MODEL, "invalid model: FullyConnected shapes do not agree");
            set_shapes(po, in, out, pm.u8);
            po.out_elems = (size_t)po.M * po.N;
            po.act = check_act(opt.scalar<int8_t>(0, 0)); // FullyConnectedOptions { act:0 }
            po.weights.assign((const int8_t *)w.data, (const int8_t *)w.data + (size_t)po.N * po.K);
            po.wzp.assign(1, zp_of(w.zp[0], pm.u8));
            std::vector<int32_t> bias(po.N);
            std::memcpy(bias.data(), b.data, (size_t)po.N * 4);
            po.c0.resize(po.N), po.c1.resize(1), po.c2.resize(po.N);
            h_preprocess_fc(in.scale[0], zp_of(in.zp[0], pm.u8), in.shape[1], po.weights.data(), pm.u8, po.K, po.N,
                            w.scale[0], zp_of(w.zp[0], pm.u8), bias.data(), b.scale[0], (int32_t)b.zp[0],
                            out.scale[0], po.c0.data(), po.c1.data(), po.c2.data(), &po.c3);
            break;
        }
        case MF_OP_CONV_2D:             // microflow-macros/src/ops/conv_2d.rs:58-83
        case MF_OP_DEPTHWISE_CONV_2D: { // microflow-macros/src/ops/depthwise_conv_2d.rs:62-89
            const bool dw = code == MF_OP_DEPTHWISE_CONV_2D;
            const char *nm = dw ? "DepthwiseConv2D" : "Conv2D";
            require_elem(in, nm, pm.u8), require_elem(out, nm, pm.u8);
            if (ins.size() < 3) fail(MF_ERR_INVALID_MODEL, std::string("invalid model: ") + nm + " needs 3 inputs");
            TensorInfo w = read_tensor(tensors, buffers, ins.get<int32_t>(1));
            TensorInfo b = read_tensor(tensors, buffers, ins.get<int32_t>(2));
            fix_rank1(b);
            need_quant(in, nm), need_quant(out, nm), need_quant(w, nm), need_quant(b, nm);
            if (in.shape.size() != 4 || out.shape.size() != 4 || w.shape.size() != 4 || w.type != (pm.u8 ? TT_UINT8 : TT_INT8) ||
                b.type != TT_INT32)
                fail(MF_ERR_UNSUPPORTED, std::string(nm) + ": unsupported tensors");
            if (in.shape[0] != 1 || out.shape[0] != 1) // src/ops/conv_2d.rs:40,49 hard-code batch 1
                fail(MF_ERR_UNSUPPORTED, std::string(nm) + ": tensor batch must be 1 (independent inferences are batched by the caller)");
            set_shapes(po, in, out, pm.u8);
            po.H = in.shape[1], po.W = in.shape[2], po.C = in.shape[3];
            po.KH = w.shape[1], po.KW = w.shape[2];
            po.OH = out.shape[1], po.OW = out.shape[2];
            // Conv2DOptions { padding:0 stride_w:1 stride_h:2 act:3 }
            // DepthwiseConv2DOptions { padding:0 stride_w:1 stride_h:2 depth_multiplier:3 act:4 }
            po.pad = check_pad(opt.scalar<int8_t>(0, 0));
            po.sw = opt.scalar<int32_t>(1, 0);
            po.sh = opt.scalar<int32_t>(2, 0);
            po.act = check_act(opt.scalar<int8_t>(dw ? 4 : 3, 0));
            if (po.sh <= 0 || po.sw <= 0) fail(MF_ERR_INVALID_MODEL, "invalid model: bad strides");
            if (dw) {
                if (w.shape[0] != 1) fail(MF_ERR_INVALID_MODEL, "invalid model: depthwise weights batch != 1");
                po.N = w.shape[3];
            } else {
                po.N = w.shape[0];
                if (w.shape[3] != po.C) fail(MF_ERR_INVALID_MODEL, "invalid model: Conv2D channel mismatch");
            }
            if (out.shape[3] != po.N) fail(MF_ERR_INVALID_MODEL, std::string("invalid model: ") + nm + " output channels");
            if (w.data_len < w.elems() || b.data_len < (size_t)po.N * 4)
                fail(MF_ERR_INVALID_MODEL, "invalid model: short weight buffer");
            po.weights.assign((const int8_t *)w.data, (const int8_t *)w.data + w.elems());
            for (int64_t z : w.zp) po.wzp.push_back(zp_of(z, pm.u8));
            std::vector<int32_t> bias(po.N);
            std::memcpy(bias.data(), b.data, (size_t)po.N * 4);
            std::vector<int32_t> bzp;
            for (int64_t z : b.zp) bzp.push_back((int32_t)z);
            po.c0.resize(po.N), po.c1.resize(w.scale.size());
            h_preprocess_conv(in.scale[0], po.N, bias.data(), b.scale.data(), (int)b.scale.size(), bzp.data(),
                              (int)bzp.size(), w.scale.data(), (int)w.scale.size(), out.scale[0], po.c0.data(),
                              po.c1.data());
            break;
        }
        case MF_OP_AVERAGE_POOL_2D: { // microflow-macros/src/ops/average_pool_2d.rs:47-66
            require_elem(in, "AveragePool2D", pm.u8), require_elem(out, "AveragePool2D", pm.u8);
            need_quant(in, "AveragePool2D"), need_quant(out, "AveragePool2D");
            if (in.shape.size() != 4 || out.shape.size() != 4)
                fail(MF_ERR_UNSUPPORTED, "AveragePool2D: unsupported tensors");
            if (in.shape[0] != 1 || out.shape[0] != 1)
                fail(MF_ERR_UNSUPPORTED, "AveragePool2D: tensor batch must be 1");
            set_shapes(po, in, out, pm.u8);
            po.H = in.shape[1], po.W = in.shape[2], po.C = in.shape[3], po.N = po.C;
            po.OH = out.shape[1], po.OW = out.shape[2];
            // Pool2DOptions { padding:0 stride_w:1 stride_h:2 filter_width:3 filter_height:4 act:5 }
            po.pad = check_pad(opt.scalar<int8_t>(0, 0));
            po.sw = opt.scalar<int32_t>(1, 0);
            po.sh = opt.scalar<int32_t>(2, 0);
            po.KW = opt.scalar<int32_t>(3, 0);
            po.KH = opt.scalar<int32_t>(4, 0);
            po.act = check_act(opt.scalar<int8_t>(5, 0));
            if (po.sh <= 0 || po.sw <= 0 || po.KH <= 0 || po.KW <= 0)
                fail(MF_ERR_INVALID_MODEL, "invalid model: bad pool geometry");
            if (out.shape[3] != po.C) fail(MF_ERR_INVALID_MODEL, "invalid model: pool channels");
            po.c0.resize(1), po.c1.resize(1);
            h_preprocess_pool(in.scale[0], zp_of(in.zp[0], pm.u8), out.scale[0], zp_of(out.zp[0], pm.u8), po.c0.data(),
                              po.c1.data());
            break;
        }
        case MF_OP_SOFTMAX: { // microflow-macros/src/ops/softmax.rs:20-49
            require_elem(in, "Softmax", pm.u8), require_elem(out, "Softmax", pm.u8);
            fix_rank1(in), fix_rank1(out);
            need_quant(in, "Softmax"), need_quant(out, "Softmax");
            if (out.shape.size() != 2) fail(MF_ERR_UNSUPPORTED, "Softmax: output tensor must have rank 2");
            set_shapes(po, in, out, pm.u8);
            po.M = out.shape[0], po.N = out.shape[1];
            if (in.elems() != out.elems()) fail(MF_ERR_INVALID_MODEL, "invalid model: softmax shapes");
            break;
        }
        case MF_OP_RESHAPE: { // microflow-macros/src/ops/reshape.rs:33-60
            if (out.shape.size() != 2 && out.shape.size() != 4)
                fail(MF_ERR_UNSUPPORTED, "Reshape supports only output tensor ranks 2 and 4, got rank " +
                                             std::to_string(out.shape.size()));
            set_shapes(po, in, out, pm.u8);
            if (in.elems() != out.elems()) fail(MF_ERR_INVALID_MODEL, "invalid model: reshape changes size");
            break;
        }
        default:
            fail(MF_ERR_UNSUPPORTED, "unsupported operator: " + std::to_string(code)); // lib.rs:148
        }
        if (po.out_elems > pm.max_elems) pm.max_elems = po.out_elems;
        if (po.in_elems > pm.max_elems) pm.max_elems = po.in_elems;
        pm.ops.push_back(std::move(po));
    }
    {
        std::vector<int> gout(pm.out_shape, pm.out_shape + pm.out_rank);
        if (!operators.size() || (prev_index != g_out.get<int32_t>(0) && !same_tensor(prev_shape, gout)))
            fail(MF_ERR_UNSUPPORTED, "the model output is not the last operator's output: only linear operator "
                                     "chains are supported");
    }
    return pm;
}

} // namespace mf
