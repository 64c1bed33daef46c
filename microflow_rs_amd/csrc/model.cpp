// model.cpp -- the runtime half of what #[model("x.tflite")] generates
// (microflow-macros/src/lib.rs:185-203): predict / predict_quantized /
// predict_inner over a BATCH of independent inferences.
//
// Device memory plan (sized for 288 GB of HBM, no reuse tricks needed):
//   weights + folded constants : resident per op (person_detect: ~0.3 MB)
//   act[0], act[1]             : ping-pong activation buffers, batch x max tensor bytes
//   in_q / io_f32              : staging for host-fed batches
// Every op reads one buffer and writes the other; Reshape is an alias (the 2D<->4D
// maps of src/tensor.rs:103-141 are the identity in NHWC memory).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <memory>
#include <vector>

#include "mf_internal.hpp"

namespace mf {

#define MF_HIP(call)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess)                                                                 \
            fail(e_ == hipErrorOutOfMemory ? MF_ERR_OOM : MF_ERR_HIP,                         \
                 std::string(#call) + ": " + hipGetErrorString(e_));                          \
    } while (0)

struct ModelImpl {
    ParsedModel pm;
    int device = -1;
    bool prepared = false;
    bool generic = false;
    hipStream_t stream = nullptr;
    std::vector<OpImpl *> ops; // nullptr for Reshape
    // fused[i] != nullptr: ops i .. fused_last[i] run as ONE kernel (depthwise + 1x1 conv pairs;
    // the pool -> head conv -> [reshape] -> softmax tail)
    std::vector<FusedImpl *> fused;
    std::vector<int> fused_last;
    // second level: ops first .. last (several first-level groups / operators) as ONE kernel: a run of identical pair
    // groups (k_stage.hip), a one-input-channel depthwise + FullyConnected + Softmax (k_dwfc.hip), the last pair + the
    // tail (k_tail3.hip).  They do not overlap; what is inside stays available for mf_model_run_until.
    struct Stage {
        FusedImpl *f;
        int first, last;
    };
    std::vector<Stage> stages;
    bool fusion = true;
    bool autotune = false; // model_set_autotune: time the run-time-geometry chain candidates at creation
    size_t cap_batch = 0;
    int8_t *act[2] = {nullptr, nullptr};
    int8_t *in_q = nullptr;   // quantized input staging (host-fed or f32 path)
    float *io_f32 = nullptr;  // f32 input / output staging
    size_t io_f32_elems = 0;

    // hipGraph replay of the device-resident path (mf_model_set_graph): the launch sequence of one
    // (input, output, batch) triple is captured the second time it is seen and replayed afterwards,
    // so a small-batch predict costs one graph launch instead of one launch per operator.
    struct GraphKey {
        const void *in = nullptr, *out = nullptr;
        size_t batch = 0;
        int last_op = 0;
        bool in_f32 = false, out_f32 = false;
        uint64_t epoch = 0;
        bool operator==(const GraphKey &o) const {
            return in == o.in && out == o.out && batch == o.batch && last_op == o.last_op &&
                   in_f32 == o.in_f32 && out_f32 == o.out_f32 && epoch == o.epoch;
        }
    };
    bool use_graph = false;
    uint64_t epoch = 0; // bumped whenever buffers or kernel routing change
    hipStream_t cap_stream = nullptr;
    hipStream_t copy_stream = nullptr; // H2D of host-fed batches, overlapped with compute
    hipGraphExec_t gexec = nullptr;
    GraphKey gkey, gcand;
    bool gvalid = false, gcand_valid = false;
    uint64_t graph_launches = 0;

    ~ModelImpl() {
        if (device >= 0) (void)hipSetDevice(device);
        drop_graph();
        if (cap_stream) (void)hipStreamDestroy(cap_stream);
        if (copy_stream) (void)hipStreamDestroy(copy_stream);
        for (const Stage &st : stages) fused_destroy(st.f); // they borrow the groups' buffers: first
        for (FusedImpl *f : fused) fused_destroy(f);
        for (OpImpl *o : ops) op_destroy(o);
        free_buffers();
    }
    void drop_graph() {
        if (gexec) (void)hipGraphExecDestroy(gexec);
        gexec = nullptr;
        gvalid = gcand_valid = false;
        ++epoch;
    }
    void free_buffers() {
        for (auto &p : act) {
            if (p) (void)hipFree(p);
            p = nullptr;
        }
        if (in_q) (void)hipFree(in_q);
        if (io_f32) (void)hipFree(io_f32);
        in_q = nullptr;
        io_f32 = nullptr;
        cap_batch = 0;
    }
};

ModelImpl *model_create(const uint8_t *buf, size_t len) {
    std::unique_ptr<ModelImpl> m(new ModelImpl);
    m->pm = parse_tflite(buf, len);
    return m.release();
}
void model_destroy(ModelImpl *m) { delete m; }
const ParsedModel &model_parsed(const ModelImpl *m) { return m->pm; }
static bool fused_at(const ModelImpl *m, int i) {
    return m->fusion && !m->generic && i >= 0 && i < (int)m->fused.size() && m->fused[(size_t)i];
}
// the second-level group that starts at op i and ends at or before last_op, if any
static const ModelImpl::Stage *stage_at(const ModelImpl *m, int i, int last_op) {
    if (!m->fusion || m->generic) return nullptr;
    for (const ModelImpl::Stage &st : m->stages)
        if (st.first == i && st.last <= last_op) return &st;
    return nullptr;
}
// index of the fused group that swallows op i (without being its first op), or -1
static int fused_owner(const ModelImpl *m, int i) {
    for (int j = i - 1; j >= 0 && j >= i - 4; --j)
        if (fused_at(m, j) && m->fused_last[(size_t)j] >= i) return j;
    return -1;
}
const char *model_op_kernel(const ModelImpl *m, int i) {
    if (!m->prepared || i < 0 || i >= (int)m->ops.size() || !m->ops[i]) return "";
    if (m->fusion && !m->generic) { // the stage a run from operator 0 uses for op i: the covering one that starts first
        const ModelImpl::Stage *best = nullptr;
        for (const ModelImpl::Stage &st : m->stages)
            if (i >= st.first && i <= st.last && (!best || st.first < best->first)) best = &st;
        if (best) return best->first == i ? fused_kernel_name(best->f) : "(fused into the previous operator)";
    }
    if (fused_at(m, i)) return fused_kernel_name(m->fused[(size_t)i]);
    if (fused_owner(m, i) >= 0) return "(fused into the previous operator)";
    return op_kernel_name(m->ops[i]);
}

int model_op_epilogue_mode(const ModelImpl *m, int i) {
    if (!m->prepared || i < 0 || i >= (int)m->ops.size() || !m->ops[i]) return -1;
    if (m->fusion && !m->generic) {
        const ModelImpl::Stage *best = nullptr;
        for (const ModelImpl::Stage &st : m->stages)
            if (i >= st.first && i <= st.last && (!best || st.first < best->first)) best = &st;
        if (best && best->first == i) return fused_epilogue_mode(best->f);
        if (!best && fused_at(m, i)) return fused_epilogue_mode(m->fused[(size_t)i]);
    }
    return op_epilogue_mode(m->ops[i]);
}

static void ensure_capacity(ModelImpl *m, size_t batch) {
    if (batch <= m->cap_batch) return;
    MF_HIP(hipSetDevice(m->device));
    MF_HIP(hipStreamSynchronize(m->stream));
    m->drop_graph();
    m->free_buffers();
    const ParsedModel &pm = m->pm;
    const size_t act_bytes = ((batch * pm.max_elems + 255) / 256) * 256 + 256;
    for (auto &p : m->act) MF_HIP(hipMalloc((void **)&p, act_bytes));
    MF_HIP(hipMalloc((void **)&m->in_q, ((batch * pm.in_elems + 255) / 256) * 256 + 256));
    m->io_f32_elems = batch * std::max(pm.in_elems, pm.out_elems);
    MF_HIP(hipMalloc((void **)&m->io_f32, m->io_f32_elems * sizeof(float) + 256));
    m->cap_batch = batch;
}

void model_prepare(ModelImpl *m, int device, size_t max_batch) {
    dev_require(device);
    if (m->prepared && m->device != device)
        fail(MF_ERR_INVALID_ARG, "model already prepared on another device");
    if (!m->prepared) {
        // Everything is built in locals and committed only when all of it exists: a failure half way
        // (a HIP error, OOM, a geometry the operator rejects) leaves the model unprepared and retryable,
        // on this or another device.
        struct Pending {
            std::vector<OpImpl *> ops;
            std::vector<FusedImpl *> fused;
            ~Pending() {
                for (FusedImpl *f : fused) fused_destroy(f);
                for (OpImpl *o : ops) op_destroy(o);
            }
        } pend;
        std::vector<OpImpl *> &ops = pend.ops;
        // predict_inner's running tensor: every op stamps its output scale / zero point on
        // it (Tensor::new(output, output_scale, output_zero_point)); Reshape keeps them.
        float cur_scale = m->pm.in_scale;
        int cur_zp = m->pm.in_zp;
        for (const ParsedOp &po : m->pm.ops) {
            if (po.kind == MF_OP_RESHAPE) {
                ops.push_back(nullptr);
                continue;
            }
            OpSpec s;
            s.kind = po.kind;
            s.u8 = m->pm.u8;
            s.M = po.M, s.K = po.K, s.N = po.N;
            s.H = po.H, s.W = po.W, s.C = po.C, s.KH = po.KH, s.KW = po.KW;
            s.sh = po.sh ? po.sh : 1, s.sw = po.sw ? po.sw : 1, s.pad = po.pad;
            s.OH = po.OH, s.OW = po.OW, s.act = po.act;
            s.izp = cur_zp;          // input.zero_point[0] of the running tensor (conv_2d.rs:56)
            s.in_scale = cur_scale;  // input.scale[0] of the running tensor (softmax.rs:20)
            s.oscale = po.out_scale, s.ozp = po.out_zp;
            s.weights = po.weights.empty() ? nullptr : po.weights.data();
            s.wzp = po.wzp.empty() ? nullptr : po.wzp.data();
            s.nq = (int)po.wzp.size();
            s.c0 = po.c0.empty() ? nullptr : po.c0.data();
            s.c1 = po.c1.empty() ? nullptr : po.c1.data();
            s.nc1 = (int)po.c1.size();
            s.c2 = po.c2.empty() ? nullptr : po.c2.data();
            s.c3 = po.c3;
            if (po.kind == MF_OP_AVERAGE_POOL_2D) s.pool_c0 = po.c0[0], s.pool_c1 = po.c1[0];
            ops.push_back(nullptr); // reserve the slot first: a throwing push_back must not leak the operator
            ops.back() = op_create(device, s);
            cur_scale = po.out_scale;
            cur_zp = po.out_zp;
        }
        // peephole (0): the boundary quantize of M::predict folds into operator 0 when that is the stem
        if (!ops.empty() && ops[0]) (void)op_set_input_quant(ops[0], m->pm.in_scale, m->pm.in_zp, m->pm.u8);
        // peepholes.  (1) DepthwiseConv2D 3x3 directly followed by a 1x1 Conv2D -> one fused kernel.
        // (2) AveragePool2D (1x1 output) -> Conv2D 1x1 -> [Reshape] -> Softmax -> one tail kernel.
        // (3) FullyConnected (few outputs, one row) -> [Reshape] -> Softmax -> one kernel.
        const size_t n = ops.size();
        std::vector<FusedImpl *> &fused = pend.fused;
        fused.assign(n, nullptr);
        std::vector<int> fused_last(n, -1);
        for (size_t i = 0; i + 1 < n; ++i) {
            if (m->pm.ops[i].kind == MF_OP_DEPTHWISE_CONV_2D && m->pm.ops[i + 1].kind == MF_OP_CONV_2D) {
                if ((fused[i] = fused_create(ops[i], ops[i + 1]))) fused_last[i] = (int)i + 1;
            } else if (m->pm.ops[i].kind == MF_OP_AVERAGE_POOL_2D && m->pm.ops[i + 1].kind == MF_OP_CONV_2D) {
                size_t j = i + 2;
                while (j < n && m->pm.ops[j].kind == MF_OP_RESHAPE) ++j;
                if (j < n && m->pm.ops[j].kind == MF_OP_SOFTMAX &&
                    (fused[i] = fused_tail_create(ops[i], ops[i + 1], ops[j])))
                    fused_last[i] = (int)j;
            } else if (m->pm.ops[i].kind == MF_OP_FULLY_CONNECTED) { // (3) FC -> [Reshape] -> Softmax
                size_t j = i + 1;
                while (j < n && m->pm.ops[j].kind == MF_OP_RESHAPE) ++j;
                if (j < n && m->pm.ops[j].kind == MF_OP_SOFTMAX &&
                    (fused[i] = fused_fc_softmax_create(ops[i], ops[j])))
                    fused_last[i] = (int)j;
            }
            if (fused[i]) i = (size_t)fused_last[i]; // groups do not overlap
        }
        // second level (the guard destroys what was built if anything below throws)
        struct StageGuard {
            std::vector<ModelImpl::Stage> v;
            ~StageGuard() {
                for (const ModelImpl::Stage &st : v) fused_destroy(st.f);
            }
        } sg;
        auto covered = [&](size_t i) {
            for (const ModelImpl::Stage &st : sg.v)
                if ((int)i >= st.first && (int)i <= st.last) return true;
            return false;
        };
        // (4) runs of consecutive pair groups on one tensor shape -> a persistent stage kernel, if one exists for that
        // shape and count (ops.hip: fused_stage_create)
        for (size_t i = 0; i + 1 < n; ++i) {
            if (!fused[i] || fused_last[i] != (int)i + 1) continue;
            std::vector<FusedImpl *> run;
            size_t a = i;
            while (a + 1 < n && fused[a] && fused_last[a] == (int)a + 1 && m->pm.ops[a].H == m->pm.ops[i].H &&
                   m->pm.ops[a].W == m->pm.ops[i].W && m->pm.ops[a].C == m->pm.ops[i].C &&
                   m->pm.ops[a + 1].N == m->pm.ops[i].C && m->pm.ops[a].sh == 1) {
                run.push_back(fused[a]);
                a += 2;
            }
            // the longest prefix of the run a stage kernel accepts (differing zero points, a run longer than the kernel's
            // limit ...); what is left forms the next run
            size_t used = 0;
            for (size_t k = run.size(); k >= 2 && !used; --k)
                if (FusedImpl *f = fused_stage_create(run.data(), (int)k)) {
                    sg.v.push_back({f, (int)i, (int)(i + 2 * k) - 1});
                    used = k;
                }
            if (used) i += 2 * used - 1;             // continue with the pair after the stage
            else if (!run.empty()) i = a - 1;        // no stage for any prefix: continue after the run
        }
        // (4b) two consecutive pair groups whose shapes have a quad kernel (k_quad.hip: a VALU-bound stride-1 pair with the
        // HBM-bound stride-2 pair behind it; the tensor between them stays in LDS)
        for (size_t i = 0; i + 3 < n; ++i) {
            if (!fused[i] || fused_last[i] != (int)i + 1 || covered(i) || covered(i + 2)) continue;
            if (!fused[i + 2] || fused_last[i + 2] != (int)i + 3) continue;
            if (FusedImpl *f = fused_quad_create(fused[i], fused[i + 2])) {
                sg.v.push_back({f, (int)i, (int)i + 3});
                // ... and with the stem in front of it: ops i - 1 .. i + 3 in one launch.  Both stay: a run that does not
                // start at the stem (mf_model_run_until pieces, the f32 entry whose stem kernel quantises) uses the quad
                if (i >= 1 && ops[i - 1] && !fused[i - 1] && !covered(i - 1))
                    if (FusedImpl *g = fused_quad_stem_create(ops[i - 1], f)) sg.v.push_back({g, (int)i - 1, (int)i + 3});
                i += 3;
            }
        }
        // (4b') the last pair group directly followed by the tail group -> one kernel (ops.hip: fused_pair_tail_create).  Before the
        // chain partition below, so that a run-time-geometry pair taken here is not also a candidate there
        for (size_t i = 0; i + 2 < n; ++i) {
            if (!fused[i] || fused_last[i] != (int)i + 1 || covered(i)) continue;
            const size_t j = i + 2;
            if (fused[j] && !covered(j))
                if (FusedImpl *f = fused_pair_tail_create(fused[i], fused[j])) {
                    sg.v.push_back({f, (int)i, fused_last[j]});
                    // ... and with the pair group in front of it: ops i - 2 .. the tail's end in one launch.  Both stay (a run that
                    // ends before the tail uses the pair groups, mf_model_run_until pieces that start at i the pair + tail stage)
                    if (i >= 2 && fused[i - 2] && fused_last[i - 2] == (int)i - 1 && !covered(i - 2))
                        if (FusedImpl *g = fused_front_pair_tail_create(fused[i - 2], f)) sg.v.push_back({g, (int)i - 2, fused_last[j]});
                }
        }
        // (4c) runs of consecutive run-time-geometry pairs (single-pair chain groups, k_chain.hip): the planner's cost model
        // decides how the run is cut into chain launches (ops.hip: fused_chain_partition); a pair that is cheapest as two
        // separate launches loses its group
        for (size_t i = 0; i + 1 < n; ++i) {
            if (!fused_is_chain_single(fused[i]) || fused_last[i] != (int)i + 1 || covered(i)) continue;
            std::vector<FusedImpl *> run;
            size_t a = i;
            while (a + 1 < n && fused_is_chain_single(fused[a]) && fused_last[a] == (int)a + 1 && !covered(a)) {
                run.push_back(fused[a]);
                a += 2;
            }
            const int rn = (int)run.size();
            std::vector<int> seg((size_t)rn, 0);
            std::unique_ptr<bool[]> unf(new bool[(size_t)rn]);
            std::vector<int> segG((size_t)rn, 0);
            fused_chain_partition(run.data(), rn, seg.data(), unf.get(), segG.data(), m->autotune);
            for (int k = 0; k < rn; ++k) {
                const size_t at = i + 2 * (size_t)k;
                if (seg[(size_t)k] >= 2) {
                    if (FusedImpl *f = fused_chain_create(run.data() + k, seg[(size_t)k], segG[(size_t)k])) sg.v.push_back({f, (int)at, (int)(at + 2 * (size_t)seg[(size_t)k]) - 1});
                } else if (seg[(size_t)k] == 1 && unf[(size_t)k]) {
                    fused_destroy(fused[at]);
                    fused[at] = nullptr, fused_last[at] = -1;
                }
            }
            i = a - 1;
        }
        // (5) DepthwiseConv2D with one input channel -> [Reshape] -> the FullyConnected + Softmax group -> one kernel
        for (size_t i = 0; i + 1 < n; ++i) {
            if (!ops[i] || fused[i] || covered(i) || m->pm.ops[i].kind != MF_OP_DEPTHWISE_CONV_2D) continue;
            size_t j = i + 1;
            while (j < n && m->pm.ops[j].kind == MF_OP_RESHAPE) ++j;
            if (j < n && fused[j] && !covered(j))
                if (FusedImpl *f = fused_dwfc_create(ops[i], fused[j])) sg.v.push_back({f, (int)i, fused_last[j]});
        }
        // commit (nothing below throws)
        m->stages.swap(sg.v);
        m->device = device;
        m->ops.swap(pend.ops);
        m->fused.swap(pend.fused);
        m->fused_last.swap(fused_last);
        m->prepared = true;
        model_set_generic(m, m->generic);
    }
    if (max_batch) ensure_capacity(m, max_batch);
}

void model_set_stream(ModelImpl *m, void *stream) { m->stream = (hipStream_t)stream; }

void model_sync(ModelImpl *m) {
    if (!m->prepared) return;
    MF_HIP(hipSetDevice(m->device));
    MF_HIP(hipStreamSynchronize(m->stream));
}

void model_set_autotune(ModelImpl *m, bool enabled) {
    if (m->prepared) fail(MF_ERR_INVALID_ARG, "mf_model_set_autotune: call it before mf_model_prepare");
    m->autotune = enabled;
}

void model_set_fusion(ModelImpl *m, bool enabled) {
    m->fusion = enabled;
    m->drop_graph();
}
void model_set_graph(ModelImpl *m, bool enabled) {
    m->use_graph = enabled;
    if (!enabled) m->drop_graph();
}
uint64_t model_graph_launches(const ModelImpl *m) { return m->graph_launches; }

void model_set_generic(ModelImpl *m, bool generic) {
    m->generic = generic;
    m->drop_graph();
    for (OpImpl *o : m->ops)
        if (o) op_set_generic(o, generic);
}

// run ops [0..last_op] reading from `src`; returns the buffer holding the result
// Is the boundary quantize fused into operator 0 for this call?  (peephole: f32 input, fusion on,
// operator 0 has an f32-input kernel)
static bool f32_first(const ModelImpl *m, int last_op) {
    return m->fusion && !m->generic && last_op >= 0 && !m->ops.empty() && m->ops[0] && op_accepts_f32(m->ops[0]);
}

// run ops [0..last_op]; `src_f32` != nullptr: operator 0 consumes the f32 input directly.  `final_dst` != nullptr: the
// launch that produces operator last_op's tensor writes it there instead of into an activation buffer (no copy of
// the result afterwards: 16 MB per step for the 4096^3 FullyConnected); the return value says where the result is.
static const int8_t *run_ops(ModelImpl *m, const int8_t *src, size_t batch, int last_op, hipStream_t stream,
                             const float *src_f32 = nullptr, int8_t *final_dst = nullptr) {
    const int8_t *cur = src;
    int which = 0;
    int first = 0;
    // the last operator that launches anything (trailing Reshapes alias its output)
    int last_real = last_op;
    while (last_real >= 0 && !m->ops[last_real]) --last_real;
    if (src_f32) {
        // the second-level group that starts at operator 0 takes the f32 image itself if it can (penta_rr, k_quad.hip), else the
        // stem kernel quantises while it stages and the groups behind it run as usual
        const ModelImpl::Stage *st0 = stage_at(m, 0, last_op);
        if (st0 && fused_accepts_f32(st0->f)) {
            int8_t *dst = (final_dst && st0->last >= last_real) ? final_dst : m->act[0];
            fused_run_f32(st0->f, src_f32, batch, dst, stream);
            cur = dst;
            first = st0->last + 1;
        } else {
            op_run_f32(m->ops[0], src_f32, batch, (final_dst && last_real == 0) ? final_dst : m->act[0], stream);
            cur = (final_dst && last_real == 0) ? final_dst : m->act[0];
            first = 1;
        }
        which = 1;
    }
    for (int i = first; i <= last_op; ++i) {
        OpImpl *o = m->ops[i];
        if (!o) continue; // Reshape: alias
        int8_t *dst = m->act[which];
        if (dst == cur) {
            which ^= 1;
            dst = m->act[which];
        }
        const ModelImpl::Stage *staged = stage_at(m, i, last_op);
        const bool grouped = !staged && fused_at(m, i) && m->fused_last[(size_t)i] <= last_op;
        const int end = staged ? staged->last : (grouped ? m->fused_last[(size_t)i] : i);
        if (final_dst && end >= last_real) dst = final_dst;
        if (staged) { // several groups in one launch
            fused_run(staged->f, cur, batch, dst, stream);
        } else if (grouped) { // the whole group in one launch
            fused_run(m->fused[(size_t)i], cur, batch, dst, stream);
        } else {
            op_run(o, cur, batch, dst, stream);
        }
        i = end;
        cur = dst;
        which ^= 1;
    }
    return cur;
}

// The whole device-resident sequence (boundary conversion -> ops -> boundary conversion) on
// `s`: every pointer is a device pointer, nothing synchronizes -- so it can be stream-captured.
static void enqueue_device(ModelImpl *m, const float *in_f32, const int8_t *in_i8, size_t batch,
                           float *out_f32, int8_t *out_i8, int last_op, hipStream_t s) {
    const ParsedModel &pm = m->pm;
    const int nops = (int)pm.ops.size();
    const size_t out_elems = last_op == nops - 1 ? pm.out_elems : pm.ops[last_op].out_elems;
    const int8_t *q_in;
    const bool fuse_q = in_f32 && ((uintptr_t)in_f32 & 15) == 0 && f32_first(m, last_op); // (float4 loads)
    if (fuse_q) {
        q_in = nullptr; // operator 0 quantises while it stages (dw3x3_stem8<.., F32IN>)
    } else if (in_f32) { // Tensor::quantize(input, scale, zero_point)  (lib.rs:189)
        dev_quantize(m->device, in_f32, batch * pm.in_elems, pm.in_scale, pm.in_zp, pm.u8, m->in_q, s);
        q_in = m->in_q;
    } else if (pm.u8) { // u8 -> internal i8 domain
        dev_xor80(m->device, in_i8, batch * pm.in_elems, m->in_q, s);
        q_in = m->in_q;
    } else if (((uintptr_t)in_i8 & 15) != 0) { // the fast kernels read 16-byte words: realign an odd caller pointer
        MF_HIP(hipMemcpyAsync(m->in_q, in_i8, batch * pm.in_elems, hipMemcpyDeviceToDevice, s));
        q_in = m->in_q;
    } else {
        q_in = in_i8; // consumed in place
    }
    // An i8 model's result is written straight into the caller's buffer by the last launch -- unless that buffer overlaps
    // the caller's input, which the first launch may still be reading (in_i8 is consumed in place): a model that is ONE
    // launch would then read and write the same memory.  Overlapping buffers take the activation buffer + copy path.
    int8_t *direct = (out_i8 && !pm.u8 && ((uintptr_t)out_i8 & 15) == 0) ? out_i8 : nullptr; // (16-byte vector stores)
    if (direct) {
        const uintptr_t ob = (uintptr_t)out_i8, oe = ob + batch * out_elems;
        const uintptr_t ib = (uintptr_t)(in_f32 ? (const void *)in_f32 : (const void *)in_i8);
        const uintptr_t ie = ib + batch * pm.in_elems * (in_f32 ? sizeof(float) : 1);
        if (ob < ie && ib < oe) direct = nullptr;
    }
    const int8_t *res = run_ops(m, q_in, batch, last_op, s, fuse_q ? in_f32 : nullptr, direct);
    if (out_i8) {
        if (pm.u8) dev_xor80(m->device, res, batch * out_elems, out_i8, s);
        else if (res != out_i8) MF_HIP(hipMemcpyAsync(out_i8, res, batch * out_elems, hipMemcpyDeviceToDevice, s));
    } else { // .dequantize()  (lib.rs:190) with the parameters the last op stamped on the tensor
        float oscale = pm.out_scale;
        int ozp = pm.out_zp;
        if (last_op != nops - 1) oscale = pm.ops[last_op].out_scale, ozp = pm.ops[last_op].out_zp;
        dev_dequantize(m->device, res, batch * out_elems, oscale, ozp, pm.u8, out_f32, s);
    }
}

// device path with graph replay: eager the first time a key is seen, captured the second time
static void run_device(ModelImpl *m, const float *in_f32, const int8_t *in_i8, size_t batch,
                       float *out_f32, int8_t *out_i8, int last_op) {
    hipStream_t s = m->stream;
    if (!m->use_graph) return enqueue_device(m, in_f32, in_i8, batch, out_f32, out_i8, last_op, s);
    ModelImpl::GraphKey k;
    k.in = in_f32 ? (const void *)in_f32 : (const void *)in_i8;
    k.out = out_f32 ? (const void *)out_f32 : (const void *)out_i8;
    k.batch = batch, k.last_op = last_op, k.in_f32 = in_f32 != nullptr, k.out_f32 = out_f32 != nullptr;
    k.epoch = m->epoch;
    if (m->gvalid && m->gkey == k) {
        MF_HIP(hipGraphLaunch(m->gexec, s));
        ++m->graph_launches;
        return;
    }
    if (!(m->gcand_valid && m->gcand == k)) { // first sighting: run eagerly (lets operators size their scratch)
        m->gcand = k, m->gcand_valid = true;
        return enqueue_device(m, in_f32, in_i8, batch, out_f32, out_i8, last_op, s);
    }
    if (!m->cap_stream) MF_HIP(hipStreamCreateWithFlags(&m->cap_stream, hipStreamNonBlocking));
    MF_HIP(hipStreamBeginCapture(m->cap_stream, hipStreamCaptureModeRelaxed));
    hipGraph_t g = nullptr;
    try {
        enqueue_device(m, in_f32, in_i8, batch, out_f32, out_i8, last_op, m->cap_stream);
    } catch (...) {
        (void)hipStreamEndCapture(m->cap_stream, &g);
        if (g) (void)hipGraphDestroy(g);
        throw;
    }
    MF_HIP(hipStreamEndCapture(m->cap_stream, &g));
    hipGraphExec_t e = nullptr;
    hipError_t err = hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (err != hipSuccess) fail(MF_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(err));
    if (m->gexec) (void)hipGraphExecDestroy(m->gexec);
    m->gexec = e, m->gkey = k, m->gvalid = true, m->gcand_valid = false;
    MF_HIP(hipGraphLaunch(m->gexec, s));
    ++m->graph_launches;
}

void model_run(ModelImpl *m, const float *in_f32, const int8_t *in_i8, size_t batch, float *out_f32,
               int8_t *out_i8, int mem, int last_op) {
    if (!m->prepared) fail(MF_ERR_INVALID_ARG, "model not prepared: call mf_model_prepare first");
    if (!batch) return; // zero inferences: nothing to read or write (empty buffers may be null)
    if ((in_f32 == nullptr) == (in_i8 == nullptr) || (out_f32 == nullptr) == (out_i8 == nullptr))
        fail(MF_ERR_INVALID_ARG, "null buffer");
    if (mem != MF_MEM_HOST && mem != MF_MEM_DEVICE) fail(MF_ERR_INVALID_ARG, "bad mem kind");
    const ParsedModel &pm = m->pm;
    const int nops = (int)pm.ops.size();
    if (last_op < 0) last_op = nops - 1;
    if (last_op >= nops) fail(MF_ERR_INVALID_ARG, "op index out of range");
    if (!batch) return;
    MF_HIP(hipSetDevice(m->device));
    ensure_capacity(m, batch);
    const size_t out_elems = last_op == nops - 1 ? pm.out_elems : pm.ops[last_op].out_elems;
    const bool host = mem == MF_MEM_HOST;
    hipStream_t s = m->stream;
    if (!host) return run_device(m, in_f32, in_i8, batch, out_f32, out_i8, last_op);

    // ---- host buffers staged through HBM ----
    // The batch is cut into chunks of ~64 MB of input: chunk c+1 crosses PCIe on the copy stream
    // while chunk c computes (pinned host memory makes the copies truly asynchronous; pageable
    // memory still works, serialized by the runtime).  Each chunk has its own slice of the
    // staging buffers; the activation ping-pong buffers are reused chunk after chunk, in order.
    const size_t in_bytes = pm.in_elems * (in_f32 ? sizeof(float) : 1);
    const size_t out_bytes = out_elems * (out_f32 ? sizeof(float) : 1);
    size_t chunk = batch;
    if (in_bytes && out_bytes <= in_bytes) { // (a chunk's f32 output slice must not reach later chunks' input slices)
        const size_t per = std::max<size_t>(256, ((64u << 20) / in_bytes) & ~(size_t)255);
        if (batch > per + per / 2) chunk = per;
    }
    if (chunk < batch && !m->copy_stream) MF_HIP(hipStreamCreateWithFlags(&m->copy_stream, hipStreamNonBlocking));
    std::vector<hipEvent_t> evs;
    float oscale = pm.out_scale;
    int ozp = pm.out_zp;
    if (last_op != nops - 1) oscale = pm.ops[last_op].out_scale, ozp = pm.ops[last_op].out_zp;
    try {
        if (chunk < batch) { // the copy stream starts after whatever the caller queued on the compute stream
            hipEvent_t e;
            MF_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            evs.push_back(e);
            MF_HIP(hipEventRecord(e, s));
            MF_HIP(hipStreamWaitEvent(m->copy_stream, e, 0));
        }
        for (size_t first = 0; first < batch; first += chunk) {
            const size_t n = std::min(chunk, batch - first);
            hipStream_t cs = chunk < batch ? m->copy_stream : s;
            int8_t *q = m->in_q + first * pm.in_elems;
            float *f = m->io_f32 + first * pm.in_elems;
            if (in_f32) MF_HIP(hipMemcpyAsync(f, in_f32 + first * pm.in_elems, n * in_bytes, hipMemcpyHostToDevice, cs));
            else MF_HIP(hipMemcpyAsync(q, in_i8 + first * pm.in_elems, n * in_bytes, hipMemcpyHostToDevice, cs));
            if (cs != s) {
                hipEvent_t e;
                MF_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                evs.push_back(e);
                MF_HIP(hipEventRecord(e, cs));
                MF_HIP(hipStreamWaitEvent(s, e, 0));
            }
            const bool fuse_q = in_f32 && f32_first(m, last_op);
            if (in_f32 && !fuse_q) // Tensor::quantize(input, scale, zero_point)  (lib.rs:189)
                dev_quantize(m->device, f, n * pm.in_elems, pm.in_scale, pm.in_zp, pm.u8, q, s);
            else if (!in_f32 && pm.u8)
                dev_xor80(m->device, q, n * pm.in_elems, q, s); // u8 -> internal i8 domain
            const int8_t *res = run_ops(m, q, n, last_op, s, fuse_q ? f : nullptr); // predict_inner
            if (out_i8) {
                if (pm.u8) { // internal i8 domain -> u8, in place in the scratch buffer holding the result
                    int8_t *tmp = res == q ? m->act[0] : const_cast<int8_t *>(res);
                    dev_xor80(m->device, res, n * out_elems, tmp, s);
                    res = tmp;
                }
                MF_HIP(hipMemcpyAsync(out_i8 + first * out_elems, res, n * out_elems, hipMemcpyDeviceToHost, s));
            } else { // .dequantize()  (lib.rs:190) with the parameters the last op stamped on the tensor
                float *fo = m->io_f32 + first * out_elems;
                dev_dequantize(m->device, res, n * out_elems, oscale, ozp, pm.u8, fo, s);
                MF_HIP(hipMemcpyAsync(out_f32 + first * out_elems, fo, n * out_bytes, hipMemcpyDeviceToHost, s));
            }
        }
        MF_HIP(hipStreamSynchronize(s));
    } catch (...) {
        if (m->copy_stream) (void)hipStreamSynchronize(m->copy_stream);
        (void)hipStreamSynchronize(s);
        for (hipEvent_t e : evs) (void)hipEventDestroy(e);
        throw;
    }
    for (hipEvent_t e : evs) (void)hipEventDestroy(e);
}

void model_time_device(ModelImpl *m, const int8_t *d_in, size_t batch, int8_t *d_out, int warmup,
                       int iters, float *avg_ms, float *per_op_ms) {
    if (!m->prepared) fail(MF_ERR_INVALID_ARG, "model not prepared");
    if (iters <= 0 || !batch) fail(MF_ERR_INVALID_ARG, "iters and batch must be positive");
    MF_HIP(hipSetDevice(m->device));
    ensure_capacity(m, batch);
    const int nops = (int)m->pm.ops.size();
    hipStream_t s = m->stream;
    hipEvent_t e0, e1;
    MF_HIP(hipEventCreate(&e0));
    MF_HIP(hipEventCreate(&e1));
    for (int i = 0; i < warmup; ++i) model_run(m, nullptr, d_in, batch, nullptr, d_out, MF_MEM_DEVICE, -1);
    MF_HIP(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) model_run(m, nullptr, d_in, batch, nullptr, d_out, MF_MEM_DEVICE, -1);
    MF_HIP(hipEventRecord(e1, s));
    MF_HIP(hipEventSynchronize(e1));
    float ms = 0;
    MF_HIP(hipEventElapsedTime(&ms, e0, e1));
    if (avg_ms) *avg_ms = ms / (float)iters;

    if (per_op_ms) { // second sweep: one event pair per op, on the same stream
        std::vector<hipEvent_t> ev((size_t)nops + 1);
        for (auto &e : ev) MF_HIP(hipEventCreate(&e));
        std::vector<std::vector<float>> samples((size_t)nops); // per operator: one duration per iteration
        for (int it = 0; it < iters; ++it) {
            const int8_t *cur = d_in;
            int which = 0;
            MF_HIP(hipEventRecord(ev[0], s));
            for (int i = 0; i < nops; ++i) {
                OpImpl *o = m->ops[i];
                if (o) {
                    int8_t *dst = m->act[which];
                    if (dst == cur) {
                        which ^= 1;
                        dst = m->act[which];
                    }
                    if (const ModelImpl::Stage *st = stage_at(m, i, nops - 1)) { // timed as one unit (index i), its other ops 0
                        fused_run(st->f, cur, batch, dst, s);
                        for (int j = i; j < st->last; ++j) MF_HIP(hipEventRecord(ev[(size_t)j + 1], s));
                        i = st->last;
                    } else if (fused_at(m, i)) { // the group is timed as one unit (index i), its other ops 0
                        fused_run(m->fused[(size_t)i], cur, batch, dst, s);
                        for (int j = i; j < m->fused_last[(size_t)i]; ++j) MF_HIP(hipEventRecord(ev[(size_t)j + 1], s));
                        i = m->fused_last[(size_t)i];
                    } else {
                        op_run(o, cur, batch, dst, s);
                    }
                    cur = dst;
                    which ^= 1;
                }
                MF_HIP(hipEventRecord(ev[(size_t)i + 1], s));
            }
            MF_HIP(hipEventSynchronize(ev[(size_t)nops]));
            for (int i = 0; i < nops; ++i) {
                float t = 0;
                MF_HIP(hipEventElapsedTime(&t, ev[(size_t)i], ev[(size_t)i + 1]));
                samples[(size_t)i].push_back(t);
            }
        }
        for (int i = 0; i < nops; ++i) { // median over the iterations
            std::vector<float> &v = samples[(size_t)i];
            std::sort(v.begin(), v.end());
            const size_t n = v.size();
            per_op_ms[i] = n % 2 ? v[n / 2] : 0.5f * (v[n / 2 - 1] + v[n / 2]);
        }
        for (auto &e : ev) (void)hipEventDestroy(e);
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
}

} // namespace mf
