"""The other workloads and sub-records of bench.py's default line.  ctx["checker"] is the CPU restatement bench.py imported (these
legs spot-check their outputs against it; they never import it themselves)."""
import json
import os
import subprocess
import sys
import time

import numpy as np

from . import roofline
from .roofline import (DOT4_PEAK_TMACS, HBM_PEAK_GBS, INT8_MFMA_PEAK_GUIDE_FLOOR, INT8_MFMA_PEAK_NOMINAL, ROOT, event_times, median,
                       pmc_traffic, requant_peak, rocprof_name, sq_counters)

def speech_record(ctx):
    """BASELINE config 2: speech.tflite (TinyConv), batch 4096, device-resident int8 -> int8; plus the same model at
    batch 65536 (the throughput regime: 4096 inferences are ONE 16-image step per CU, i.e. launch + latency)."""
    mf, _lib, torch, synth_i8, SEED = ctx["mf"], ctx["_lib"], ctx["torch"], ctx["synth_i8"], ctx["SEED"]
    O = ctx["checker"]  # the CPU restatement, imported by bench.py
    path = os.path.join(ROOT, "models", "speech.tflite")
    L = _lib.lib()
    om = O.Model(path)

    def run(B, iters):
        m = mf.model(path)
        m.prepare(B, device=ctx["local_rank"])
        _lib.check(L.mf_model_set_stream(m._h, torch.cuda.current_stream().cuda_stream))
        x = synth_i8(SEED + 2, 0, B * m.input_elems)
        y = torch.empty(B * m.output_elems, dtype=torch.int8, device="cuda")
        step = lambda: _lib.check(L.mf_model_run_quantized(m._h, x.data_ptr(), B, y.data_ptr(), _lib.MF_MEM_DEVICE))  # noqa: E731
        roofline.prewarm(torch, step)
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        ev = event_times(torch, step, iters)
        ms = median(ev)
        _, per_op = m.time_device(x, y, B, warmup=2, iters=20)
        descs = [m.op(i) for i in range(m.num_ops)]
        kernels = [{"op": i, "kernel": d["kernel"], "ms": round(per_op[i], 5)} for i, d in enumerate(descs)
                   if d["kernel"] and not d["kernel"].startswith("(fused")]
        idx = list(range(0, B, max(1, B // 16) + 1))
        ok = bool(np.array_equal(y.reshape(B, -1)[idx].cpu().numpy(), om.run_quantized_batch(x.reshape(B, -1)[idx].cpu().numpy())))
        return m, ms, len(ev), kernels, ok, len(idx)

    B = 4096
    m, ms, nev, kernels, ok, nidx = run(B, 50)
    # one launch (k_dwfc.hip): the depthwise taps run on the matrix pipe, what is left on the VALU is the
    # requantisation of the 4000 depthwise outputs per inference -> the same ceiling as the fused person_detect kernels
    one = next((k for k in kernels if k["kernel"].startswith("dwc1_fc")), None)
    nbytes = (m.input_elems + m.output_elems) * B  # model input + output: all the HBM traffic there is
    rec = {"metric": "inferences/sec (int8) for speech.tflite", "value": round(B / (ms * 1e-3), 1), "value_batch": B, "unit": "inferences/s",
           "ms_per_step": round(ms, 5), "config": {"workload": "speech.tflite batch=%d, predict_inner int8->int8" % B},
           "kernels": kernels, "timing": "HIP events on the launch stream, median of %d steps" % nev,
           "parity": {"bit_exact_vs_oracle": ok, "sampled_images": nidx}}
    if one:
        B2 = 65536
        _m2, ms2, nev2, k2, ok2, nidx2 = run(B2, 20)
        rq = 4000.0 * B2 / (ms2 * 1e-3) / 1e9
        rec["roofline"] = {"bound": "valu", "kernel": one["kernel"], "batch": B2, "ms": round(ms2, 5),
                           "achieved": round(rq, 1), "peak": roofline.REQUANT_PEAK_GBS, "unit": "GB/s of requantised int8",
                           "frac": round(rq / roofline.REQUANT_PEAK_GBS, 4),
                           "note": "measured at batch %d (16 steps per workgroup); at batch %d every CU runs ONE 16-image "
                                   "step, so that time is launch + one load/compute latency chain, not a rate" % (B2, B),
                           "hbm_GBps": round((m.input_elems + m.output_elems) * B2 / (ms2 * 1e-3) / 1e9, 1),
                           "hbm_frac": round((m.input_elems + m.output_elems) * B2 / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        rec["batch_%d" % B2] = {"value": round(B2 / (ms2 * 1e-3), 1), "unit": "inferences/s", "ms_per_step": round(ms2, 5),
                                "kernels": k2, "parity": {"bit_exact_vs_oracle": ok2, "sampled_images": nidx2}}
    else:  # operator-by-operator kernels (MF_NO_DWFC): the depthwise conv is 320 000 MAC / inference on v_dot4
        dw = next((k for k in kernels if k["kernel"].startswith("dw_c1")), kernels[0])
        tmacs = 320000.0 * B / (dw["ms"] * 1e-3) / 1e12 if dw["ms"] > 0 else 0.0
        rec["roofline"] = {"bound": "valu", "kernel": dw["kernel"], "achieved": round(tmacs, 2), "peak": DOT4_PEAK_TMACS,
                           "unit": "TMAC/s", "frac": round(tmacs / DOT4_PEAK_TMACS, 4), "ms": dw["ms"],
                           "note": "54 MAC per input byte: bounded by the v_dot4_i32_i8 issue rate (measured, scripts/ubench)"}
    rec["roofline"]["hbm_GBps_batch_%d" % B] = round(nbytes / (ms * 1e-3) / 1e9, 1)
    return rec


def fc4096_record(ctx, steps, warmup, wzp=0):
    """BASELINE config 5: FullyConnected 4096x4096x4096 through the model API: one step = one predict_inner
    over a [4096, 4096] int8 input (one dense int8 GEMM + fused requantize epilogue; with a non-zero weight
    zero point also the row-sum pre-pass of src/ops/fully_connected.rs:60-72)."""
    mf, _lib, torch = ctx["mf"], ctx["_lib"], ctx["torch"]
    from tools.make_fc_model import synthetic_fc
    O = ctx["checker"]  # the CPU restatement, imported by bench.py
    M = K = N = 4096
    blob = synthetic_fc(M, K, N, wzp=wzp, seed=5)
    m = mf.model(blob)
    m.prepare(1, device=ctx["local_rank"])
    L = _lib.lib()
    _lib.check(L.mf_model_set_stream(m._h, torch.cuda.current_stream().cuda_stream))
    g = torch.Generator(device="cuda").manual_seed(1234 + ctx["rank"])
    x = torch.randint(-128, 128, (M, K), dtype=torch.int8, device="cuda", generator=g)  # random operands
    y = torch.empty(M * N, dtype=torch.int8, device="cuda")
    step = lambda: _lib.check(L.mf_model_run_quantized(m._h, x.data_ptr(), 1, y.data_ptr(), _lib.MF_MEM_DEVICE))  # noqa: E731
    roofline.prewarm(torch, step)  # (the device idled while the host built and verified the model: benchlib/roofline.py prewarm)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ev = event_times(torch, step, max(20, steps))
    ms = median(ev)
    # ... and the same launch inside ONE event pair per 20 back-to-back steps (a per-step event pair adds its own packet processing
    # to a 65 us launch): reported beside the per-step figure
    nreg = max(20, steps)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    regs = []
    for _ in range(3):
        e0.record()
        for _ in range(nreg):
            step()
        e1.record()
        torch.cuda.synchronize()
        regs.append(e0.elapsed_time(e1) / nreg)
    ops = 2.0 * M * K * N
    tops = ops / (ms * 1e-3) / 1e12
    rows = sorted(set([0, 777, 4095] + list(range(5, M, 131))))[:40]
    om = O.Model(synthetic_fc(len(rows), K, N, wzp=wzp, seed=5))
    want = om.run_quantized(x[rows].cpu().numpy()).reshape(len(rows), N)
    ok = bool(np.array_equal(y.reshape(M, N)[rows].cpu().numpy(), want))
    crosscheck = None
    if wzp == 0:
        w_nk = torch.from_numpy(np.random.default_rng(5).integers(-128, 128, (N, K), dtype=np.int8)).cuda()  # = synthetic_fc's W
        crosscheck = int8_gemm_crosscheck(torch, x, w_nk)
        del w_nk
    return {
        "crosscheck": crosscheck,
        "metric": "int8 GEMM TOP/s, FullyConnected 4096x4096x4096 via predict_inner",
        "value": round(ops / (elapsed / steps) / 1e12, 1), "unit": "TOP/s",
        "steps": steps, "warmup": warmup, "ms_per_step": round(elapsed / steps * 1e3, 4),
        "config": {"workload": "FullyConnected int8 M=K=N=4096 (generated single-op .tflite), weight zero point %d, "
                               "uniform random int8 operands" % wzp},
        "roofline": {"bound": "mfma", "kernel": m.op(0)["kernel"], "achieved": round(tops, 1),
                     "peak": INT8_MFMA_PEAK_NOMINAL, "unit": "TOP/s", "frac": round(tops / INT8_MFMA_PEAK_NOMINAL, 4),
                     "peak_guide_floor": INT8_MFMA_PEAK_GUIDE_FLOOR,
                     "frac_of_guide_floor": round(tops / INT8_MFMA_PEAK_GUIDE_FLOOR, 4),
                     "traffic": pmc_traffic("fc_mfma", None)[0], "traffic_src": pmc_traffic("fc_mfma", None)[1],
                     "algorithmic_bytes": M * K + N * K + M * N,
                     "ms": round(ms, 4), "algorithmic_ops": ops,
                     "region": {"ms_per_step": round(median(regs), 4), "TOPs": round(ops / (median(regs) * 1e-3) / 1e12, 1),
                                "note": "3 regions of %d back-to-back steps, one event pair per region" % nreg},
                     "method": "HIP events on the launch stream, median of %d steps (whole predict_inner: the GEMM launch"
                               "%s)" % (len(ev), ", row sums formed in its prologue" if wzp else ""),
                     "peak_note": "5033 = 2 x the ~2.5 PF dense bf16 MFMA peak (nominal); 3944 = the guide's measured "
                                  "int8 floor (MI355X_MICROARCH.md)"},
        "parity": {"bit_exact_vs_oracle": ok, "sampled_rows": len(rows)},
    }


def int8_gemm_crosscheck(torch, x, w_nk, iters=20):
    """What does a vendor int8 GEMM sustain on this chip on the SAME random operands?  (SURVEY.md 7 allows the BLAS
    libraries as a cross-check; nothing here is linked into libmicroflow_amd.so.)  Tries torch._int_mm (hipBLASLt) and
    rocBLAS gemm_ex through ctypes; int8 x int8 -> int32, NT layout like the FullyConnected kernel (both K-contiguous)."""
    M, K = x.shape
    N = w_nk.shape[0]
    ops = 2.0 * M * K * N
    out = {"operands": "the step's own x [%d,%d] and W [%d,%d] (uniform random int8)" % (M, K, N, K), "results": {}}
    rows = [0, 1, M // 2, M - 1]
    want = x[rows].cpu().numpy().astype(np.int32) @ w_nk.cpu().numpy().astype(np.int32).T

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        return median(event_times(torch, fn, iters))

    try:
        wt = w_nk.t()  # [K, N] view, column-major = W's own memory
        c = torch._int_mm(x, wt)
        ok = bool(np.array_equal(c[rows].cpu().numpy(), want))
        ms = timed(lambda: torch._int_mm(x, wt))
        out["results"]["torch._int_mm"] = {"ms": round(ms, 4), "TOPs": round(ops / (ms * 1e-3) / 1e12, 1), "correct": ok,
                                           "epilogue": "none (int32 out, 4x the output bytes of the fused kernel)"}
    except Exception as e:  # noqa: BLE001
        out["results"]["torch._int_mm"] = {"error": str(e)[:200]}
    try:
        import ctypes
        rb = ctypes.CDLL("librocblas.so")
        h = ctypes.c_void_p()
        assert rb.rocblas_create_handle(ctypes.byref(h)) == 0
        rb.rocblas_set_stream(h, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        c32 = torch.empty((M, N), dtype=torch.int32, device="cuda")
        alpha, beta = ctypes.c_int32(1), ctypes.c_int32(0)
        I8, I32, OP_N, OP_T = 160, 162, 111, 112  # rocblas_datatype_i8_r / i32_r, rocblas_operation_none / transpose
        # column-major C^T [N, M] = W [N, K] (as A^T of a K x N column-major matrix) * X^T: A = W memory (K x N, lda K,
        # transposed), B = X memory (K x M, ldb K, not transposed), C memory = row-major [M, N]
        def gemm():
            return rb.rocblas_gemm_ex(h, OP_T, OP_N, N, M, K, ctypes.byref(alpha),
                                      ctypes.c_void_p(w_nk.data_ptr()), I8, K, ctypes.c_void_p(x.data_ptr()), I8, K,
                                      ctypes.byref(beta), ctypes.c_void_p(c32.data_ptr()), I32, N,
                                      ctypes.c_void_p(c32.data_ptr()), I32, N, I32, 0, 0, 0)
        st = gemm()
        torch.cuda.synchronize()
        if st != 0:
            raise RuntimeError("rocblas_gemm_ex status %d" % st)
        ok = bool(np.array_equal(c32[rows].cpu().numpy(), want))
        ms = timed(gemm)
        out["results"]["rocblas_gemm_ex"] = {"ms": round(ms, 4), "TOPs": round(ops / (ms * 1e-3) / 1e12, 1), "correct": ok,
                                             "epilogue": "none (int32 out)"}
        rb.rocblas_destroy_handle(h)
    except Exception as e:  # noqa: BLE001
        out["results"]["rocblas_gemm_ex"] = {"error": str(e)[:200]}
    best = [v["TOPs"] for v in out["results"].values() if v.get("correct")]
    out["best_TOPs"] = max(best) if best else None
    return out


def op_bytes_table(m, per_op, count):
    """per launch: algorithmic bytes (unique in + out), GB/s -- for the layer-wise (one kernel per operator) sub-records"""
    rows = []
    for i in range(m.num_ops):
        d = m.op(i)
        if not d["kernel"] or d["kernel"].startswith("(fused") or per_op[i] <= 0:
            continue
        nbytes = (int(np.prod(d["in_shape"])) + d["out_elems"]) * count
        rows.append({"op": i, "kind": d["name"], "kernel": d["kernel"], "ms": round(per_op[i], 4), "bytes": nbytes,
                     "GBps": round(nbytes / (per_op[i] * 1e-3) / 1e9, 1)})
    return rows


def kind_agg(rows, kind):
    ks = [k for k in rows if k["kind"] == kind]
    ms, by = sum(k["ms"] for k in ks), sum(k["bytes"] for k in ks)
    return {"kernels": len(ks), "ms": round(ms, 4), "GBps": round(by / (ms * 1e-3) / 1e9, 1) if ms > 0 else 0.0,
            "frac": round(by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ms > 0 else 0.0}


def runtime_geometry_record(ctx, table_layerwise):
    """Shapes outside person_detect's tables (the reference compiles for any shape: src/ops/depthwise_conv_2d.rs:28-49):
    (1) person_detect itself with the table kernels switched off (MF_NO_TABLE=1, a subprocess because routing is decided
    when an operator is created) against the table kernels' layer-wise numbers of this run -- same shapes, like for like;
    (2) generated person_detect-shaped models at other input sizes / widths (tools/tflite_writer.person_detect_like)."""
    mf, _lib, torch, synth_i8, SEED = ctx["mf"], ctx["_lib"], ctx["torch"], ctx["synth_i8"], ctx["SEED"]
    rec = {"note": "run-time-geometry kernels (k_rt.hip: dw3x3_rt, pw_rt, conv_rows_lds); GB/s = algorithmic bytes / median launch time"}
    try:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "time_kernels.py"), "20", "layerwise", "--json"],
                             env=dict(os.environ, MF_DEV="1", MF_NO_TABLE="1"), capture_output=True, text=True, timeout=300)
        j = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
        tab = {k["op"]: k for k in table_layerwise}
        rows = []
        for k in j["kernels"]:
            t = tab.get(k["op"])
            if t and k["kernel"] != t["kernel"]:
                rows.append({"op": k["op"], "kind": t["kind"], "kernel": k["kernel"], "ms": round(k["ms"], 4), "bytes": t["bytes"],
                             "GBps": round(t["bytes"] / (k["ms"] * 1e-3) / 1e9, 1), "table_kernel": t["kernel"], "table_ms": t["ms"],
                             "slowdown": round(k["ms"] / t["ms"], 3) if t["ms"] > 0 else None})
        cmp_ = {"layerwise_ms": round(j["ms_per_step"], 4), "kernels": rows}
        for kind in ("depthwise_conv_2d", "conv_2d"):
            ks = [r for r in rows if r["kind"] == kind]
            if ks:
                ms, tms, by = sum(r["ms"] for r in ks), sum(r["table_ms"] for r in ks), sum(r["bytes"] for r in ks)
                cmp_[kind] = {"kernels": len(ks), "ms": round(ms, 4), "GBps": round(by / (ms * 1e-3) / 1e9, 1), "table_ms": round(tms, 4),
                              "table_GBps": round(by / (tms * 1e-3) / 1e9, 1), "slowdown": round(ms / tms, 3)}
        rec["person_detect_without_tables"] = cmp_
    except Exception as e:  # noqa: BLE001
        rec["person_detect_without_tables"] = {"error": str(e)[:300]}
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import tflite_writer as tw
    O = ctx["checker"]  # the CPU restatement, imported by bench.py
    L = _lib.lib()
    models = {}
    for side, width in ((128, 1.0), (64, 1.0), (96, 0.5)):
        blob = tw.person_detect_like(np.random.default_rng(side), side, width)
        m, om = mf.Model(blob, autotune=True), O.Model(blob)  # (opt-in: the chain candidates are timed at creation, mf_model_set_autotune)
        B = int(65536 * 96 * 96 / (side * side))
        m.prepare(B, device=ctx["local_rank"])
        _lib.check(L.mf_model_set_stream(m._h, torch.cuda.current_stream().cuda_stream))
        x = synth_i8(SEED + 6, 0, B * m.input_elems)
        y = torch.empty(B * m.output_elems, dtype=torch.int8, device="cuda")
        fused_ms, _ = m.time_device(x, y, B, warmup=2, iters=10, per_op=False)
        fused_kernels = sorted({m.op(i)["kernel"].split("<")[0] for i in range(m.num_ops) if m.op(i)["kernel"] and not m.op(i)["kernel"].startswith("(fused")})
        chains = [m.op(i)["kernel"] for i in range(m.num_ops) if m.op(i)["kernel"].startswith("chain_rt<")]
        idx = [0, 1, B // 2, B - 1]
        ok = bool(np.array_equal(y.reshape(B, -1)[idx].cpu().numpy(), om.run_quantized_batch(x.reshape(B, -1)[idx].cpu().numpy())))
        m.set_fusion(False)
        lw_ms, per = m.time_device(x, y, B, warmup=1, iters=10)
        rows = op_bytes_table(m, per, B)
        generic = [r["kernel"] for r in rows if r["kernel"].endswith("_generic")]
        models["%dx%d_width%s" % (side, side, width)] = {
            "batch": B, "value": round(B / (fused_ms * 1e-3), 1), "unit": "inferences/s", "ms_per_step": round(fused_ms, 4),
            "layerwise_ms": round(lw_ms, 4), "depthwise": kind_agg(rows, "depthwise_conv_2d"), "conv_2d": kind_agg(rows, "conv_2d"),
            "kernels_used": fused_kernels, "chains": chains, "layerwise_kernels": sorted({r["kernel"].split("<")[0] for r in rows}),
            "speedup_vs_layerwise": round(lw_ms / fused_ms, 3), "generic_kernels": generic,
            "parity": {"bit_exact_vs_oracle": ok, "sampled_images": len(idx)}}
        del m, x, y
        torch.cuda.empty_cache()
    rec["generated_models"] = models
    return rec


def general_conv_record(ctx):
    """Conv2D beyond 1x1 (src/ops/conv_2d.rs:28-108 is generic in filter size): a colour MobileNet stem (3x3x3 -> 16, stride 2,
    batch 65536: conv_rows_lds), a ResNet-8-style 3x3 block (16 -> 16 on 32x32: conv_mm_rt, the MFMA product over
    K = KH KW C) and a 64 -> 64 one on 8x8, each against the shape-generic kernel on a slice of its batch."""
    mf, torch = ctx["mf"], ctx["torch"]
    O = ctx["checker"]  # the CPU restatement, imported by bench.py
    rng = np.random.default_rng(3)
    out = {}
    for name, (H, W, C, N, K, S, B) in {"stem_96x96x3_to_16_s2": (96, 96, 3, 16, 3, 2, 65536), "block_32x32x16_to_16": (32, 32, 16, 16, 3, 1, 16384),
                                        "block_8x8x64_to_64": (8, 8, 64, 64, 3, 1, 65536)}.items():
        OH, OW = -(-H // S), -(-W // S)
        f = rng.integers(-128, 128, (N, K, K, C)).astype(np.int8)
        c0 = rng.uniform(-30, 30, N).astype(np.float32)
        c1 = (rng.uniform(0.5, 1.5, N) * 40.0 / (5476.0 * np.sqrt(K * K * C))).astype(np.float32)
        opts = mf.ops.Conv2DOptions(mf.FusedActivation(3), mf.TensorViewPadding.SAME, (S, S))
        op = mf.ops.prepare_conv_2d((H, W, C), f, np.zeros(N, np.int8), -128, 0.0235294122, -128, opts, (c0, c1), (OH, OW))
        x = torch.randint(-128, 128, (B, H, W, C), dtype=torch.int8, device="cuda")
        y = op(x)
        ms = median(event_times(torch, lambda: op(x), 12))
        idx = [0, B // 2, B - 1]
        want = np.stack([O.conv_2d(x[i].cpu().numpy(), f, np.zeros(N, np.int8), -128, 0.0235294122, -128, 3, 0, (S, S), (OH, OW), c0, c1) for i in idx])
        ok = bool(np.array_equal(y[idx].cpu().numpy(), want))
        kernel = op.kernel
        nb = min(B, 512)
        op.set_generic(True)
        gms = median(event_times(torch, lambda: op(x[:nb]), 3)) * (B / nb)
        nbytes, macs = B * (H * W * C + OH * OW * N), float(B) * OH * OW * N * K * K * C
        out[name] = {"kernel": kernel, "batch": B, "ms": round(ms, 4), "GBps": round(nbytes / (ms * 1e-3) / 1e9, 1),
                     "hbm_frac": round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "TMACps": round(macs / (ms * 1e-3) / 1e12, 2),
                     "generic_ms_scaled": round(gms, 2), "speedup_vs_generic": round(gms / ms, 1), "bit_exact_vs_oracle": ok}
        del x, y
        torch.cuda.empty_cache()
    return out


def general_depthwise_record(ctx):
    """DepthwiseConv2D beyond 3x3 SAME (src/ops/depthwise_conv_2d.rs:28-49 is generic in filter, stride and padding): a 5x5 stride-1 layer
    on 24x24x32 and a 3x3 VALID one, both conv_mm_rt's depthwise mode (taps of a 16-channel group on the matrix pipe), each against the
    shape-generic kernel on a slice of its batch."""
    mf, torch = ctx["mf"], ctx["torch"]
    O = ctx["checker"]  # the CPU restatement, imported by bench.py
    rng = np.random.default_rng(4)
    out = {}
    for name, (H, W, C, KH, KW, S, pad, B) in {"dw5x5_24x24x32_s1": (24, 24, 32, 5, 5, 1, 0, 65536), "dw3x3_valid_24x24x64": (24, 24, 64, 3, 3, 1, 1, 32768)}.items():
        OH, OW = (-(-H // S), -(-W // S)) if pad == 0 else ((H - KH) // S + 1, (W - KW) // S + 1)
        w = rng.integers(-128, 128, (KH, KW, C)).astype(np.int8)
        c0 = rng.uniform(-30, 30, C).astype(np.float32)
        c1 = (rng.uniform(0.5, 1.5, C) * 40.0 / (5476.0 * np.sqrt(KH * KW))).astype(np.float32)
        opts = mf.ops.DepthwiseConv2DOptions(mf.FusedActivation(3), mf.TensorViewPadding(pad), (S, S))
        op = mf.ops.prepare_depthwise_conv_2d((H, W, C), w, np.zeros(C, np.int8), -128, 0.0235294122, -128, opts, (c0, c1), (OH, OW))
        x = torch.randint(-128, 128, (B, H, W, C), dtype=torch.int8, device="cuda")
        y = op(x)
        ms = median(event_times(torch, lambda: op(x), 12))
        idx = [0, B // 2, B - 1]
        want = np.stack([O.depthwise_conv_2d(x[i].cpu().numpy(), w, np.zeros(C, np.int8), -128, 0.0235294122, -128, 3, pad, (S, S), (OH, OW), c0, c1)
                         for i in idx])
        ok = bool(np.array_equal(y[idx].cpu().numpy(), want))
        kernel = op.kernel
        nb = min(B, 512)
        op.set_generic(True)
        gms = median(event_times(torch, lambda: op(x[:nb]), 3)) * (B / nb)
        nbytes = B * (H * W * C + OH * OW * C)
        out[name] = {"kernel": kernel, "batch": B, "ms": round(ms, 4), "GBps": round(nbytes / (ms * 1e-3) / 1e9, 1),
                     "hbm_frac": round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                     "generic_ms_scaled": round(gms, 2), "speedup_vs_generic": round(gms / ms, 1), "bit_exact_vs_oracle": ok}
        del x, y
        torch.cuda.empty_cache()
    return out


def generic_fallback_record(ctx, m, x, count, fast_ms):
    """The cliff: person_detect on the byte-wise shape-generic kernels (mf_model_set_generic), on a slice of the batch."""
    torch = ctx["torch"]
    n = min(count, 2048)
    y = torch.empty(n * m.output_elems, dtype=torch.int8, device="cuda")
    m.set_generic(True)
    try:
        ms, _ = m.time_device(x[: n * m.input_elems], y, n, warmup=1, iters=3, per_op=False)
    finally:
        m.set_generic(False)
    fast, _ = m.time_device(x[: n * m.input_elems], y, n, warmup=2, iters=10, per_op=False)
    return {"batch": n, "ms_per_step": round(ms, 3), "value": round(n / (ms * 1e-3), 1), "unit": "inferences/s",
            "fast_path_ms_same_batch": round(fast, 4), "slowdown": round(ms / fast, 1),
            "note": "every operator on its `*_generic` kernel (one thread per output element, byte loads): what a shape with "
                    "no fast kernel costs; the fused step of the full batch takes %.3f ms" % fast_ms}
