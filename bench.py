#!/usr/bin/env python3
"""bench.py -- throughput of the quantized-operator hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
        N > 1 launches N ranks itself (torch.distributed.run, one rank per GPU over RCCL); the same
        file also runs under an external launcher (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the env).
    python bench.py --workload speech|fc4096        # one of the other BASELINE configs as the headline

Workload (BASELINE.json): person_detect.tflite, int8, 65536 independent inferences per GPU (configs[2];
configs[3] = the same shard on each of 8 GPUs: weak scaling, no data-path collective).  One "step" = one
pass of predict_inner (31 operators: 14 DepthwiseConv2D, 14 Conv2D, AveragePool2D, Reshape, Softmax) over
the GPU's batch, int8 in -> int8 out, with the synthetic input batch resident in HBM when the timed region
starts.  `value` follows the driver's contract: W untimed steps, then exactly K steps between two
barrier + synchronize fences, wall clock, max over ranks.  Beside it the line carries
  event_median : the same step timed per iteration with HIP events on the launch stream, median
  roofline     : the dominant kernel -- algorithmic bytes / its median HIP-event duration vs the 8 TB/s HBM peak
  kernels      : the same figure for every launch of the step; layerwise: with the fusions switched off
  cpu_baseline : the reference-faithful C restatement (oracle/, "port", gcc -O3) on this box's host cores
  parity       : sampled bit-exact comparison of the GPU outputs with that oracle
  speech       : BASELINE config 2 (speech.tflite, batch 4096) -- VALU-bound, priced against the v_dot4 rate
  fc4096       : BASELINE config 5 (FullyConnected 4096^3 through predict_inner) -- int8 MFMA roofline, with
                 weight zero point 0 and != 0
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# the measurement legs live in benchlib/ (this file: the launch path, the timed region, the parity legs and the CPU baseline)
from benchlib import roofline as rl  # noqa: E402
from benchlib.compact import COMPACT_CAP, DETAILS_FILE, compact_record  # noqa: E402,F401  (tests import compact_record from here)
from benchlib.records import (fc4096_record, general_conv_record, general_depthwise_record, generic_fallback_record,  # noqa: E402,F401
                              runtime_geometry_record, speech_record)
from benchlib.roofline import (HBM_PEAK_GBS, event_times, measure_requant_ceiling, median, pmc_traffic, requant_peak,  # noqa: E402,F401
                               rocprof_name, sq_counters)

WORKLOADS = {
    # name: (model file, BASELINE config index used as stream id, per-GPU batch)
    "person_detect": ("person_detect.tflite", 3, 65536),
    "speech": ("speech.tflite", 2, 4096),
    # BASELINE config 5: generated single-op FullyConnected model, one [4096,4096] input per step
    "fc4096": (None, 5, 1),
}
PARITY_NOTE = ("bit-exact vs the restated CPU oracle (oracle/mf_oracle.c, pinned to every reference KAT; "
               "softmax's expf is pinned at the reference's 9 points only)")


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args):
    """`python bench.py --gpus N` without an external launcher: one rank per GPU under
    torch.distributed.run, rendezvous on 127.0.0.1."""
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < args.gpus and not (args.share_device and have >= 1):
        raise SystemExit("bench.py --gpus %d: only %d GPU device(s) visible on this box" % (args.gpus, have))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="person_detect", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the BASELINE size)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-fed", action="store_true", help="skip the PCIe-inclusive and f32 legs")
    ap.add_argument("--no-extra", action="store_true", help="skip the speech / fc4096 sub-records")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend of the N > 1 path (nccl = RCCL; gloo: tests on a 1-GPU box)")
    ap.add_argument("--dist", action="store_true",
                    help="with --gpus 1: still initialise the process group (world size 1) and run the barrier, the max-over-ranks "
                         "all_reduce and the checksum all_gather -- RCCL's library load, device binding and teardown on a 1-GPU box")
    ap.add_argument("--share-device", action="store_true",
                    help="every rank uses GPU 0 (only with --backend gloo: exercises the real multi-rank launch path on one GPU)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    # (the host driver only does dmabuf IPC; RCCL's peer mapping fails with hipIpcGetMemHandle otherwise -- also under an external
    # torchrun that did not export it.  Must be in the environment before HIP is initialised.)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if args.share_device:
        if args.backend != "gloo":
            raise SystemExit("--share-device needs --backend gloo (RCCL refuses two ranks on one device)")
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d has no GPU (only %d device(s) visible)" % (rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    if rank != 0:
        # Only rank 0 speaks on stdout, and its JSON line must be the LAST thing there.  Libraries write to the C-level stdout behind
        # Python's back (RCCL prints "Librccl path : ..." through a buffered printf that surfaces when the process exits -- seen
        # the first time RCCL ran here, round 5): every other rank's stdout goes to stderr from the start.
        os.dup2(2, 1)
    coll_dev = "cuda" if args.backend == "nccl" else "cpu"  # where the three tiny collectives' tensors live
    use_dist = world > 1 or args.dist  # (--dist: the same collectives over a world of one)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", str(free_port()))
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    import microflow_rs_amd as mf
    from microflow_rs_amd import _lib
    from microflow_rs_amd.model import checksum_i8, synth_i8
    from microflow_rs_amd.shard import gather_checksums, max_over_ranks, shard_range
    from microflow_rs_amd.synth import SEED

    # (the CPU restatement: imported HERE and nowhere else outside tests/ and smoke(); the benchlib legs that spot-check their outputs
    # get it as ctx["checker"], the CPU baseline and the step's parity legs below use it directly)
    from oracle import oracle as O
    ctx = dict(args=args, mf=mf, _lib=_lib, torch=torch, dist=dist, world=world, rank=rank, local_rank=local_rank, use_dist=use_dist,
               synth_i8=synth_i8, checksum_i8=checksum_i8, SEED=SEED, checker=O)
    fname, cfg, base_batch = WORKLOADS[args.workload]
    B = args.batch or base_batch
    if args.workload == "fc4096":
        result = headline_from_sub(args, fc4096_record(ctx, args.steps, args.warmup, wzp=0), world)
        finish(ctx, result)
        return
    m = mf.model(os.path.join(ROOT, "models", fname))
    m.prepare(B, device=local_rank)
    L = _lib.lib()
    stream = torch.cuda.current_stream()
    _lib.check(L.mf_model_set_stream(m._h, stream.cuda_stream))

    # this rank's shard of the global synthetic stream, generated directly in HBM
    first, count = shard_range(B * world, rank, world)
    x = synth_i8(SEED + cfg, first * m.input_elems, count * m.input_elems)
    y = torch.empty(count * m.output_elems, dtype=torch.int8, device="cuda")

    def step():
        _lib.check(L.mf_model_run_quantized(m._h, x.data_ptr(), count, y.data_ptr(), _lib.MF_MEM_DEVICE))

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # untimed launches until the device's clock has recovered from the idle period of model preparation (benchlib/roofline.py prewarm:
    # without them the W = 5 warm-up steps the driver asks for end inside the ramp and the K timed steps read 2.3 % slow), then the W
    # warm-up steps, then exactly K timed steps
    prewarm_steps = rl.prewarm(torch, step)
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        elapsed = max_over_ranks(dist, elapsed, device=coll_dev)
    ms_per_step = elapsed / args.steps * 1e3
    value = B * world / (elapsed / args.steps)

    # the same step, one HIP event pair per iteration on the launch stream: median
    ev = event_times(torch, step, max(20, args.steps))
    ev_med = median(ev)
    if use_dist:
        ev_med = max_over_ranks(dist, ev_med, device=coll_dev)

    # output checksums of every shard (outside the timed region; RCCL all_gather of 8 bytes)
    ck = checksum_i8(y)
    cks = [ck]
    if use_dist:
        cks = gather_checksums(dist, ck, device=coll_dev)
        # The distributed part of the run ends HERE: everything below (per-kernel tables, parity, CPU baseline, the
        # other workloads) is rank 0's alone and takes tens of seconds, so the process group is torn down first
        # instead of leaving the other ranks parked in a collective.
        dist.barrier()
        dist.destroy_process_group()
        ctx["group_closed"] = True

    result = None
    if rank == 0:
        measure_requant_ceiling()
        # ---- per-kernel HIP-event timing on the launch stream (median over the iterations) ----
        iters = max(5, min(args.steps, 20))

        def kernel_table():
            avg_ms, per_op = m.time_device(x, y, count, warmup=1, iters=iters)
            descs = [m.op(i) for i in range(m.num_ops)]
            rows = []
            for i, d in enumerate(descs):
                if not d["kernel"] or d["kernel"].startswith("(fused"):
                    continue
                in_elems = int(np.prod(d["in_shape"]))
                out_elems, kind = d["out_elems"], d["name"]
                # a fused group reads its first operator's input and writes its last operator's output
                last = i
                for j in range(i + 1, len(descs)):
                    if descs[j]["kernel"].startswith("(fused"):
                        last = j
                    elif descs[j]["kernel"]:
                        break
                if last != i:
                    out_elems = descs[last]["out_elems"]
                    kind = "+".join(descs[j]["name"] for j in range(i, last + 1) if descs[j]["name"] != "reshape")
                nbytes = (in_elems + out_elems) * count  # algorithmic: unique in + out bytes
                gbs = nbytes / (per_op[i] * 1e-3) / 1e9 if per_op[i] > 0 else 0.0
                # every int8 tensor the launch produces, on chip or not, goes through the reference's f32 epilogue
                rq = sum(descs[j]["out_elems"] for j in range(i, last + 1) if descs[j]["name"] not in ("reshape",)) * count
                rq_gbs = rq / (per_op[i] * 1e-3) / 1e9 if per_op[i] > 0 else 0.0
                nops_in_group = sum(1 for j in range(i, last + 1) if descs[j]["name"] != "reshape")
                # the launch's own requantisation form (a fused launch runs the weakest form any of its operators needs) and
                # that form's measured ceiling
                mode = m.op_epilogue_mode(i)
                peak = requant_peak(mode)
                # the binding roof is the one that gives the longer time floor: HBM for the algorithmic bytes, or the
                # VALU for the bytes that go through the reference's f32 requantisation
                bound = "valu" if (peak and rq / peak > nbytes / HBM_PEAK_GBS) else "hbm"
                if d["kernel"].startswith(("quad_rr", "quad_mm")):
                    kind = "quad(2 pairs)"   # two depthwise+pointwise pairs in one launch (k_quad.hip, k_quad_mm.hip)
                elif d["kernel"].startswith("penta_rr"):
                    kind = "penta(stem + 2 pairs)"   # ... with the network's first operator in front of them
                elif nops_in_group > 3:
                    kind = "stage(%d ops)" % nops_in_group
                rows.append({"op": i, "kind": kind,
                             "kernel": d["kernel"], "ms": round(per_op[i], 4), "bound": bound,
                             "bytes": nbytes, "GBps": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4),
                             "hbm_frac": round(gbs / HBM_PEAK_GBS, 4),
                             "requant_bytes": rq, "requant_GBps": round(rq_gbs, 1),
                             "epilogue_mode": mode, "requant_peak_GBps": peak,
                             "requant_frac": round(rq_gbs / peak, 4) if peak else None,
                             "valu_frac": round(rq_gbs / peak, 4) if peak else None, "sq": sq_counters(d["kernel"])})
            return avg_ms, rows

        def agg(rows, kind):
            ks = [k for k in rows if k["kind"] == kind]
            ms = sum(k["ms"] for k in ks)
            by = sum(k["bytes"] for k in ks)
            gbs = by / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            return {"kernels": len(ks), "ms": round(ms, 4), "bytes": by, "GBps": round(gbs, 1),
                    "frac": round(gbs / HBM_PEAK_GBS, 4)}

        avg_ms, kernels = kernel_table()
        # `roofline` = the LONGEST launch of the step, whatever bounds it, with both fractions: algorithmic bytes / time
        # against the 8 TB/s HBM peak (`frac` = `hbm_frac`) and requantised bytes / time against the requantisation
        # ceiling measured in this run (`valu_frac`); `bound` names the roof with the longer time floor.
        dom = max(kernels, key=lambda k: k["ms"])
        # HBM bytes per launch from the committed rocprofv3 PMC passes of this same command
        # (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE; scripts/pmc_summary.py), if the batch matches
        traffic, traffic_src = pmc_traffic(dom["kernel"], count)
        # `bound` names the roof that achieved / peak / frac are stated against -- the contract's roofline of this byte-moving path is
        # HBM --; `binding_roof` says which of the two roofs (HBM, or the builder-defined requantisation ceiling) gives this launch
        # the longer time floor.
        roofline = {"bound": "hbm", "binding_roof": dom["bound"], "kernel": dom["kernel"], "op": dom["op"], "kind": dom["kind"], "ms": dom["ms"],
                    "achieved": dom["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dom["frac"],
                    "hbm_frac": dom["hbm_frac"], "valu_frac": dom["valu_frac"],
                    "requant_GBps": dom["requant_GBps"], "requant_peak_GBps": dom["requant_peak_GBps"], "epilogue_mode": dom["epilogue_mode"],
                    "valu_inst_per_clk_per_simd": (dom["sq"] or {}).get("valu_inst_per_clk_per_simd"),
                    "mfma_busy_frac": (dom["sq"] or {}).get("mfma_busy_frac"),
                    "issue_busy_frac": (dom["sq"] or {}).get("issue_busy_frac"),
                    "sq_src": (dom["sq"] or {}).get("source"),
                    "algorithmic_bytes": dom["bytes"], "requant_bytes": dom["requant_bytes"],
                    "traffic": traffic, "traffic_src": traffic_src, "traffic_source": traffic_src,
                    "rocprof_name": rocprof_name(dom["kernel"], dom["epilogue_mode"], m.dtype == np.uint8),
                    "requant_ceiling_src": (rl.REQUANT_CEILING or {}).get("used"),
                    "method": "HIP events on the launch stream, median of %d launches" % iters,
                    "note": "the longest launch of the step.  achieved / peak / frac: algorithmic bytes per launch / its "
                            "duration vs the 8 TB/s HBM peak; valu_frac: every int8 byte the launch requantises (on chip or "
                            "not) / its duration vs the ceiling, measured in this run, of the requantisation form the launch runs "
                            "(`epilogue_mode`, `requant_ceiling`, `requant_ceiling_src`); traffic, valu_inst_per_clk_per_simd and mfma_busy_frac (the "
                            "share of the launch's cycles its matrix pipes were busy) and issue_busy_frac (= mfma_busy_frac + 2.75 clocks x VALU instructions per "
                            "clock: a SIMD's matrix pipe and VALU do not overlap for this instruction mix, scripts/ubench/mfma_valu_overlap.hip, "
                            "so that sum is the share of the launch's cycles its SIMDs were issuing -- the roof this launch is under, DESIGN 4.4f) are "
                            "REPLAYED from committed counter passes (`traffic_src`, `sq_src` say which, and whether they are stale); "
                            "rocprof_name = the kernel's name in profiles/*kernel_stats.csv; "
                            "`bound` = the roof with the longer time floor"}
        step_bytes = sum(k["bytes"] for k in kernels)
        step_rq = sum(k["requant_bytes"] for k in kernels)
        have_ceiling = all(k["requant_peak_GBps"] for k in kernels)  # (None: a launch's form was not measured in this run)
        floor_ms = sum(max(k["bytes"] / HBM_PEAK_GBS, k["requant_bytes"] / k["requant_peak_GBps"] if k["requant_peak_GBps"] else 0.0)
                       for k in kernels) / 1e6
        whole_step = {"ms": round(ev_med, 4), "launches": len(kernels),
                      "algorithmic_bytes": step_bytes, "GBps": round(step_bytes / (ev_med * 1e-3) / 1e9, 1),
                      "frac": round(step_bytes / (ev_med * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                      "hbm_frac": round(step_bytes / (ev_med * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                      "requant_bytes": step_rq, "requant_GBps": round(step_rq / (ev_med * 1e-3) / 1e9, 1),
                      "valu_frac": (round(sum(k["requant_bytes"] / k["requant_peak_GBps"] for k in kernels) / 1e6 / ev_med, 4)
                                    if have_ceiling else None),
                      "roof_floor_ms": round(floor_ms, 4) if have_ceiling else None,
                      "frac_of_roof_floor": round(floor_ms / ev_med, 4) if have_ceiling else None,
                      # fusing launches removes algorithmic bytes, so hbm_frac falls as the step gets faster; for comparison
                      # with earlier rounds: the bytes of round 2's ten launches (pairs + stage + tail) over this step's time
                      "hbm_frac_at_round2_bytes": (round(253442 * count / (ev_med * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                                                   if args.workload == "person_detect" else None),
                      "note": "all launches of the timed step: their algorithmic bytes / the step's median time vs 8 TB/s, their "
                              "requantised bytes / the same time vs the measured ceiling; roof_floor_ms = sum over launches of "
                              "max(bytes / 8 TB/s, requantised bytes / ceiling)"}

        # the same step with the fusions switched off: one kernel per reference operator
        # (the layer-wise DepthwiseConv2D / Conv2D roofline figures of BASELINE.json's targets)
        m.set_fusion(False)
        lw_ms, lw_kernels = kernel_table()
        m.set_fusion(True)
        layerwise = {"ms_per_step": round(lw_ms, 4), "depthwise": agg(lw_kernels, "depthwise_conv_2d"),
                     "conv_2d": agg(lw_kernels, "conv_2d"), "kernels": lw_kernels}

        # ---- parity: sampled bit-exact comparison with the CPU oracle ----
        om = O.Model(os.path.join(ROOT, "models", fname))
        idx = sorted(set([0, 1, count // 3, count // 2, count - 2, count - 1] + list(range(7, count, max(1, count // 42)))))[:48]
        xs = x.reshape(count, -1)[idx].cpu().numpy()
        ys = y.reshape(count, -1)[idx].cpu().numpy()
        parity_ok = bool(np.array_equal(ys, om.run_quantized_batch(xs)))
        # Uniform-noise images drive this network into nearly the same output for every image, which makes the
        # final 2 bytes a weak witness of the late layers.  Structured images (constant levels, ramps, checkerboards,
        # blobs) spread the outputs over the whole range: they go through the same handle as one extra small batch.
        structured_ok, n_struct, n_distinct = True, 0, 0
        if m.input_elems == 96 * 96:
            from microflow_rs_amd.synth import structured_images
            xs2 = structured_images(96)
            imgs = xs2
            want2 = om.run_quantized_batch(xs2)
            got2 = m.run_quantized(torch.from_numpy(xs2).cuda().reshape((len(imgs),) + m.input_shape)).reshape(len(imgs), -1).cpu().numpy()
            structured_ok = bool(np.array_equal(got2, want2))
            n_struct, n_distinct = len(imgs), len({tuple(r) for r in want2.tolist()})
            parity_ok = parity_ok and structured_ok

        cpu = cpu_mt = None
        if not args.no_cpu_baseline:
            cpu = cpu_baseline(om, x.reshape(count, -1), args.cpu_seconds)
            cpu_mt = cpu_baseline_all_cores(om, x.reshape(count, -1))

        # ---- PCIe-inclusive rate (never `value`): the same batch fed from pinned host memory,
        # H2D of the inputs and D2H of the outputs inside the timed region (MF_MEM_HOST) ----
        host_fed = None
        if not args.no_host_fed:
            xh = torch.empty(x.shape, dtype=torch.int8).pin_memory()
            xh.copy_(x)
            yh = torch.empty(y.shape, dtype=torch.int8).pin_memory()
            run_host = lambda: _lib.check(L.mf_model_run_quantized(  # noqa: E731
                m._h, xh.data_ptr(), count, yh.data_ptr(), _lib.MF_MEM_HOST))
            run_host()
            t0 = time.perf_counter()
            for _ in range(3):
                run_host()
            dt = (time.perf_counter() - t0) / 3
            host_fed = {"value": round(count / dt, 1), "unit": "inferences/s", "ms_per_step": round(dt * 1e3, 3),
                        "h2d_bytes": int(x.numel()), "d2h_bytes": int(y.numel()),
                        "bit_exact_vs_device_path": bool(torch.equal(yh, y.cpu())),
                        "note": "pinned host buffers through mf_model_run_quantized(MF_MEM_HOST): the batch "
                                "is cut into ~64 MB chunks whose H2D copies overlap the previous chunk's compute"}
            del xh, yh

        # ---- the f32 entry point (M::predict): quantize -> ops -> dequantize, device-resident ----
        predict_f32 = None
        if not args.no_host_fed:
            xf = (x.reshape(count, -1).float() - float(m.input_zero_point)) * float(m.input_scale)
            yf = torch.empty((count, m.output_elems), dtype=torch.float32, device="cuda")
            run_f = lambda: _lib.check(L.mf_model_predict(  # noqa: E731
                m._h, xf.data_ptr(), count, yf.data_ptr(), _lib.MF_MEM_DEVICE))
            for _ in range(3):
                run_f()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                run_f()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 20    # (wall clock over 20 back-to-back calls, like the headline)
            yq = (yf / float(m.output_scale) + float(m.output_zero_point)).round().to(torch.int8).reshape(-1)
            predict_f32 = {"value": round(count / dt, 1), "unit": "inferences/s", "ms_per_step": round(dt * 1e3, 3), "iterations": 20,
                           "input_bytes": int(xf.numel() * 4), "same_outputs_as_int8_path": bool(torch.equal(yq, y)),
                           "note": "f32 in HBM -> predict (boundary quantize inside the first launch where the model starts "
                                   "with a stem it has an f32 instance for) -> dequantize kernel"}
            del xf, yf

        generic_fb = rt_rec = None
        if args.workload == "person_detect" and not args.no_extra:
            generic_fb = generic_fallback_record(ctx, m, x, count, ev_med)
            rt_rec = runtime_geometry_record(ctx, lw_kernels)
            rt_rec["general_conv"] = general_conv_record(ctx)
            rt_rec["general_depthwise"] = general_depthwise_record(ctx)

        result = {
            "metric": "inferences/sec (int8) for %s" % fname, "value": round(value, 1),
            "unit": "inferences/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
            "prewarm_steps": prewarm_steps,  # untimed launches in front of the W warm-up steps (benchlib/roofline.py prewarm: 60 ms, clock ramp)
            "vs_baseline": None, "dtype": "i8", "data": "synthetic",
            "config": {"workload": "%s batch=%d per GPU, predict_inner int8->int8, inputs resident in HBM"
                                   % (fname, B), "per_gpu_batch": B, "global_batch": B * world,
                       "parallelism": "batch shard x%d, no data-path collective" % world,
                       "backend": (args.backend if (world > 1 or args.dist) else None),
                       "shards": [list(shard_range(B * world, r, world)) for r in range(world)]},
            "roofline": roofline,
            "requant_ceiling": rl.REQUANT_CEILING,
            "event_median": {"ms_per_step": round(ev_med, 4), "value": round(B * world / (ev_med * 1e-3), 1),
                             "iterations": len(ev), "min_ms": round(min(ev), 4), "max_ms": round(max(ev), 4),
                             "note": "HIP event pair per step on the launch stream, median; max over ranks"},
            "whole_step": whole_step,
            "fused_dwpw": agg(kernels, "depthwise_conv_2d+conv_2d"),
            "fused_quad": agg(kernels, "quad(2 pairs)"),
            "fused_penta": agg(kernels, "penta(stem + 2 pairs)"),
            "depthwise": layerwise["depthwise"], "conv_2d": layerwise["conv_2d"],
            "event_ms_per_step": round(avg_ms, 4),
            "kernels": kernels,
            "layerwise": layerwise,
            "cpu_baseline": cpu,
            "cpu_baseline_all_cores": cpu_mt,
            "host_fed": host_fed,
            "predict_f32": predict_f32,
            "generic_fallback": generic_fb,
            "runtime_geometry": rt_rec,
            "parity": {"bit_exact_vs_oracle": parity_ok, "what": PARITY_NOTE, "sampled_images": len(idx),
                       "structured_images": n_struct, "structured_distinct_outputs": n_distinct,
                       "structured_bit_exact": structured_ok,
                       "output_checksums": ["%016x" % c for c in cks]},
        }
        x = y = m = None
        torch.cuda.empty_cache()
        if args.workload == "person_detect" and not args.no_extra:
            # the other single-GPU BASELINE configs, driver-visible in the same line
            result["speech"] = speech_record(ctx)
            result["fc4096"] = fc4096_record(ctx, 10, 2, wzp=0)
            result["fc4096_wzp"] = fc4096_record(ctx, 10, 2, wzp=-3)
        if not parity_ok:
            result["value"] = 0.0
            result["error"] = "GPU outputs differ from the CPU oracle: number withheld"
    finish(ctx, result)


def finish(ctx, result):
    dist, world, rank = ctx["dist"], ctx["world"], ctx["rank"]
    if (world > 1 or ctx.get("use_dist")) and not ctx.get("group_closed"):
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # the full record goes to a file next to bench.py (and to gpurun_out/ when that exists); stdout carries ONE
        # compact line, the last thing printed
        for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
            if os.path.isdir(d):
                try:
                    with open(os.path.join(d, DETAILS_FILE), "w") as f:
                        json.dump(result, f, indent=1)
                except OSError as e:
                    print("bench.py: cannot write %s: %s" % (os.path.join(d, DETAILS_FILE), e), file=sys.stderr)
        _, line = compact_record(result)
        sys.stderr.flush()
        try:  # whatever native libraries have buffered for stdout comes out BEFORE the line ...
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(line, flush=True)
        os.dup2(2, 1)  # ... and whatever they still print at exit (RCCL does) goes to stderr: the JSON line stays the last on stdout
        ok = result.get("parity", {}).get("bit_exact_vs_oracle", False)
        for sub in ("speech", "fc4096", "fc4096_wzp"):
            if sub in result and not result[sub]["parity"]["bit_exact_vs_oracle"]:
                ok = False
        if not ok:
            sys.exit(1)


def headline_from_sub(args, rec, world):
    """--workload fc4096: promote the sub-record to a headline-shaped line"""
    out = {"metric": rec["metric"], "value": rec["value"], "unit": rec["unit"], "n_gpus": world, "steps": rec["steps"],
           "warmup": rec["warmup"], "ms_per_step": rec["ms_per_step"], "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "i8", "data": "synthetic", "config": rec["config"]}
    out.update({k: v for k, v in rec.items() if k not in out})
    return out


def cpu_baseline(om, x_dev_rows, seconds):
    """Time the oracle (oracle/mf_oracle.c: scalar restatement of the reference algorithm,
    gcc -O3, one thread) on this box's host over a bounded sample of the same stream."""
    from oracle import oracle as O
    probe = x_dev_rows[:8].cpu().numpy()
    t0 = time.perf_counter()
    om.run_quantized_batch(probe)
    per_img = (time.perf_counter() - t0) / 8
    n = int(max(16, min(x_dev_rows.shape[0], seconds / max(per_img, 1e-6))))
    xs = x_dev_rows[:n].cpu().numpy()
    t0 = time.perf_counter()
    om.run_quantized_batch(xs)
    dt = time.perf_counter() - t0
    cpu_model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": round(n / dt, 2), "unit": "inferences/s", "cores": 1, "kind": "port",
            "sample": "%d images of the same synthetic stream, 1 thread, %.1f s" % (n, dt),
            "build": "gcc " + " ".join(O.CFLAGS),
            "host": {"cpu": cpu_model, "logical_cores": os.cpu_count()},
            "note": "C restatement of the reference algorithm (oracle/), not the Rust binary"}


def cpu_baseline_all_cores(om, x_dev_rows, per_thread=448):
    """the same oracle with the batch split across every host thread (the C
    call releases the GIL; each thread runs whole inferences, like the reference would per core)."""
    from concurrent.futures import ThreadPoolExecutor
    nthr = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota_note = ""
    try:  # a container's CPU quota (cgroup v2 cpu.max = "<quota> <period>") caps the usable cores
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            cap = max(1, -(-int(q) // int(per)))
            if cap < nthr:
                quota_note = " (cgroup cpu.max allows %d of the host's %d logical cores)" % (cap, nthr)
                nthr = cap
    except (OSError, ValueError):
        pass
    n = min(x_dev_rows.shape[0], nthr * per_thread)
    xs = x_dev_rows[:n].cpu().numpy()
    chunks = [c for c in np.array_split(xs, nthr) if len(c)]
    with ThreadPoolExecutor(len(chunks)) as ex:
        list(ex.map(om.run_quantized_batch, [c[:1] for c in chunks]))  # spin the threads up
        t0 = time.perf_counter()
        list(ex.map(om.run_quantized_batch, chunks))
        dt = time.perf_counter() - t0
    return {"value": round(n / dt, 1), "unit": "inferences/s", "cores": len(chunks), "kind": "port",
            "sample": "%d images split over %d threads, %.1f s%s" % (n, len(chunks), dt, quota_note)}


if __name__ == "__main__":
    main()
