#!/usr/bin/env python3
"""Writes small .tflite models made of the six operators MicroFlow supports
(microflow-macros/src/lib.rs:138-148), for randomized model-level tests: parser + constant
preparation + kernel routing + run, product vs oracle.  Only the fields the MicroFlow front
end reads are written (tflite.fbs field ids as used by microflow-macros/src/ops/*.rs).

    layers = [dict(op="depthwise_conv_2d", ...), dict(op="conv_2d", ...), ...]
    blob = build_model(input_shape, input_q, layers)

Each layer dict (all quantization parameters are given, nothing is derived):
    conv_2d           filters int8 [N,KH,KW,C], fscale [N or 1], fzp [N or 1], bias int32 [N],
                      bscale [N or 1], bzp [N or 1], padding "same"|"valid", strides (h,w), act,
                      out_shape (1,OH,OW,N), out_q (scale, zp)
    depthwise_conv_2d weights int8 [1,KH,KW,C] + the same fields
    average_pool_2d   filter (h,w), padding, strides, act, out_shape, out_q
    fully_connected   weights int8 [N,K], wscale, wzp, bias int32 [N], bscale, bzp, act, out_shape (M,N), out_q
    reshape           out_shape
    softmax           out_shape, out_q
"""
import struct

import numpy as np

from make_fc_model import FB, Table, TableVec, Vec, emit

INT32, UINT8, INT8 = 2, 3, 9
OPCODES = {"average_pool_2d": 1, "conv_2d": 3, "depthwise_conv_2d": 4, "fully_connected": 9, "reshape": 22,
           "softmax": 25}
# BuiltinOptions union type ids (tflite.fbs)
OPT_TYPE = {"conv_2d": 1, "depthwise_conv_2d": 2, "average_pool_2d": 5, "fully_connected": 8, "softmax": 9,
            "reshape": 17}
ACT = {None: 0, "none": 0, "relu": 1, "relu6": 3, 0: 0, 1: 1, 3: 3}
PAD = {"same": 0, "valid": 1, 0: 0, 1: 1}


def _quant(scales, zps):
    # QuantizationParameters { scale:2 zero_point:3 }
    scales, zps = np.atleast_1d(scales), np.atleast_1d(zps)
    return Table({2: ("ref", Vec("<f", [float(np.float32(v)) for v in scales])),
                  3: ("ref", Vec("<q", [int(v) for v in zps], align=8))})


def _tensor(shape, ttype, buffer, scales, zps):
    # Tensor { shape:0 type:1 buffer:2 quantization:4 }
    return Table({0: ("ref", Vec("<i", [int(v) for v in shape])), 1: ("i8", ttype), 2: ("u32", buffer),
                  4: ("ref", _quant(scales, zps))})


def build_model(input_shape, input_q, layers, elem=INT8):
    tensors, buffers, operators = [], [Table({})], []
    used_ops = []

    def add_buffer(arr):
        buffers.append(Table({0: ("ref", Vec(None, np.ascontiguousarray(arr).tobytes(), align=16))}))
        return len(buffers) - 1

    def add_tensor(shape, ttype, buf, scales, zps):
        tensors.append(_tensor(shape, ttype, buf, scales, zps))
        return len(tensors) - 1

    cur = add_tensor(input_shape, elem, 0, input_q[0], input_q[1])
    first = cur
    wdt = np.uint8 if elem == UINT8 else np.int8
    for L in layers:
        op = L["op"]
        if "detach_input" in L:  # negative tests: this operator reads a fresh tensor instead of the running one
            cur = add_tensor(L["detach_input"], elem, 0, input_q[0], input_q[1])
        if op not in used_ops:
            used_ops.append(op)
        oc = used_ops.index(op)
        if op in ("conv_2d", "depthwise_conv_2d"):
            w = np.asarray(L["filters" if op == "conv_2d" else "weights"], wdt)
            tw = add_tensor(w.shape, elem, add_buffer(w), L["fscale"], L["fzp"])
            tb = add_tensor((w.shape[0] if op == "conv_2d" else w.shape[3],), INT32,
                            add_buffer(np.asarray(L["bias"], np.int32)), L["bscale"], L["bzp"])
            out = add_tensor(L["out_shape"], elem, 0, *L["out_q"])
            if op == "conv_2d":  # Conv2DOptions { padding:0 stride_w:1 stride_h:2 act:3 }
                opts = Table({0: ("i8", PAD[L["padding"]]), 1: ("i32", L["strides"][1]), 2: ("i32", L["strides"][0]),
                              3: ("i8", ACT[L.get("act")])})
            else:                # DepthwiseConv2DOptions { padding:0 stride_w:1 stride_h:2 depth_multiplier:3 act:4 }
                opts = Table({0: ("i8", PAD[L["padding"]]), 1: ("i32", L["strides"][1]), 2: ("i32", L["strides"][0]),
                              3: ("i32", L.get("depth_multiplier", 1)), 4: ("i8", ACT[L.get("act")])})
            ins = [cur, tw, tb]
        elif op == "average_pool_2d":
            out = add_tensor(L["out_shape"], elem, 0, *L["out_q"])
            # Pool2DOptions { padding:0 stride_w:1 stride_h:2 filter_width:3 filter_height:4 act:5 }
            opts = Table({0: ("i8", PAD[L["padding"]]), 1: ("i32", L["strides"][1]), 2: ("i32", L["strides"][0]),
                          3: ("i32", L["filter"][1]), 4: ("i32", L["filter"][0]), 5: ("i8", ACT[L.get("act")])})
            ins = [cur]
        elif op == "fully_connected":
            w = np.asarray(L["weights"], wdt)
            tw = add_tensor(w.shape, elem, add_buffer(w), L["wscale"], L["wzp"])
            tb = add_tensor((w.shape[0],), INT32, add_buffer(np.asarray(L["bias"], np.int32)), L["bscale"], L["bzp"])
            out = add_tensor(L["out_shape"], elem, 0, *L["out_q"])
            opts = Table({0: ("i8", ACT[L.get("act")])})
            ins = [cur, tw, tb]
        elif op == "reshape":
            src_q = L.get("out_q")
            out = add_tensor(L["out_shape"], elem, 0, *(src_q or (1.0, 0)))
            shp = add_tensor((len(L["out_shape"]),), INT32, add_buffer(np.asarray(L["out_shape"], np.int32)), 1.0, 0)
            opts = Table({0: ("ref", Vec("<i", [int(v) for v in L["out_shape"]]))})
            ins = [cur, shp]
        elif op == "softmax":
            out = add_tensor(L["out_shape"], elem, 0, *L["out_q"])
            opts = Table({0: ("f32", 1.0)})
            ins = [cur]
        else:
            raise ValueError(op)
        # Operator { opcode_index:0 inputs:1 outputs:2 builtin_options_type:3 builtin_options:4 }
        operators.append(Table({0: ("u32", oc), 1: ("ref", Vec("<i", ins)), 2: ("ref", Vec("<i", [out])),
                                3: ("u8", OPT_TYPE[op]), 4: ("ref", opts)}))
        cur = out
    subgraph = Table({0: ("ref", TableVec(tensors)), 1: ("ref", Vec("<i", [first])), 2: ("ref", Vec("<i", [cur])),
                      3: ("ref", TableVec(operators))})
    # OperatorCode { deprecated_builtin_code:0 builtin_code:3 }
    codes = TableVec([Table({0: ("i8", OPCODES[o]), 3: ("i32", OPCODES[o])}) for o in used_ops])
    model = Table({0: ("u32", 3), 1: ("ref", codes), 2: ("ref", TableVec([subgraph])), 4: ("ref", TableVec(buffers))})
    fb = FB()
    fb.buf += bytes(4) + b"TFL3"
    root = emit(fb, model)
    struct.pack_into("<I", fb.buf, 0, root)
    return bytes(fb.buf)


def random_cnn(rng, elem=INT8, per_channel=True, wzp_nonzero=False, tail="fc"):
    """A random small network touching every operator: conv (KxK) -> depthwise -> conv 1x1 ->
    average pool -> reshape -> fully connected -> softmax (or conv 1x1 head + reshape + softmax)."""
    lo, hi = (0, 256) if elem == UINT8 else (-128, 128)
    mid = (lo + hi) // 2
    H, W, C = int(rng.integers(7, 13)), int(rng.integers(7, 13)), int(rng.integers(1, 4))

    def q(scale_lo=0.02, scale_hi=0.08, relu=False):
        return (float(np.float32(rng.uniform(scale_lo, scale_hi))), int(lo if relu else rng.integers(lo + 20, hi - 20)))

    def wq(n):
        sc = rng.uniform(0.002, 0.01, n if per_channel else 1).astype(np.float32)
        zp = rng.integers(mid - 10, mid + 10, n if per_channel else 1) if wzp_nonzero else np.full(n if per_channel else 1, mid)
        return sc, zp

    def conv_like(op, in_shape, n, kh, kw, strides, padding, act, in_scale):
        _, h, w, c = in_shape
        if padding == "same":
            oh, ow = -(-h // strides[0]), -(-w // strides[1])
        else:
            oh, ow = (h - kh) // strides[0] + 1, (w - kw) // strides[1] + 1
        sc, zp = wq(n)
        shape = (n, kh, kw, c) if op == "conv_2d" else (1, kh, kw, n)
        wts = rng.integers(lo, hi, shape)
        bias = rng.integers(-500, 500, n)
        bscale = (sc * np.float32(in_scale)).astype(np.float32)
        d = dict(op=op, fscale=sc, fzp=zp, bias=bias, bscale=bscale, bzp=np.zeros_like(zp), padding=padding,
                 strides=strides, act=act, out_shape=(1, oh, ow, n), out_q=q(0.05, 0.2, relu=act in ("relu", "relu6")))
        d["filters" if op == "conv_2d" else "weights"] = wts
        return d

    in_q = q()
    layers = []
    L = conv_like("conv_2d", (1, H, W, C), int(rng.integers(2, 9)), int(rng.integers(1, 4)), int(rng.integers(1, 4)),
                  (int(rng.integers(1, 3)), int(rng.integers(1, 3))), str(rng.choice(["same", "valid"])),
                  str(rng.choice(["none", "relu", "relu6"])), in_q[0])
    layers.append(L)
    shp = L["out_shape"]
    L = conv_like("depthwise_conv_2d", shp, shp[3], int(rng.integers(1, 4)), int(rng.integers(1, 4)), (1, 1), "same",
                  str(rng.choice(["none", "relu6"])), layers[-1]["out_q"][0])
    layers.append(L)
    shp = L["out_shape"]
    L = conv_like("conv_2d", shp, int(rng.integers(2, 7)), 1, 1, (1, 1), "same", "relu", layers[-1]["out_q"][0])
    layers.append(L)
    shp = L["out_shape"]
    fh, fw = min(2, shp[1]), min(2, shp[2])
    pool_out = (1, (shp[1] - fh) // fh + 1, (shp[2] - fw) // fw + 1, shp[3])
    layers.append(dict(op="average_pool_2d", filter=(fh, fw), padding="valid", strides=(fh, fw), act="none",
                       out_shape=pool_out, out_q=q(0.05, 0.2)))
    flat = int(np.prod(pool_out))
    classes = int(rng.integers(2, 6))
    if tail == "fc":
        layers.append(dict(op="reshape", out_shape=(1, flat), out_q=layers[-1]["out_q"]))
        sc, zp = wq(1)
        layers.append(dict(op="fully_connected", weights=rng.integers(lo, hi, (classes, flat)), wscale=sc[:1], wzp=zp[:1],
                           bias=rng.integers(-300, 300, classes), bscale=sc[:1] * np.float32(layers[-1]["out_q"][0]),
                           bzp=[0], act="none", out_shape=(1, classes), out_q=q(0.1, 0.3)))
    else:
        L = conv_like("conv_2d", pool_out, classes, 1, 1, (1, 1), "same", "none", layers[-1]["out_q"][0])
        layers.append(L)
        layers.append(dict(op="reshape", out_shape=(1, int(np.prod(L["out_shape"]))), out_q=L["out_q"]))
        classes = int(np.prod(L["out_shape"]))
    layers.append(dict(op="softmax", out_shape=(1, classes), out_q=(1.0 / 256.0, lo)))
    return build_model((1, H, W, C), in_q, layers, elem)


def speech_like(rng, elem=INT8, fc_wzp=0, per_channel=True, act="relu"):
    """The layer structure and shapes of speech.tflite (TinyConv: Reshape -> DepthwiseConv2D 10x8 stride 2 SAME with one
    input channel and 8 output channels -> FullyConnected 4000 -> 4 -> Softmax) with random weights and quantization,
    incl. a non-zero FullyConnected weight zero point, which the shipped model does not have."""
    lo, hi = (0, 256) if elem == UINT8 else (-128, 128)
    mid = (lo + hi) // 2
    in_q = (float(np.float32(rng.uniform(0.02, 0.08))), int(rng.integers(lo, lo + 40)))
    dsc = rng.uniform(0.002, 0.01, 8 if per_channel else 1).astype(np.float32)
    dzp = np.full(8 if per_channel else 1, mid)
    dw_out = (float(np.float32(rng.uniform(0.05, 0.2))), int(lo if act in ("relu", "relu6") else rng.integers(lo + 20, hi - 20)))
    layers = [dict(op="reshape", out_shape=(1, 49, 40, 1), out_q=in_q),
              dict(op="depthwise_conv_2d", weights=rng.integers(lo, hi, (1, 10, 8, 8)), fscale=dsc, fzp=dzp,
                   bias=rng.integers(-2000, 2000, 8), bscale=(dsc * np.float32(in_q[0])).astype(np.float32),
                   bzp=np.zeros_like(dzp), padding="same", strides=(2, 2), act=act, out_shape=(1, 25, 20, 8), out_q=dw_out)]
    fsc = np.float32(rng.uniform(0.001, 0.004))
    layers.append(dict(op="fully_connected", weights=rng.integers(lo, hi, (4, 4000)), wscale=[fsc], wzp=[mid + fc_wzp],
                       bias=rng.integers(-3000, 3000, 4), bscale=[fsc * np.float32(dw_out[0])], bzp=[0], act="none",
                       out_shape=(1, 4), out_q=(float(np.float32(rng.uniform(0.3, 0.9))), int(rng.integers(lo + 60, hi - 60)))))
    layers.append(dict(op="softmax", out_shape=(1, 4), out_q=(1.0 / 256.0, lo)))
    return build_model((1, 1960), in_q, layers, elem)


def person_detect_like(rng, side=96, width=1.0, elem=INT8, wzp_nonzero=False, n_stage=5, classes=2, wmax=None):
    """The layer structure of person_detect.tflite (MobileNet-v1 0.25, grey input: a one-channel 3x3 stride-2 stem, then
    depthwise 3x3 + 1x1 pairs with strides 1 2 1 2 1 2 [1 x n_stage] 2 1, AveragePool2D over what is left, a 1x1 head,
    Reshape, Softmax) at another input size / channel width / run length, with random weights.  Activations keep the
    shipped model's quantization (scale 6/255, zero point = the type's minimum, relu6), weights are per-channel.
    `wmax`: weights within +-wmax of the type's middle (trained networks have small weights; None = the full range, the worst
    case for the accumulator bounds that decide the epilogue form)."""
    lo, hi = (0, 256) if elem == UINT8 else (-128, 128)
    mid = (lo + hi) // 2
    act_q = (0.0235294122, lo)
    in_q = (0.00784313772, mid - 1)
    ch = lambda c: max(4, int(round(c * width / 4.0)) * 4)  # noqa: E731
    layers = []

    def conv(op, in_shape, n, k, stride, act, in_scale, out_q):
        _, h, w, c = in_shape
        oh, ow = -(-h // stride), -(-w // stride)
        fan = k * k * (1 if op == "depthwise_conv_2d" else c)
        sc = (rng.uniform(0.6, 1.4, n) * 2.2 / (74.0 * np.sqrt(fan))).astype(np.float32)   # keeps the activations spread
        zp = rng.integers(mid - 12, mid + 12, n) if wzp_nonzero else np.full(n, mid)
        shape = (n, k, k, c) if op == "conv_2d" else (1, k, k, n)
        d = dict(op=op, fscale=sc, fzp=zp, bias=rng.integers(-300, 300, n), bscale=(sc * np.float32(in_scale)).astype(np.float32),
                 bzp=np.zeros(n, np.int64), padding="same", strides=(stride, stride), act=act, out_shape=(1, oh, ow, n), out_q=out_q)
        d["filters" if op == "conv_2d" else "weights"] = rng.integers(lo, hi, shape) if wmax is None else rng.integers(mid - wmax, mid + wmax + 1, shape)
        layers.append(d)
        return d["out_shape"]

    shp = conv("depthwise_conv_2d", (1, side, side, 1), ch(8), 3, 2, "relu6", in_q[0], act_q)
    plan = [(1, 16), (2, 32), (1, 32), (2, 64), (1, 64), (2, 128)] + [(1, 128)] * n_stage + [(2, 256), (1, 256)]
    for stride, n in plan:
        shp = conv("depthwise_conv_2d", shp, shp[3], 3, stride, "relu6", act_q[0], act_q)
        shp = conv("conv_2d", shp, ch(n), 1, 1, "relu6", act_q[0], act_q)
    layers.append(dict(op="average_pool_2d", filter=(shp[1], shp[2]), padding="valid", strides=(2, 2), act="none",
                       out_shape=(1, 1, 1, shp[3]), out_q=(0.0186093301, lo)))
    shp = (1, 1, 1, shp[3])
    head_q = (0.0125187514, mid - 1)
    sc = (rng.uniform(0.6, 1.4, classes) * 0.002).astype(np.float32)
    zp = rng.integers(mid - 12, mid + 12, classes) if wzp_nonzero else np.full(classes, mid)
    layers.append(dict(op="conv_2d", filters=rng.integers(lo, hi, (classes, 1, 1, shp[3])), fscale=sc, fzp=zp,
                       bias=rng.integers(-300, 300, classes), bscale=(sc * np.float32(0.0186093301)).astype(np.float32),
                       bzp=np.zeros(classes, np.int64), padding="same", strides=(1, 1), act="none",
                       out_shape=(1, 1, 1, classes), out_q=head_q))
    layers.append(dict(op="reshape", out_shape=(1, classes), out_q=head_q))
    layers.append(dict(op="softmax", out_shape=(1, classes), out_q=(1.0 / 256.0, lo)))
    return build_model((1, side, side, 1), in_q, layers, elem)
