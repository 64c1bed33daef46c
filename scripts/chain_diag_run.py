import os, sys, numpy as np
ROOT="/root/repo"; sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+"/tools")
import torch, microflow_rs_amd as mf, tflite_writer as tw
from microflow_rs_amd.model import synth_i8
side=int(sys.argv[1]); width=float(sys.argv[2])
m = mf.model(tw.person_detect_like(np.random.default_rng(side), side, width))
B = int(65536 * 96 * 96 / (side * side)); m.prepare(B)
x = synth_i8(9, 0, B * m.input_elems); y = torch.empty(B * m.output_elems, dtype=torch.int8, device="cuda")
m.time_device(x, y, B, warmup=0, iters=1, per_op=False)
for i in range(m.num_ops):
    k=m.op(i)["kernel"]
    if k.startswith("chain_rt"): print(i, k)
