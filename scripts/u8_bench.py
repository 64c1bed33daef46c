import sys, time, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import microflow_rs_amd as mf
from make_u8_model import to_u8
data = to_u8(open("models/person_detect.tflite", "rb").read())
m = mf.Model(data); B = 65536; m.prepare(B)
x = torch.randint(0, 256, (B, m.input_elems), dtype=torch.uint8, device="cuda")
out = torch.empty((B, m.output_elems), dtype=torch.uint8, device="cuda")
for _ in range(3): m.run_quantized(x, out=out)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): m.run_quantized(x, out=out)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print("u8 person_detect batch 65536: %.3f ms/step, %.2f M inf/s" % (dt * 1e3, B / dt / 1e6))
