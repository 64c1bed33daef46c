"""bench.py's real multi-rank launch path, on the one GPU a test box has: `--gpus 2` self-launches two ranks under
torch.distributed.run exactly as the driver's 8-GPU line does; `--backend gloo --share-device` only swaps RCCL (which
refuses two ranks on one device) for gloo and puts both ranks on GPU 0.  Everything else -- rendezvous, shard ranges,
the barrier + max-over-ranks timing, the checksum gather, the teardown before rank 0's single-rank extras -- is the
code an 8-GPU node runs."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from tests.conftest import ROOT, model_path
from tests.synth import layer_checksum, synth_i8

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("per_gpu,world", [(64, 2), (8192, 8)])
def test_bench_ranks_share_one_device(O, per_gpu, world):
    """world = 8: what an 8-GPU run can fail at for host-side reasons -- eight concurrent model creations (each runs the
    device verifier of the single-fma epilogue), eight step queues on one device, rendezvous, gather, teardown, one line."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1",
           "--batch", str(per_gpu), "--no-extra", "--no-cpu-baseline", "--no-host-fed", "--backend", "gloo", "--share-device"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # rank 0 prints ONE line, the other rank none
    assert r.stdout.strip().splitlines()[-1] == lines[0]   # ... and it is the LAST thing on stdout
    rec = json.loads(lines[0])
    from tests.test_bench_line import check_compact
    rec.setdefault("cpu_baseline", None)
    check_compact(rec, lines[0])                        # < 8 KB, the contract's keys, roofline block
    assert len(lines[0]) < 4096
    details = json.load(open(os.path.join(ROOT, rec["details"])))
    assert details["value"] == rec["value"] and len(details["kernels"]) >= 1
    assert rec["n_gpus"] == world and rec["steps"] == 3 and rec["warmup"] == 1
    assert rec["config"]["per_gpu_batch"] == per_gpu and rec["config"]["global_batch"] == world * per_gpu
    assert rec["config"]["shards"] == [[r * per_gpu, per_gpu] for r in range(world)] and rec["config"]["backend"] == "gloo"
    assert rec["scaling"] == "weak" and rec["parity"]["bit_exact_vs_oracle"] is True
    assert abs(rec["value"] - world * per_gpu / (rec["ms_per_step"] * 1e-3)) <= 1e-3 * rec["value"]
    # every shard's output checksum against a run over the same slice of the global stream: the oracle's (world 2), or -- 65 536
    # images are minutes of oracle time -- one in-process pass of the library over the whole stream (world 8; rank 0's own oracle
    # samples are in the line's parity block)
    from microflow_rs_amd.shard import shard_range
    want = []
    if world * per_gpu <= 1024:
        om = O.Model(model_path("person_detect"))
        for rank in range(world):
            first, count = shard_range(world * per_gpu, rank, world)
            y = om.run_quantized_batch(synth_i8(3, first, count, om.in_elems))
            want.append("%016x" % (int(layer_checksum(y)) & 0x7FFFFFFFFFFFFFFF))
    else:
        import microflow_rs_amd as mf
        from microflow_rs_amd.model import checksum_i8, synth_i8 as dev_synth
        m = mf.model(model_path("person_detect"))
        total = world * per_gpu
        m.prepare(total)
        x = dev_synth(3 + 0x4D4643, 0, total * m.input_elems).reshape((total,) + m.input_shape)
        y = m.run_quantized(x).reshape(total, -1)
        for rank in range(world):
            first, count = shard_range(total, rank, world)
            want.append("%016x" % (checksum_i8(y[first:first + count].reshape(-1)) & 0x7FFFFFFFFFFFFFFF))
    assert rec["parity"]["output_checksums"] == want
    assert len(set(want)) == world


_RCCL_CHILD = r"""
import os, sys
sys.path.insert(0, {root!r})
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "{port}")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch, torch.distributed as dist
from microflow_rs_amd.shard import gather_checksums, max_over_ranks, shard_range
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl"
dist.barrier()
assert max_over_ranks(dist, 1.25, device="cuda") == 1.25
assert gather_checksums(dist, 0x7123456789ABCDEF, device="cuda") == [0x7123456789ABCDEF]
assert shard_range(65536, 0, 1) == (0, 65536)
dist.barrier()
dist.destroy_process_group()
print("rccl-ok", torch.cuda.nccl.version())
"""


def test_rccl_world_of_one():
    """RCCL itself, on the one GPU this box has: library load, device binding, the three collectives of the multi-GPU path
    (barrier, all_reduce MAX, all_gather) and the teardown order bench.py uses -- a world size of one is legal for RCCL."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-c", _RCCL_CHILD.format(root=ROOT, port=port)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "rccl-ok" in r.stdout, r.stdout[-1000:] + r.stderr[-3000:]


def test_bench_single_rank_over_rccl(O):
    """`bench.py --gpus 1 --dist`: the driver's multi-GPU line with N = 1 -- init_process_group("nccl", device_id=...), the
    barrier-bracketed timed region, max over ranks, the checksum all_gather, teardown before the single-rank extras."""
    per_gpu = 64
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--dist", "--steps", "3", "--warmup", "1",
           "--batch", str(per_gpu), "--no-extra", "--no-cpu-baseline", "--no-host-fed"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    rec = json.loads(r.stdout.strip().splitlines()[-1])   # the JSON line is the LAST line although RCCL prints to stdout at exit
    assert rec["n_gpus"] == 1 and rec["config"]["backend"] == "nccl" and rec["parity"]["bit_exact_vs_oracle"] is True
    om = O.Model(model_path("person_detect"))
    y = om.run_quantized_batch(synth_i8(3, 0, per_gpu, om.in_elems))
    assert rec["parity"]["output_checksums"] == ["%016x" % (int(layer_checksum(y)) & 0x7FFFFFFFFFFFFFFF)]
