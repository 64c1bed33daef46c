"""bench.py's stdout contract: the LAST line is one compact JSON record the driver can parse out of its 8 KB stdout
window (round 3's 26.5 KB line was lost); the per-kernel tables live in bench_details.json."""
import json
import os

from tests.conftest import ROOT

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity")


def check_compact(rec, line):
    assert len(line) < 8192 and "\n" not in line
    for k in REQUIRED:
        assert k in rec, k
    assert set(("workload", "per_gpu_batch", "global_batch", "parallelism")) <= set(rec["config"])
    assert "model" not in rec["config"]
    rl = rec["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rl, k
    assert rl["bound"] in ("hbm", "mfma")            # the roof achieved / peak / frac are stated against (the contract's two)
    assert rl.get("binding_roof", rl["bound"]) in ("hbm", "valu", "mfma")
    assert abs(rl["frac"] - rl["achieved"] / rl["peak"]) < 1e-3


def test_compact_line_from_a_full_record():
    import bench
    full = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_full_sample.json")))
    assert len(json.dumps(full)) > 20000          # the record that broke the driver's parser
    rec, line = bench.compact_record(full)
    assert json.loads(line) == rec
    assert len(line) < 4096
    check_compact(rec, line)
    assert rec["value"] == full["value"] and rec["ms_per_step"] == full["ms_per_step"]
    assert rec["cpu_baseline"]["kind"] == "port" and rec["cpu_baseline"]["cores"] == 1
    assert rec["whole_step"]["launches"] == full["whole_step"]["launches"]
    for sub in ("speech", "fc4096", "fc4096_wzp"):
        assert rec[sub]["parity"] is True and "frac" in rec[sub]["roofline"]
    for dropped in ("kernels", "layerwise", "runtime_geometry", "requant_ceiling", "generic_fallback"):
        assert dropped not in rec


def test_compact_line_survives_an_eight_gpu_record():
    import bench
    full = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_full_sample.json")))
    full["n_gpus"] = 8
    full["config"]["shards"] = [[r * 65536, 65536] for r in range(8)]
    full["parity"]["output_checksums"] = ["%016x" % (r * 0x123456789) for r in range(8)]
    rec, line = bench.compact_record(full)
    check_compact(rec, line)
    assert len(line) < 4096
