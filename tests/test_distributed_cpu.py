"""The multi-GPU decomposition (batch shard, no data-path collective) exercised with
world_size 2 on CPU over gloo: same shard helper and collectives bench.py uses; the
compute stand-in is the CPU oracle (tests may use it)."""
import os
import socket
import sys

import numpy as np
import pytest

from tests.conftest import ROOT, model_path
from tests.synth import layer_checksum, synth_i8


def test_shard_range_partitions_exactly():
    from microflow_rs_amd.shard import shard_range
    for total in (0, 1, 7, 8, 65536, 524288, 1000003):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0
            for (f0, c0), (f1, _) in zip(spans, spans[1:]):
                assert f0 + c0 == f1
            assert spans[-1][0] + spans[-1][1] == total
            counts = [c for _, c in spans]
            assert max(counts) - min(counts) <= 1
    assert shard_range(524288, 3, 8) == (3 * 65536, 65536)   # BASELINE config 4
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from microflow_rs_amd.shard import gather_checksums, max_over_ranks, shard_range
    from oracle import oracle as O
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        m = O.Model(model_path("speech"))
        first, count = shard_range(total, rank, world)
        # every rank regenerates exactly its slice of the global stream
        x = synth_i8(2, first, count, m.in_elems)
        y = m.run_quantized_batch(x)
        dist.barrier()
        cks = gather_checksums(dist, int(layer_checksum(y)))
        tmax = max_over_ranks(dist, 1.0 + rank)
        if rank == 0:
            q.put((cks, tmax))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_batch_shard_over_gloo():
    import torch.multiprocessing as mp
    from microflow_rs_amd.shard import shard_range
    from oracle import oracle as O
    world, total = 2, 13   # uneven on purpose
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    cks, tmax = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # the unsharded run, cut at the same boundaries, must give the same per-shard checksums
    m = O.Model(model_path("speech"))
    y = m.run_quantized_batch(synth_i8(2, 0, total, m.in_elems))
    want = []
    for r in range(world):
        f, c = shard_range(total, r, world)
        want.append(int(layer_checksum(y[f:f + c])) & 0x7FFFFFFFFFFFFFFF)
    assert cks == want
    assert tmax == 2.0   # max over ranks of (1.0, 2.0)
