"""N>1 path on CPU: world_size-2 (and 3) gloo process groups exercise the query
sharding + the single gather of placement records.  The per-rank placer is the
C oracle here (no GPU in this container); on a GPU box the same code runs with
the engine and the nccl (RCCL) backend -- see test_dist_engine_gloo_on_gpu."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from classeq2_amd import dist as cdist
from classeq2_amd.synth import SynthDb
from tests.helpers import records_equal


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_reads, use_engine, q, backend="gloo"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    device = None
    if backend == "nccl":  # RCCL: one GPU per rank, records gathered device to device
        import torch
        torch.cuda.set_device(rank)
        device = torch.device("cuda", rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        s = SynthDb(60, 300, 8, 4, threads=1)
        bases, offsets, _ = s.reads(n_reads, 90)
        if use_engine:
            from classeq2_amd import engine
            db = engine.PlacementDb(s.flat, device=rank if backend == "nccl" else 0)
            place = db.place_batch
        else:
            from oracle.oracle_port import OraclePort
            place = OraclePort(s.flat).place_batch
        got = cdist.place_sharded(place, bases, offsets, device=device)
        if rank == 0:
            from oracle.oracle_port import OraclePort
            want = OraclePort(s.flat).place_batch(bases, offsets)
            q.put(int(len(records_equal(got, want)) + abs(len(got) - len(want))))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _run(world, n_reads, use_engine=False, backend="gloo"):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_reads, use_engine, q, backend)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert q.get(timeout=5) == 0


def test_shard_ranges_cover_everything():
    for n in (0, 1, 7, 64, 1001):
        for world in (1, 2, 3, 8):
            spans = [cdist.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))


@pytest.mark.parametrize("world,n_reads", [(2, 101), (3, 50), (2, 1)])
def test_sharded_gather_matches_unsharded(world, n_reads):
    _run(world, n_reads)


@pytest.mark.gpu
def test_dist_engine_gloo_on_gpu():
    """Two ranks share the one GPU of the test box: engine placements + gloo gather."""
    _run(2, 301, use_engine=True)


@pytest.mark.gpu
def test_dist_engine_nccl_two_gpus():
    """The production shape: one rank per GPU, the gather over RCCL.  Needs two GPUs (the round's test box has one:
    skipped there; N > 1 on hardware is only ever run by the driver's scaling bench)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    _run(2, 4001, use_engine=True, backend="nccl")
