"""CPU, world_size 2 over gloo: sharding + gather + reassembly of the multi-GPU path.  The
per-rank compute is stood in for by the oracle (this is test infrastructure: the product's
per-rank compute is the HIP plan, exercised in the -m gpu tests and bench.py)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from bamsignals_amd.dist import gather_signals, shard_indices
        from bamsignals_amd.synth import synth_ranges, synth_reads
        from oracle import oracle_c
        cols = synth_reads(50_000, [400_000, 90_000], seed=4)
        rg = synth_ranges(301, 700, cols["ref_len"], seed=9, jitter=300)
        rg["len"][5] = 0
        orc = oracle_c.OracleReads(cols["ref_off"], cols["pos"], cols["end"], cols["flag"], cols["mapq"], cols["tlen"])
        results = {}
        for name, fn, args, (bs, ss) in (
            ("profile_ss", oracle_c.pileup_core, dict(binsize=3, ss=True, shift=20), (3, True)),
            ("count", oracle_c.pileup_core, dict(binsize=-1), (-1, False)),
            ("coverage", oracle_c.coverage_core, dict(), (1, False)),
        ):
            mine = shard_indices(rg["rid"], rg["loc"], rank, world)
            sub = {k: v[mine] for k, v in rg.items()}
            local, _ = fn(orc, sub, **args)
            out, off = gather_signals(local, rg, bs, ss, dst=0)
            if rank == 0:
                want, woff = fn(orc, rg, **args)
                results[name] = bool(np.array_equal(out, want) and np.array_equal(off, woff))
        # the shards partition the ranges
        allidx = np.concatenate([shard_indices(rg["rid"], rg["loc"], r, world) for r in range(world)])
        results["partition"] = bool(np.array_equal(np.sort(allidx), np.arange(len(rg["rid"]))))
        if rank == 0:
            q.put(results)
    finally:
        dist.destroy_process_group()


def _gather_over_gloo(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == {"profile_ss": True, "count": True, "coverage": True, "partition": True}


@pytest.mark.timeout(300)
def test_two_rank_gather_over_gloo():
    _gather_over_gloo(2)


@pytest.mark.timeout(420)
def test_eight_rank_gather_over_gloo():
    """BASELINE config 5's rank count: 301 ragged ranges dealt round-robin to 8 ranks (37-38 each)."""
    _gather_over_gloo(8)


def test_scatter_segments_rejects_bad_shapes():
    from bamsignals_amd import _lib
    lib = _lib.load()
    src = np.arange(10, dtype=np.int32)
    soff = np.asarray([0, 4, 10], dtype=np.int64)
    dst = np.zeros(10, dtype=np.int32)
    doff = np.asarray([0, 6, 10], dtype=np.int64)
    which = np.asarray([1, 0], dtype=np.int64)
    assert lib.bsig_scatter_segments(2, src.ctypes.data, soff.ctypes.data, dst.ctypes.data, doff.ctypes.data, which.ctypes.data) == 0
    assert list(dst) == [4, 5, 6, 7, 8, 9, 0, 1, 2, 3]
    which = np.asarray([0, 1], dtype=np.int64)
    assert lib.bsig_scatter_segments(2, src.ctypes.data, soff.ctypes.data, dst.ctypes.data, doff.ctypes.data, which.ctypes.data) != 0
