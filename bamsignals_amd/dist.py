"""Multi-GPU: ranges are independent units (each owns its own output, ref:
src/bamsignals.cpp:164,181,186), so they are dealt round-robin, in sorted order, to the ranks
(one process per GPU, ``torch.distributed``; backend "nccl" = RCCL over xGMI).  There is no
collective on the data path; the only exchange is the final gather of the per-rank results to
one rank, which reassembles them in the caller's range order.
"""
from __future__ import annotations

import numpy as np

from . import _lib


def shard_indices(rid, loc, rank, world):
    """Indices of the ranges rank ``rank`` owns: every ``world``-th range of the (rid, loc) order
    (the order the reference sorts them in, src/bamsignals.cpp:222-226,246)."""
    order = np.lexsort((np.asarray(loc), np.asarray(rid)))
    return order[rank::world]


def shard_layout(length, binsize, ss, rank, world, rid, loc):
    """(indices, local offsets) of one rank's shard; every rank can compute everybody's."""
    from .device import layout
    idx = shard_indices(rid, loc, rank, world)
    return idx, layout(np.asarray(length)[idx], binsize, ss)


def gather_signals(local_out, ranges, binsize, ss, dst=0, group=None):
    """Gather the per-rank flat results to ``dst`` and put them back into the caller's range order.

    ``local_out``: this rank's flat int32 result (torch tensor on the backend's device, or numpy).
    ``ranges``: the FULL range set (dict rid, loc, len) known to every rank.
    Returns ``(out, off)`` on ``dst`` (numpy int32 / int64), ``(None, off)`` elsewhere.
    """
    import torch
    import torch.distributed as dist
    from .device import layout

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    off = layout(ranges["len"], binsize, ss)
    shards = [shard_layout(ranges["len"], binsize, ss, r, world, ranges["rid"], ranges["loc"]) for r in range(world)]
    sizes = [int(o[-1]) for _, o in shards]
    pad = max(max(sizes), 1)
    t = local_out if isinstance(local_out, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(local_out, dtype=np.int32))
    if t.numel() != sizes[rank]:
        raise ValueError(f"rank {rank}: result has {t.numel()} cells, its shard needs {sizes[rank]}")
    buf = torch.zeros(pad, dtype=torch.int32, device=t.device)
    buf[:t.numel()] = t
    bufs = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, bufs, dst=dst, group=group)
    if rank != dst:
        return None, off
    lib = _lib.load()
    out = np.zeros(int(off[-1]), dtype=np.int32)
    for r in range(world):
        idx, loff = shards[r]
        src = bufs[r][:sizes[r]].cpu().numpy()
        which = np.ascontiguousarray(idx, dtype=np.int64)
        _lib.check(lib.bsig_scatter_segments(len(which), src.ctypes.data, loff.ctypes.data, out.ctypes.data,
                                             off.ctypes.data, which.ctypes.data))
    return out, off


def rank_device(group=None):
    """The GPU this rank's file-level calls should run on: an explicit BAMSIGNALS_DEVICE(S) wins (-1 =
    leave it to the library); with the nccl backend torch's current device (the one the communicator
    is bound to); otherwise LOCAL_RANK modulo the number of GPUs, so that ranks never pile up on
    GPU 0 by default."""
    import os

    import torch
    import torch.distributed as dist
    if os.environ.get("BAMSIGNALS_DEVICES") or os.environ.get("BAMSIGNALS_DEVICE"):
        return -1
    n = torch.cuda.device_count()
    if n <= 0:
        return -1
    if dist.get_backend(group) == "nccl":
        return torch.cuda.current_device()
    return int(os.environ.get("LOCAL_RANK", dist.get_rank(group))) % n


def _sharded(kind, bampath, gr, dst, group, **kw):
    """Run one of the user-level calls on this rank's shard of ``gr`` and gather on ``dst``."""
    import torch
    import torch.distributed as dist

    from . import wrappers
    from .countsignals import CountSignals

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    wrappers.set_default_device(rank_device(group))
    levels, codes, start, width, strand = gr.flatten()
    ranges = dict(rid=codes, loc=start - 1, len=width)          # any consistent (seq, start) order works
    mine = shard_indices(ranges["rid"], ranges["loc"], rank, world)
    sub = gr[mine]
    ss = bool(kw.get("ss", False))
    if kind == "count":
        res = wrappers.bamCount(bampath, sub, verbose=False, **kw)
        local = (res.T.reshape(-1) if ss else res).astype(np.int32, copy=False)
        binsize = -1
    else:
        fn = wrappers.bamProfile if kind == "profile" else wrappers.bamCoverage
        sig = fn(bampath, sub, verbose=False, **kw)
        parts = [(m.T.reshape(-1) if ss else m) for m in sig.as_list()]
        local = np.concatenate(parts).astype(np.int32, copy=False) if parts else np.zeros(0, np.int32)
        binsize = int(kw.get("binsize", 1)) if kind == "profile" else 1
    backend = dist.get_backend(group)
    t = torch.from_numpy(np.ascontiguousarray(local))
    if backend == "nccl":
        t = t.cuda()
    out, off = gather_signals(t, ranges, binsize, ss and kind != "coverage", dst=dst, group=group)
    if rank != dst:
        return None
    if kind == "count":
        return out.reshape(-1, 2).T if ss else out
    sigs = [out[off[i]:off[i + 1]] for i in range(len(off) - 1)]
    if ss and kind == "profile":
        sigs = [v.reshape(-1, 2).T for v in sigs]
    return CountSignals(sigs, ss and kind == "profile")


def bamProfile_sharded(bampath, gr, dst=0, group=None, **kw):  # noqa: N802
    """``bamProfile`` over all ranks of the process group (one process per GPU): ranges dealt
    round-robin, every rank reads the BAM, results gathered to ``dst`` (a CountSignals there,
    ``None`` elsewhere).  Keyword arguments as ``bamProfile`` (without ``verbose``)."""
    return _sharded("profile", bampath, gr, dst, group, **kw)


def bamCount_sharded(bampath, gr, dst=0, group=None, **kw):  # noqa: N802
    return _sharded("count", bampath, gr, dst, group, **kw)


def bamCoverage_sharded(bampath, gr, dst=0, group=None, **kw):  # noqa: N802
    return _sharded("coverage", bampath, gr, dst, group, **kw)
