"""Seeded synthetic reads and ranges for the parity tests and bench.py.

Reads: position uniform on the genome, sorted by (reference, pos); read length 100;
CIGAR ``100M`` for 90 %, the rest uniformly from ``50M10D50M``, ``40M2000N60M``, ``5S95M``,
``48M4I48M``; mapq uniform 0..60; 2 % flagged 0x400; strand Bernoulli(0.5).
Paired-end: fragment length ``1 + NegBin(mu=149, size=10)`` as the reference's toy-data generator
(tests/testthat/utils.R:138), flags 99/147/83/163, tlen = +-fragment.
"""
from __future__ import annotations

import numpy as np

# (packed ops, reference span) ; op codes MIDNSHP=X = 0..8
_M, _I, _D, _N, _S = 0, 1, 2, 3, 4


def _ops(*pairs):
    return [ln << 4 | op for ln, op in pairs]


CIGAR_MENU = [
    _ops((100, _M)),
    _ops((50, _M), (10, _D), (50, _M)),
    _ops((40, _M), (2000, _N), (60, _M)),
    _ops((5, _S), (95, _M)),
    _ops((48, _M), (4, _I), (48, _M)),
]
CIGAR_SPAN = np.asarray([100, 110, 2100, 95, 96], dtype=np.int32)


def synth_reads(n_reads, ref_len, seed=0xBA51, paired=False, with_cigar=True):
    """Columns of a coordinate-sorted synthetic BAM.

    Returns dict(ref_len, ref_off, pos, flag, mapq, tlen, end, [cigar_off, cigar]).
    With ``paired`` ``n_reads`` must be even (n_reads/2 pairs).
    """
    rng = np.random.default_rng(seed)
    ref_len = np.asarray(ref_len, dtype=np.int64)
    n_ref = len(ref_len)
    gstart = np.concatenate([[0], np.cumsum(ref_len)])
    total = int(gstart[-1])

    if not paired:
        g = np.sort(rng.integers(0, total, n_reads, dtype=np.int64))
        neg = rng.random(n_reads) < 0.5
        flag = np.where(neg, 16, 0).astype(np.uint16)
        tlen = np.zeros(n_reads, dtype=np.int32)
    else:
        if n_reads % 2:
            raise ValueError("paired reads come in pairs")
        npair = n_reads // 2
        gp = rng.integers(0, total, npair, dtype=np.int64)
        # NegBin(mu, size): p = size / (size + mu)
        frag = (1 + rng.negative_binomial(10, 10.0 / (10.0 + 149.0), npair)).astype(np.int64)
        frag = np.maximum(frag, 100)
        first_is_read1 = rng.random(npair) < 0.5
        gm = np.minimum(gp + frag - 100, total - 1)
        g = np.concatenate([gp, gm])
        flag = np.concatenate([np.where(first_is_read1, 99, 163), np.where(first_is_read1, 147, 83)]).astype(np.uint16)
        tlen = np.concatenate([frag, -frag]).astype(np.int32)
        order = np.argsort(g, kind="stable")
        g, flag, tlen = g[order], flag[order], tlen[order]
        del order

    rid = (np.searchsorted(gstart, g, side="right") - 1).astype(np.int32)
    pos = (g - gstart[rid]).astype(np.int32)
    # keep every read inside its reference (mates of pairs near a boundary are clipped back)
    over = pos > (ref_len[rid] - 1)
    pos[over] = (ref_len[rid[over]] - 1).astype(np.int32)
    del g
    # (rid, pos) order survives the clipping except for a handful of boundary reads: re-sort if needed
    key = rid.astype(np.int64) << 32 | pos
    if np.any(np.diff(key) < 0):
        order = np.argsort(key, kind="stable")
        rid, pos, flag, tlen = rid[order], pos[order], flag[order], tlen[order]
    del key
    ref_off = np.searchsorted(rid, np.arange(n_ref + 1)).astype(np.int64)

    which = np.where(rng.random(n_reads) < 0.9, 0, rng.integers(1, 5, n_reads)).astype(np.int8)
    end = (pos + CIGAR_SPAN[which] - 1).astype(np.int32)
    mapq = rng.integers(0, 61, n_reads).astype(np.uint8)
    dup = rng.random(n_reads) < 0.02
    flag = (flag | np.where(dup, 0x400, 0).astype(np.uint16)).astype(np.uint16)
    out = dict(ref_len=ref_len.astype(np.int32), ref_off=ref_off, rid=rid, pos=pos, flag=flag, mapq=mapq,
               tlen=tlen, end=end, cigar_menu=which)
    if with_cigar:
        add_cigar(out)
    return out


def add_cigar(cols):
    """Adds the packed CIGAR columns (``cigar_off``, ``cigar``) to reads made with ``with_cigar=False``
    (the choice from CIGAR_MENU is kept in ``cols["cigar_menu"]``)."""
    which = cols["cigar_menu"]
    nops = np.asarray([len(m) for m in CIGAR_MENU], dtype=np.int64)[which]
    cigar_off = np.empty(len(which) + 1, dtype=np.int64)
    cigar_off[0] = 0
    np.cumsum(nops, out=cigar_off[1:])
    del nops
    cigar = np.full(int(cigar_off[-1]), CIGAR_MENU[0][0], dtype=np.uint32)     # 90 % are the one-op 100M
    for k, ops in enumerate(CIGAR_MENU):
        if k == 0:
            continue
        sel = cigar_off[:-1][which == k]
        for t, op in enumerate(ops):
            cigar[sel + t] = op
    cols["cigar_off"] = cigar_off
    cols["cigar"] = cigar
    return cols


def synth_ranges(n, width, ref_len, seed=0xBA52, strands=(1, -1, 0), jitter=0):
    """``n`` ranges of ``width`` bp (+- ``jitter``), start uniform, strand uniform on ``strands``."""
    rng = np.random.default_rng(seed)
    ref_len = np.asarray(ref_len, dtype=np.int64)
    gstart = np.concatenate([[0], np.cumsum(ref_len)])
    w = np.full(n, width, dtype=np.int64)
    if jitter:
        w = np.maximum(w + rng.integers(-jitter, jitter + 1, n), 0)
    g = rng.integers(0, int(gstart[-1]), n, dtype=np.int64)
    rid = (np.searchsorted(gstart, g, side="right") - 1).astype(np.int32)
    loc = g - gstart[rid]
    loc = np.minimum(loc, np.maximum(ref_len[rid] - w, 0))
    strand = np.asarray(strands, dtype=np.int32)[rng.integers(0, len(strands), n)]
    return dict(rid=rid, loc=loc.astype(np.int32), len=w.astype(np.int32), strand=strand)


def tile_ranges(ref_len, width, strand=0):
    """Consecutive ``width``-bp tiles covering every reference (last tile of a reference short)."""
    rid, loc, ln = [], [], []
    for r, L in enumerate(np.asarray(ref_len, dtype=np.int64)):
        s = np.arange(0, L, width, dtype=np.int64)
        rid.append(np.full(len(s), r, dtype=np.int32))
        loc.append(s.astype(np.int32))
        ln.append(np.minimum(width, L - s).astype(np.int32))
    rid, loc, ln = np.concatenate(rid), np.concatenate(loc), np.concatenate(ln)
    return dict(rid=rid, loc=loc, len=ln, strand=np.full(len(rid), strand, dtype=np.int32))
