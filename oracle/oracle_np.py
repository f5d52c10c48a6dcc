"""TEST INFRASTRUCTURE ONLY — numpy restatement of the reference hot path.

This module is the *checker*, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  The product path (``bamsignals_amd``) must never import anything from
``oracle/``.

It restates, in "all pairs" form (for every range, for every read on that
reference: filter, then pile up), the arithmetic of

* ``Pileupper::setRead`` / ``Pileupper::pileup``   src/bamsignals.cpp:326-363
* ``Coverager::setRead`` / ``Coverager::pileup``   src/bamsignals.cpp:392-438
* ``allocateList`` output shapes                   src/bamsignals.cpp:139-192
* ``pileup_core`` / ``coverage_core`` orchestration src/bamsignals.cpp:444-494
* ``cumsum``                                        src/bamsignals.cpp:464-470
* htslib ``bam_endpos`` (third-party, Rhtslib >= 1.13.1, not vendored in the
  reference): ``pos + sum(len(op) for op in M,D,N,=,X)``; unmapped (0x4) or a
  zero sum gives ``pos + 1``.

The all-pairs form is equivalent to the reference's chunked driver
``overlapAndPileup`` (src/bamsignals.cpp:240-291): chunks partition the
ranges, so every (read, range) pair is evaluated at most once, and the index
query window (``±ext``) always contains every read that can hit.  The faithful
chunked form lives in ``oracle/bamsignals_oracle.c``; tests require both to
agree.

PARITY UNPINNED by the evidence rule (no expected value was computed by the reference or by R:
the reference holds no literal outputs and cannot run in this image; see bamsignals_oracle.h).
What exists instead: this restatement is checked in ``tests/test_oracle_golden.py``
against the reference's own fixtures (``inst/extdata/randomBam.bam``,
``tests/testthat/randomReads.RData``, ``inst/extdata/randomAnnot.Rdata``)
through ``oracle/r_oracle.py``, an independent restatement of the reference's
test oracle (``tests/testthat/utils.R:178-311``).  The reference itself cannot
be built here (needs Rcpp + htslib; neither R nor htslib are in the image).
"""
from __future__ import annotations

import numpy as np

# CIGAR op codes "MIDNSHP=X" = 0..8 ; reference-consuming = M, D, N, =, X
_REF_CONSUMING = np.zeros(16, dtype=bool)
_REF_CONSUMING[[0, 2, 3, 7, 8]] = True

BAM_FUNMAP = 0x4
BAM_FREVERSE = 0x10


def cigar_end(pos, flag, cigar_off, cigar):
    """Inclusive 0-based read end = bam_endpos(b) - 1 (src/bamsignals.cpp:16-18).

    ``cigar`` holds packed ops ``len << 4 | op``; read *i* owns
    ``cigar[cigar_off[i]:cigar_off[i+1]]``.
    """
    pos = np.asarray(pos, dtype=np.int64)
    flag = np.asarray(flag, dtype=np.int64)
    cigar = np.asarray(cigar, dtype=np.uint32)
    cigar_off = np.asarray(cigar_off, dtype=np.int64)
    oplen = (cigar >> 4).astype(np.int64)
    consuming = _REF_CONSUMING[(cigar & 0xF).astype(np.int64)]
    contrib = np.where(consuming, oplen, 0)
    csum = np.concatenate([[0], np.cumsum(contrib)])
    rlen = csum[cigar_off[1:]] - csum[cigar_off[:-1]]
    rlen = np.where((flag & BAM_FUNMAP) != 0, 0, rlen)
    rlen = np.where(rlen == 0, 1, rlen)
    return (pos + rlen - 1).astype(np.int32)


def filter_mask(flag, mapq, tlen, mapqual, requiredF, filteredF, tlen_filter):
    """True where the read is KEPT (src/bamsignals.cpp:328-333 / 394-399)."""
    flag = np.asarray(flag).astype(np.int64)
    mapq = np.asarray(mapq).astype(np.int64)
    tlen = np.asarray(tlen).astype(np.int64)
    notflag = (~flag) & 0xFFFFFFFF  # ~flag after int promotion, as uint32
    req = np.int64(requiredF) & 0xFFFFFFFF
    fil = np.int64(filteredF) & 0xFFFFFFFF
    rej = mapq < mapqual
    rej |= (req & notflag) != 0          # invalidFlag(read, requiredF)
    rej |= (fil & notflag) == 0          # !invalidFlag(read, filteredF)
    if tlen_filter is not None and len(tlen_filter) > 0:
        a = np.abs(tlen)
        rej |= (a < int(tlen_filter[0])) | (a > int(tlen_filter[1]))
    return ~rej


def profile_layout(widths, binsize, ss):
    """Cells per range and flat offsets (allocateList, src/bamsignals.cpp:139-192)."""
    widths = np.asarray(widths, dtype=np.int64)
    mult = 2 if ss else 1
    if binsize <= 0:
        cells = np.ones(len(widths), dtype=np.int64)
    else:
        cells = -(-widths // binsize)  # ceil(len / binsize)
    off = np.concatenate([[0], np.cumsum(cells * mult)])
    return cells, off


def pileup_core(reads, ranges, tlen_filter=(), mapqual=0, binsize=1, shift=0,
                ss=False, requiredF=0, filteredF=-1, pe_mid=False):
    """Flat int32 result of ``pileup_core`` (src/bamsignals.cpp:444-461).

    ``reads``: dict of arrays rid,pos,end,flag,mapq,tlen.
    ``ranges``: dict of arrays rid,loc(0-based),len,strand(-1,0,+1).
    Returns ``(out, off)``: range *i* owns ``out[off[i]:off[i+1]]``; with ``ss``
    the element ``2*bin + antisense`` (column-major 2 x width matrix).  With
    ``binsize <= 0`` (bamCount) every range owns ``mult`` cells.
    """
    rid = np.asarray(reads["rid"]); pos = np.asarray(reads["pos"], dtype=np.int64)
    end = np.asarray(reads["end"], dtype=np.int64)
    flag = np.asarray(reads["flag"]).astype(np.int64)
    tlen = np.asarray(reads["tlen"]).astype(np.int64)
    keep = filter_mask(flag, reads["mapq"], tlen, mapqual, requiredF, filteredF, tlen_filter)
    if pe_mid and (tlen_filter is None or len(tlen_filter) < 2):
        raise ValueError("pe_mid needs a 2-element tlen_filter")
    neg = (flag & BAM_FREVERSE) != 0
    # C integer division of a non-negative value
    offset = (np.abs(tlen) // 2 + shift) if pe_mid else np.full(len(pos), shift, dtype=np.int64)
    p5 = np.where(neg, end - offset, pos + offset)

    r_rid = np.asarray(ranges["rid"]); r_loc = np.asarray(ranges["loc"], dtype=np.int64)
    r_len = np.asarray(ranges["len"], dtype=np.int64)
    r_strand = np.asarray(ranges["strand"], dtype=np.int64)
    n = len(r_loc)
    mult = 2 if ss else 1
    cells, off = profile_layout(r_len, binsize, ss)
    if binsize <= 0:
        bs = int(r_len.max()) if n else -1   # maxw starts at -1 (:160-167)
    else:
        bs = int(binsize)
    out = np.zeros(int(off[-1]), dtype=np.int32)
    for i in range(n):
        m = keep & (rid == r_rid[i])
        rel = p5[m] - r_loc[i]
        ng = neg[m]
        ok = (rel >= 0) & (rel < r_len[i])
        rel = rel[ok]; anti = ng[ok].astype(np.int64)
        if r_strand[i] < 0:
            rel = r_len[i] - rel - 1
            anti = 1 - anti
        idx = (rel // bs) * mult + (anti if ss else 0)
        np.add.at(out, off[i] + idx, 1)
    return out, off


def coverage_core(reads, ranges, tlen_filter=(), mapqual=0, requiredF=0,
                  filteredF=-1, tspan=False):
    """Flat int32 result of ``coverage_core`` (src/bamsignals.cpp:474-494)."""
    rid = np.asarray(reads["rid"]); pos = np.asarray(reads["pos"], dtype=np.int64)
    rend = np.asarray(reads["end"], dtype=np.int64)
    flag = np.asarray(reads["flag"]).astype(np.int64)
    tlen = np.asarray(reads["tlen"]).astype(np.int64)
    keep = filter_mask(flag, reads["mapq"], tlen, mapqual, requiredF, filteredF, tlen_filter)
    if tspan and (tlen_filter is None or len(tlen_filter) < 2):
        raise ValueError("tspan needs a 2-element tlen_filter")
    neg = (flag & BAM_FREVERSE) != 0
    start = pos.copy(); end = rend.copy()
    if tspan:
        a = neg & (tlen < 0)
        b = (~neg) & (tlen > 0)
        start = np.where(a, end + tlen + 1, start)
        end = np.where(b, start + tlen - 1, end)

    r_rid = np.asarray(ranges["rid"]); r_loc = np.asarray(ranges["loc"], dtype=np.int64)
    r_len = np.asarray(ranges["len"], dtype=np.int64)
    r_strand = np.asarray(ranges["strand"], dtype=np.int64)
    n = len(r_loc)
    cells, off = profile_layout(r_len, 1, False)
    out = np.zeros(int(off[-1]), dtype=np.int32)
    for i in range(n):
        L = int(r_len[i])
        if L <= 0:
            continue  # reference writes out of bounds here (UB); nothing to fill
        rend_i = r_loc[i] + L
        m = keep & (rid == r_rid[i]) & ~((start >= rend_i) | (end < r_loc[i]))
        s = start[m]; e = end[m]
        d = np.zeros(L + 1, dtype=np.int64)
        if r_strand[i] >= 0:
            a = np.maximum(s - r_loc[i], 0)
            b = e + 1 - r_loc[i]
        else:
            a = np.maximum(rend_i - 1 - e, 0)
            b = rend_i - s
        np.add.at(d, a, 1)
        b = b[b < L]
        np.add.at(d, b, -1)
        out[off[i]:off[i + 1]] = np.cumsum(d[:L]).astype(np.int32)
    return out, off


def split_signals(out, off, ss):
    """List of per-range vectors / 2 x w matrices, as the R list the reference returns."""
    sigs = []
    for i in range(len(off) - 1):
        v = out[off[i]:off[i + 1]]
        if ss:
            v = v.reshape(-1, 2).T  # column-major 2 x width
        sigs.append(v)
    return sigs
