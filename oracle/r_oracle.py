"""TEST INFRASTRUCTURE ONLY — restatement of the reference's *test* oracle.

The reference's tests do not hold literal expected numbers: they compare the
native code against an R re-implementation on a data.frame of the same reads
(``tests/testthat/utils.R:178-311``: ``df2gr``, ``countR``, ``profileR``,
``coverageR``).  This module restates those four R functions with numpy, on the
data.frame columns of ``tests/testthat/randomReads.RData`` (1-based ``pos``,
``qwidth``, ``strand``, ``isize``, ``read1``, ``mapq``).  It is deliberately
written from the R text, *not* from the C++ text, so that agreement between
this module and ``oracle_np.py`` / ``bamsignals_oracle.c`` pins the C++
restatements to the reference's own test oracle.

Coordinates here are 1-based inclusive, as in GenomicRanges.
"""
from __future__ import annotations

import numpy as np


def df2gr(df, paired_end="ignore", shift=0, mapqual=0, tlenFilter=None):
    """utils.R:178-225.  Returns dict(rname, start, end, neg) of 1-based inclusive ranges."""
    if paired_end not in ("ignore", "filter", "midpoint", "extend"):
        raise ValueError("invalid paired.end option")
    rname = np.asarray(df["rname"]); pos = np.asarray(df["pos"], dtype=np.int64)
    qwidth = np.asarray(df["qwidth"], dtype=np.int64)
    neg = np.asarray(df["neg"], dtype=bool)
    isize = np.asarray(df["isize"], dtype=np.int64)
    read1 = np.asarray(df["read1"], dtype=bool)
    mapq = np.asarray(df["mapq"], dtype=np.int64)

    keep = mapq >= mapqual                                   # :184
    if paired_end != "ignore":                               # :187-195
        keep &= read1
        lo, hi = (0, 1000) if tlenFilter is None else tlenFilter
        keep &= (np.abs(isize) >= lo) & (np.abs(isize) <= hi)
    rname, pos, qwidth, neg, isize = (a[keep] for a in (rname, pos, qwidth, neg, isize))

    if paired_end in ("extend", "midpoint"):                 # :198-204
        pos = np.where(neg, pos - np.abs(isize) + qwidth, pos)
        qwidth = np.abs(isize)

    start = pos
    end = pos + qwidth - 1

    if paired_end == "midpoint":                             # :207-218
        mids = (start + end) / 2.0
        signed = mids * np.where(neg, -1.0, 1.0)
        mids = np.abs(np.ceil(signed)).astype(np.int64)
        start = mids
        end = mids.copy()

    sh = np.where(neg, -shift, shift)                        # :221-224
    return dict(rname=rname, start=start + sh, end=end + sh, neg=neg)


def _overlapping(gr, chrom, gstart, gend):
    # findOverlaps(type="any", ignore.strand=TRUE) for one gene
    return (gr["rname"] == chrom) & (gr["start"] <= gend) & (gr["end"] >= gstart)


def countR(df, genes, ss=False, **kw):
    """utils.R:228-252.  genes: dict(chrom, start, end, neg) (1-based inclusive)."""
    gr = df2gr(df, **kw)
    n = len(genes["start"])
    res = np.zeros((2, n), dtype=np.int64)
    for g in range(n):
        gs, ge = genes["start"][g], genes["end"][g]
        ov = _overlapping(gr, genes["chrom"][g], gs, ge)
        s = gr["start"][ov]; e = gr["end"][ov]; ng = gr["neg"][ov]
        res[0, g] = np.sum((s >= gs) & (s <= ge) & ~ng)
        res[1, g] = np.sum((e >= gs) & (e <= ge) & ng)
        if genes["neg"][g]:
            res[:, g] = res[::-1, g]
    if not ss:
        return res.sum(axis=0)
    return res


def profileR(df, genes, ss=False, **kw):
    """utils.R:254-290."""
    gr = df2gr(df, **kw)
    out = []
    for g in range(len(genes["start"])):
        gs, ge = genes["start"][g], genes["end"][g]
        glen = ge - gs + 1
        ov = _overlapping(gr, genes["chrom"][g], gs, ge)
        s = gr["start"][ov]; e = gr["end"][ov]; ng = gr["neg"][ov]
        ps = s[~ng] - gs + 1
        ne = e[ng] - gs + 1
        ps = ps[(ps >= 1) & (ps <= glen)]        # factor(levels=1:gLen) drops the rest
        ne = ne[(ne >= 1) & (ne <= glen)]
        mat = np.zeros((2, glen), dtype=np.int64)
        np.add.at(mat[0], ps - 1, 1)
        np.add.at(mat[1], ne - 1, 1)
        if genes["neg"][g]:
            # rev() of the column-major 2 x gLen matrix: reverses columns AND swaps rows
            mat = mat[::-1, ::-1]
        out.append(mat.copy() if ss else mat.sum(axis=0))
    return out


def coverageR(df, genes, **kw):
    """utils.R:292-311 (GenomicRanges::coverage ignores strand)."""
    gr = df2gr(df, **kw)
    out = []
    for g in range(len(genes["start"])):
        gs, ge = genes["start"][g], genes["end"][g]
        on = gr["rname"] == genes["chrom"][g]
        s = gr["start"][on]; e = gr["end"][on]
        p = np.arange(gs, ge + 1)
        # coverage at each position = #reads with start <= p <= end
        ss_ = np.sort(s); es_ = np.sort(e)
        cov = np.searchsorted(ss_, p, side="right") - np.searchsorted(es_, p, side="left")
        if genes["neg"][g]:
            cov = cov[::-1]
        out.append(cov.astype(np.int64))
    return out
