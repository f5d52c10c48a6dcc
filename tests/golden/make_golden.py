#!/usr/bin/env python3
"""Generate the committed golden fixtures from the reference's own test data.

Runs ONLY in the dev container (needs /root/reference).  Nothing from the
reference's *sources* is copied: the outputs are data —

* ``randomBam.bam`` / ``.bai``   byte copies of the reference's data fixtures
                                 (``inst/extdata``), used to test the build's BAM
                                 reader against a file written by htslib;
* ``fixture_reads.npz``          the same 99,000 reads as columns, decoded twice
                                 and independently: from the BAM with Python's
                                 gzip/struct (``bam_*``) and from
                                 ``tests/testthat/randomReads.RData`` (``df_*``);
* ``regions.json``               ``grgenes`` (inst/extdata/randomAnnot.Rdata, 20
                                 ranges) + 30 seeded ranges drawn like
                                 tests/testthat/test_methods.R:11-20;
* ``expected_grid.npz``          outputs of the reference's TEST ORACLE
                                 (utils.R:178-311, restated in oracle/r_oracle.py)
                                 on the full parameter grid of
                                 test_methods.R:33-104;
* ``expected_extra.npz``         cases no reference test covers ('*' strand,
                                 binsize>1, filteredFlag variants, synthetic
                                 CIGARs...), from the two C++ restatements, which
                                 must agree with each other to be written.

Before anything is written, the script asserts that the C restatement
(oracle/bamsignals_oracle.c), the numpy restatement (oracle/oracle_np.py) and
the R-test-oracle restatement (oracle/r_oracle.py) agree on every grid point.
"""
from __future__ import annotations

import gzip
import itertools
import json
import os
import shutil
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from oracle import oracle_c, oracle_np, r_oracle  # noqa: E402
from rdata_reader import load_rdata  # noqa: E402

REF = "/root/reference"


def decode_bam(path):
    """BAM -> header refs + columns, with gzip/struct only (SAM spec section 4)."""
    d = gzip.open(path, "rb").read()
    assert d[:4] == b"BAM\1"
    l_text = struct.unpack_from("<i", d, 4)[0]
    o = 8 + l_text
    n_ref = struct.unpack_from("<i", d, o)[0]
    o += 4
    refs = []
    for _ in range(n_ref):
        l_name = struct.unpack_from("<i", d, o)[0]
        name = d[o + 4:o + 4 + l_name - 1].decode()
        l_ref = struct.unpack_from("<i", d, o + 4 + l_name)[0]
        refs.append((name, l_ref))
        o += 8 + l_name
    cols = {k: [] for k in ("rid", "pos", "flag", "mapq", "tlen", "bin")}
    cigar, cigar_off = [], [0]
    while o < len(d):
        bs = struct.unpack_from("<i", d, o)[0]
        rid, pos, l_rn, mapq, bin_, n_cig, flag, l_seq, nrid, npos, tlen = \
            struct.unpack_from("<iiBBHHHiiii", d, o + 4)
        co = o + 4 + 32 + l_rn
        cigar.extend(struct.unpack_from("<%dI" % n_cig, d, co))
        cigar_off.append(len(cigar))
        for k, v in zip(("rid", "pos", "flag", "mapq", "tlen", "bin"), (rid, pos, flag, mapq, tlen, bin_)):
            cols[k].append(v)
        o += 4 + bs
    out = {k: np.asarray(v, dtype=np.int32) for k, v in cols.items()}
    out["cigar"] = np.asarray(cigar, dtype=np.uint32)
    out["cigar_off"] = np.asarray(cigar_off, dtype=np.int64)
    return refs, out


def main():
    # ---- reads, decoded twice ------------------------------------------------------
    refs, bam = decode_bam(f"{REF}/inst/extdata/randomBam.bam")
    ref_names = [r[0] for r in refs]
    bam["end"] = oracle_np.cigar_end(bam["pos"], bam["flag"], bam["cigar_off"], bam["cigar"])
    assert np.array_equal(bam["end"], oracle_c.cigar_end(bam["pos"], bam["flag"], bam["cigar_off"], bam["cigar"]))
    n = len(bam["pos"])
    assert n == 99000
    # sorted by (rid, pos) as a coordinate-sorted BAM must be
    key = bam["rid"].astype(np.int64) << 32 | bam["pos"]
    assert np.all(np.diff(key) >= 0)
    ref_off = np.searchsorted(bam["rid"], np.arange(len(refs) + 1)).astype(np.int64)

    rd = load_rdata(f"{REF}/tests/testthat/randomReads.RData")["reads"]
    dfc = dict(zip(rd.attr["names"].value, rd.value))
    rn_levels = dfc["rname"].attr["levels"].value
    st_levels = dfc["strand"].attr["levels"].value
    df = dict(
        rname=np.asarray([ref_names.index(rn_levels[c - 1]) for c in dfc["rname"].value], dtype=np.int32),
        pos=dfc["pos"].value.astype(np.int32),
        qwidth=dfc["qwidth"].value.astype(np.int32),
        neg=np.asarray([st_levels[c - 1] == "-" for c in dfc["strand"].value]),
        isize=dfc["isize"].value.astype(np.int32),
        read1=dfc["read1"].value.astype(bool),
        mapq=dfc["mapq"].value.astype(np.int32),
        flag=dfc["flag"].value.astype(np.int32),
    )
    assert len(df["pos"]) == n
    # the data.frame and the BAM hold the same reads (different order)
    a = np.lexsort((df["flag"], df["isize"], df["pos"], df["rname"]))
    b = np.lexsort((bam["flag"], bam["tlen"], bam["pos"], bam["rid"]))
    assert np.array_equal(df["pos"][a] - 1, bam["pos"][b])
    assert np.array_equal(df["flag"][a], bam["flag"][b])
    assert np.array_equal(df["isize"][a], bam["tlen"][b])
    assert np.array_equal(df["mapq"][a], bam["mapq"][b])
    assert np.array_equal(df["pos"][a] + df["qwidth"][a] - 2, bam["end"][b])

    np.savez_compressed(
        f"{HERE}/fixture_reads.npz",
        ref_names=np.asarray(ref_names), ref_len=np.asarray([r[1] for r in refs], dtype=np.int32),
        ref_off=ref_off,
        **{"bam_" + k: v for k, v in bam.items()},
        **{"df_" + k: v for k, v in df.items()})
    for f in ("randomBam.bam", "randomBam.bam.bai"):
        shutil.copyfile(f"{REF}/inst/extdata/{f}", f"{HERE}/{f}")

    # ---- regions: grgenes + 30 seeded (test_methods.R:11-20 style) ----------------------
    g = load_rdata(f"{REF}/inst/extdata/randomAnnot.Rdata")["grgenes"]

    def rle(o):
        codes = o.attr["values"].value
        lv = o.attr["values"].attr["levels"].value
        return [lv[c - 1] for c, l in zip(codes, o.attr["lengths"].value) for _ in range(l)]

    chrom = rle(g.attr["seqnames"])
    strand = rle(g.attr["strand"])
    start = [int(x) for x in g.attr["ranges"].attr["start"].value]
    width = [int(x) for x in g.attr["ranges"].attr["width"].value]
    rng = np.random.default_rng(0xBA51)
    for _ in range(30):
        chrom.append(ref_names[rng.integers(3)])
        strand.append("+-"[rng.integers(2)])
        start.append(int(rng.integers(1, 1001)))
        width.append(int(1 + rng.poisson(199)))
    regions = dict(chrom=chrom, strand=strand, start=start, width=width)
    json.dump(regions, open(f"{HERE}/regions.json", "w"), indent=0)

    genes = dict(chrom=np.asarray([ref_names.index(c) for c in chrom]),
                 start=np.asarray(start), end=np.asarray(start) + np.asarray(width) - 1,
                 neg=np.asarray([s == "-" for s in strand]))
    ranges = dict(rid=genes["chrom"].astype(np.int32), loc=(np.asarray(start) - 1).astype(np.int32),
                  len=np.asarray(width, dtype=np.int32),
                  strand=np.asarray([{"+": 1, "-": -1}.get(s, 0) for s in strand], dtype=np.int32))
    reads_np = dict(rid=bam["rid"], pos=bam["pos"], end=bam["end"], flag=bam["flag"],
                    mapq=bam["mapq"], tlen=bam["tlen"])
    reads_c = oracle_c.OracleReads(ref_off, bam["pos"], bam["end"], bam["flag"], bam["mapq"], bam["tlen"])

    def r_args(pe, tf):
        """R/wrappers.R:76-98: flagMask + tlenFilter normalisation."""
        req = 66 if pe != "ignore" else 0
        tlf = () if pe == "ignore" else ((0, 1000) if tf is None else tf)
        return req, tlf

    expected = {}
    npts = 0
    # ---- bamCount / bamProfile grid: test_methods.R:33-67 ----------------------------------
    for shift, mapq, ss, pe, tf in itertools.product((0, 100), (0, 100), (False, True),
                                                     ("ignore", "filter", "midpoint"),
                                                     (None, (50, 200))):
        req, tlf = r_args(pe, tf)
        key = f"shift={shift},mapq={mapq},ss={int(ss)},pe={pe},tf={'NULL' if tf is None else '50_200'}"
        kw = dict(shift=shift, paired_end=pe, mapqual=mapq, tlenFilter=tf)
        # count
        cr = r_oracle.countR(df, genes, ss=ss, **kw)
        want = (cr.T.reshape(-1) if ss else cr).astype(np.int32)
        args = dict(tlen_filter=tlf, mapqual=mapq, binsize=-1, shift=shift, ss=ss,
                    requiredF=req, filteredF=-1, pe_mid=(pe == "midpoint"))
        got_np, _ = oracle_np.pileup_core(reads_np, ranges, **args)
        got_c, _ = oracle_c.pileup_core(reads_c, ranges, **args)
        assert np.array_equal(want, got_np), ("count np", key)
        assert np.array_equal(want, got_c), ("count c", key)
        expected["count|" + key] = want
        # profile (binsize=1)
        pr = r_oracle.profileR(df, genes, ss=ss, **kw)
        want = np.concatenate([(m.T.reshape(-1) if ss else m) for m in pr]).astype(np.int32)
        args["binsize"] = 1
        got_np, off = oracle_np.pileup_core(reads_np, ranges, **args)
        got_c, off_c = oracle_c.pileup_core(reads_c, ranges, **args)
        assert np.array_equal(off, off_c)
        assert np.array_equal(want, got_np), ("profile np", key)
        assert np.array_equal(want, got_c), ("profile c", key)
        expected["profile|" + key] = want
        npts += 2
    # ---- bamCoverage grid: test_methods.R:70-83 ---------------------------------------------
    for mapq, pe, tf in itertools.product((0, 100), ("ignore", "extend"), (None, (50, 200))):
        req, tlf = r_args(pe, tf)
        key = f"mapq={mapq},pe={pe},tf={'NULL' if tf is None else '50_200'}"
        cv = r_oracle.coverageR(df, genes, paired_end=pe, mapqual=mapq, tlenFilter=tf)
        want = np.concatenate(cv).astype(np.int32)
        args = dict(tlen_filter=tlf, mapqual=mapq, requiredF=req, filteredF=-1, tspan=(pe == "extend"))
        got_np, _ = oracle_np.coverage_core(reads_np, ranges, **args)
        got_c, _ = oracle_c.coverage_core(reads_c, ranges, **args)
        assert np.array_equal(want, got_np), ("coverage np", key)
        assert np.array_equal(want, got_c), ("coverage c", key)
        expected["coverage|" + key] = want
        npts += 1
    # ---- filteredFlag=16 with all-'+' regions == sense row: test_methods.R:85-104 ------------
    genes_plus = dict(genes, neg=np.zeros(len(start), dtype=bool))
    ranges_plus = dict(ranges, strand=np.ones(len(start), dtype=np.int32))
    for shift, mapq, pe, tf in itertools.product((0, 100), (0, 100), ("ignore", "filter", "midpoint"),
                                                 (None, (50, 200))):
        req, tlf = r_args(pe, tf)
        key = f"shift={shift},mapq={mapq},pe={pe},tf={'NULL' if tf is None else '50_200'}"
        want = r_oracle.countR(df, genes_plus, ss=True, shift=shift, paired_end=pe, mapqual=mapq,
                               tlenFilter=tf)[0].astype(np.int32)
        args = dict(tlen_filter=tlf, mapqual=mapq, binsize=-1, shift=shift, ss=False,
                    requiredF=req, filteredF=16, pe_mid=(pe == "midpoint"))
        got_np, _ = oracle_np.pileup_core(reads_np, ranges_plus, **args)
        got_c, _ = oracle_c.pileup_core(reads_c, ranges_plus, **args)
        assert np.array_equal(want, got_np), ("ff16 np", key)
        assert np.array_equal(want, got_c), ("ff16 c", key)
        expected["ff16|" + key] = want
        npts += 1
    np.savez_compressed(f"{HERE}/expected_grid.npz", **expected)
    print(f"grid: {npts} points pinned against the R test oracle; "
          f"default profile total = {int(expected['profile|shift=0,mapq=0,ss=0,pe=ignore,tf=NULL'].sum())}")

    # ---- extra cases (not covered by any reference test; np == c required) -------------------
    extra = {}
    ranges_star = dict(ranges, strand=np.asarray([(1, -1, 0)[i % 3] for i in range(len(start))], dtype=np.int32))

    def both(kind, name, rg, **args):
        f_np = oracle_np.pileup_core if kind == "pileup" else oracle_np.coverage_core
        f_c = oracle_c.pileup_core if kind == "pileup" else oracle_c.coverage_core
        a, _ = f_np(reads_np, rg, **args)
        b, _ = f_c(reads_c, rg, **args)
        assert np.array_equal(a, b), name
        extra[name] = a

    both("pileup", "profile_star_bs1", ranges_star, binsize=1)
    both("pileup", "profile_star_bs7_ss", ranges_star, binsize=7, ss=True)
    both("pileup", "profile_bs50_shift-30", ranges, binsize=50, shift=-30)
    both("pileup", "profile_ff0", ranges, binsize=1, filteredF=0)
    both("pileup", "profile_ff1024", ranges, binsize=1, filteredF=1024)
    both("pileup", "profile_ff1040", ranges, binsize=1, filteredF=1040)
    both("pileup", "count_star_ss", ranges_star, binsize=-1, ss=True)
    both("pileup", "profile_mid_bs3_ss", ranges_star, binsize=3, ss=True, requiredF=66,
         tlen_filter=(30, 300), pe_mid=True, shift=5)
    both("coverage", "coverage_star", ranges_star)
    both("coverage", "coverage_star_extend", ranges_star, requiredF=66, tlen_filter=(0, 1000), tspan=True)
    # binsize=7 ss result == per-base ss result summed in range orientation (survey App. B)
    pb, off1 = oracle_np.pileup_core(reads_np, ranges_star, binsize=1, ss=True)
    b7, off7 = oracle_np.pileup_core(reads_np, ranges_star, binsize=7, ss=True)
    for i in range(len(start)):
        m = pb[off1[i]:off1[i + 1]].reshape(-1, 2)
        w = m.shape[0]
        pad = (-w) % 7
        m = np.concatenate([m, np.zeros((pad, 2), dtype=m.dtype)]).reshape(-1, 7, 2).sum(axis=1)
        assert np.array_equal(m.reshape(-1), b7[off7[i]:off7[i + 1]])
    assert int(extra["profile_ff0"].sum()) == 0
    np.savez_compressed(f"{HERE}/expected_extra.npz", **extra)
    print("extra:", {k: int(v.sum()) for k, v in extra.items()})


if __name__ == "__main__":
    main()
