#!/usr/bin/env python3
"""Diagnostic: BASELINE config 5's call through the single-process multi-GPU route at full size --
1e9 SE reads on 24 references (3.1 Gbp) as a BAM file, 1,000,000 x 1 kb ranges, bamProfile(binsize=1) --
with the box's ONE GPU listed several times (4 slots by default: every slot keeps its own resident copy,
12.5 GB, and its share's scratch).  Checks the whole 4-GB result against the C oracle."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402

from bamsignals_amd import GRanges, _lib, bamProfile, write_columns_as_bam  # noqa: E402
from bamsignals_amd.synth import synth_ranges, synth_reads  # noqa: E402
from bamsignals_amd.wrappers import last_call_route, last_call_timing  # noqa: E402
from oracle import oracle_c  # noqa: E402

HG38 = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636,
        138394717, 133797422, 135086622, 133275309, 114364328, 107043718, 101991189, 90338345,
        83257441, 80373285, 58617616, 64444167, 46709983, 50818468, 156040895, 57227415]
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
slots = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n_ranges = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
names = ["chr%d" % (i + 1) for i in range(22)] + ["chrX", "chrY"]
t = time.time(); cols = synth_reads(n_reads, HG38, seed=0xC5); print("generate", round(time.time() - t, 1), flush=True)
bam = "/tmp/c5_synth.bam"
t = time.time(); write_columns_as_bam(bam, names, cols, level=1); print("write", round(time.time() - t, 1), os.path.getsize(bam), flush=True)
cols.pop("cigar"); cols.pop("cigar_off")
rg = synth_ranges(n_ranges, 1000, HG38, seed=0xC6)
if os.environ.get("C5_SORTED", "1") != "0":
    # the caller's ranges in (chromosome, start) order, as a tiling or a sorted peak list comes: the order in which
    # every block of the sharding is one contiguous slice of the result ("blocks")
    o = np.lexsort((rg["loc"], rg["rid"]))
    rg = {k: v[o] for k, v in rg.items()}
t = time.time()
gr = GRanges([names[r] for r in rg["rid"]], rg["loc"] + 1, width=rg["len"], strand=[{1: "+", -1: "-", 0: "*"}[int(s)] for s in rg["strand"]])
print("GRanges", round(time.time() - t, 1), flush=True)
os.environ["BAMSIGNALS_DEVICES"] = ",".join(["0"] * slots)
os.environ["BAMSIGNALS_DECODE"] = "all"
res = {}
want = None
for gather in ("xgmi", "pcie", "blocks", ""):   # (same-device slots: "xgmi" = the first GPU reads the shards in place; "": the library's own choice)
    os.environ["BAMSIGNALS_GATHER"] = gather
    for rep in (("cold", "resident", "resident") if gather == "xgmi" else ("resident", "resident")):
        if rep == "cold":
            _lib.load().bsig_cache_clear()
        t = time.time(); sig = bamProfile(bam, gr, verbose=False); dt = time.time() - t
        print(gather, rep, round(dt, 3), "s", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in last_call_timing().items()}, "|", last_call_route(), flush=True)
        res[gather] = sig
    flat = np.concatenate(res[gather].as_list())
    if want is None:
        t = time.time()
        orc = oracle_c.OracleReads(cols["ref_off"], cols["pos"], cols["end"], cols["flag"], cols["mapq"], cols["tlen"])
        want, _ = oracle_c.pileup_core(orc, rg, binsize=1)
        print("oracle", round(time.time() - t, 1), "s;", len(want), "cells", flush=True)
    assert np.array_equal(flat, want), gather
    print(gather, "identical to the oracle (%d cells, sum %d)" % (len(flat), int(flat.astype(np.int64).sum())), flush=True)
    del flat
_lib.load().bsig_cache_clear()
os.remove(bam); os.remove(bam + ".bai")
