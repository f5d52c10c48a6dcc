#!/usr/bin/env python3
"""Diagnostic: latency of the user-level calls on BASELINE config 1's shape (the reference's fixture BAM,
99,000 reads, 50 ranges): first call, later calls (BAM resident), and the CPU oracle on resident columns."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402

from bamsignals_amd import GRanges, _lib, bamCount, bamCoverage, bamProfile  # noqa: E402
from bamsignals_amd.wrappers import last_call_timing  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
reg = json.load(open(os.path.join(G, "regions.json")))
gr = GRanges(reg["chrom"], reg["start"], width=reg["width"], strand=reg["strand"])
bam = os.path.join(G, "randomBam.bam")
_lib.load().bsig_cache_clear()
t = time.perf_counter(); bamCount(bam, gr, verbose=False); print("first bamCount (context, open, decode, run): %.1f ms" % ((time.perf_counter() - t) * 1e3))
_lib.load().bsig_cache_clear()
t = time.perf_counter(); bamCount(bam, gr, verbose=False); print("cold bamCount, context alive: %.2f ms" % ((time.perf_counter() - t) * 1e3), last_call_timing())
for name, fn, kw in (("bamCount", bamCount, {}), ("bamCount ss midpoint", bamCount, dict(ss=True, paired_end="midpoint")),
                     ("bamProfile", bamProfile, {}), ("bamProfile ss binsize 10", bamProfile, dict(ss=True, binsize=10)),
                     ("bamCoverage", bamCoverage, {})):
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ts = []
        for _ in range(200):
            t = time.perf_counter(); fn(bam, gr, verbose=False, **kw); ts.append(time.perf_counter() - t)
    print("%-26s resident: median %.3f ms, p90 %.3f ms" % (name, np.median(ts) * 1e3, np.percentile(ts, 90) * 1e3))
print("stages of the last call:", last_call_timing())
z = np.load(os.path.join(G, "fixture_reads.npz"))
from oracle import oracle_c  # noqa: E402
orc = oracle_c.OracleReads(z["ref_off"], z["bam_pos"], z["bam_end"], z["bam_flag"], z["bam_mapq"], z["bam_tlen"])
names = [str(s) for s in z["ref_names"]]
rg = dict(rid=np.asarray([names.index(c) for c in reg["chrom"]], np.int32), loc=np.asarray(reg["start"], np.int32) - 1,
          len=np.asarray(reg["width"], np.int32), strand=np.asarray([{"+": 1, "-": -1}.get(s, 0) for s in reg["strand"]], np.int32))
ts = []
for _ in range(200):
    t = time.perf_counter(); oracle_c.pileup_core(orc, rg, binsize=-1); ts.append(time.perf_counter() - t)
print("CPU oracle bamCount on resident columns: median %.3f ms" % (np.median(ts) * 1e3))
