#!/usr/bin/env python3
"""Diagnostic: repeated bamCount / bamProfile on 50 ranges of a 5e7-read BAM (index-driven decode): the first
call decodes the BAI islands of the ranges, the later ones find those reads resident (BAMSIGNALS_REGION_CACHE);
with the cache off every call decodes again, as the reference does (ref: src/bamsignals.cpp:252-271)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402

from bamsignals_amd import GRanges, _lib, bamCount, bamProfile, write_columns_as_bam  # noqa: E402
from bamsignals_amd.synth import synth_ranges, synth_reads  # noqa: E402
from bamsignals_amd.wrappers import last_call_route, last_call_timing  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
ref = [250_000_000]
cols = synth_reads(n, ref, seed=21)
bam = "/tmp/region_cache.bam"
write_columns_as_bam(bam, ["chr1"], cols)
del cols
rg = synth_ranges(50, 2000, ref, seed=22)
gr = GRanges(["chr1"] * 50, rg["loc"] + 1, width=rg["len"], strand=["+"] * 50)
for cache in ("8", "0"):
    os.environ["BAMSIGNALS_REGION_CACHE"] = cache
    _lib.load().bsig_cache_clear()
    t = time.perf_counter(); bamProfile(bam, gr, shift=50, verbose=False); dt = time.perf_counter() - t
    print(f"cache={cache} first call (bamProfile shift=50): {dt * 1e3:.2f} ms; library {last_call_timing()['total'] * 1e3:.2f} ms | {last_call_route()}")
    ts, lib = [], []
    for _ in range(100):
        t = time.perf_counter(); bamCount(bam, gr, verbose=False); ts.append(time.perf_counter() - t)
        lib.append(last_call_timing()["total"])
    print(f"cache={cache} repeated bamCount: median {np.median(ts) * 1e3:.3f} ms (library {np.median(lib) * 1e3:.3f} ms, p90 {np.percentile(lib, 90) * 1e3:.3f}) | {last_call_route()}")
_lib.load().bsig_cache_clear()
os.remove(bam); os.remove(bam + ".bai")
