#!/usr/bin/env python3
"""Diagnostic: the warm file-level call at the north star's shape (BAM resident in HBM: plan + kernels + download of the
800-MB result into a fresh host buffer) under environment variants, alternating, one fresh process per variant and
round (its first call is the cold one; the five behind it are printed).
VARIANTS="BAMSIGNALS_COPY_THREADS=8;BAMSIGNALS_COPY_THREADS=16" (default)  python scripts/warm_call_ab.py [reads] [rounds]
(Round 4: huge pages requested for the destination, madvise(MADV_HUGEPAGE), changed nothing -- the download already
runs at 52 GB/s of the link's 56: 0.0335 s against 0.0333 s.)"""
import os
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402
from bamsignals_amd.bamio import write_columns_as_bam  # noqa: E402
from bamsignals_amd.synth import synth_ranges, synth_reads  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000_000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ref = [250_000_000] * 10
cols = synth_reads(n, ref, seed=9)
d = tempfile.mkdtemp(prefix="bsig_warm_", dir="/tmp")
bam = os.path.join(d, "ns.bam")
names = ["c%d" % i for i in range(10)]
write_columns_as_bam(bam, names, cols, level=1)
del cols
bench._settle(bam)
rg = synth_ranges(100_000, 2000, ref, seed=10)
call = dict(tlen_filter=(), device=0)
variants = [dict(kv.split("=", 1) for kv in v.split(",") if kv)
            for v in os.environ.get("VARIANTS", "BAMSIGNALS_COPY_THREADS=8;BAMSIGNALS_COPY_THREADS=16").split(";")]
res = {i: [] for i in range(len(variants))}
_real = sys.stderr
for r in range(rounds):
    for i, v in enumerate(variants):
        sys.stderr = open(os.devnull, "w")
        try:
            child, _ = bench.cold_call_in_fresh_process(d, "w", bam, names, rg, call, 0, env=v, reps=6, want_result=False)
        finally:
            sys.stderr.close()
            sys.stderr = _real
        warm = child["calls"][1:]
        res[i] += [c["call_s"] for c in warm]
        print(v, "round", r, "warm calls", [round(c["call_s"], 4) for c in warm], "download", [round(c["stages_s"].get("download", 0), 4) for c in warm],
              "cold", round(child["calls"][0]["call_s"], 3), flush=True)
        if r == 0:
            print("   stages of the last warm call:", {k: (round(x, 4) if isinstance(x, float) else x) for k, x in warm[-1]["stages_s"].items()}, flush=True)
for i, v in enumerate(variants):
    print(v, "median warm call %.4f s" % float(np.median(res[i])))
shutil.rmtree(d, ignore_errors=True)
