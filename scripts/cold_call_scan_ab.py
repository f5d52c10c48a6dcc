#!/usr/bin/env python3
"""Diagnostic: the cold file-level call of the north star's BAM in fresh child processes (bench.py's
cold_call_in_fresh_process) with the block table built through pread() or through the populated mapping,
on a file that is settled in the page cache (flushed, read twice) -- which walk should be the default?"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402
from bamsignals_amd.bamio import write_columns_as_bam  # noqa: E402
from bamsignals_amd.synth import synth_ranges, synth_reads  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000_000
ref = [250_000_000] * 10
cols = synth_reads(n, ref, seed=9)
d = tempfile.mkdtemp(prefix="bsig_ab_", dir="/tmp")
bam = os.path.join(d, "ns.bam")
names = ["c%d" % i for i in range(10)]
write_columns_as_bam(bam, names, cols, level=1)
del cols
bench._settle(bam)
rg = synth_ranges(100_000, 2000, ref, seed=10)
call = dict(tlen_filter=(), device=0)
variants = [("pread", {"BAMSIGNALS_SCAN": "pread"}, 0), ("mmap", {"BAMSIGNALS_SCAN": "mmap"}, 0), ("pread + 44-GB arena", {"BAMSIGNALS_SCAN": "pread"}, 44)]
for rnd in range(3):
    for how, env, arena in variants:
        child, flat = bench.cold_call_in_fresh_process(d, "ab", bam, names, rg, call, 0, env=env, reps=1, arena_gb=arena)
        c = child["calls"][0]
        dd = c["stages_s"]["decode_stages_s"]
        print(how, rnd, "call %.3f s" % c["call_s"], "context %.2f s" % child["hip_context_s"], {k: round(v, 3) for k, v in dd.items()},
              "plan+run+download %.3f" % c["stages_s"]["plan_run_download"], flush=True)
import shutil  # noqa: E402
shutil.rmtree(d, ignore_errors=True)
