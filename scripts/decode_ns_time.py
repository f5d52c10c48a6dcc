#!/usr/bin/env python3
"""Diagnostic: the cold decode of the north-star BAM (5e8 bare reads, 3 GB file, 26 GB of stream)
three times in a row, with stage times and BSIG_DIAG_INFLATE lines."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402

from bamsignals_amd.bamio import BamFile, write_columns_as_bam  # noqa: E402
from bamsignals_amd.device import Context, Reads  # noqa: E402
from bamsignals_amd.synth import synth_reads  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000_000
bam = "/tmp/ns_synth.bam"
keep = os.environ.get("BSIG_KEEP_BAM") == "1"            # (A/B runs: several processes over the same file)
if not (keep and os.path.exists(bam) and os.path.exists(bam + ".bai")):
    t = time.time(); cols = synth_reads(n, [250_000_000] * 10, seed=9); print("generate", round(time.time() - t, 1), flush=True)
    t = time.time(); write_columns_as_bam(bam, ["c%d" % i for i in range(10)], cols, level=1); print("write", round(time.time() - t, 1), os.path.getsize(bam), flush=True)
    del cols
ctx = Context(0)
b = BamFile(bam)
os.environ["BSIG_DIAG_DECODE"] = "1"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
for rep in range(reps):
    t = time.time(); r = Reads.from_bam(ctx, b); dt = time.time() - t
    print("decode", rep, round(dt, 3), {k: round(v, 3) for k, v in Reads.device_decode_timing().items()}, flush=True)
    t = time.time(); r.close(); print("  free of the resident reads", round((time.time() - t) * 1e3, 1), "ms", flush=True)
if not keep:
    os.remove(bam); os.remove(bam + ".bai")
