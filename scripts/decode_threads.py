#!/usr/bin/env python3
"""Diagnostic: whole-file decode time of a synthetic BAM against the number of decode threads."""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bamsignals_amd.bamio import BamFile, write_columns_as_bam  # noqa: E402
from bamsignals_amd.synth import synth_reads  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
d = tempfile.mkdtemp(prefix="bsig_dec_", dir="/tmp")
bam = os.path.join(d, "s.bam")
cols = synth_reads(n, [250_000_000])
t = time.time(); write_columns_as_bam(bam, ["chr1"], cols, level=1); print("write_s", round(time.time() - t, 2), os.path.getsize(bam))
b = BamFile(bam)
b.decode(threads=8)
for th in (8, 16, 32, 48):
    best = None
    for _ in range(2):
        b.decode(threads=th)
        tm = b.decode_timing()
        if best is None or tm["total"] < best["total"]:
            best = tm
    print(th, {k: round(v, 3) for k, v in best.items()})
os.remove(bam); os.remove(bam + ".bai")
