#!/usr/bin/env python3
"""Diagnostic: stage times of the device-side decode (whole file and index-driven regions) on the
5e7-read synthetic BAM of config 2 (written to /tmp on first use)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402

from bamsignals_amd.bamio import BamFile, write_columns_as_bam  # noqa: E402
from bamsignals_amd.device import Context, Reads  # noqa: E402
from bamsignals_amd.synth import synth_ranges, synth_reads  # noqa: E402

bam = "/tmp/dd_synth.bam"
if not os.path.exists(bam):
    cols = synth_reads(50_000_000, [250_000_000])
    write_columns_as_bam(bam, ["chr1"], cols, level=1)
ctx = Context(0)
b = BamFile(bam)
os.environ["BAMSIGNALS_DEVICE_DECODE"] = "require"
for rep in range(3):
    t = time.time(); r = Reads.from_bam(ctx, b); dt = time.time() - t
    print("whole", rep, round(dt, 4), {k: round(v, 4) for k, v in Reads.device_decode_timing().items()})
    r.close()
for n in (100, 1000, 10000):
    rg = synth_ranges(n, 2000, [250_000_000], seed=77)
    beg = rg["loc"].astype(np.int64) - 0
    end = beg + rg["len"]
    for mode in ("require", "0"):
        os.environ["BAMSIGNALS_DEVICE_DECODE"] = mode
        for rep in range(2):
            t = time.time(); r = Reads.from_bam_regions(ctx, b, rg["rid"], beg, end); dt = time.time() - t
            print("regions", n, "device" if mode == "require" else "cpu", rep, round(dt, 4), r.n_reads,
                  {k: round(v, 4) for k, v in Reads.device_decode_timing().items()})
            r.close()
