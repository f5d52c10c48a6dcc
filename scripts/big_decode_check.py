#!/usr/bin/env python3
"""Diagnostic: the device-side decode at scale -- a 5e8-read synthetic BAM (3 GB file, 26 GB of
uncompressed stream: four 8-GiB passes through HBM, GPU inflate) against the CPU decode of the same
file: identical resident reads and identical results on a batch of ranges."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402

from bamsignals_amd import _lib  # noqa: E402
from bamsignals_amd.bamio import BamFile, write_columns_as_bam  # noqa: E402
from bamsignals_amd.device import Context, Plan, Reads, make_params  # noqa: E402
from bamsignals_amd.synth import synth_ranges, synth_reads  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000_000
ref_len = [250_000_000] * 10
bam = "/tmp/big_synth.bam"
t = time.time(); cols = synth_reads(n, ref_len, seed=9); print("generate", round(time.time() - t, 1), flush=True)
t = time.time(); write_columns_as_bam(bam, ["c%d" % i for i in range(10)], cols, level=1); print("write", round(time.time() - t, 1), os.path.getsize(bam), flush=True)
del cols
ctx = Context(0)
b = BamFile(bam)
rg = synth_ranges(20000, 2000, ref_len, seed=3)
res = {}
for mode in ("require", "0"):
    os.environ["BAMSIGNALS_DEVICE_DECODE"] = mode
    t = time.time(); r = Reads.from_bam(ctx, b); dt = time.time() - t
    print("device" if mode == "require" else "cpu", round(dt, 3), r.info(), {k: round(v, 3) for k, v in Reads.device_decode_timing().items()}, flush=True)
    p = Plan(ctx, r, rg["rid"], rg["loc"], rg["len"], rg["strand"], make_params(_lib.MODE_PROFILE, binsize=1, ss=True, shift=5))
    res[mode] = (r.info(), p.run_host().copy())
    p.close(); r.close()
assert res["require"][0] == res["0"][0]
assert np.array_equal(res["require"][1], res["0"][1])
print("identical")
os.remove(bam); os.remove(bam + ".bai")
