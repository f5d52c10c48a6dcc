#!/usr/bin/env python3
"""Diagnostic (and the rocprofv3 target of scripts/profile_round.sh `decodereal`): the device-side decode of a
REAL-SHAPED BAM -- 2e7 single-end 100-bp reads with names, bases, qualities and an NM tag (204-byte records,
2.1 GB file, 4.1 GB of stream in 62,000 literal-heavy blocks) -- with the GPU inflating, four times."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402

from bamsignals_amd.bamio import BamFile, write_columns_as_bam  # noqa: E402
from bamsignals_amd.device import Context, Reads  # noqa: E402
from bamsignals_amd.synth import synth_reads  # noqa: E402

bam = "/tmp/dd_real.bam"
if not os.path.exists(bam):
    cols = synth_reads(20_000_000, [250_000_000], seed=12)
    write_columns_as_bam(bam, ["chr1"], cols, level=1, l_seq=100, seed=3)
    del cols
ctx = Context(0)
b = BamFile(bam)
os.environ["BAMSIGNALS_DEVICE_DECODE"] = "require"
os.environ["BAMSIGNALS_INFLATE"] = "gpu"
stream_bytes = 20_000_000 * 204 + 1000
for rep in range(4):
    t = time.time(); r = Reads.from_bam(ctx, b); dt = time.time() - t
    d = Reads.device_decode_timing()
    print(json.dumps(dict(rep=rep, decode_s=round(dt, 4), file_bytes=os.path.getsize(bam), stream_bytes=stream_bytes,
                          inflate_output_GBps=round(stream_bytes / d["inflate"] / 1e9, 1), **{k: round(v, 4) for k, v in d.items()})), flush=True)
    r.close()
