import os, sys, time, json
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
from bamsignals_amd.bamio import BamFile, write_columns_as_bam
from bamsignals_amd.device import Context, Reads
from bamsignals_amd.synth import synth_reads
bam = "/tmp/dd_synth.bam"
if not os.path.exists(bam):
    cols = synth_reads(50_000_000, [250_000_000])
    write_columns_as_bam(bam, ["chr1"], cols, level=1)
ctx = Context(0)
b = BamFile(bam)
os.environ["BAMSIGNALS_DEVICE_DECODE"] = "require"
for rep in range(4):
    t = time.time(); r = Reads.from_bam(ctx, b); dt = time.time() - t
    print(os.environ.get("BAMSIGNALS_BATCH_BLOCKS", "default"), rep, round(dt, 4), {k: round(v, 4) for k, v in Reads.device_decode_timing().items()})
    r.close()
