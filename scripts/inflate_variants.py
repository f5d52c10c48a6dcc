#!/usr/bin/env python3
"""Diagnostic: k_inflate on a bare-record and a real-shaped BAM (BSIG_LIB_PATH selects the build, e.g. one
made with -DBSIG_LFAST=7 or 9).  Prints the inflate stage times."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402

from bamsignals_amd.bamio import BamFile, write_columns_as_bam  # noqa: E402
from bamsignals_amd.device import Context, Reads  # noqa: E402
from bamsignals_amd.synth import synth_reads  # noqa: E402

os.environ["BAMSIGNALS_INFLATE"] = "gpu"
os.environ["BAMSIGNALS_DEVICE_DECODE"] = "require"
ctx = Context(0)
variants = [("160 lanes/CU (old LDS footprint)", {"BAMSIGNALS_INFLATE_LDS_PAD": "416"}), ("288 lanes/CU", {}),
            ("64-lane waves, 256 lanes/CU", {"BAMSIGNALS_INFLATE_LANES": "64"}),
            ("64-lane waves, 128 lanes/CU", {"BAMSIGNALS_INFLATE_LANES": "64", "BAMSIGNALS_INFLATE_LDS_PAD": "416"})]
for tag, n, l_seq in (("bare", 100_000_000, 0), ("real", 20_000_000, 100)):
    path = f"/tmp/iv_{tag}.bam"
    if not os.path.exists(path):
        cols = synth_reads(n, [250_000_000], seed=12)
        write_columns_as_bam(path, ["c"], cols, l_seq=l_seq, seed=3)
        del cols
    b = BamFile(path)
    for name, env in variants + variants[:2]:
        os.environ.update(env)
        for rep in range(2):
            t = time.time(); r = Reads.from_bam(ctx, b); dt = time.time() - t
            d = Reads.device_decode_timing()
            print(tag, "|", name, "|", rep, round(dt, 4), "inflate", round(d["inflate"], 4), "wait", round(d["copy_wait"], 4), "parse", round(d["gpu_parse"], 4), flush=True)
            r.close()
        for k in env:
            os.environ.pop(k)
    b.close()
