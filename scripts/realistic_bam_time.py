#!/usr/bin/env python3
"""Diagnostic: decode stage times on a BAM that carries sequences and qualities (literal-heavy
DEFLATE blocks, 250-byte records) instead of bench.py's 52-byte records: GPU vs CPU inflate."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402

from bamsignals_amd.bamio import BamFile, writeSamAsBamAndIndex  # noqa: E402
from bamsignals_amd.device import Context, Reads  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000
bam = "/tmp/real_synth_%d.bam" % n
if not os.path.exists(bam):
    rng = np.random.default_rng(7)
    L = 100
    pos = np.sort(rng.integers(1, 200_000_000, n))
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)
    seq = bases[rng.integers(0, 4, (n, L))]
    # qualities: mostly high, with a tail -- about 3 bits of entropy per base like Illumina binned data
    q = np.clip(np.round(40 - np.abs(rng.normal(0, 6, (n, L)))), 2, 41).astype(np.uint8) + 33
    flag = rng.choice([0, 16], n)
    mapq = rng.integers(0, 61, n)
    t = time.time()
    with open(bam + ".sam", "w") as f:
        f.write("@HD\tVN:1.0\tSO:coordinate\n@SQ\tSN:chr1\tLN:200000000\n")
        for i in range(n):
            f.write("r%d\t%d\tchr1\t%d\t%d\t100M\t*\t0\t0\t%s\t%s\tNM:i:%d\n" % (i, flag[i], pos[i], mapq[i], seq[i].tobytes().decode(), q[i].tobytes().decode(), i % 5))
    print("sam written", round(time.time() - t, 1), flush=True)
    t = time.time(); writeSamAsBamAndIndex(bam + ".sam", bam); print("bam written", round(time.time() - t, 1), os.path.getsize(bam), flush=True)
    os.remove(bam + ".sam")
ctx = Context(0)
b = BamFile(bam)
for eng in ("gpu", "cpu"):
    os.environ["BAMSIGNALS_INFLATE"] = eng
    os.environ["BAMSIGNALS_DEVICE_DECODE"] = "require"
    for rep in range(3):
        t = time.time(); r = Reads.from_bam(ctx, b); dt = time.time() - t
        print(eng, rep, round(dt, 4), r.n_reads, {k: round(v, 4) for k, v in Reads.device_decode_timing().items()}, flush=True)
        r.close()
