#!/usr/bin/env python3
"""Diagnostic: k_inflate against the CPU decode on BAMs of several shapes -- bare records and records with names,
bases and qualities of 36 / 100 / 250 bp, single- and paired-end, written at zlib levels 1, 6 and 9 (different
match lengths, distances and code tables) -- a few million reads each: identical resident reads and results."""
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

ctx = Context(0)
ref_len = [40_000_000, 3_000_000, 25_000_000]
names = ["a", "b", "c"]
rg = synth_ranges(3000, 1500, ref_len, seed=5, jitter=700)
n_cases = 0
for paired in (False, True):
    cols = synth_reads(3_000_000, ref_len, seed=21 + paired, paired=paired)
    for l_seq in (0, 36, 100, 250):
        for level in (1, 6, 9):
            if l_seq == 250 and level == 9:
                continue                                    # (minutes of zlib on the host)
            bam = "/tmp/shape.bam"
            t = time.time()
            write_columns_as_bam(bam, names, dict(cols), level=level, l_seq=l_seq, seed=3)
            tw = time.time() - t
            b = BamFile(bam)
            res = {}
            for how, env in (("gpu inflate", dict(BAMSIGNALS_DEVICE_DECODE="require", BAMSIGNALS_INFLATE="gpu")),
                             ("cpu decode", dict(BAMSIGNALS_DEVICE_DECODE="0"))):
                for k in ("BAMSIGNALS_DEVICE_DECODE", "BAMSIGNALS_INFLATE"):
                    os.environ.pop(k, None)
                os.environ.update(env)
                t = time.time(); r = Reads.from_bam(ctx, b); dt = time.time() - t
                p = Plan(ctx, r, rg["rid"], rg["loc"], rg["len"], rg["strand"], make_params(_lib.MODE_PROFILE, binsize=1, ss=True, shift=3))
                res[how] = (r.info(), p.run_host().copy(), dt, dict(Reads.device_decode_timing()))
                p.close(); r.close()
            assert res["gpu inflate"][0] == res["cpu decode"][0], (paired, l_seq, level)
            assert np.array_equal(res["gpu inflate"][1], res["cpu decode"][1]), (paired, l_seq, level)
            d = res["gpu inflate"][3]
            print("paired" if paired else "single", "l_seq %3d" % l_seq, "level", level, "file %5.0f MB (written in %4.1f s):" % (os.path.getsize(bam) / 1e6, tw),
                  "gpu-inflate decode %.3f s (inflate %.3f), cpu decode %.3f s -- identical" % (res["gpu inflate"][2], d["inflate"], res["cpu decode"][2]), flush=True)
            b.close()
            os.remove(bam); os.remove(bam + ".bai")
            n_cases += 1
print(n_cases, "shapes: k_inflate == CPU decode")
