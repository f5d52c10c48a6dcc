#!/usr/bin/env python3
"""Kernel times and algorithmic-byte rates for the non-headline configurations (coverage tiling,
paired-end strand-split profile, count).  Diagnostic companion of bench.py; prints JSON lines."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def time_plan(torch, stream, plan, out, steps=30, warmup=5):
    for _ in range(warmup):
        plan.run_device(out.data_ptr())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(steps):
        plan.run_device(out.data_ptr())
    e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


def main():
    import torch

    from bamsignals_amd import _lib
    from bamsignals_amd.device import Context, Plan, Reads, make_params
    from bamsignals_amd.synth import synth_ranges, synth_reads, tile_ranges

    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        ctx = Context(0, stream=stream.cuda_stream)
        L = 248_956_422
        cases = []
        cols = synth_reads(100_000_000, [L], seed=3, with_cigar=False)
        reads = Reads(ctx, cols["ref_len"], cols["ref_off"], cols["pos"], cols["flag"], cols["mapq"], cols["tlen"], end=cols["end"])
        tiles = tile_ranges([L], 2000)
        cases.append(("C3 coverage, chr1 2-kb tiling, 1e8 SE reads", reads, tiles, make_params(_lib.MODE_COVERAGE)))
        cases.append(("profile binsize=1 on the same tiling", reads, tiles, make_params(_lib.MODE_PROFILE, binsize=1)))
        cases.append(("count on the same tiling", reads, tiles, make_params(_lib.MODE_COUNT, binsize=-1)))
        one = dict(rid=np.zeros(1, np.int32), loc=np.zeros(1, np.int32), len=np.asarray([L], np.int32), strand=np.zeros(1, np.int32))
        cases.append(("coverage of ONE whole-chromosome range (tiled internally)", reads, one, make_params(_lib.MODE_COVERAGE)))
        cases.append(("genome-wide 8-bp bins: ONE chr1 range, binsize=8", reads, one, make_params(_lib.MODE_PROFILE, binsize=8)))
        cases.append(("genome-wide 32-bp bins: ONE chr1 range, binsize=32", reads, one, make_params(_lib.MODE_PROFILE, binsize=32)))
        cases.append(("genome-wide 64-bp bins: ONE chr1 range, binsize=64", reads, one, make_params(_lib.MODE_PROFILE, binsize=64)))
        cases.append(("genome-wide 200-bp bins, ss: ONE chr1 range, binsize=200", reads, one, make_params(_lib.MODE_PROFILE, binsize=200, ss=True)))
        cases.append(("genome-wide 1-kb bins: ONE chr1 range, binsize=1000", reads, one, make_params(_lib.MODE_PROFILE, binsize=1000)))
        cases.append(("genome-wide 100-kb bins: ONE chr1 range, binsize=100000", reads, one, make_params(_lib.MODE_PROFILE, binsize=100000)))
        for name, rd, rg, prm in cases:
            plan = Plan(ctx, rd, rg["rid"], rg["loc"], rg["len"], rg["strand"], prm)
            out = torch.empty(max(plan.cells, 4), dtype=torch.int32, device="cuda")
            ms = time_plan(torch, stream, plan, out)
            st = plan.stats()
            print(json.dumps(dict(case=name, kernel_ms=ms, algorithmic_GBps=st["algorithmic_bytes"] / ms / 1e6,
                                  frac_of_8TBps=st["algorithmic_bytes"] / ms / 1e6 / 8000, **st)))
            plan.close()
        reads.close()
        del cols
        cols = synth_reads(100_000_000, [250_000_000], seed=9, paired=True, with_cigar=False)
        reads = Reads(ctx, cols["ref_len"], cols["ref_off"], cols["pos"], cols["flag"], cols["mapq"], cols["tlen"], end=cols["end"])
        rg = synth_ranges(100_000, 2000, [250_000_000], seed=10)
        for mid in (False, True):
            prm = make_params(_lib.MODE_PROFILE, binsize=1, ss=True, shift=75, requiredF=66, tlen_filter=(50, 500), pe_mid=mid)
            plan = Plan(ctx, reads, rg["rid"], rg["loc"], rg["len"], rg["strand"], prm)
            out = torch.empty(max(plan.cells, 4), dtype=torch.int32, device="cuda")
            ms = time_plan(torch, stream, plan, out)
            st = plan.stats()
            print(json.dumps(dict(case=f"C4 call (PE {'midpoint' if mid else 'filter'}, tlenFilter 50-500, shift 75, ss), 100k x 2kb, 1e8 PE reads",
                                  kernel_ms=ms, algorithmic_GBps=st["algorithmic_bytes"] / ms / 1e6,
                                  frac_of_8TBps=st["algorithmic_bytes"] / ms / 1e6 / 8000, **st)))
            plan.close()


if __name__ == "__main__":
    main()
