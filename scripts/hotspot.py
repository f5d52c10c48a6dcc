#!/usr/bin/env python3
"""Diagnostic: one read hotspot (a chrM-like 16-kb reference holding 30 % of all reads) next to
uniform coverage.  Kernel time of profile / coverage / count over ranges that include it."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    from bamsignals_amd import _lib
    from bamsignals_amd.device import Context, Plan, Reads, make_params
    from bamsignals_amd.synth import synth_ranges, synth_reads, tile_ranges
    from oracle import oracle_c
    from scripts.kernel_times import time_plan
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        ctx = Context(0, stream=stream.cuda_stream)
        a = synth_reads(20_000_000, [250_000_000], seed=1, with_cigar=False)
        b = synth_reads(8_000_000, [16_569], seed=2, with_cigar=False)
        cols = {k: np.concatenate([a[k], b[k]]) for k in ("pos", "flag", "mapq", "tlen", "end")}
        ref_len = np.asarray([250_000_000, 16_569], np.int32)
        ref_off = np.asarray([0, len(a["pos"]), len(a["pos"]) + len(b["pos"])], np.int64)
        reads = Reads(ctx, ref_len, ref_off, cols["pos"], cols["flag"], cols["mapq"], cols["tlen"], end=cols["end"])
        orc = oracle_c.OracleReads(ref_off, cols["pos"], cols["end"], cols["flag"], cols["mapq"], cols["tlen"])
        rg = synth_ranges(10_000, 2000, [250_000_000], seed=5)
        hot = tile_ranges([16_569], 2000)
        hot["rid"][:] = 1
        both = {k: np.concatenate([rg[k], hot[k]]) for k in rg}
        for name, ranges in (("10k uniform ranges", rg), ("10k uniform ranges + 9 tiles of the hotspot", both)):
            for mode, mname, args in ((_lib.MODE_PROFILE, "profile", dict(binsize=1)), (_lib.MODE_COVERAGE, "coverage", dict()),
                                      (_lib.MODE_COUNT, "count", dict(binsize=-1))):
                plan = Plan(ctx, reads, ranges["rid"], ranges["loc"], ranges["len"], ranges["strand"], make_params(mode, **args))
                out = torch.empty(max(plan.cells, 4), dtype=torch.int32, device="cuda")
                ms = time_plan(torch, stream, plan, out, steps=10, warmup=2)
                got = plan.run_host()
                fn = oracle_c.coverage_core if mode == _lib.MODE_COVERAGE else oracle_c.pileup_core
                want, _ = fn(orc, ranges, **args)
                print(json.dumps(dict(case=f"{mname}: {name}", kernel_ms=ms, exact=bool(np.array_equal(got, want)))))
                plan.close()


if __name__ == "__main__":
    main()
