#!/usr/bin/env python3
"""Diagnostic: profile / coverage kernel time on DENSE data (several thousand reads per tile)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    from bamsignals_amd import _lib
    from bamsignals_amd.device import Context, Plan, Reads, make_params
    from bamsignals_amd.synth import synth_reads, tile_ranges
    from scripts.kernel_times import time_plan
    L = 30_000_000
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        ctx = Context(0, stream=stream.cuda_stream)
        for n in (30_000_000, 100_000_000):
            cols = synth_reads(n, [L], seed=3, with_cigar=False)
            reads = Reads(ctx, cols["ref_len"], cols["ref_off"], cols["pos"], cols["flag"], cols["mapq"], cols["tlen"], end=cols["end"])
            tiles = tile_ranges([L], 2000)
            for mode, name in ((_lib.MODE_PROFILE, "profile"), (_lib.MODE_COVERAGE, "coverage")):
                plan = Plan(ctx, reads, tiles["rid"], tiles["loc"], tiles["len"], tiles["strand"], make_params(mode, binsize=1))
                out = torch.empty(plan.cells, dtype=torch.int32, device="cuda")
                ms = time_plan(torch, stream, plan, out)
                st = plan.stats()
                print(json.dumps(dict(case=f"{name}, {n / L:.1f} reads/bp ({st['visits'] // st['n_items']} reads per tile)",
                                      kernel_ms=ms, GBps=st["algorithmic_bytes"] / ms / 1e6)))
                plan.close()
            reads.close()


if __name__ == "__main__":
    main()
