#!/usr/bin/env python3
"""Diagnostic: 1,000,000 ranges of 100 bp (peak / TSS windows) — count, profile, coverage."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    from bamsignals_amd import _lib
    from bamsignals_amd.device import Context, Plan, Reads, make_params
    from bamsignals_amd.synth import synth_ranges, synth_reads
    from oracle import oracle_c
    from scripts.kernel_times import time_plan
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        ctx = Context(0, stream=stream.cuda_stream)
        cols = synth_reads(50_000_000, [250_000_000], seed=1, with_cigar=False)
        reads = Reads(ctx, cols["ref_len"], cols["ref_off"], cols["pos"], cols["flag"], cols["mapq"], cols["tlen"], end=cols["end"])
        orc = oracle_c.OracleReads(cols["ref_off"], cols["pos"], cols["end"], cols["flag"], cols["mapq"], cols["tlen"])
        rg = synth_ranges(1_000_000, 100, [250_000_000], seed=5)
        for mode, name, args in ((_lib.MODE_COUNT, "count", dict(binsize=-1)), (_lib.MODE_PROFILE, "profile", dict(binsize=1)),
                                 (_lib.MODE_COVERAGE, "coverage", dict())):
            t0 = time.time()
            plan = Plan(ctx, reads, rg["rid"], rg["loc"], rg["len"], rg["strand"], make_params(mode, **args))
            t_plan = time.time() - t0
            out = torch.empty(max(plan.cells, 4), dtype=torch.int32, device="cuda")
            ms = time_plan(torch, stream, plan, out, steps=10, warmup=2)
            t0 = time.time()
            fn = oracle_c.coverage_core if mode == _lib.MODE_COVERAGE else oracle_c.pileup_core
            want, _ = fn(orc, rg, **args)
            t_cpu = time.time() - t0
            got = plan.run_host()
            st = plan.stats()
            print(json.dumps(dict(case=f"{name}: 1M x 100 bp", kernel_ms=ms, plan_s=t_plan, cpu_oracle_s=t_cpu,
                                  exact=bool(np.array_equal(got, want)), GBps=st["algorithmic_bytes"] / ms / 1e6)))
            plan.close()


if __name__ == "__main__":
    main()
