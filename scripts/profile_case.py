#!/usr/bin/env python3
"""One BASELINE configuration's kernel in a loop, for rocprofv3 (scripts/profile_round.sh): prints one
JSON line with the HIP-event time per launch and the plan's algorithmic bytes.
   C3     bamCoverage, chr1 (248,956,422 bp) in 2-kb tiles, 1e8 SE reads      -> k_coverage<64>
   C4     bamProfile PE filter tlenFilter=c(50,500) shift=75 ss=TRUE, 100k x 2 kb, 1e8 PE reads on 250 Mbp -> k_profile<64, true>
   C4mid  the same with paired.end="midpoint"
   count  bamCount on C3's tiling                                             -> k_count_multi
   bins   bamProfile binsize=200 ss=TRUE, 100k x 2 kb on the C3 reads (the wide-bin form a ChIP-seq caller uses) -> k_profile_small
   t500 / t1000   bamProfile binsize=1 over 400,000 x 500 bp / 200,000 x 1 kb (2e8 bases) on the C3 reads -> k_profile_multi / k_profile<.., 8, true>"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    case = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    import torch

    from bamsignals_amd import _lib
    from bamsignals_amd.device import Context, Plan, Reads, make_params
    from bamsignals_amd.synth import synth_ranges, synth_reads, tile_ranges

    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        ctx = Context(0, stream=stream.cuda_stream)
        if case in ("C3", "count"):
            L = 248_956_422
            cols = synth_reads(100_000_000, [L], seed=3, with_cigar=False)
            rgs = [tile_ranges([L], 2000)]
            prm = make_params(_lib.MODE_COVERAGE) if case == "C3" else make_params(_lib.MODE_COUNT, binsize=-1)
            name = ("C3: bamCoverage, chr1 2-kb tiling, 1e8 SE reads" if case == "C3" else "bamCount on C3's tiling, 1e8 SE reads")
        elif case == "bins":
            L = 248_956_422
            cols = synth_reads(100_000_000, [L], seed=3, with_cigar=False)
            rgs = [synth_ranges(100_000, 2000, [L], seed=20 + b) for b in range(2)]
            prm = make_params(_lib.MODE_PROFILE, binsize=200, ss=True)
            name = "bamProfile binsize=200 ss=TRUE, 100k x 2kb, 1e8 SE reads on 249 Mbp"
        elif case in ("t500", "t1000"):
            L, w = 248_956_422, int(case[1:])
            cols = synth_reads(100_000_000, [L], seed=3, with_cigar=False)
            rgs = [synth_ranges(200_000_000 // w, w, [L], seed=30 + b) for b in range(2)]
            prm = make_params(_lib.MODE_PROFILE, binsize=1)
            name = f"bamProfile binsize=1, {200_000_000 // w} x {w} bp, 1e8 SE reads on 249 Mbp"
        elif case in ("C4", "C4mid"):
            cols = synth_reads(100_000_000, [250_000_000], seed=9, paired=True, with_cigar=False)
            rgs = [synth_ranges(100_000, 2000, [250_000_000], seed=10 + b) for b in range(2)]
            prm = make_params(_lib.MODE_PROFILE, binsize=1, ss=True, shift=75, requiredF=66, tlen_filter=(50, 500), pe_mid=(case == "C4mid"))
            name = f"C4: bamProfile PE {'midpoint' if case == 'C4mid' else 'filter'} tlenFilter=c(50,500) shift=75 ss=TRUE, 100k x 2kb, 1e8 PE reads, 250 Mbp"
        else:
            raise SystemExit("unknown case " + case)
        reads = Reads(ctx, cols["ref_len"], cols["ref_off"], cols["pos"], cols["flag"], cols["mapq"], cols["tlen"], end=cols["end"])
        plans = [Plan(ctx, reads, rg["rid"], rg["loc"], rg["len"], rg["strand"], prm) for rg in rgs]
        outs = [torch.empty(max(p.cells, 4), dtype=torch.int32, device="cuda") for p in plans]
        nb = len(plans)
        for q in range(4):
            plans[q % nb].run_device(outs[q % nb].data_ptr())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for q in range(steps):
            plans[q % nb].run_device(outs[q % nb].data_ptr())
        e1.record(stream)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / steps
        st = plans[0].stats()
        print(json.dumps(dict(case=case, workload=name, steps=steps, kernel_ms=ms, algorithmic_bytes=st["algorithmic_bytes"],
                              frac_of_8TBps=st["algorithmic_bytes"] / ms / 1e6 / 8000, **{k: st[k] for k in ("n_items", "cells", "visits", "streamed")})))


if __name__ == "__main__":
    main()
