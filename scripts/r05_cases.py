#!/usr/bin/env python3
"""Diagnostic: the kernels VERDICT r04 lists below the roofline, in ONE process on one set of reads (1e8 SE reads on
chr1's 248,956,422 bp): C3 (bamCoverage, 2-kb tiling), count (bamCount on that tiling), bins (bamProfile binsize=200
ss=TRUE, 100k x 2 kb), t500 (bamProfile binsize=1 over 400,000 x 500 bp = 2e8 bases), t1000 (200,000 x 1 kb).
Prints one JSON line per case: HIP-event ms per step, algorithmic bytes, fraction of 8 TB/s.  BSIG_LIB_PATH selects a build;
cases on the command line (default: all)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    cases = sys.argv[1:] or ["C3", "count", "bins", "t500", "t1000"]
    steps = int(os.environ.get("BSIG_CASE_STEPS", "40"))
    import numpy as np
    import torch

    from bamsignals_amd import _lib
    from bamsignals_amd.device import Context, Plan, Reads, make_params
    from bamsignals_amd.synth import synth_ranges, synth_reads, tile_ranges
    from oracle import oracle_c

    L = 248_956_422
    if os.environ.get("BSIG_CASE_RESOLVE_MIN"):              # (diagnostic: the two-launch form from that many tiles on, bamCount included)
        import ctypes
        fn = _lib.load().bsig_debug_set_resolve_min
        fn.argtypes = [ctypes.c_longlong]
        fn(int(os.environ["BSIG_CASE_RESOLVE_MIN"]))
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        ctx = Context(0, stream=stream.cuda_stream)
        cols = synth_reads(100_000_000, [L], seed=3, with_cigar=False)
        reads = Reads(ctx, cols["ref_len"], cols["ref_off"], cols["pos"], cols["flag"], cols["mapq"], cols["tlen"], end=cols["end"])
        orc = oracle_c.OracleReads(cols["ref_off"], cols["pos"], cols["end"], cols["flag"], cols["mapq"], cols["tlen"]) if os.environ.get("BSIG_CASE_CHECK") else None
        for case in cases:
            if case == "C3":
                rgs, prm, a, kind = [tile_ranges([L], 2000)], make_params(_lib.MODE_COVERAGE), dict(), "coverage"
            elif case == "count":
                rgs, prm, a, kind = [tile_ranges([L], 2000)], make_params(_lib.MODE_COUNT, binsize=-1), dict(binsize=-1), "pileup"
            elif case == "bins":
                rgs, a, kind = [synth_ranges(100_000, 2000, [L], seed=20 + b) for b in range(2)], dict(binsize=200, ss=True), "pileup"
                prm = make_params(_lib.MODE_PROFILE, **a)
            elif case in ("bins16", "bins50ss", "bins2000ss", "bins500"):
                a = {"bins16": dict(binsize=16), "bins50ss": dict(binsize=50, ss=True), "bins2000ss": dict(binsize=2000, ss=True),
                     "bins500": dict(binsize=500)}[case]
                rgs, kind = [synth_ranges(100_000, 2000, [L], seed=20 + b) for b in range(2)], "pileup"
                prm = make_params(_lib.MODE_PROFILE, **a)
            elif case in ("t500", "t1000"):
                w = int(case[1:])
                rgs, a, kind = [synth_ranges(200_000_000 // w, w, [L], seed=30 + b) for b in range(2)], dict(binsize=1), "pileup"
                prm = make_params(_lib.MODE_PROFILE, **a)
            else:
                raise SystemExit("unknown case " + case)
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
            ok = None
            if orc is not None:
                rg = rgs[0]
                sub = {k: v[:20000] for k, v in rg.items()}
                want, woff = (oracle_c.coverage_core if kind == "coverage" else oracle_c.pileup_core)(orc, sub, **a)
                got = outs[0][:len(want)].cpu().numpy() if a.get("binsize", 1) > 0 else None
                if got is None:      # bamCount: one flat vector over all ranges
                    got = outs[0][:plans[0].cells].cpu().numpy()[:len(want)]
                ok = bool(np.array_equal(got, want))
            print(json.dumps(dict(case=case, ms=round(ms, 5), frac_of_8TBps=round(st["algorithmic_bytes"] / ms / 1e6 / 8000, 4),
                                  algorithmic_bytes=st["algorithmic_bytes"], items=st["n_items"], cells=st["cells"], visits=st["visits"],
                                  first_20000_ranges_match_the_oracle=ok)), flush=True)
            for p in plans:
                p.close()
            del outs
        reads.close()


if __name__ == "__main__":
    main()
