#!/usr/bin/env python3
"""Diagnostic (NOT a benchmark): time k_profile with its read streaming and/or its stores removed.
Needs:  make -C bamsignals_amd/csrc stamps ; BSIG_LIB_PATH=.../libbamsignals_hip_stamps.so"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=50_000_000)
    ap.add_argument("--ranges", type=int, default=10_000)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--tile-cells", type=int, default=0)
    ap.add_argument("--batches", type=int, default=8)
    ap.add_argument("--refs", type=int, default=1, help="references of 250 Mbp (the north star: 10, with --reads 500000000 --ranges 100000)")
    a = ap.parse_args()
    import torch

    from bamsignals_amd import _lib
    from bamsignals_amd.device import Context, Plan, Reads, make_params
    from bamsignals_amd.synth import synth_ranges, synth_reads

    ref_len = [250_000_000] * a.refs
    cols = synth_reads(a.reads, ref_len, with_cigar=False)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        ctx = Context(0, stream=stream.cuda_stream)
        reads = Reads(ctx, cols["ref_len"], cols["ref_off"], cols["pos"], cols["flag"], cols["mapq"], cols["tlen"], end=cols["end"])
        plans, outs = [], []
        for b in range(a.batches):
            rg = synth_ranges(a.ranges, 2000, ref_len, seed=100 + b)
            p = Plan(ctx, reads, rg["rid"], rg["loc"], rg["len"], rg["strand"],
                     make_params(_lib.MODE_PROFILE, binsize=1, threads=a.threads, tile_cells=a.tile_cells))
            plans.append(p)
            outs.append(torch.empty(p.cells, dtype=torch.int32, device="cuda"))
        lib = _lib.load()
        nb = a.batches
        for bits, name in ((0, "full"), (1, "no reads"), (2, "no stores"), (3, "neither"), (0, "full")):
            assert lib.bsig_debug_set_ablate(bits) == 0
            for s in range(16):
                plans[s % nb].run_device(outs[s % nb].data_ptr())
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for s in range(64):
                plans[s % nb].run_device(outs[s % nb].data_ptr())
            e1.record(stream)
            torch.cuda.synchronize()
            print(f"{name:10s} {e0.elapsed_time(e1) / 64 * 1e3:8.1f} us per step (k_profile)")


if __name__ == "__main__":
    main()
