#!/usr/bin/env python3
"""Diagnostic: where does a k_profile workgroup spend its lifetime?  (NOT a benchmark.)

Needs the stamps build:  make -C bamsignals_amd/csrc stamps
Run:  BSIG_LIB_PATH=bamsignals_amd/libbamsignals_hip_stamps.so python scripts/stamps.py
"""
import argparse
import ctypes as C
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
    ap.add_argument("--refs", type=int, default=1, help="references of 250 Mbp (10 with --reads 500000000 = the north star's shape)")
    ap.add_argument("--width", type=int, default=2000)
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
        for b in range(8 if a.reads <= 100_000_000 else 2):
            rg = synth_ranges(a.ranges, a.width, ref_len, seed=100 + b)
            p = Plan(ctx, reads, rg["rid"], rg["loc"], rg["len"], rg["strand"],
                     make_params(_lib.MODE_PROFILE, binsize=1, threads=a.threads, tile_cells=a.tile_cells))
            plans.append(p)
            outs.append(torch.empty(p.cells, dtype=torch.int32, device="cuda"))
        n_items = plans[0].stats()["n_items"]
        for s in range(24):
            plans[s % len(plans)].run_device(outs[s % len(plans)].data_ptr())
        torch.cuda.synchronize()
        stamps = torch.zeros(n_items * 8, dtype=torch.int64, device="cuda")
        lib = _lib.load()
        rc = lib.bsig_debug_set_stamp_buffer(C.c_void_p(stamps.data_ptr()))
        assert rc == 0, rc
        plans[0].run_device(outs[0].data_ptr())
        torch.cuda.synchronize()
        st = stamps.cpu().numpy().reshape(-1, 8)
    # s_memrealtime: 100 MHz reference clock shared by all XCDs -> microseconds
    rel = (st[:, :5] - st[:, 0].min()).astype(np.float64) / 100.0
    print("items", n_items, " kernel span (us):", rel[:, 4].max())
    names = ["start->item+windows+zero", "loads+process", "LDS->global issue", "stores drain"]
    for k in range(4):
        d = rel[:, k + 1] - rel[:, k]
        print(f"{names[k]:28s} median {np.median(d):9.2f}  p10 {np.percentile(d,10):9.2f}  p90 {np.percentile(d,90):9.2f}  max {d.max():9.2f}")
    life = rel[:, 4] - rel[:, 0]
    print(f"{'lifetime':28s} median {np.median(life):9.2f}  p10 {np.percentile(life,10):9.2f}  p90 {np.percentile(life,90):9.2f}")
    starts = np.sort(rel[:, 0])
    print("start time deciles:", [round(float(x), 2) for x in np.percentile(starts, np.arange(0, 101, 10))])
    ends = np.sort(rel[:, 4])
    print("end   time deciles:", [round(float(x), 2) for x in np.percentile(ends, np.arange(0, 101, 10))])
    xcc = (st[:, 5] >> 32) & 0xF
    print("workgroups per XCC:", np.bincount(xcc.astype(np.int64), minlength=8))
    # concurrency: how many workgroups alive over time
    ev = np.concatenate([np.stack([rel[:, 0], np.ones(len(rel))], 1), np.stack([rel[:, 4], -np.ones(len(rel))], 1)])
    ev = ev[np.argsort(ev[:, 0])]
    alive = np.cumsum(ev[:, 1])
    print("max workgroups alive:", int(alive.max()), " mean alive:", float(np.sum(alive[:-1] * np.diff(ev[:, 0])) / ev[-1, 0]))
    # timeline in 20 slices of the kernel span: workgroups alive, and how many are in each phase
    span = rel[:, 4].max()
    print("slice  alive  in:setup  in:reads  in:store-issue  in:drain   started  finished")
    for k in range(20):
        t = (k + 0.5) * span / 20
        al = (rel[:, 0] <= t) & (rel[:, 4] > t)
        ph = [int(np.sum((rel[:, j] <= t) & (rel[:, j + 1] > t))) for j in range(4)]
        print(f"{k:5d} {int(al.sum()):6d} {ph[0]:9d} {ph[1]:9d} {ph[2]:15d} {ph[3]:9d} {int((rel[:, 0] <= t).sum()):9d} {int((rel[:, 4] <= t).sum()):9d}")


if __name__ == "__main__":
    main()
