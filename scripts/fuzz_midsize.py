#!/usr/bin/env python3
"""Diagnostic: seeded random MID-SIZE inputs, HIP against the C oracle bit for bit -- what the small fuzz
(tests/test_gpu_parity.py::test_fuzz_small_inputs) cannot reach: a few hundred thousand reads with piles of up to 80,000
on one position (the 16-bit images' slices), thousands of ranges of 1 ... 60,000 bases (bamCount's sub-intervals, several
tiles per range), every bin width class (per-base images, the wide-bin kernel and its replicas, bins wider than a tile),
each plan run twice (the second run reads the windows kept by the first) in the fused and in the resolved launch forms.
usage: fuzz_midsize.py [first seed] [seeds]"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one_seed(ctx, seed, fn):
    from bamsignals_amd import _lib
    from bamsignals_amd.device import Plan, Reads, make_params
    from oracle import oracle_c
    rng = np.random.default_rng(seed)
    n_ref = int(rng.integers(1, 5))
    ref_len = rng.integers(50_000, 3_000_000, n_ref).astype(np.int32)
    n = int(rng.integers(50_000, 600_000))
    rid = rng.integers(0, n_ref, n).astype(np.int32)
    pos = (rng.random(n) * ref_len[rid]).astype(np.int32)
    for _ in range(int(rng.integers(0, 4))):                     # piles: many reads on one position or on a few hundred
        k = int(rng.choice([20_000, 32_767, 32_768, 32_769, 50_000, 80_000]))
        r = int(rng.integers(0, n_ref))
        p = int(rng.integers(0, ref_len[r]))
        sel = rng.choice(n, min(k, n), replace=False)
        rid[sel] = r
        pos[sel] = np.minimum(p + rng.integers(0, int(rng.choice([1, 1, 300])), len(sel)), ref_len[r] - 1)
    order = np.lexsort((pos, rid))
    rid, pos = rid[order], pos[order]
    u = rng.random(n)
    span = np.where(u < 0.9, rng.integers(20, 160, n), np.where(u < 0.98, rng.integers(160, 3000, n), rng.integers(3000, 120_000, n)))
    end = (pos + span - 1).astype(np.int32)
    flag = (rng.integers(0, 2, n) * 16 + (rng.random(n) < 0.05) * 1024 + rng.integers(0, 2, n) * 64 + rng.integers(0, 2, n) * 2
            + (rng.random(n) < 0.01) * 4 + (rng.random(n) < 0.01) * 256).astype(np.uint16)
    mapq = np.where(rng.random(n) < 0.8, 60, rng.integers(0, 256, n)).astype(np.uint8)
    tlen = rng.integers(-900, 900, n).astype(np.int32)
    ref_off = np.searchsorted(rid, np.arange(n_ref + 1)).astype(np.int64)
    gpu = Reads(ctx, ref_len, ref_off, pos, flag, mapq, tlen, end=end)
    orc = oracle_c.OracleReads(ref_off, pos, end, flag, mapq, tlen)
    done = 0
    for rep in range(6):
        m = int(rng.integers(1, 3000))
        wide = rng.random() < 0.3
        rr = rng.integers(0, n_ref, m).astype(np.int32)
        ln = (rng.integers(1, 60_000, m) if wide else rng.integers(1, int(rng.choice([300, 2100, 9000])), m)).astype(np.int32)
        loc = (rng.random(m) * (ref_len[rr] + 600) - 300).astype(np.int32)
        rg = dict(rid=rr, loc=loc, len=ln, strand=rng.integers(-1, 2, m).astype(np.int32))
        pe = rng.random() < 0.4
        tf = tuple(sorted(int(x) for x in rng.integers(0, 900, 2))) if pe else ()
        common = dict(mapqual=int(rng.choice([0, 0, 20, 61])), requiredF=int(rng.choice([0, 2, 66])),
                      filteredF=int(rng.choice([-1, 0, 16, 1024, 1040])), tlen_filter=tf)
        pile = dict(common, binsize=int(rng.choice([-1, 1, 1, 2, 7, 16, 50, 200, 1000, 8192, 10_000, 70_000])), shift=int(rng.integers(-150, 150)),
                    ss=bool(rng.integers(0, 2)), pe_mid=bool(pe and rng.integers(0, 2)))
        cov = dict(common, tspan=bool(pe and rng.integers(0, 2)))
        for kind, a in (("pileup", pile), ("coverage", cov)):
            want, woff = (oracle_c.pileup_core if kind == "pileup" else oracle_c.coverage_core)(orc, rg, **a)
            for form in ("fused", "resolved"):
                fn(1 if form == "resolved" else -1)
                b = dict(a)
                if kind == "coverage":
                    prm = make_params(_lib.MODE_COVERAGE, tile_cells=int(rng.choice([0, 0, 256, 1000])), threads=int(rng.choice([0, 64, 128, 256])), **b)
                else:
                    bs = b.pop("binsize")
                    prm = make_params(_lib.MODE_COUNT if bs <= 0 else _lib.MODE_PROFILE, binsize=bs, tile_cells=int(rng.choice([0, 0, 256, 1000])),
                                      threads=int(rng.choice([0, 64, 128, 256])), **b)
                plan = Plan(ctx, gpu, rg["rid"], rg["loc"], rg["len"], rg["strand"], prm)
                for run in range(2):
                    got = plan.run_host()
                    if not (np.array_equal(plan.offsets, woff) and np.array_equal(got, want)):
                        bad = np.flatnonzero(got != want)
                        raise SystemExit(f"seed {seed} rep {rep} {kind} {form} run {run}: {len(bad)} cells differ (first {bad[:5]}), args {a}, "
                                         f"heavy tiles {plan.stats()['heavy_tiles']}")
                    done += 1
                plan.close()
                fn(-1)
    gpu.close()
    return done, n


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    from bamsignals_amd import _lib
    from bamsignals_amd.device import Context
    fn = _lib.load().bsig_debug_set_resolve_min
    fn.argtypes = [ctypes.c_longlong]
    ctx = Context(0)
    for seed in range(first, first + seeds):
        t0 = time.time()
        done, n = one_seed(ctx, seed, fn)
        print(f"seed {seed}: {n} reads, {done} runs identical to the oracle ({time.time() - t0:.1f} s)", flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
