#!/usr/bin/env python3
"""Diagnostic: the north star's launch (or --config C2 / C4 / C5) under k_profile's launch-shape knobs --
class-0 passes in flight (knob 0: 4 / 3 / 2 = 77 / 64 / 53 VGPRs) and the 8-waves-per-SIMD build of the
two-pass variant (knob 3 = 8: 96 SGPRs, the rest in VGPR lanes, 32 instead of 24-28 workgroups per CU).
One resident workload, every variant timed on it in turn (twice, interleaved), results compared bit for bit."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="NS")
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--reads", type=int, default=0)
    ap.add_argument("--ranges", type=int, default=0)
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--only", default="", help="comma-separated variant names")
    a = ap.parse_args()
    import torch

    import bench
    from bamsignals_amd import _lib
    args = argparse.Namespace(seed=0xBA51, tile_cells=0, threads=0)
    stream = torch.cuda.Stream()
    w = bench.Workload(args, a.config, 0, 1, 0, stream, a.reads, a.ranges, a.width, 0)
    lib = _lib.load()
    # (name, class passes in flight, 8-wave build, tiles from which the windows are looked up by a launch of their own)
    never = 1 << 30
    # ... and, last, consecutive tiles per wave (k_profile_multi, knob 5)
    variants = [("pre2", 2, 1, never, 1), ("pre2+resolve-launch", 2, 1, 0, 1), ("pre2+8waves+resolve-launch", 2, 8, 0, 1),
                ("resolve-launch+2tiles", 2, 1, 0, 2), ("resolve-launch+4tiles", 2, 1, 0, 4),
                ("resolve-launch+8waves+2tiles", 2, 8, 0, 2), ("resolve-launch+8waves+4tiles", 2, 8, 0, 4), ("default", 2, 0, -1, 0)]
    if a.only:
        variants = [v for v in variants if v[0] in a.only.split(",")]
    ref = None
    for rnd in range(2):
        for name, pre, w8, rmin, pt in variants:
            assert lib.bsig_debug_set_knob(0, pre) == 0 and lib.bsig_debug_set_knob(3, w8) == 0 and lib.bsig_debug_set_knob(4, rmin) == 0
            assert lib.bsig_debug_set_knob(5, pt) == 0
            _, ms = w.timed(a.steps, 10, stream, lambda: None)
            got = [o.clone() for o in w.outs]
            if ref is None:
                ref = got
            same = all(torch.equal(x, y) for x, y in zip(got, ref))
            print(json.dumps(dict(config=a.config, variant=name, round=rnd, kernel_ms=ms,
                                  frac=w.stats["algorithmic_bytes"] / ms / 1e6 / 8000, identical=same)), flush=True)
            if not same:
                raise SystemExit("variant differs")


if __name__ == "__main__":
    main()
