#!/usr/bin/env python3
"""Diagnostic: config 3's tiling (chr1 in 2-kb tiles, 1e8 SE reads) under the launch-shape knobs of the count and
coverage kernels -- bamCount: tiles per wave (knob 1) x passes in flight (knob 2); both: windows looked up inside the
kernel or by a launch of their own (knob 4).  One resident workload, every variant timed on it in turn (twice),
results compared bit for bit."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    import torch

    from bamsignals_amd import _lib
    from bamsignals_amd.device import Context, Plan, Reads, make_params
    from bamsignals_amd.synth import synth_reads, tile_ranges
    lib = _lib.load()
    L = 248_956_422
    cols = synth_reads(100_000_000, [L], seed=3, with_cigar=False)
    rg = tile_ranges([L], 2000)
    stream = torch.cuda.Stream()
    never = 1 << 30
    with torch.cuda.stream(stream):
        ctx = Context(0, stream=stream.cuda_stream)
        reads = Reads(ctx, cols["ref_len"], cols["ref_off"], cols["pos"], cols["flag"], cols["mapq"], cols["tlen"], end=cols["end"])
        for mode, prm, variants in (
            ("count", make_params(_lib.MODE_COUNT, binsize=-1),
             [(t, p, r) for r in (never, 0) for t in (1, 2, 4, 8) for p in (2, 3)]),
            ("coverage", make_params(_lib.MODE_COVERAGE), [(0, 0, never), (0, 0, 0)]),
        ):
            plan = Plan(ctx, reads, rg["rid"], rg["loc"], rg["len"], rg["strand"], prm)
            outs = [torch.empty(max(plan.cells, 4), dtype=torch.int32, device="cuda") for _ in range(2)]
            ref = None
            for rnd in range(2):
                for t, p, r in variants:
                    if t:
                        assert lib.bsig_debug_set_knob(1, t) == 0 and lib.bsig_debug_set_knob(2, p) == 0
                    assert lib.bsig_debug_set_knob(4, r) == 0
                    for q in range(6):
                        plan.run_device(outs[q % 2].data_ptr())
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(stream)
                    for q in range(steps):
                        plan.run_device(outs[q % 2].data_ptr())
                    e1.record(stream)
                    torch.cuda.synchronize()
                    ms = e0.elapsed_time(e1) / steps
                    got = outs[0].clone()
                    if ref is None:
                        ref = got
                    same = bool(torch.equal(got, ref))
                    st = plan.stats()
                    print(json.dumps(dict(mode=mode, tiles_per_wave=t, passes=p, resolve_launch=(r == 0), round=rnd, kernel_ms=ms,
                                          frac=st["algorithmic_bytes"] / ms / 1e6 / 8000, identical=same)), flush=True)
                    if not same:
                        raise SystemExit("variant differs")
            lib.bsig_debug_set_knob(4, -1)
            plan.close()


if __name__ == "__main__":
    main()
