#!/usr/bin/env python3
"""Diagnostic: config 2's step time against (a) the tile count relative to the resident workgroups
(ranges split so that the launch fills whole rounds) and (b) the bucket width of the read index
(env BAMSIGNALS_BUCKET_READS, read when the reads are laid out).  Prints JSON lines."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    from bamsignals_amd import _lib
    from bamsignals_amd.device import Context, Plan, Reads, make_params
    from bamsignals_amd.synth import synth_ranges, synth_reads

    stream = torch.cuda.Stream()
    cols = synth_reads(50_000_000, [250_000_000], seed=0xBA51, with_cigar=False)
    nb = 8
    base = [synth_ranges(10_000, 2000, [250_000_000], seed=0xBA52 + 7919 * b) for b in range(nb)]

    def split(rg, n_split):
        """the first n_split ranges (in sorted order: spread evenly) cut into two 1,000-bp halves"""
        order = np.lexsort((rg["loc"], rg["rid"]))
        pick = np.zeros(len(order), bool)
        if n_split:
            pick[order[np.linspace(0, len(order) - 1, n_split).astype(np.int64)]] = True
        keep = {k: v[~pick] for k, v in rg.items()}
        a = {k: v[pick].copy() for k, v in rg.items()}
        b = {k: v[pick].copy() for k, v in rg.items()}
        a["len"][:] = 1000
        b["len"][:] = 1000
        b["loc"] = b["loc"] + 1000
        return {k: np.concatenate([keep[k], a[k], b[k]]) for k in rg}

    with torch.cuda.stream(stream):
        ctx = Context(0, stream=stream.cuda_stream)
        for bucket in (os.environ.get("BUCKETS", "16,8,4,2").split(",")):
            os.environ["BAMSIGNALS_BUCKET_READS"] = bucket
            reads = Reads(ctx, cols["ref_len"], cols["ref_off"], cols["pos"], cols["flag"], cols["mapq"], cols["tlen"], end=cols["end"])
            for n_split in ([0, 1144, 2288, 4000, 10000] if bucket == "16" else [0, 2288]):
                batches = [split(g, n_split) for g in base]
                prm = make_params(_lib.MODE_PROFILE, binsize=1)
                plans = [Plan(ctx, reads, g["rid"], g["loc"], g["len"], g["strand"], prm) for g in batches]
                outs = [torch.empty(p.cells, dtype=torch.int32, device="cuda") for p in plans]
                st = plans[0].stats()
                for r in range(3):
                    for q in range(40):
                        plans[q % nb].run_device(outs[q % nb].data_ptr())
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(stream)
                    K = 400
                    for q in range(K):
                        plans[q % nb].run_device(outs[q % nb].data_ptr())
                    e1.record(stream)
                    torch.cuda.synchronize()
                    ms = e0.elapsed_time(e1) / K
                print(json.dumps(dict(bucket_reads=bucket, n_split=n_split, items=st["n_items"], us=ms * 1e3,
                                      alg_bytes=st["algorithmic_bytes"], streamed=st["streamed"], visits=st["visits"],
                                      hbm_bytes=reads.info()["hbm_bytes"], frac=st["algorithmic_bytes"] / ms / 1e6 / 8000)), flush=True)
                for p in plans:
                    p.close()
            reads.close()


if __name__ == "__main__":
    main()
