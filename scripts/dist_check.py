#!/usr/bin/env python3
"""Launched by torch.distributed.run: the sharded user-level calls against the single-rank ones
on the fixture BAM.  `--backend gloo` lets two ranks share one GPU; nccl needs a GPU per rank."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="nccl")
    a = ap.parse_args()
    import torch
    import torch.distributed as dist
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    os.environ["BAMSIGNALS_DEVICE"] = str(local)
    torch.cuda.set_device(local)
    if a.backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group("gloo")
    from bamsignals_amd import GRanges, bamCount, bamCoverage, bamProfile
    from bamsignals_amd.dist import bamCount_sharded, bamCoverage_sharded, bamProfile_sharded
    reg = json.load(open(os.path.join(ROOT, "tests", "golden", "regions.json")))
    bam = os.path.join(ROOT, "tests", "golden", "randomBam.bam")
    gr = GRanges(reg["chrom"], reg["start"], width=reg["width"], strand=reg["strand"])
    ok = True
    p = bamProfile_sharded(bam, gr, binsize=5, ss=True, shift=20)
    c = bamCount_sharded(bam, gr, ss=True, paired_end="midpoint")
    v = bamCoverage_sharded(bam, gr, paired_end="extend", tlenFilter=(50, 300))
    if dist.get_rank() == 0:
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            p1 = bamProfile(bam, gr, binsize=5, ss=True, shift=20, verbose=False)
        c1 = bamCount(bam, gr, ss=True, paired_end="midpoint", verbose=False)
        v1 = bamCoverage(bam, gr, paired_end="extend", tlenFilter=(50, 300), verbose=False)
        ok = all(np.array_equal(x, y) for x, y in zip(p, p1)) and np.array_equal(c, c1) and \
            all(np.array_equal(x, y) for x, y in zip(v, v1)) and p.ss and len(p) == len(gr)
        print("dist_check", "OK" if ok else "MISMATCH", "world", dist.get_world_size(), a.backend)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
