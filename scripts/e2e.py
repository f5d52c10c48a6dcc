#!/usr/bin/env python3
"""End-to-end timings around the hot path (NOT the bench metric): synthetic BAM on disk ->
decode -> HBM -> kernels -> host result, next to the CPU path (decode + oracle) on the same box.
Prints one JSON object; numbers quoted in DESIGN.md come from here."""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=50_000_000)
    ap.add_argument("--ranges", type=int, default=10_000)
    ap.add_argument("--genome", type=int, default=250_000_000)
    ap.add_argument("--dir", default=None)
    a = ap.parse_args()
    import torch  # noqa: F401  (first, so that one libamdhip64 is shared)

    from bamsignals_amd import GRanges, _lib, bamProfile
    from bamsignals_amd.bamio import BamFile, write_columns_as_bam
    from bamsignals_amd.device import Context, Plan, Reads, make_params
    from bamsignals_amd.synth import synth_ranges, synth_reads
    from oracle import oracle_c

    res = {"reads": a.reads, "ranges": a.ranges, "host_threads": os.cpu_count()}
    d = a.dir or tempfile.mkdtemp(prefix="bsig_e2e_", dir="/tmp")
    bam = os.path.join(d, "synth.bam")
    t = time.time(); cols = synth_reads(a.reads, [a.genome]); res["generate_s"] = time.time() - t
    t = time.time(); write_columns_as_bam(bam, ["chr1"], cols, level=1); res["write_bam_s"] = time.time() - t
    res["bam_bytes"] = os.path.getsize(bam)
    rg = synth_ranges(a.ranges, 2000, [a.genome], seed=77)
    bases = int(rg["len"].astype(np.int64).sum())

    # --- decode stage alone -----------------------------------------------------------------
    b = BamFile(bam)
    t = time.time(); dec = b.decode(); res["decode_all_py_s"] = time.time() - t     # + numpy copies
    res["decode_all_stages_s"] = b.decode_timing()
    res["decode_all_s"] = res["decode_all_stages_s"]["total"]
    assert np.array_equal(dec["pos"], cols["pos"]) and np.array_equal(dec["cigar"], cols["cigar"])
    dec1 = b.decode(threads=1)
    res["decode_all_1thread_s"] = b.decode_timing()["total"]
    t = time.time(); sub = b.decode(rg["rid"], rg["loc"].astype(np.int64), (rg["loc"] + rg["len"]).astype(np.int64))
    res["decode_regions_s"] = time.time() - t
    res["decode_regions_reads"] = int(len(sub["pos"]))
    del dec1, sub

    # --- GPU: upload, plan, run to host --------------------------------------------------------
    ctx = Context(0)
    t = time.time()
    reads = Reads(ctx, dec["ref_len"], dec["ref_off"], dec["pos"], dec["flag"], dec["mapq"], dec["tlen"],
                  cigar_off=dec["cigar_off"], cigar=dec["cigar"])
    res["upload_and_layout_s"] = time.time() - t
    # --- the same BAM decoded on the device (CPU inflate -> pinned -> HBM, records -> columns on the GPU)
    os.environ["BAMSIGNALS_DEVICE_DECODE"] = "require"
    for rep in ("first", "again"):                   # the first call also pins the staging buffers
        t = time.time(); rdev = Reads.from_bam(ctx, b); res[f"reads_from_bam_device_{rep}_s"] = time.time() - t
        res[f"reads_from_bam_device_{rep}_stages_s"] = Reads.device_decode_timing()
        assert rdev.info() == reads.info()
        rdev.close()
    os.environ["BAMSIGNALS_DEVICE_DECODE"] = "0"
    t = time.time(); rcpu = Reads.from_bam(ctx, b); res["reads_from_bam_cpu_s"] = time.time() - t
    rcpu.close()
    os.environ.pop("BAMSIGNALS_DEVICE_DECODE")
    t = time.time()
    plan = Plan(ctx, reads, rg["rid"], rg["loc"], rg["len"], rg["strand"], make_params(_lib.MODE_PROFILE, binsize=1))
    res["plan_s"] = time.time() - t
    out = plan.run_host()
    ts = []
    for _ in range(5):
        t = time.time(); out = plan.run_host(); ts.append(time.time() - t)
    res["run_to_host_s"] = min(ts)                     # kernel + D2H of the int32 result over PCIe
    res["run_to_host_Mbases_s"] = bases / min(ts) / 1e6
    from bamsignals_amd.device import pinned_empty
    outp = pinned_empty(plan.cells, np.int32)
    ts = []
    for _ in range(5):
        t = time.time(); plan.run_host(out=outp); ts.append(time.time() - t)
    assert np.array_equal(outp, out)
    res["run_to_pinned_host_s"] = min(ts)              # ... into a reused page-locked buffer
    res["run_to_pinned_host_Mbases_s"] = bases / min(ts) / 1e6

    # --- file-level API: cold (decode + upload + run) and warm (BAM cached in HBM) ---------------
    gr = GRanges(["chr1"] * a.ranges, rg["loc"] + 1, width=rg["len"],
                 strand=[{1: "+", -1: "-", 0: "*"}[int(s)] for s in rg["strand"]])
    _lib.load().bsig_cache_clear()
    for mode in ("regions", "all"):
        os.environ["BAMSIGNALS_DECODE"] = mode
        _lib.load().bsig_cache_clear()
        t = time.time(); sig = bamProfile(bam, gr, verbose=False); res[f"bamProfile_cold_{mode}_s"] = time.time() - t
        from bamsignals_amd.wrappers import last_call_timing
        res[f"bamProfile_cold_{mode}_stages_s"] = last_call_timing()
        t = time.time(); sig = bamProfile(bam, gr, verbose=False); res[f"bamProfile_again_{mode}_s"] = time.time() - t
    got = np.concatenate(sig.as_list())
    assert np.array_equal(got, out)

    # --- CPU path on the same box: decode (1 thread) + oracle (1 thread) -------------------------
    end = oracle_c.cigar_end(dec["pos"], dec["flag"], dec["cigar_off"], dec["cigar"])
    orc = oracle_c.OracleReads(dec["ref_off"], dec["pos"], end, dec["flag"], dec["mapq"], dec["tlen"])
    t = time.time(); want, _ = oracle_c.pileup_core(orc, rg, binsize=1); res["oracle_columns_s"] = time.time() - t
    assert np.array_equal(want, out)
    res["parity"] = True
    res["cpu_path_decode_plus_oracle_s"] = res["decode_all_1thread_s"] + res["oracle_columns_s"]
    res["cpu_path_Mbases_s"] = bases / res["cpu_path_decode_plus_oracle_s"] / 1e6
    print(json.dumps(res))
    os.remove(bam); os.remove(bam + ".bai")


if __name__ == "__main__":
    main()
