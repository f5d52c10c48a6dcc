#!/usr/bin/env python3
"""bench.py — Mbases profiled / s for bamProfile(binsize=1) on MI355X (BASELINE.json's metric).

A "step" is one pass of the hot path (HIP kernel k_profile behind bsig_plan_run) over one batch
of ranges, with the read columns and the range work items already resident in HBM and the result
left in HBM.  Default workload = the shape BASELINE.json's north star quotes its 1-GPU target on
("NS"): 100,000 x 2 kb ranges, 5e8 synthetic single-end reads on 10 x 250 Mbp (6.2 GB resident).
At N = 1 the same run also reports

  * "also"/"C2": the kernel-only step of BASELINE config 2 (10k x 2 kb, 5e7 reads) — the small launch;
  * "end_to_end": the file-level call bamProfile(bampath, GRanges) on the SAME workload written to
    local disk as a BAM (cold: open + GPU inflate + parse + HBM layout + kernels + result in host
    memory; warm: BAM resident), next to the CPU path including the BAM decode (`cpu_baseline.
    with_bam_decode`): informational, never `value`.

Multi-GPU (launched by torch.distributed.run, one rank per GPU): ranges are independent units,
so they are sharded round-robin over the ranks with NO data-path collective; every rank holds
the reads (weak scaling: each rank gets `--ranges` ranges).  After the timed region the per-rank
results are gathered to rank 0 over RCCL once (the north star's final reassembly step); its time
is reported separately in "gather" and is not part of `value`.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

CONFIGS = {
    # name: (reads, reference lengths, ranges, range width, paired, profile args)
    "C2": dict(reads=50_000_000, ref_len=[250_000_000], ranges=10_000, width=2000, paired=False,
               args=dict(binsize=1), desc="bamProfile binsize=1, 10k x 2kb ranges, 5e7 SE reads, 250 Mbp"),
    # the shape BASELINE.json's north_star quotes its 1-GPU target on (10 x C2 in reads and ranges)
    "NS": dict(reads=500_000_000, ref_len=[250_000_000] * 10, ranges=100_000, width=2000, paired=False,
               args=dict(binsize=1), desc="bamProfile binsize=1, 100k x 2kb ranges, 5e8 SE reads, 10 x 250 Mbp"),
    "C2small": dict(reads=2_000_000, ref_len=[10_000_000], ranges=10_000, width=2000, paired=False,
                    args=dict(binsize=1), desc="bamProfile binsize=1, 10k x 2kb ranges, 2e6 SE reads, 10 Mbp"),
    # hg38-like: 24 references, 3.1 Gbp; per GPU 125,000 x 1 kb ranges (1M over 8 GPUs)
    "C5": dict(reads=1_000_000_000,
               ref_len=[248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636,
                        138394717, 133797422, 135086622, 133275309, 114364328, 107043718, 101991189, 90338345,
                        83257441, 80373285, 58617616, 64444167, 46709983, 50818468, 156040895, 57227415],
               ranges=125_000, width=1000, paired=False, args=dict(binsize=1),
               desc="bamProfile binsize=1, 1M x 1kb ranges over 8 GPUs (125k per GPU), 1e9 SE reads, 24 refs / 3.1 Gbp"),
    "C4": dict(reads=500_000_000, ref_len=[250_000_000] * 10, ranges=100_000, width=2000, paired=True,
               args=dict(binsize=1, ss=True, shift=75, requiredF=66, tlen_filter=(50, 500)),
               desc="bamProfile PE filter tlenFilter=c(50,500) shift=75 ss=TRUE, 100k x 2kb, 5e8 PE reads"),
}


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def end_to_end(cfg, cols, rg, want_flat, device, oracle_c):
    """Informational (NOT `value`): the file-level call bamProfile(bampath, gr) on the bench's own reads
    written to local disk as a BAM -- BGZF inflate, records -> columns, HBM layout, kernels, result in
    host memory -- cold and again with the BAM resident in HBM, next to the CPU path that includes the
    BAM decode (one thread: this repo's BGZF/BAM reader, htslib being absent, + the oracle), as the
    reference's own call does (ref: src/bamsignals.cpp:271 bam_itr_next inside the pileup loop)."""
    import shutil
    import tempfile

    from bamsignals_amd import GRanges, _lib
    from bamsignals_amd.bamio import BamFile, write_columns_as_bam
    from bamsignals_amd.synth import add_cigar
    from bamsignals_amd.wrappers import last_call_timing, pileup_core
    args = cfg["args"]           # (oracle_c: the checker and CPU baseline, handed in by the cpu_baseline leg)
    d = tempfile.mkdtemp(prefix="bsig_bench_", dir=os.environ.get("TMPDIR", "/tmp"))
    try:
        if "cigar" not in cols:
            add_cigar(cols)
        names = ["ref%d" % (i + 1) for i in range(len(cfg["ref_len"]))]
        bam = os.path.join(d, "synth.bam")
        t0 = time.perf_counter(); write_columns_as_bam(bam, names, cols, level=1); t_write = time.perf_counter() - t0
        cols.pop("cigar"); cols.pop("cigar_off")
        log(f"end_to_end: wrote {os.path.getsize(bam) / 1e6:.0f} MB BAM in {t_write:.1f} s")
        gr = GRanges([names[r] for r in rg["rid"]], rg["loc"] + 1, width=rg["len"],
                     strand=[{1: "+", -1: "-", 0: "*"}[int(x)] for x in rg["strand"]])
        call = dict(tlen_filter=args.get("tlen_filter", ()), mapqual=args.get("mapqual", 0), binsize=args.get("binsize", 1),
                    shift=args.get("shift", 0), ss=args.get("ss", False), requiredF=args.get("requiredF", 0),
                    filteredF=args.get("filteredF", -1), pe_mid=args.get("pe_mid", False), device=device)
        _lib.load().bsig_cache_clear()
        t0 = time.perf_counter(); sig = pileup_core(bam, gr, **call); t_cold = time.perf_counter() - t0
        stages = last_call_timing()
        from bamsignals_amd.device import Reads
        stages["decode_stages_s"] = Reads.device_decode_timing()
        del sig
        t0 = time.perf_counter(); sig2 = pileup_core(bam, gr, **call); t_warm = time.perf_counter() - t0
        flat = np.concatenate([np.asarray(m).T.reshape(-1) if call["ss"] else np.asarray(m) for m in sig2])
        if not np.array_equal(flat, want_flat):
            raise SystemExit("file-level result differs from the resident-column result")
        del sig2, flat
        _lib.load().bsig_cache_clear()
        log(f"end_to_end: cold {t_cold:.3f} s, warm {t_warm:.3f} s; now the 1-thread CPU path on the same BAM")
        b = BamFile(bam)
        dec = b.decode(threads=1)
        t_dec1 = b.decode_timing()["total"]
        t0 = time.perf_counter()
        end = oracle_c.cigar_end(dec["pos"], dec["flag"], dec["cigar_off"], dec["cigar"])
        orc = oracle_c.OracleReads(dec["ref_off"], dec["pos"], end, dec["flag"], dec["mapq"], dec["tlen"])
        want, _ = oracle_c.pileup_core(orc, rg, **args)
        t_orc = time.perf_counter() - t0
        if not np.array_equal(want, want_flat):
            raise SystemExit("CPU path on the decoded BAM differs from the GPU result")
        del dec, orc, end
        b.close()
        bases = int(rg["len"].astype(np.int64).sum())
        gpu = dict(bam_bytes=os.path.getsize(bam), write_bam_s=t_write,
                   cold_call_s=t_cold, cold_call_stages_s=stages, warm_call_s=t_warm,
                   cold_Mbases_s=bases / t_cold / 1e6, warm_Mbases_s=bases / t_warm / 1e6,
                   vs_cpu_path_cold=(t_dec1 + t_orc) / t_cold, vs_cpu_path_warm=(t_dec1 + t_orc) / t_warm,
                   note="pileup_core(bampath, GRanges) -> per-range arrays in host memory (PCIe-inclusive); compared "
                        "with cpu_baseline.with_bam_decode; parity-checked against the timed result")
        cpu = dict(value=bases / (t_dec1 + t_orc) / 1e6, unit="Mbases/s", cores=1, seconds=t_dec1 + t_orc,
                   decode_s=t_dec1, pileup_s=t_orc,
                   sample="1 x the per-GPU workload from the BAM file: single-thread BGZF/BAM decode (this repo's "
                          "reader; htslib is absent) + oracle/bamsignals_oracle.c, as the reference decodes inside "
                          "its pileup loop")
        return gpu, cpu
    finally:
        shutil.rmtree(d, ignore_errors=True)


def end_to_end_realistic(seed, device, oracle_c):
    """Informational: the cold file-level call on a BAM shaped like real data -- 2e7 single-end 100-bp
    reads on 250 Mbp WITH read names, bases, qualities and an NM tag (204-byte records, 4 GB of stream
    that compresses about 1.9 : 1), 10k x 2 kb ranges -- under both inflate engines, next to the same
    single-thread CPU path as above.  The bench's own BAM carries bare 52-byte records;
    this is what the decode stage costs on literal-heavy DEFLATE blocks."""
    import shutil
    import tempfile

    from bamsignals_amd import GRanges, _lib
    from bamsignals_amd.bamio import BamFile, write_columns_as_bam
    from bamsignals_amd.synth import synth_ranges, synth_reads
    from bamsignals_amd.wrappers import last_call_route, last_call_timing, pileup_core
    from bamsignals_amd.device import Reads
    ref_len, n_reads, l_seq = [250_000_000], 20_000_000, 100
    d = tempfile.mkdtemp(prefix="bsig_bench_real_", dir=os.environ.get("TMPDIR", "/tmp"))
    try:
        cols = synth_reads(n_reads, ref_len, seed=seed + 77, with_cigar=True)
        bam = os.path.join(d, "real.bam")
        t0 = time.perf_counter(); write_columns_as_bam(bam, ["ref1"], cols, level=1, l_seq=l_seq, seed=seed); t_write = time.perf_counter() - t0
        rg = synth_ranges(10_000, 2000, ref_len, seed=seed + 78)
        gr = GRanges(["ref1"] * len(rg["rid"]), rg["loc"] + 1, width=rg["len"], strand=[{1: "+", -1: "-", 0: "*"}[int(x)] for x in rg["strand"]])
        orc = oracle_c.OracleReads(cols["ref_off"], cols["pos"], cols["end"], cols["flag"], cols["mapq"], cols["tlen"])
        want, _ = oracle_c.pileup_core(orc, rg, binsize=1)
        del orc
        out = dict(bam_bytes=os.path.getsize(bam), reads=n_reads, record_bytes=108 + l_seq - 4, write_bam_s=t_write,
                   workload=f"bamProfile binsize=1, 10k x 2kb ranges, {n_reads:.0e} SE {l_seq}-bp reads with names, bases, qualities")
        bases = int(rg["len"].astype(np.int64).sum())
        old = os.environ.get("BAMSIGNALS_INFLATE")
        for eng in ("default", "gpu", "cpu"):
            if eng == "default":
                os.environ.pop("BAMSIGNALS_INFLATE", None)
            else:
                os.environ["BAMSIGNALS_INFLATE"] = eng
            _lib.load().bsig_cache_clear()
            t0 = time.perf_counter(); sig = pileup_core(bam, gr, (), device=device); t_cold = time.perf_counter() - t0
            if not np.array_equal(np.concatenate(sig), want):
                raise SystemExit("file-level result on the real-shaped BAM differs from the oracle")
            dd = Reads.device_decode_timing()
            out["cold_" + eng] = dict(call_s=t_cold, Mbases_s=bases / t_cold / 1e6, stages_s=last_call_timing(),
                                      decode_stages_s=dd, route=last_call_route())
        if old is None:
            os.environ.pop("BAMSIGNALS_INFLATE", None)
        else:
            os.environ["BAMSIGNALS_INFLATE"] = old
        # the on-disk reads file: written by one cold call, loaded by the next "process" (cleared cache)
        os.environ["BAMSIGNALS_SIDECAR_DIR"] = d
        try:
            _lib.load().bsig_cache_clear()
            t0 = time.perf_counter(); pileup_core(bam, gr, (), device=device); t_make = time.perf_counter() - t0
            _lib.load().bsig_cache_clear()
            t0 = time.perf_counter(); sig = pileup_core(bam, gr, (), device=device); t_load = time.perf_counter() - t0
            if not np.array_equal(np.concatenate(sig), want) or "sidecar" not in last_call_route():
                raise SystemExit("the call from the reads file differs from the oracle")
            side = [f for f in os.listdir(d) if f.endswith(".bsig")]
            out["sidecar"] = dict(bytes=os.path.getsize(os.path.join(d, side[0])), cold_call_writing_it_s=t_make,
                                  cold_call_loading_it_s=t_load, Mbases_s=bases / t_load / 1e6)
            del sig
        finally:
            os.environ.pop("BAMSIGNALS_SIDECAR_DIR", None)
        _lib.load().bsig_cache_clear()
        b = BamFile(bam)
        dec = b.decode(threads=1)
        t_dec1 = b.decode_timing()["total"]
        del dec
        b.decode(threads=0)
        t_decN = b.decode_timing()["total"]
        b.close()
        out["cpu_decode_1_thread_s"] = t_dec1
        out["cpu_decode_all_threads_s"] = t_decN
        out["vs_cpu_path_cold"] = (t_dec1 + 0.0) / out["cold_default"]["call_s"]
        out["note"] = ("cold call under the cost model's engine choice and with each engine forced; vs_cpu_path_cold = "
                       "single-thread CPU decode alone / cold call (the pileup itself is 0.05 s on one core at this size)")
        return out
    finally:
        shutil.rmtree(d, ignore_errors=True)


class Workload:
    """One configuration resident on the GPU: reads, `nb` distinct range batches, their plans and
    result buffers."""

    def __init__(self, a, name, rank, world, local, stream, n_reads=0, n_ranges=0, width=0, nb=0):
        import torch

        from bamsignals_amd import _lib
        from bamsignals_amd.device import Context, Plan, Reads, make_params
        from bamsignals_amd.synth import synth_ranges, synth_reads
        self.name, self.cfg = name, CONFIGS[name]
        cfg = self.cfg
        self.n_reads = n_reads or cfg["reads"]
        self.n_ranges = n_ranges or cfg["ranges"]
        self.width = width or cfg["width"]
        # distinct range batches (and result buffers) the steps rotate over, so that neither reads
        # nor results sit in the 256-MiB Infinity Cache from one step to the next: 8 x ~135 MB at
        # config 2; a batch of the larger configurations (>= 1 GB per step) exceeds it by itself
        nb = nb or (8 if self.n_reads <= 100_000_000 else 2)
        t0 = time.time()
        self.cols = synth_reads(self.n_reads, cfg["ref_len"], seed=a.seed, paired=cfg["paired"], with_cigar=False)
        self.batches = []
        for b in range(nb):
            all_rg = synth_ranges(self.n_ranges * world, self.width, cfg["ref_len"], seed=a.seed + 1 + 7919 * b)
            order = np.lexsort((all_rg["loc"], all_rg["rid"]))        # sorted as the reference sorts them
            mine = order[rank::world]                                 # round-robin shard of sorted ranges
            self.batches.append({k: v[mine] for k, v in all_rg.items()})
        self.nb = nb
        self.t_gen = time.time() - t0
        log(f"{name}: generated {self.n_reads} reads, {nb} x {len(self.batches[0]['rid'])} ranges in {self.t_gen:.1f} s")
        cols = self.cols
        with torch.cuda.stream(stream):
            self.ctx = Context(local, stream=stream.cuda_stream)
            t0 = time.time()
            self.reads = Reads(self.ctx, cols["ref_len"], cols["ref_off"], cols["pos"], cols["flag"], cols["mapq"],
                               cols["tlen"], end=cols["end"])
            self.t_upload = time.time() - t0
            t0 = time.time()
            self.params = make_params(_lib.MODE_PROFILE, tile_cells=a.tile_cells, threads=a.threads, **cfg["args"])
            self.plans = [Plan(self.ctx, self.reads, g["rid"], g["loc"], g["len"], g["strand"], self.params)
                          for g in self.batches]
            self.t_plan = (time.time() - t0) / nb
            all_stats = [p.stats() for p in self.plans]
            self.stats = {k: int(round(float(np.mean([st[k] for st in all_stats])))) for k in all_stats[0]}
            self.outs = [torch.empty(max(p.cells, 4), dtype=torch.int32, device="cuda") for p in self.plans]
            self.step_bases = [int(g["len"].astype(np.int64).sum()) for g in self.batches]
            for b in range(nb):                      # every result buffer is produced at least once
                self.plans[b].run_device(self.outs[b].data_ptr())
        torch.cuda.synchronize()

    def run_steps(self, k):
        for q in range(k):
            self.plans[q % self.nb].run_device(self.outs[q % self.nb].data_ptr())

    def timed(self, steps, warmup, stream, barrier):
        """W untimed + K timed steps between barriers; returns (this rank's wall seconds, kernel_ms)."""
        import torch
        with torch.cuda.stream(stream):
            self.run_steps(warmup)
            torch.cuda.synchronize()
            barrier()
            torch.cuda.synchronize()
            # HIP events on the launch stream bracket the K launches: (e1 - e0) / K is the average
            # duration of one launch including the ~1.5 us dependent-launch boundary, i.e. an upper
            # bound of the kernel time rocprofv3 reports for the same command (profiles/)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t_start = time.perf_counter()
            e0.record(stream)
            self.run_steps(steps)
            e1.record(stream)
            torch.cuda.synchronize()
            # this rank's K steps are done: stop its clock here.  The closing barrier only lines the
            # ranks up again; the whole-job time is the MAX of the per-rank times (all-reduce in
            # main), so the collective's own latency is not billed to the steps.
            elapsed = time.perf_counter() - t_start
            barrier()
            torch.cuda.synchronize()
            return elapsed, e0.elapsed_time(e1) / steps

    def check_parity(self, oracle_c, seed, sample=500):
        """What was just timed against the oracle: a seeded random sample of `sample` ranges per batch,
        drawn over the WHOLE (rid, loc)-sorted order (all references, both ends of the genome)."""
        cols = self.cols
        orc = oracle_c.OracleReads(cols["ref_off"], cols["pos"], cols["end"], cols["flag"], cols["mapq"], cols["tlen"])
        refs_seen = set()
        for b in range(self.nb):
            g = self.batches[b]
            n = len(g["rid"])
            pick = np.sort(np.random.default_rng(seed + 31 * b).choice(n, size=min(sample, n), replace=False))
            sub = {kk: v[pick] for kk, v in g.items()}
            want, woff = oracle_c.pileup_core(orc, sub, **self.cfg["args"])
            off = self.plans[b].offsets
            got = self.outs[b][:self.plans[b].cells].cpu().numpy()
            for j, i in enumerate(pick):
                if not np.array_equal(got[off[i]:off[i + 1]], want[woff[j]:woff[j + 1]]):
                    raise SystemExit(f"HIP result differs from the oracle (batch {b}, range {i} on reference "
                                     f"{int(g['rid'][i])}): refusing to report a number")
            refs_seen.update(np.unique(sub["rid"]).tolist())
        return orc, dict(ranges_per_batch=min(sample, n), batches=self.nb, how="seeded random sample over the sorted order",
                         references_covered=len(refs_seen), references=len(self.cfg["ref_len"]))

    def roofline(self, kernel_ms, traffic=None, traffic_src=None):
        st = self.stats
        achieved = st["algorithmic_bytes"] / (kernel_ms * 1e-3) / 1e9
        return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
                "kernel": "k_profile", "kernel_ms": kernel_ms,
                "algorithmic_bytes": st["algorithmic_bytes"],
                # SURVEY 8(d)'s layout-agnostic figure 15*V + 16*I + 4*S*C for the same launch
                # (not used for `achieved`: the packed HBM layout moves fewer bytes per visit)
                "algorithmic_bytes_survey_formula": 15 * st["visits"] + 16 * st["n_ranges"] + 4 * st["cells"],
                "visits": st["visits"], "streamed_reads": st["streamed"], "visits_short": st["visits_short"],
                "bytes_per_visit_short": st["bytes_per_visit_short"],
                "bytes_per_visit_long": st["bytes_per_visit_long"],
                "items": st["n_items"], "cells": st["cells"]}

    def close(self):
        for p in self.plans:
            p.close()
        self.outs = []
        self.reads.close()
        self.ctx.close()


def committed_traffic(a, name, cfg):
    """HBM traffic per step from the committed rocprofv3 PMC passes of this same command
    (profiles/<tag>_pmc.json, made by scripts/profile_round.sh + scripts/summarize_profile.py);
    only quoted when that profile was taken on the workload and launch shape being run now."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc*.json")), reverse=True):
        try:
            pj = json.load(open(f))
        except Exception:
            continue
        if pj.get("workload") == name + ": " + cfg["desc"] and not a.threads and not a.tile_cells \
                and not a.reads and not a.ranges and not a.width and pj.get("step_hbm_bytes"):
            return pj["step_hbm_bytes"], os.path.relpath(f, ROOT)
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="NS", choices=sorted(CONFIGS))
    ap.add_argument("--reads", type=int, default=0, help="override the number of reads")
    ap.add_argument("--ranges", type=int, default=0, help="override the number of ranges per GPU")
    ap.add_argument("--width", type=int, default=0, help="override the range width")
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--tile-cells", type=int, default=0)
    ap.add_argument("--batches", type=int, default=0,
                    help="distinct range batches (and result buffers) the steps rotate over; 0 = 8 up to 1e8 reads "
                         "(so that the working set exceeds the 256 MiB Infinity Cache), 2 above")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1 (nccl = RCCL over xGMI; gloo only to exercise "
                         "the multi-rank code path on a box with fewer GPUs than ranks)")
    ap.add_argument("--pipelined", action="store_true",
                    help="also time the same steps issued alternately on two streams (informational; off by "
                         "default so that a profiler sees only the one-stream launches the metric is defined on)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed and run the collectives even with one rank (exercises RCCL "
                         "on a 1-GPU box)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true",
                    help="skip the informational end-to-end section (the workload as a BAM on disk -> host result, "
                         "next to the single-thread CPU path incl. BAM decode); it runs at N=1")
    ap.add_argument("--no-realistic", action="store_true",
                    help="skip the end-to-end section on a BAM with read names, bases and qualities")
    ap.add_argument("--no-also", action="store_true", help="skip the secondary kernel-only measurement of config 2")
    ap.add_argument("--seed", type=int, default=0xBA51)
    a = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    ngpu = torch.cuda.device_count()
    if a.backend == "nccl" and world > ngpu:
        raise SystemExit(f"{world} ranks but {ngpu} GPU(s): RCCL needs one GPU per rank")
    local = local % ngpu
    torch.cuda.set_device(local)
    cdev = "cuda" if a.backend == "nccl" else "cpu"      # where collective payloads live
    use_dist = world > 1 or a.force_dist
    if use_dist and os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":
        # RCCL prints its version banner to STDOUT at communicator creation; this program's stdout is
        # one JSON line
        os.environ["NCCL_DEBUG"] = "WARN"
    if use_dist:
        os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")      # RCCL's own messages: not on stdout
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")

    def barrier():
        if use_dist:
            dist.barrier()

    cfg = CONFIGS[a.config]
    stream = torch.cuda.Stream()
    w = Workload(a, a.config, rank, world, local, stream, a.reads, a.ranges, a.width, a.batches)
    elapsed, kernel_ms = w.timed(a.steps, a.warmup, stream, barrier)
    nb, plan, rg = w.nb, w.plans[0], w.batches[0]

    # whole-job time = max over ranks
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- informational: the same K steps issued alternately on two streams, so that the tail of
    # one launch overlaps the ramp of the next (independent batches).  NOT the reported metric:
    # `value`, `kernel_ms` and the roofline figure above are the one-stream numbers.
    pipelined = None
    if rank == 0 and nb >= 2 and a.pipelined:
        from bamsignals_amd.device import Context, Plan
        stream2 = torch.cuda.Stream()
        ctx2 = Context(local, stream=stream2.cuda_stream)
        plans2 = {b: Plan(ctx2, w.reads, w.batches[b]["rid"], w.batches[b]["loc"], w.batches[b]["len"],
                          w.batches[b]["strand"], w.params) for b in range(1, nb, 2)}

        def step2(s):
            b = s % nb
            (plans2[b] if b % 2 else w.plans[b]).run_device(w.outs[b].data_ptr())
        for s in range(min(a.warmup, 2 * nb)):
            step2(s)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for s in range(a.steps):
            step2(s)
        torch.cuda.synchronize()
        t_p = time.perf_counter() - t1
        pipelined = dict(streams=2, ms_per_step=t_p / a.steps * 1e3,
                         value=sum(w.step_bases[s % nb] for s in range(a.steps)) / t_p / 1e6, unit="Mbases/s")
        for p2 in plans2.values():
            p2.close()
        ctx2.close()

    # ---- correctness of what was just timed, then the CPU baseline (rank 0) ----------------------
    got = w.outs[0][:plan.cells].cpu().numpy()
    parity = cpu = e2e = None
    bases = w.step_bases[0]
    # (the oracle is the checker and the CPU baseline: loaded on rank 0 only, after the timed region)
    if rank == 0:
        from oracle import oracle_c
        orc, parity = w.check_parity(oracle_c, a.seed)
        log(f"parity ok ({parity}); step {kernel_ms * 1e3:.1f} us")
        if not a.no_cpu_baseline:
            # the oracle (C restatement of overlapAndPileup + Pileupper, single thread) on the
            # rank's whole workload, repeated until about 10 s have passed
            reps, t_cpu = 0, 0.0
            while t_cpu < 10.0 and reps < 200:
                t1 = time.perf_counter()
                oracle_c.pileup_core(orc, rg, **cfg["args"])
                t_cpu += time.perf_counter() - t1
                reps += 1
            cpu = dict(value=bases * reps / t_cpu / 1e6, unit="Mbases/s", cores=1, kind="port",
                       sample=f"{reps} x the full per-GPU workload ({len(rg['rid'])} ranges) on read columns "
                              f"already in RAM (BAM decode excluded), oracle/bamsignals_oracle.c, 1 thread")
            # for fairness also the same oracle sharded over the host's cores: contiguous blocks of
            # the sorted ranges, one thread each (the C call releases the GIL)
            from concurrent.futures import ThreadPoolExecutor
            ncore = max(1, min(os.cpu_count() or 1, 64))
            srt = np.lexsort((rg["loc"], rg["rid"]))
            blocks = [srt[i * len(srt) // ncore:(i + 1) * len(srt) // ncore] for i in range(ncore)]
            shards = [{kk: v[b] for kk, v in rg.items()} for b in blocks if len(b)]
            with ThreadPoolExecutor(max_workers=len(shards)) as ex:
                list(ex.map(lambda sh: oracle_c.pileup_core(orc, sh, **cfg["args"]), shards))   # warm
                reps_m, t_m = 0, 0.0
                while t_m < 3.0 and reps_m < 400:
                    t1 = time.perf_counter()
                    list(ex.map(lambda sh: oracle_c.pileup_core(orc, sh, **cfg["args"]), shards))
                    t_m += time.perf_counter() - t1
                    reps_m += 1
            cpu["multicore"] = dict(value=bases * reps_m / t_m / 1e6, unit="Mbases/s", cores=len(shards),
                                    sample=f"{reps_m} x the same workload, ranges split into {len(shards)} contiguous "
                                           f"blocks, one thread per block")
            log(f"cpu baseline: {cpu['value']:.0f} Mbases/s on 1 core, {cpu['multicore']['value']:.0f} on {len(shards)}")
        del orc

    # ---- final reassembly on rank 0 over RCCL (outside the timed region) -------------------
    gather = None
    if use_dist:
        shard = w.outs[0][:plan.cells].to(cdev)
        bufs = [torch.empty_like(shard) for _ in range(world)] if rank == 0 else None
        torch.cuda.synchronize(); dist.barrier()
        t1 = time.perf_counter()
        dist.gather(shard, bufs, dst=0)
        torch.cuda.synchronize()
        t_g = time.perf_counter() - t1
        sums = torch.tensor([int(got.astype(np.int64).sum())], dtype=torch.int64, device=cdev)
        allsums = [torch.zeros_like(sums) for _ in range(world)]
        dist.all_gather(allsums, sums)
        if rank == 0:
            ok = all(int(b.sum(dtype=torch.int64).item()) == int(s.item()) for b, s in zip(bufs, allsums))
            if not ok:
                raise SystemExit("gathered shards do not match the per-rank checksums")
            gather = dict(ms=t_g * 1e3, bytes=int(shard.numel() * 4 * (world - 1)),
                          GBps=shard.numel() * 4 * (world - 1) / t_g / 1e9, checked=True, backend=a.backend,
                          ranks=world)
        del bufs, shard

    res = None
    if rank == 0:
        traffic, traffic_src = committed_traffic(a, a.config, cfg)
        total_bases = sum(w.step_bases[s % nb] for s in range(a.steps)) * world
        res = {
            "metric": "Mbases profiled/sec (bamProfile binsize=1)", "value": total_bases / elapsed / 1e6, "unit": "Mbases/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": a.config + ": " + cfg["desc"], "reads": w.n_reads,
                       "ranges_per_gpu": len(rg["rid"]), "range_width": w.width, "batches": nb,
                       "parallelism": f"ranges round-robin over {world} GPU(s), reads replicated",
                       "launch": "one launch per step from the host loop",
                       "threads": w.params.threads or 64,
                       "tile_cells": w.params.tile_cells or "auto (widest range, at most 2048)"},
            "roofline": w.roofline(kernel_ms, traffic, traffic_src),
            "cpu_baseline": cpu,
            "end_to_end": None,
            "parity_checked": parity,
            "pipelined_two_streams": pipelined,
            "gather": gather,
            "setup_s": {"generate": w.t_gen, "upload_and_layout": w.t_upload, "plan": w.t_plan},
            "reads_in_hbm": w.reads.info(),
        }

    # ---- N = 1 extras: the same workload from a BAM file, and config 2's small launch ---------------
    if rank == 0 and world == 1 and not use_dist:
        cols = w.cols
        want_flat = got[:plan.cells]
        w.close()
        w.cols = None
        torch.cuda.empty_cache()
        if cpu is not None and not a.no_e2e:
            try:
                res["end_to_end"], cpu["with_bam_decode"] = end_to_end(cfg, cols, rg, want_flat, local, oracle_c)
            except Exception as exc:      # e.g. no room for the BAM on local disk: the metric does not depend on it
                res["end_to_end"] = {"error": f"{type(exc).__name__}: {exc}"}
        del cols, want_flat
        if cpu is not None and not a.no_e2e and not a.no_realistic:
            try:
                res["end_to_end_realistic"] = end_to_end_realistic(a.seed, local, oracle_c)
                log("end_to_end_realistic: " + ", ".join(f"{k[5:]} {v['call_s']:.3f} s" for k, v in res["end_to_end_realistic"].items()
                                                        if k.startswith("cold_")))
            except Exception as exc:
                res["end_to_end_realistic"] = {"error": f"{type(exc).__name__}: {exc}"}
        if a.config != "C2" and not a.no_also and not (a.reads or a.ranges or a.width):
            w2 = Workload(a, "C2", 0, 1, local, stream)
            k2 = max(a.steps, 200)
            el2, kms2 = w2.timed(k2, max(a.warmup, 20), stream, lambda: None)
            _, par2 = w2.check_parity(oracle_c, a.seed)
            tr2, src2 = committed_traffic(a, "C2", w2.cfg)
            res["also"] = {"C2": {"workload": "C2: " + w2.cfg["desc"], "steps": k2, "ms_per_step": el2 / k2 * 1e3,
                                  "value": sum(w2.step_bases[s % w2.nb] for s in range(k2)) / el2 / 1e6, "unit": "Mbases/s",
                                  "batches": w2.nb, "roofline": w2.roofline(kms2, tr2, src2), "parity_checked": par2}}
            w2.close()

    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # anything native code left in C stdio buffers goes out first: the JSON line is the last line
        import ctypes
        ctypes.CDLL(None).fflush(None)
        sys.stdout.flush()
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
