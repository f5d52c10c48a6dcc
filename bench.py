#!/usr/bin/env python3
"""bench.py — Mbases profiled / s for bamProfile(binsize=1) on MI355X (BASELINE.json's metric).

A "step" is one pass of the hot path (HIP kernel k_profile behind bsig_plan_run) over one batch
of ranges, with the read columns and the range work items already resident in HBM and the result
left in HBM.  Default workload = the shape BASELINE.json's north star quotes its 1-GPU target on
("NS"): 100,000 x 2 kb ranges, 5e8 synthetic single-end reads on 10 x 250 Mbp (4.2 GB resident).
At N = 1 the same run also reports

  * "also"/"C2": the kernel-only step of BASELINE config 2 (10k x 2 kb, 5e7 reads) — the small launch;
  * "end_to_end": the file-level call bamProfile(bampath, GRanges) on the SAME workload written to
    local disk as a BAM (cold: open + GPU inflate + parse + HBM layout + kernels + result in host
    memory; warm: BAM resident), next to the CPU path including the BAM decode (`cpu_baseline.
    with_bam_decode`): informational, never `value`.

Multi-GPU (launched by torch.distributed.run, one rank per GPU): the SAME workload at FIXED total size
("scaling": "strong").  Ranges are independent units (each owns its output, ref: src/bamsignals.cpp:164,181,186):
the (rid, loc)-sorted ranges (ref: :222-226,246) are dealt round-robin to the ranks, every rank holds the reads,
and a step is the whole north-star path -- the kernels on the rank's shard, the gather of the shards to rank 0
over RCCL (torch.distributed.gather: grouped send / recv, every peer straight to the root over its own xGMI link), and the
reassembly into the caller's range order in rank 0's HBM (bsig_segmap_run).  A shard travels as two bits a cell + a
list of exceptions (--wire narrow, the default: lossless, ~1/15 of the int32 bytes; packed by bsig_narrow_pack and widened
by bsig_segmap_run_narrow while it is put in place; --wire int32 sends the cells as they are).  All of it is inside the timed
region; `value` = the whole range set's bases x K / the slowest rank's time.  At N = 1 there is nothing to gather
and the step is the launch alone.  The result assembled on rank 0 is compared with the oracle, every range of
every batch.  The reads are generated ONCE per node (local rank 0 -> .npy files in /dev/shm, the other ranks map
them), so the N-rank run costs one generation and one copy of the columns in host memory, not N.

A run with N > 1 also carries (informational, never `value`):
  * "no_collective": every rank running the WHOLE range set by itself with nothing exchanged (N independent
    replicas: what rounds 1-4 reported as a weak-scaling `value`; linear by construction);
  * "strong": BASELINE config 5 at FIXED total size -- 1M x 1 kb ranges over 1e9 reads on 24
    references, 1M/N sorted ranges per rank -- with the whole north-star step in the timed region:
    kernel on the rank's shard + RCCL gather of the shards to rank 0 + reassembly into the caller's
    range order in rank 0's HBM (bsig_segmap_run); next to the same 1M ranges run by rank 0's GPU
    alone, and checked cell by cell against that result and on a sample against the oracle;
  * "in_process": the route an R session uses -- ONE process driving all N GPUs through the
    file-level call (BAMSIGNALS_DEVICES=0..N-1) on the same data written as a BAM: cold call
    (sharded decode + column all-gather) and warm calls under each gather route.

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


def _cold_call_child(spec_path):
    """Child process of the end-to-end sections: a fresh session (HIP context up, nothing else allocated or freed
    yet) makes the cold file-level call, then the same call again with the BAM resident.  No torch here."""
    spec = json.load(open(spec_path))
    for k, v in spec.get("env", {}).items():
        os.environ[k] = v
    from bamsignals_amd import GRanges, _lib
    from bamsignals_amd.device import Context, Reads
    from bamsignals_amd.wrappers import last_call_route, last_call_timing, pileup_core
    z = np.load(spec["ranges"])
    names = spec["names"]
    gr = GRanges([names[r] for r in z["rid"]], z["loc"] + 1, width=z["len"],
                 strand=[{1: "+", -1: "-", 0: "*"}[int(x)] for x in z["strand"]])
    call = spec["call"]
    call["tlen_filter"] = tuple(call["tlen_filter"])
    t0 = time.perf_counter(); Context(max(spec["device"], 0)).close(); t_ctx = time.perf_counter() - t0      # the HIP context
    out = dict(hip_context_s=t_ctx, calls=[])
    flat = None
    per_rep_env = spec.get("per_rep_env") or []
    for rep in range(spec.get("reps", 2)):
        if rep < len(per_rep_env):
            for k, v in per_rep_env[rep].items():
                os.environ[k] = v
        t0 = time.perf_counter(); sig = pileup_core(spec["bam"], gr, **call); dt = time.perf_counter() - t0
        st = last_call_timing()
        if rep == 0:
            st["decode_stages_s"] = Reads.device_decode_timing()
        out["calls"].append(dict(call_s=dt, stages_s=st, route=last_call_route()))
        cur = np.concatenate([np.asarray(m).T.reshape(-1) if call.get("ss") else np.asarray(m) for m in sig]) if len(sig) else np.zeros(0, np.int32)
        del sig
        if rep == 0:
            flat = cur
        elif not np.array_equal(cur, flat):
            raise SystemExit(f"call {rep} of the child differs from its first call")
        del cur
    out["host_cpus_used"] = int(_lib.load().bsig_effective_cpus())
    if spec.get("result"):
        np.save(spec["result"], flat)
    # (large results are compared by value count, sum and a position-weighted sum instead of travelling back)
    out["result_cells"] = int(flat.size)
    out["result_sum"] = int(flat.sum(dtype=np.int64))
    out["result_wsum"] = int((flat.astype(np.int64) * (np.arange(flat.size, dtype=np.int64) % 1000003)).sum()) if flat.size else 0
    print(json.dumps(out), flush=True)


def result_fingerprint(flat):
    flat = np.asarray(flat)
    return (int(flat.size), int(flat.sum(dtype=np.int64)),
            int((flat.astype(np.int64) * (np.arange(flat.size, dtype=np.int64) % 1000003)).sum()) if flat.size else 0)


def cold_call_in_fresh_process(workdir, tag, bam, names, rg, call, device, env=None, reps=2, arena_gb=0, per_rep_env=None,
                               want_result=True, timeout=None):
    """The cold file-level call as a NEW session sees it.  Why a child process: on this platform a hipMalloc
    stalls for 2-3 s once about 70 GB have been freed since the last stall (plain HIP, scripts/hipmalloc_stalls.py),
    and by the time the end-to-end sections run this process has allocated and freed well over that -- the
    stall would land in the timed call at random (it did: 0.47-0.79 s from run to run in round 2).  A session
    that opens its first BAM has no such debt.  Returns (child's JSON, flat result of the cold call or None)."""
    import subprocess
    spec = dict(bam=bam, names=list(names), ranges=os.path.join(workdir, tag + "_ranges.npz"), call=dict(call), device=int(device),
                env=dict(env or {}), result=os.path.join(workdir, tag + "_result.npy") if want_result else None, reps=reps,
                per_rep_env=list(per_rep_env or []))
    if arena_gb and "BAMSIGNALS_ARENA_GB" not in os.environ:
        spec["env"]["BAMSIGNALS_ARENA_GB"] = str(int(arena_gb))
    spec["call"].pop("device", None)
    spec["call"]["device"] = int(device)
    spec["call"]["tlen_filter"] = list(spec["call"].get("tlen_filter", ()))
    np.savez(spec["ranges"], **{k: np.asarray(v) for k, v in rg.items()})
    sp = os.path.join(workdir, tag + "_spec.json")
    json.dump(spec, open(sp, "w"))
    child_env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        child_env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child-cold", sp], capture_output=True, text=True, env=child_env,
                       timeout=timeout)
    if r.returncode != 0:
        raise RuntimeError(f"cold-call child failed ({r.returncode}): {r.stderr[-800:]}")
    if os.environ.get("BSIG_DIAG_DECODE"):                    # (diagnostic: the child's stage marks)
        sys.stderr.write(f"[bench] child {tag}:\n{r.stderr}\n")
    out = json.loads(r.stdout.strip().splitlines()[-1])
    if os.environ.get("BSIG_DIAG_DECODE"):
        out["stderr"] = r.stderr
    flat = None
    if want_result:
        flat = np.load(spec["result"])
        os.remove(spec["result"])
    return out, flat


def driver_reclaim_probe(gb=96):
    """Plain HIP in a process of its own (ctypes on libamdhip64, nothing of this repository): 8-GB hipMalloc calls up to
    `gb` GB, timed one by one, then freed.  On this platform the memory that EXITED processes had allocated is reclaimed
    lazily, and the next process to allocate pays for it inside one of its hipMalloc calls (scripts/hipmalloc_after_exit.py:
    3.6 s once after a process that had used 120 GB) -- the GPU test suite or a profiler run in front of the bench leaves
    such a debt, and the first cold-call session would pay it inside its decode's allocations (it did: 0.83 / 0.52 / 0.28 s
    for three otherwise identical sessions).  The probe shows whether there was a debt (its slowest call) and settles it,
    so that the sessions below measure a session's first call and not the previous tenant's clean-up."""
    import subprocess
    code = (
        "import ctypes as C, time, json\n"
        "h = C.CDLL('libamdhip64.so')\n"
        "h.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]; h.hipFree.argtypes = [C.c_void_p]\n"
        "ps, ts = [], []\n"
        "p = C.c_void_p(); h.hipMalloc(C.byref(p), 256); h.hipFree(p)\n"
        "fr, to = C.c_size_t(), C.c_size_t(); h.hipMemGetInfo(C.byref(fr), C.byref(to))\n"
        # never more than what is free now less 16 GB: other users of this GPU (the parent's resident workload) keep theirs
        f"for _ in range(min({int(gb) // 8}, max(0, (fr.value - (16 << 30)) >> 33))):\n"
        "    p = C.c_void_p(); t = time.perf_counter(); rc = h.hipMalloc(C.byref(p), 8 << 30); ts.append(time.perf_counter() - t)\n"
        "    if rc != 0: break\n"
        "    ps.append(p)\n"
        "for p in ps: h.hipFree(p)\n"
        "print(json.dumps(dict(calls=len(ts), bytes_per_call=8 << 30, slowest_s=max(ts + [0.0]), total_s=sum(ts), free_bytes_before=fr.value)))\n")
    try:
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
        out = json.loads(r.stdout.strip().splitlines()[-1])
        out["what"] = ("plain hipMalloc calls in a process of their own, in front of the cold-call sessions: a slowest call of "
                       "seconds is the driver reclaiming what earlier processes freed")
        return out
    except Exception as exc:
        return {"error": f"{type(exc).__name__}: {exc}"}


def _settle(path):
    """A BAM the bench has just written is still dirty in the page cache and its pages have never been read;
    the timed calls are meant to see a file that has been on disk for a while and has been used before (cached,
    clean, on the active list: mapping such a file is 4-8 x faster): flush it and read it twice."""
    for p in (path, path + ".bai"):
        fd = os.open(p, os.O_RDONLY)
        try:
            os.fsync(fd)
        finally:
            os.close(fd)
    for _ in range(2):
        with open(path, "rb") as fh:
            while fh.read(64 << 20):
                pass


def end_to_end(cfg, cols, rg, want_flat, device, oracle_c):
    """Informational (NOT `value`): the file-level call bamProfile(bampath, gr) on the bench's own reads
    written to local disk as a BAM -- BGZF inflate, records -> columns, HBM layout, kernels, result in
    host memory -- cold and again with the BAM resident in HBM, next to the CPU path that includes the
    BAM decode (one thread: this repo's BGZF/BAM reader, htslib being absent, + the oracle), as the
    reference's own call does (ref: src/bamsignals.cpp:271 bam_itr_next inside the pileup loop)."""
    import shutil
    import tempfile

    from bamsignals_amd.bamio import BamFile, write_columns_as_bam
    from bamsignals_amd.synth import add_cigar
    args = cfg["args"]           # (oracle_c: the checker and CPU baseline, handed in by the cpu_baseline leg)
    d = tempfile.mkdtemp(prefix="bsig_bench_", dir=os.environ.get("TMPDIR", "/tmp"))
    try:
        if "cigar" not in cols:
            add_cigar(cols)
        names = ["ref%d" % (i + 1) for i in range(len(cfg["ref_len"]))]
        bam = os.path.join(d, "synth.bam")
        t0 = time.perf_counter(); write_columns_as_bam(bam, names, cols, level=1); t_write = time.perf_counter() - t0
        _settle(bam)
        cols.pop("cigar"); cols.pop("cigar_off")
        log(f"end_to_end: wrote {os.path.getsize(bam) / 1e6:.0f} MB BAM in {t_write:.1f} s")
        call = dict(tlen_filter=args.get("tlen_filter", ()), mapqual=args.get("mapqual", 0), binsize=args.get("binsize", 1),
                    shift=args.get("shift", 0), ss=args.get("ss", False), requiredF=args.get("requiredF", 0),
                    filteredF=args.get("filteredF", -1), pe_mid=args.get("pe_mid", False), device=device)
        # (no arena: measured on the same file, scripts/cold_call_scan_ab.py, a 44-GB arena makes the call 0.04 s
        # slower -- first touches of its memory -- and it is only insurance against rare allocation stalls)
        arena_gb = 0
        # whatever ran on this GPU before the bench (the GPU tests, a profiler) left freed memory for the driver to reclaim:
        # shown, and settled, by plain hipMalloc calls in a process of their own
        # ... and one session IN FRONT of that probe: what a session's first call costs behind whatever tenant came
        # before (it is listed by itself, `cold_call_before_the_probe_s`, and is not among the three below)
        first, flat0 = cold_call_in_fresh_process(d, "ns_unsettled", bam, names, rg, call, device, arena_gb=arena_gb)
        if not np.array_equal(flat0, want_flat):
            raise SystemExit("file-level result differs from the resident-column result")
        del flat0
        reclaim = driver_reclaim_probe()
        log(f"end_to_end: first session {first['calls'][0]['call_s']:.3f} s; driver reclaim probe {reclaim}")
        # three sessions, the median one reported: a session's first call now and then catches an allocation stall
        # of the driver (r03: one run in three to five, 0.5 s instead of 0.28 s; profiles/NOTES_r01_r03.md) -- all three are listed
        child, flat = cold_call_in_fresh_process(d, "ns", bam, names, rg, call, device, arena_gb=arena_gb)
        if not np.array_equal(flat, want_flat):
            raise SystemExit("file-level result differs from the resident-column result")
        fp = result_fingerprint(flat)
        del flat
        sessions = [child]
        for k in (1, 2):
            c, _ = cold_call_in_fresh_process(d, "ns%d" % k, bam, names, rg, call, device, arena_gb=arena_gb, want_result=False)
            if (c["result_cells"], c["result_sum"], c["result_wsum"]) != fp:
                raise SystemExit("file-level result of a repeated session differs")
            sessions.append(c)
        cold_samples = [c["calls"][0]["call_s"] for c in sessions]
        # a stage of one session well above the same stage of the others names where that session's time went
        # (alloc_*: seconds inside the driver's allocator, metered by the library)
        stage_keys = ("open", "decode", "upload_and_layout", "plan", "kernels", "download", "alloc_total")
        # (against the FASTEST session's stage: with three sessions a median is itself an outlier when two of them stall)
        min_stage = {k: float(min(c["calls"][0]["stages_s"].get(k, 0.0) for c in sessions)) for k in stage_keys}
        outliers = [dict(session=i, stage=k, seconds=c["calls"][0]["stages_s"].get(k, 0.0), fastest_session=min_stage[k])
                    for i, c in enumerate(sessions) for k in stage_keys
                    if c["calls"][0]["stages_s"].get(k, 0.0) > max(2 * min_stage[k], min_stage[k] + 0.03)]
        child = sorted(sessions, key=lambda c: c["calls"][0]["call_s"])[1]
        t_cold, t_warm = child["calls"][0]["call_s"], child["calls"][1]["call_s"]
        stages = child["calls"][0]["stages_s"]
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
        dd = stages["decode_stages_s"]
        gpu = dict(bam_bytes=os.path.getsize(bam), write_bam_s=t_write,
                   cold_call_s=t_cold, cold_call_sessions_s=cold_samples, cold_call_min_s=min(cold_samples),
                   cold_call_median_s=float(np.median(cold_samples)), cold_call_max_s=max(cold_samples),
                   cold_call_sessions_stages_s=[{k: c["calls"][0]["stages_s"].get(k) for k in stage_keys + ("alloc_calls", "reserved_bytes", "reservation_wait")}
                                                for c in sessions],
                   stages_well_above_the_fastest_session=outliers, driver_reclaim_probe=reclaim,
                   cold_call_before_the_probe_s=first["calls"][0]["call_s"],
                   cold_call_before_the_probe_stages_s={k: first["calls"][0]["stages_s"].get(k) for k in stage_keys + ("alloc_calls",)},
                   cold_call_stages_s=stages, warm_call_s=t_warm,
                   # the compressed file's trip into HBM: what the call waited for it, and the file size over the
                   # whole decode (block scan + copies + inflate + parse), i.e. the ingest rate the cold call sees
                   copy_wait_s=dd.get("copy_wait"), decode_ingest_GBps=os.path.getsize(bam) / max(dd.get("total") or 1e-9, 1e-9) / 1e9,
                   host_cpus_used=child["host_cpus_used"], hip_context_s=child["hip_context_s"], route=child["calls"][0]["route"],
                   measured_in="fresh child processes with their HIP context up (a new session's first BAM; see cold_call_in_fresh_process), "
                               "run while the parent still holds its resident workload (nothing has been freed on the GPU since the "
                               "timed steps): three sessions behind the driver_reclaim_probe, the median one's call and stages reported, all "
                               "three listed with their stages; cold_call_before_the_probe_s is one more session run IN FRONT of the probe "
                               "(what a first call costs behind whatever this GPU's previous tenant left for the driver to reclaim)",
                   arena_gb=arena_gb,
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
    this is what the decode stage costs on literal-heavy DEFLATE blocks.  The file is written at zlib level 1,
    like the bench's own; one more cold call runs on the same records at level 6, htslib's default."""
    import shutil
    import tempfile

    from bamsignals_amd.bamio import BamFile, write_columns_as_bam
    from bamsignals_amd.synth import synth_ranges, synth_reads
    ref_len, n_reads, l_seq = [250_000_000], 20_000_000, 100
    d = tempfile.mkdtemp(prefix="bsig_bench_real_", dir=os.environ.get("TMPDIR", "/tmp"))
    try:
        cols = synth_reads(n_reads, ref_len, seed=seed + 77, with_cigar=True)
        bam = os.path.join(d, "real.bam")
        t0 = time.perf_counter(); write_columns_as_bam(bam, ["ref1"], cols, level=1, l_seq=l_seq, seed=seed); t_write = time.perf_counter() - t0
        rg = synth_ranges(10_000, 2000, ref_len, seed=seed + 78)
        orc = oracle_c.OracleReads(cols["ref_off"], cols["pos"], cols["end"], cols["flag"], cols["mapq"], cols["tlen"])
        want, _ = oracle_c.pileup_core(orc, rg, binsize=1)
        del orc
        out = dict(bam_bytes=os.path.getsize(bam), reads=n_reads, record_bytes=108 + l_seq - 4, write_bam_s=t_write,
                   workload=f"bamProfile binsize=1, 10k x 2kb ranges, {n_reads:.0e} SE {l_seq}-bp reads with names, bases, qualities")
        bases = int(rg["len"].astype(np.int64).sum())
        # every engine sees the file in the same state (the first of three otherwise identical cold calls paid
        # 0.08-0.14 s more in its block scan while the freshly written pages were still under write-back)
        _settle(bam)
        call = dict(tlen_filter=(), device=device)
        for eng in ("default", "gpu", "cpu"):
            env = {} if eng == "default" else {"BAMSIGNALS_INFLATE": eng}
            child, flat = cold_call_in_fresh_process(d, "real_" + eng, bam, ["ref1"], rg, call, device, env=env, reps=1)
            if not np.array_equal(flat, want):
                raise SystemExit("file-level result on the real-shaped BAM differs from the oracle")
            c0 = child["calls"][0]
            out["cold_" + eng] = dict(call_s=c0["call_s"], Mbases_s=bases / c0["call_s"] / 1e6, stages_s=c0["stages_s"],
                                      decode_stages_s=c0["stages_s"]["decode_stages_s"], route=c0["route"])
        # the on-disk reads file: written by one cold call, loaded by the next process
        child, flat = cold_call_in_fresh_process(d, "side_w", bam, ["ref1"], rg, call, device, env={"BAMSIGNALS_SIDECAR_DIR": d}, reps=1)
        t_make = child["calls"][0]["call_s"]
        child, flat = cold_call_in_fresh_process(d, "side_r", bam, ["ref1"], rg, call, device, env={"BAMSIGNALS_SIDECAR_DIR": d}, reps=1)
        t_load = child["calls"][0]["call_s"]
        if not np.array_equal(flat, want) or "sidecar" not in child["calls"][0]["route"]:
            raise SystemExit("the call from the reads file differs from the oracle")
        side = [f for f in os.listdir(d) if f.endswith(".bsig")]
        out["sidecar"] = dict(bytes=os.path.getsize(os.path.join(d, side[0])), cold_call_writing_it_s=t_make,
                              cold_call_loading_it_s=t_load, Mbases_s=bases / t_load / 1e6)
        out["measured_in"] = "fresh child processes with their HIP context up (see cold_call_in_fresh_process)"
        del flat
        # the same records at zlib level 6, htslib's default (the files above are level 1, like the bench's own BAM):
        # shorter literal runs, more matches and several deflate blocks -- code tables -- per BGZF block
        bam6 = os.path.join(d, "real6.bam")
        t0 = time.perf_counter(); write_columns_as_bam(bam6, ["ref1"], cols, level=6, l_seq=l_seq, seed=seed); t_write6 = time.perf_counter() - t0
        _settle(bam6)
        child, flat = cold_call_in_fresh_process(d, "real6", bam6, ["ref1"], rg, call, device, reps=1)
        if not np.array_equal(flat, want):
            raise SystemExit("file-level result on the level-6 real-shaped BAM differs from the oracle")
        del flat
        c0 = child["calls"][0]
        out["cold_default_zlib_level_6"] = dict(bam_bytes=os.path.getsize(bam6), write_bam_s=t_write6, call_s=c0["call_s"],
                                                Mbases_s=bases / c0["call_s"] / 1e6, decode_stages_s=c0["stages_s"]["decode_stages_s"],
                                                route=c0["route"])
        os.remove(bam6); os.remove(bam6 + ".bai")
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
        out["vs_cpu_path_cold_best_engine"] = (t_dec1 + 0.0) / min(out["cold_default"]["call_s"], out["cold_gpu"]["call_s"], out["cold_cpu"]["call_s"])
        out["note"] = ("cold call under the cost model's engine choice and with each engine forced; vs_cpu_path_cold = "
                       "single-thread CPU decode alone / cold call (the pileup itself is 0.05 s on one core at this size)")
        return out
    finally:
        shutil.rmtree(d, ignore_errors=True)


def host_ram_available():
    try:
        import psutil
        return int(psutil.virtual_memory().available)
    except Exception:
        return None


class SharedReads:
    """The synthetic read columns of one configuration, generated once per node: local rank 0 runs the
    generator and writes the columns as .npy files (to /dev/shm when it has room, else TMPDIR), the other
    ranks map them read-only after a barrier.  N ranks then cost one generation (26 s for 5e8 reads, 49 s for
    1e9) and ONE copy of the columns in host memory (15 B per read) instead of N of each."""

    KEYS = ("ref_len", "ref_off", "pos", "flag", "mapq", "tlen", "end")

    def __init__(self, tag, n_reads, ref_len, seed, paired, rank, world, barrier):
        import shutil

        from bamsignals_amd.synth import synth_reads
        self.dir = None
        self.leader = rank == 0
        t0 = time.time()
        need_gen = int(n_reads) * 48            # generator temporaries (int64 positions, sort buffers) + columns
        need_shm = int(n_reads) * 16
        if world == 1:
            self.cols = synth_reads(n_reads, ref_len, seed=seed, paired=paired, with_cigar=False)
            self.t_gen = time.time() - t0
            return
        root = None
        for cand in ("/dev/shm", os.environ.get("TMPDIR", "/tmp")):
            try:
                if shutil.disk_usage(cand).free > need_shm * 1.2:
                    root = cand
                    break
            except OSError:
                pass
        self.dir = os.path.join(root or "/tmp", f"bsig_bench_{os.environ.get('MASTER_PORT', '0')}_{tag}")
        if self.leader:
            avail = host_ram_available()
            if root is None or (avail is not None and avail < need_gen + (need_shm if root == "/dev/shm" else 0)):
                raise SystemExit(f"bench.py: not enough host memory / scratch space for {n_reads} synthetic reads "
                                 f"(available RAM {avail}, need about {need_gen + need_shm} bytes)")
            shutil.rmtree(self.dir, ignore_errors=True)
            os.makedirs(self.dir)
            self.cols = synth_reads(n_reads, ref_len, seed=seed, paired=paired, with_cigar=False)
            for k in self.KEYS:
                np.save(os.path.join(self.dir, k + ".npy"), self.cols[k])
            log(f"{tag}: generated {n_reads} reads once for {world} ranks in {time.time() - t0:.1f} s -> {self.dir}")
        barrier()
        if not self.leader:
            self.cols = {k: np.load(os.path.join(self.dir, k + ".npy"), mmap_mode="r") for k in self.KEYS}
        self.t_gen = time.time() - t0

    def release(self, barrier):
        """After every rank has uploaded: the files go (the leader keeps its in-memory columns)."""
        import shutil
        if self.dir is None:
            return
        if not self.leader:
            self.cols = None
        barrier()
        if self.leader:
            shutil.rmtree(self.dir, ignore_errors=True)
        self.dir = None


class Workload:
    """One configuration resident on the GPU: reads, `nb` distinct range batches, their plans and result buffers.
    Every batch is ONE range set of `n_ranges` ranges in (rid, loc) order (the order the reference sorts them in);
    with `world` > 1 this rank's plans hold its round-robin shard of each batch (ranges rank, rank + world, ...)
    and, on every rank, a second set of plans holds the whole batch (rank 0: what the assembled result must equal;
    all ranks: the informational replica run)."""

    def __init__(self, a, name, rank, world, local, stream, n_reads=0, n_ranges=0, width=0, nb=0, cols=None, t_gen=0.0, gathered=None):
        import torch

        from bamsignals_amd import _lib
        from bamsignals_amd.device import Context, Plan, Reads, layout, make_params
        from bamsignals_amd.synth import synth_ranges, synth_reads
        self.name, self.cfg = name, CONFIGS[name]
        cfg = self.cfg
        self.rank, self.world = rank, world
        # is a step "shard -> gather to rank 0 -> reassembly"?  With more than one rank always; with one rank only when
        # asked (--force-dist: the same code path over a one-rank communicator, for a box with one GPU)
        self.gathered = (world > 1) if gathered is None else bool(gathered)
        self.n_reads = n_reads or cfg["reads"]
        self.n_ranges = n_ranges or cfg["ranges"]            # of the WHOLE job, whatever the number of ranks
        self.width = width or cfg["width"]
        # distinct range batches (and result buffers) the steps rotate over, so that neither reads
        # nor results sit in the 256-MiB Infinity Cache from one step to the next: 8 x ~135 MB at
        # config 2; a batch of the larger configurations (>= 1 GB per step) exceeds it by itself
        nb = nb or (8 if self.n_reads <= 100_000_000 else 2)
        t0 = time.time()
        self.cols = cols if cols is not None else synth_reads(self.n_reads, cfg["ref_len"], seed=a.seed, paired=cfg["paired"], with_cigar=False)
        self.full, self.batches = [], []
        for b in range(nb):
            all_rg = synth_ranges(self.n_ranges, self.width, cfg["ref_len"], seed=a.seed + 1 + 7919 * b)
            order = np.lexsort((all_rg["loc"], all_rg["rid"]))        # sorted as the reference sorts them
            full = {k: v[order] for k, v in all_rg.items()}
            self.full.append(full)
            self.batches.append({k: v[rank::world] for k, v in full.items()} if world > 1 else full)   # round-robin shard
        self.nb = nb
        self.t_gen = time.time() - t0 + t_gen
        if rank == 0:
            log(f"{name}: {self.n_reads} reads, {nb} x {len(self.full[0]['rid'])} ranges"
                + (f" ({len(self.batches[0]['rid'])} per rank)" if world > 1 else "") + f" ready in {self.t_gen:.1f} s")
        cols = self.cols
        ss = bool(cfg["args"].get("ss", False))
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
            self.step_bases = [int(g["len"].astype(np.int64).sum()) for g in self.full]      # of the whole job
            # the shards travel as equal-sized messages (the largest shard of any batch, in cells)
            self.shard_offs = [[layout(g["len"][r::world], cfg["args"].get("binsize", 1), ss) for r in range(world)] for g in self.full] \
                if self.gathered else None
            self.pad = max(4, max(int(o[-1]) for offs in self.shard_offs for o in offs)) if self.gathered else 0
            # (zeros: the cells between a shard's end and the common message length travel too)
            self.outs = [torch.zeros(max(p.cells, 4, self.pad), dtype=torch.int32, device="cuda") for p in self.plans]
            for b in range(nb):                      # every result buffer is produced at least once
                self.plans[b].run_device(self.outs[b].data_ptr())
            # (after a first run: a plan's later steps -- the timed ones -- may read kept windows where its first looked them
            # up, and the algorithmic bytes are those of the form that is timed)
            all_stats = [p.stats() for p in self.plans]
            self.stats = {k: int(round(float(np.mean([st[k] for st in all_stats])))) for k in all_stats[0]}
            self.full_plans, self.full_outs = self.plans, self.outs
            if self.gathered:
                self.full_plans = [Plan(self.ctx, self.reads, g["rid"], g["loc"], g["len"], g["strand"], self.params) for g in self.full]
                self.full_outs = [torch.empty(max(p.cells, 4), dtype=torch.int32, device="cuda") for p in self.full_plans]
        self.final = self.outs                        # where a step's whole result ends up (N > 1: setup_gather)
        self.bufs = self.maps = None
        torch.cuda.synchronize()

    def setup_gather(self, dist, backend, wire="int32", host_group=None):
        """N > 1: rank 0's receive buffers (one per peer), the whole job's result buffer of every batch, and the
        segment maps that put shard r's ranges at their place in the caller's order (range i of the batch is range
        i // world of shard i % world)."""
        import torch

        from bamsignals_amd.device import SegmentMap, layout
        self.dist, self.backend, self.wire = dist, backend, wire
        self.cap = 0
        if wire == "narrow":
            # The narrow wire's exception lists must have room: a plan's result is a function of plan and reads, so the
            # count of this rank's shard of every batch (all of which have been run once) is exact; every rank sends
            # messages of ONE length, sized by the largest shard and the longest list of any rank and batch.
            from bamsignals_amd.device import narrow_bytes, narrow_count, narrow_pack
            probe = torch.empty(narrow_bytes(self.pad, 0) // 4, dtype=torch.int32, device="cuda")
            worst = 0
            for b in range(self.nb):
                narrow_pack(self.ctx, self.outs[b].data_ptr(), self.pad, probe.data_ptr(), 0)
                worst = max(worst, narrow_count(self.ctx, probe.data_ptr()))
            allw = [None] * self.world
            dist.all_gather_object(allw, int(worst), group=host_group)
            self.cap = int(max(allw))
            self.msg_words = narrow_bytes(self.pad, self.cap) // 4
            self.msg = [torch.empty(self.msg_words, dtype=torch.int32, device="cuda") for _ in range(self.nb)]
            del probe
        if self.rank != 0:
            return
        cfg, world = self.cfg, self.world
        ss = bool(cfg["args"].get("ss", False))
        n = len(self.full[0]["rid"])
        self.bufs = [torch.empty(self.msg_words if wire == "narrow" else self.pad, dtype=torch.int32, device="cuda") for _ in range(world)]
        self.final = [torch.zeros(max(p.cells, 4), dtype=torch.int32, device="cuda") for p in self.full_plans]
        self.maps = []
        for b in range(self.nb):
            off_all = layout(self.full[b]["len"], cfg["args"].get("binsize", 1), ss)
            assert int(off_all[-1]) == self.full_plans[b].cells
            self.maps.append([SegmentMap(self.ctx, self.shard_offs[b][r], off_all, np.arange(r, n, world, dtype=np.int64))
                              for r in range(world)])
        self.gather_bytes = [int(sum(int(o[-1]) for o in offs[1:]) * 4) if wire != "narrow" else 4 * self.msg_words * (world - 1)
                             for offs in self.shard_offs]
        torch.cuda.synchronize()

    def strong_step(self, q, ev=None):
        """One step of the N-rank job: kernels on the shard, shards to rank 0, every range to its place there."""
        import torch
        dist, b = self.dist, q % self.nb
        root = self.rank == 0
        if ev:
            ev[0].record()
        self.plans[b].run_device(self.outs[b].data_ptr())
        if ev:
            ev[1].record()
        narrow = self.wire == "narrow"
        # The root's own shard crosses no link: with peers about, it is put in place as it is (int32, bsig_segmap_run)
        # while their messages are on the way, and the root packs nothing.  (One rank alone -- --force-dist on a one-GPU
        # box -- keeps the whole wire in its step: that run is how the wire's kernels are measured.)
        own_direct = narrow and root and self.world > 1
        if narrow and not own_direct:
            from bamsignals_amd.device import narrow_pack
            narrow_pack(self.ctx, self.outs[b].data_ptr(), self.pad, self.msg[b].data_ptr(), self.cap)
        send = self.msg[b] if narrow else self.outs[b][:self.pad]
        if ev:
            ev[2].record()
        if self.backend == "nccl":
            # torch.distributed.gather over RCCL: grouped ncclSend / ncclRecv, every peer straight to the root over its
            # own link (the root's own entry is a copy on the device); asynchronous, on RCCL's stream, so that the root's
            # own placement runs under it
            work = dist.gather(send, self.bufs if root else None, dst=0, async_op=True)
            if own_direct:
                self.maps[b][0].run(self.outs[b].data_ptr(), self.final[b].data_ptr())
            work.wait()                                  # (this stream waits for RCCL's; the host does not)
        else:
            # gloo (the code path on a box with fewer GPUs than ranks): through host memory
            if own_direct:
                self.maps[b][0].run(self.outs[b].data_ptr(), self.final[b].data_ptr())
            torch.cuda.synchronize()
            mine = send.cpu()
            got = [torch.empty_like(mine) for _ in range(self.world)] if root else None
            dist.gather(mine, got, dst=0)
            if root:
                for r in range(1 if own_direct else 0, self.world):
                    self.bufs[r].copy_(got[r])
        if ev:
            ev[3].record()
        if root:
            for r in range(1 if own_direct else 0, self.world):
                if narrow:
                    self.maps[b][r].run_narrow(self.bufs[r].data_ptr(), self.pad, self.cap, self.final[b].data_ptr())
                else:
                    self.maps[b][r].run(self.bufs[r].data_ptr(), self.final[b].data_ptr())
        if ev:
            ev[4].record()

    def run_steps(self, k, full=False):
        if self.gathered and not full:
            for q in range(k):
                self.strong_step(q)
            return
        plans, outs = (self.full_plans, self.full_outs) if full else (self.plans, self.outs)
        for q in range(k):
            plans[q % self.nb].run_device(outs[q % self.nb].data_ptr())

    def timed(self, steps, warmup, stream, barrier, full=False):
        """W untimed + K timed steps between barriers; returns (this rank's wall seconds, kernel_ms).  `full`: the
        whole range set on this rank alone, nothing exchanged (N > 1: the informational replica run)."""
        import torch
        with torch.cuda.stream(stream):
            self.run_steps(warmup, full)
            torch.cuda.synchronize()
            barrier()
            torch.cuda.synchronize()
            # HIP events on the launch stream bracket the K launches: (e1 - e0) / K is the average
            # duration of one launch including the ~1.5 us dependent-launch boundary, i.e. an upper
            # bound of the kernel time rocprofv3 reports for the same command (profiles/)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t_start = time.perf_counter()
            e0.record(stream)
            self.run_steps(steps, full)
            e1.record(stream)
            torch.cuda.synchronize()
            # this rank's K steps are done: stop its clock here.  The closing barrier only lines the
            # ranks up again; the whole-job time is the MAX of the per-rank times (all-reduce in
            # main), so the barrier's own latency is not billed to the steps.
            elapsed = time.perf_counter() - t_start
            barrier()
            torch.cuda.synchronize()
            return elapsed, e0.elapsed_time(e1) / steps

    def phases(self, steps, stream, barrier):
        """After the timed region (N > 1): the same steps once more with HIP events between their parts; returns
        this rank's mean (kernel, pack, gather, place) in ms -- rank 0's say where a step's time goes, and the kernel
        part is the launch duration the roofline figure is quoted on (pack: the narrow wire's bsig_narrow_pack, 0 for
        the int32 wire)."""
        import torch
        with torch.cuda.stream(stream):
            torch.cuda.synchronize(); barrier(); torch.cuda.synchronize()
            evs = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(steps)]
            for q in range(steps):
                self.strong_step(q, evs[q])
            torch.cuda.synchronize(); barrier(); torch.cuda.synchronize()
            return np.asarray([[e[i].elapsed_time(e[i + 1]) for i in range(4)] for e in evs]).mean(axis=0)

    def check_parity(self, oracle_c, seed=0, sample=None):
        """What was just timed against the oracle: EVERY range of every batch, cell by cell (the oracle runs all of
        a batch in about half a second at the north star's size).  Returns the oracle's reads handle, a summary, and
        the oracle's result for batch 0 (the CPU baseline times that very computation)."""
        cols = self.cols
        orc = oracle_c.OracleReads(cols["ref_off"], cols["pos"], cols["end"], cols["flag"], cols["mapq"], cols["tlen"])
        refs_seen = set()
        cells = 0
        for b in range(self.nb):
            g = self.full[b]
            want, woff = oracle_c.pileup_core(orc, g, **self.cfg["args"])
            off = self.full_plans[b].offsets
            got = self.final[b][:self.full_plans[b].cells].cpu().numpy()
            if not np.array_equal(off, woff):
                raise SystemExit(f"HIP result layout differs from the oracle's (batch {b}): refusing to report a number")
            if not np.array_equal(got, want):
                bad = np.flatnonzero(got != want)
                i = int(np.searchsorted(off, bad[0], side="right") - 1)
                raise SystemExit(f"HIP result differs from the oracle (batch {b}, range {i} on reference "
                                 f"{int(g['rid'][i])}, {len(bad)} cells): refusing to report a number")
            cells += int(got.size)
            refs_seen.update(np.unique(g["rid"]).tolist())
        return orc, dict(ranges_per_batch=len(self.full[0]["rid"]), batches=self.nb, cells=cells,
                         how="every range of every batch, cell by cell, against oracle/bamsignals_oracle.c"
                             + (f" (the result assembled on rank 0 from the {self.world} ranks' shards)" if self.gathered else ""),
                         references_covered=len(refs_seen), references=len(self.cfg["ref_len"]))

    def roofline(self, kernel_ms, traffic=None, traffic_src=None):
        st = self.stats
        achieved = st["algorithmic_bytes"] / (kernel_ms * 1e-3) / 1e9
        return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
                # what the launch really moves per second (PMC bytes / this run's time), beside what the guide calls
                # achievable on this part (MI355X_MICROARCH.md: 8 TB/s peak, about 6.3 TB/s achievable); informational
                "traffic_GBps": (traffic / (kernel_ms * 1e-3) / 1e9) if traffic else None, "achievable_GBps": 6300.0,
                "kernel": "k_profile", "kernel_ms": kernel_ms,
                "algorithmic_bytes": st["algorithmic_bytes"],
                # SURVEY 8(d)'s layout-agnostic figure 15*V + 16*I + 4*S*C for the same launch
                # (not used for `achieved`: the packed HBM layout moves fewer bytes per visit)
                "algorithmic_bytes_survey_formula": 15 * st["visits"] + 16 * st["n_ranges"] + 4 * st["cells"],
                "visits": st["visits"], "streamed_reads": st["streamed"], "visits_short": st["visits_short"],
                "visits_packed": st["visits_packed"], "bytes_per_visit_packed": st["bytes_per_visit_packed"],
                "bytes_per_visit_short": st["bytes_per_visit_short"],
                "bytes_per_visit_long": st["bytes_per_visit_long"],
                "items": st["n_items"], "cells": st["cells"]}

    def close(self):
        for m in [m for ms in (self.maps or []) for m in ms]:
            m.close()
        for p in self.plans + ([] if self.full_plans is self.plans else self.full_plans):
            p.close()
        self.outs = self.full_outs = self.final = self.bufs = []
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


def strong_block(a, rank, world, stream, ctx, reads, cols, use_dist, dist, cdev, barrier, backend, n_ranges_total, steps):
    """BASELINE config 5 at FIXED total size over `world` GPUs, the whole north-star step timed:
       kernel on this rank's shard of the (rid, loc)-sorted ranges (ref: src/bamsignals.cpp:222-226,246)
       -> gather of the shards to rank 0 (RCCL over xGMI: grouped send/recv, the peers send straight to the root)
       -> reassembly into the caller's range order in rank 0's HBM (bsig_segmap_run; each range owns its
          output, ref: :164,181,186).
    Next to it the same ranges run by rank 0's GPU alone (the 1-GPU time the speed-up is quoted against);
    the assembled result must equal that result cell by cell, and a sample is checked against the oracle."""
    import torch

    from bamsignals_amd import _lib
    from bamsignals_amd.device import Plan, SegmentMap, layout, make_params
    from bamsignals_amd.synth import synth_ranges
    cfg = CONFIGS["C5"]
    width = cfg["width"]
    prm = make_params(_lib.MODE_PROFILE, **cfg["args"])
    rg = synth_ranges(n_ranges_total, width, cfg["ref_len"], seed=a.seed + 555)        # the caller's (unsorted) order
    order = np.lexsort((rg["loc"], rg["rid"]))
    shards = [order[r::world] for r in range(world)]
    off_all = layout(rg["len"], 1, False)
    loffs = [layout(rg["len"][sh], 1, False) for sh in shards]
    sizes = [int(o[-1]) for o in loffs]
    pad = max(max(sizes), 4)
    mine = shards[rank]
    out = {}
    with torch.cuda.stream(stream):
        plan = Plan(ctx, reads, rg["rid"][mine], rg["loc"][mine], rg["len"][mine], rg["strand"][mine], prm)
        assert plan.cells == sizes[rank]
        shard = torch.zeros(pad, dtype=torch.int32, device="cuda")
        # the wire (see Workload.setup_gather): two bits a cell + exceptions, the room for them counted on a first run
        narrow = a.wire == "narrow"
        cap, send = 0, shard
        if narrow:
            from bamsignals_amd.device import narrow_bytes, narrow_count, narrow_pack
            plan.run_device(shard.data_ptr())
            probe = torch.empty(narrow_bytes(pad, 0) // 4, dtype=torch.int32, device="cuda")
            narrow_pack(ctx, shard.data_ptr(), pad, probe.data_ptr(), 0)
            cap = narrow_count(ctx, probe.data_ptr())
            del probe
            if use_dist:
                t = torch.tensor([cap], dtype=torch.int64, device=cdev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                cap = int(t.item())
            send = torch.empty(narrow_bytes(pad, cap) // 4, dtype=torch.int32, device="cuda")
        bufs = maps = final = None
        if rank == 0:
            bufs = [torch.empty(len(send), dtype=torch.int32, device=cdev) for _ in range(world)]
            final = torch.zeros(max(int(off_all[-1]), 4), dtype=torch.int32, device="cuda")
            maps = [SegmentMap(ctx, loffs[r], off_all, shards[r]) for r in range(world)]
        dev_bufs = bufs

        def produce():
            plan.run_device(shard.data_ptr())
            if narrow:
                narrow_pack(ctx, shard.data_ptr(), pad, send.data_ptr(), cap)

        def place():
            for r in range(world):
                if narrow:
                    maps[r].run_narrow(dev_bufs[r].data_ptr(), pad, cap, final.data_ptr())
                else:
                    maps[r].run(dev_bufs[r].data_ptr(), final.data_ptr())

        def step():
            nonlocal dev_bufs
            produce()
            if use_dist:
                if backend == "nccl":
                    dist.gather(send, bufs, dst=0)
                else:                           # gloo (testing the code path on a box with fewer GPUs): via host memory
                    torch.cuda.synchronize()
                    dist.gather(send.cpu(), bufs, dst=0)
                    if rank == 0:
                        dev_bufs = [b.cuda() for b in bufs]
            else:
                dev_bufs = [send]
            if rank == 0:
                place()

        for _ in range(2):
            step()
        torch.cuda.synchronize(); barrier(); torch.cuda.synchronize()
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(steps)] if rank == 0 and backend == "nccl" else None
        t0 = time.perf_counter()
        for q in range(steps):
            if ev:
                ev[q][0].record(stream)
                produce()
                ev[q][1].record(stream)
                if use_dist:
                    dist.gather(send, bufs, dst=0)
                else:
                    dev_bufs = [send]
                ev[q][2].record(stream)
                place()
                ev[q][3].record(stream)
            else:
                step()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        barrier(); torch.cuda.synchronize()
        if use_dist:
            t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        bases = int(rg["len"].astype(np.int64).sum())
        if rank == 0:
            out = dict(workload=f"C5 strong: bamProfile binsize=1, {n_ranges_total} x {width} bp ranges in total over {world} GPU(s), "
                                f"{reads.n_reads} SE reads, {len(cfg['ref_len'])} refs",
                       n_gpus=world, ranges_total=n_ranges_total, ranges_per_gpu=len(mine), reads=int(reads.n_reads), steps=steps,
                       timed_region="kernel on the rank's shard + gather of the shards to rank 0 + reassembly in rank 0's HBM",
                       backend=backend, ms_per_step=elapsed / steps * 1e3, value=bases * steps / elapsed / 1e6, unit="Mbases/s",
                       gather_bytes=int(sum(sizes[1:]) * 4) if not narrow else 4 * len(send) * (world - 1), wire=a.wire,
                       wire_message_bytes=4 * len(send), wire_int32_bytes=4 * pad)
            if narrow and any(m.narrow_overflowed() for m in maps):
                raise SystemExit("strong: a narrow message had more exceptions than its list holds")
            if ev:
                ph = np.asarray([[e[i].elapsed_time(e[i + 1]) for i in range(3)] for e in ev]).mean(axis=0)
                out["phases_ms_rank0"] = dict(kernel_and_pack=float(ph[0]), gather=float(ph[1]), place=float(ph[2]))
                if ph[1] > 0:
                    out["gather_GBps"] = out["gather_bytes"] / (ph[1] * 1e-3) / 1e9
            # the same ranges on rank 0's GPU alone
            plan1 = Plan(ctx, reads, rg["rid"], rg["loc"], rg["len"], rg["strand"], prm)
            ref = torch.zeros(max(plan1.cells, 4), dtype=torch.int32, device="cuda")
            for _ in range(2):
                plan1.run_device(ref.data_ptr())
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(steps):
                plan1.run_device(ref.data_ptr())
            e1.record(stream)
            torch.cuda.synchronize()
            one = e0.elapsed_time(e1) / steps
            st1 = plan1.stats()
            out["one_gpu_ms"] = one
            out["one_gpu_frac_of_hbm_peak"] = st1["algorithmic_bytes"] / (one * 1e-3) / 1e9 / HBM_PEAK_GBPS
            out["speedup_vs_1gpu"] = one / out["ms_per_step"]
            out["efficiency_vs_1gpu"] = one / out["ms_per_step"] / world
            if not torch.equal(final[:plan1.cells], ref[:plan1.cells]):
                raise SystemExit("strong block: the result assembled from the ranks' shards differs from the 1-GPU result")
            out["checked"] = f"assembled result identical to the 1-GPU result ({plan1.cells} cells)"
            out["_final_host"] = ref[:plan1.cells].cpu().numpy()
            plan1.close()
            for m in maps:
                m.close()
        plan.close()
    out_rg = rg if rank == 0 else None
    return out, out_rg


def in_process_block(a, world, ngpu, cols, cfg, rg, want_flat):
    """ONE process, N GPU slots, the file-level call (what an R session with BAMSIGNALS_DEVICES=0..N-1 does):
    the strong block's reads written to local disk as a BAM, cold call (every GPU inflates and parses its
    share, column all-gather over xGMI, one layout per GPU) and warm calls under each gather route.  The session
    is a child process with a time limit: RCCL inside the library has never met N > 1 physical GPUs on the
    builder's boxes, and a communicator that does not come up must cost this block, not the bench's line."""
    import shutil
    import subprocess
    import tempfile

    from bamsignals_amd.bamio import write_columns_as_bam
    from bamsignals_amd.synth import add_cigar
    d = tempfile.mkdtemp(prefix="bsig_bench_inproc_", dir=os.environ.get("TMPDIR", "/tmp"))
    try:
        if "cigar" not in cols:
            add_cigar(cols)
        names = ["ref%d" % (i + 1) for i in range(len(cfg["ref_len"]))]
        bam = os.path.join(d, "c5.bam")
        t0 = time.perf_counter(); write_columns_as_bam(bam, names, cols, level=1); t_write = time.perf_counter() - t0
        cols.pop("cigar"); cols.pop("cigar_off")
        _settle(bam)
        devices = ",".join(str(r % ngpu) for r in range(world))
        out = dict(devices=devices, bam_bytes=os.path.getsize(bam), write_bam_s=t_write, ranges=len(rg["rid"]),
                   note="pileup_core(bampath, GRanges) from ONE process (a child of rank 0, 240-s limit) over all listed GPU "
                        "slots; results in host memory (PCIe-inclusive); every call checked against the strong block's 1-GPU "
                        "result (cell count, sum, position-weighted sum; the warm calls cell by cell against the cold one)")
        bases = int(rg["len"].astype(np.int64).sum())
        want = result_fingerprint(want_flat)
        call = dict(tlen_filter=(), device=-1)
        env = {"BAMSIGNALS_DEVICES": devices, "BAMSIGNALS_DECODE": "all", "BAMSIGNALS_GATHER": "xgmi"}
        # ("auto": no request -- what the library picks by itself; the strong block's ranges come in random order, so a
        # second pair of calls passes them sorted, as a tiling or a sorted peak list would be: "auto_sorted", "blocks")
        kinds = ["cold", "warm", "warm", "warm", "warm", "warm", "warm", "warm", "warm", "warm"]
        gathers = ["xgmi", "xgmi", "xgmi", "direct", "direct", "pcie", "pcie", "blocks", "blocks", "auto"]
        try:
            child, _ = cold_call_in_fresh_process(d, "inproc", bam, names, rg, call, -1, env=env, reps=len(kinds),
                                                  per_rep_env=[{"BAMSIGNALS_GATHER": "" if g == "auto" else g} for g in gathers], want_result=False,
                                                  timeout=240)
        except subprocess.TimeoutExpired:
            out["error"] = "the in-process session did not finish within 240 s and was stopped"
            return out
        if (child["result_cells"], child["result_sum"], child["result_wsum"]) != want:
            raise SystemExit("in-process calls differ from the 1-GPU result")
        for kind, g, c in zip(kinds, gathers, child["calls"]):
            out.setdefault(g, []).append(dict(kind=kind, call_s=c["call_s"], Mbases_s=bases / c["call_s"] / 1e6, stages_s=c["stages_s"],
                                              route=c["route"]))
        return out
    finally:
        shutil.rmtree(d, ignore_errors=True)


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--child-cold":
        return _cold_call_child(sys.argv[2])
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="NS", choices=sorted(CONFIGS))
    ap.add_argument("--reads", type=int, default=0, help="override the number of reads")
    ap.add_argument("--ranges", type=int, default=0, help="override the number of ranges (of the whole job: N ranks share them)")
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
    ap.add_argument("--wire", default="narrow", choices=["narrow", "int32"],
                    help="how a rank's result shard travels to rank 0 at N > 1: 'narrow' = two bits a cell + a list of exceptions "
                         "(bsig_narrow_pack / bsig_segmap_run_narrow: lossless, ~1/15 of the bytes for a per-base profile), "
                         "'int32' = the cells as they are")
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
    ap.add_argument("--strong", action="store_true", help="run the strong-scaling block (config 5 at fixed total size) "
                    "even with one rank; with N > 1 it always runs unless --no-strong")
    ap.add_argument("--no-strong", action="store_true")
    ap.add_argument("--strong-reads", type=int, default=0, help="reads of the strong block's data set (default: config 5's 1e9; "
                    "with --config C5 the block reuses the main workload's reads)")
    ap.add_argument("--strong-ranges", type=int, default=1_000_000, help="TOTAL ranges of the strong block")
    ap.add_argument("--strong-steps", type=int, default=20)
    ap.add_argument("--no-in-process", action="store_true", help="skip the single-process multi-GPU file-level block (N > 1)")
    ap.add_argument("--in-process-slots", type=int, default=0, help="run the in-process block with this many GPU slots even "
                    "with one rank (slots beyond the box's GPUs reuse them: testing)")
    ap.add_argument("--budget-s", type=float, default=420.0, help="optional blocks are skipped once the run is older than this")
    a = ap.parse_args()
    t_program = time.time()

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

    # long host-side waits (one rank generating reads, rank 0 running the in-process block) go through a
    # gloo group: an RCCL barrier would park a spinning kernel on every waiting GPU
    host_group = dist.new_group(backend="gloo") if use_dist and a.backend == "nccl" else None

    def barrier():
        if use_dist:
            dist.barrier()

    def host_barrier():
        if use_dist:
            dist.barrier(group=host_group) if host_group is not None else dist.barrier()

    def all_agree(flag):
        # rank 0 decides (time budget): every rank takes the same branch
        if not use_dist:
            return bool(flag)
        t = torch.tensor([1 if flag else 0], dtype=torch.int32)
        dist.broadcast(t, src=0, group=host_group) if host_group is not None else dist.broadcast(t, src=0)
        return bool(t.item())

    cfg = CONFIGS[a.config]
    stream = torch.cuda.Stream()
    shared = SharedReads(a.config, a.reads or cfg["reads"], cfg["ref_len"], a.seed, cfg["paired"], rank, world, host_barrier)
    gathered = world > 1 or a.force_dist
    w = Workload(a, a.config, rank, world, local, stream, a.reads, a.ranges, a.width, a.batches, cols=shared.cols, t_gen=shared.t_gen,
                 gathered=gathered)
    shared.release(host_barrier)
    if rank != 0:
        w.cols = None                    # (only rank 0 checks against the oracle: the mapped columns can go)
    if gathered:
        w.setup_gather(dist, a.backend, a.wire, host_group)
        with torch.cuda.stream(stream):
            for q in range(w.nb):                # every batch's whole result is assembled at least once (all ranks take part)
                w.strong_step(q)
            torch.cuda.synchronize()
    elapsed, kernel_ms = w.timed(a.steps, a.warmup, stream, barrier)
    nb, plan, rg = w.nb, w.plans[0], w.full[0]

    # whole-job time = max over ranks
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- N > 1, informational: where a step's time goes (events between its parts, a second run of the same steps),
    # and every rank running the WHOLE range set alone with nothing exchanged (N replicas; rounds 1-4's `value`)
    phases = replicas = None
    if gathered:
        ksteps = min(a.steps, 20)
        ph = w.phases(ksteps, stream, barrier)
        allph = [None] * world
        dist.all_gather_object(allph, [float(x) for x in ph], group=host_group)
        kernel_ms = float(ph[0])                    # this rank's shard launch: what the roofline figure is quoted on
        el_r, kms_r = w.timed(ksteps, min(a.warmup, 4), stream, barrier, full=True)
        t = torch.tensor([el_r], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if rank == 0:
            phases = dict(steps=ksteps, rank0_ms=dict(kernel=float(ph[0]), pack=float(ph[1]), gather=float(ph[2]), place=float(ph[3])),
                          kernel_ms_by_rank=[p[0] for p in allph],
                          gather_bytes_into_rank0=w.gather_bytes[0],
                          gather_GBps=(w.gather_bytes[0] / (float(ph[2]) * 1e-3) / 1e9) if ph[2] > 0 else None,
                          note="HIP events between the parts of a step, on a second run of the same steps: the launches on the "
                               "rank's shard, bsig_narrow_pack (the narrow wire only; with peers about the root packs nothing), "
                               "torch.distributed.gather to rank 0 (the root puts its own shard in place under it), "
                               "bsig_segmap_run[_narrow] over the peers' shards")
            replicas = dict(value=sum(w.step_bases[q % nb] for q in range(ksteps)) * world / float(t.item()) / 1e6, unit="Mbases/s",
                            ms_per_step=float(t.item()) / ksteps * 1e3, steps=ksteps, kernel_ms_rank0=kms_r,
                            what=f"every one of the {world} ranks runs the WHOLE range set by itself, nothing exchanged: {world} "
                                 f"independent replicas (linear by construction; rounds 1-4 reported this as a weak-scaling `value`)")

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
    got = w.outs[0][:plan.cells].cpu().numpy() if not gathered else None
    parity = cpu = e2e = None
    bases = w.step_bases[0]
    # (the oracle is the checker and the CPU baseline: loaded on rank 0 only, after the timed region)
    if rank == 0:
        from oracle import oracle_c
        orc, parity = w.check_parity(oracle_c)
        if gathered and a.wire == "narrow" and any(m.narrow_overflowed() for ms in w.maps for m in ms):
            raise SystemExit("a narrow message had more exceptions than its list holds: refusing to report a number")
        log(f"parity ok ({parity}); step {kernel_ms * 1e3:.1f} us")
        if not a.no_cpu_baseline and world == 1:          # (the contract: rank 0 at N = 1 only)
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

    res = None
    if rank == 0:
        traffic, traffic_src = committed_traffic(a, a.config, cfg)
        total_bases = sum(w.step_bases[s % nb] for s in range(a.steps))         # the whole job's, whatever the number of ranks
        res = {
            "metric": "Mbases profiled/sec (bamProfile binsize=1)", "value": total_bases / elapsed / 1e6, "unit": "Mbases/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": a.config + ": " + cfg["desc"], "reads": w.n_reads,
                       "ranges_total": len(rg["rid"]), "ranges_per_gpu": len(w.batches[0]["rid"]), "range_width": w.width, "batches": nb,
                       "parallelism": (f"fixed total size: sorted ranges round-robin over {world} GPUs, reads replicated; a step = "
                                       f"kernels on the shard + {'the narrow pack (two bits a cell + exceptions) + ' if a.wire == 'narrow' else ''}"
                                       f"gather to rank 0 ({a.backend}: "
                                       + ("RCCL grouped send/recv" if a.backend == "nccl" else "gloo through host memory, testing only")
                                       + ") + reassembly in rank 0's HBM, all inside the timed region") if gathered
                                      else "one GPU: a step = the launch, nothing to gather",
                       "launch": "one bsig_plan_run per step from the host loop; the tiles' index windows are looked up (k_resolve_tiles) in "
                                 "a plan's first run (its second, below 32,768 tiles) and kept with the plan, which is immutable like the "
                                 "layout of the reads: the timed steps read them back" if os.environ.get("BAMSIGNALS_CACHE_WINDOWS") != "0" else
                                 "one bsig_plan_run per step from the host loop: k_resolve_tiles + the pileup launch in every step (BAMSIGNALS_CACHE_WINDOWS=0)",
                       "threads": w.params.threads or 64,
                       "tile_cells": w.params.tile_cells or "auto (widest range, at most 2048)"},
            "roofline": w.roofline(kernel_ms, traffic, traffic_src),
            "cpu_baseline": cpu,
            "end_to_end": None,
            "parity_checked": parity,
            "pipelined_two_streams": pipelined,
            "wire": (dict(kind=a.wire, exceptions_room=w.cap, message_bytes=4 * getattr(w, "msg_words", 0) if a.wire == "narrow" else 4 * w.pad,
                          int32_bytes=4 * w.pad,
                          what="a rank's shard travels as two bits a cell (0, 1, 2 or 'see the list') + a list of (cell, value) "
                               "exceptions sized by a first run: lossless; rank 0 widens it while placing it (bsig_segmap_run_narrow)"
                               if a.wire == "narrow" else "a rank's shard travels as its int32 cells") if gathered else None),
            "step_phases": phases,
            "no_collective": replicas,
            "strong": None,
            "in_process": None,
            "setup_s": {"generate": w.t_gen, "upload_and_layout": w.t_upload, "plan": w.t_plan},
            "reads_in_hbm": w.reads.info(),
        }

    # ---- N > 1: config 5 at fixed total size with the collective in the timed region; the in-process route ------
    want_strong = (world > 1 or a.strong) and not a.no_strong
    want_inproc = (world > 1 and not a.no_in_process) or a.in_process_slots > 0
    if want_strong or want_inproc:
        go = all_agree(time.time() - t_program < a.budget_s)
        if not go:
            if rank == 0:
                res["strong"] = res["in_process"] = {"skipped": f"time budget ({a.budget_s:.0f} s) used up before the block"}
        else:
            c5 = CONFIGS["C5"]
            if a.config == "C5":
                reads5, cols5, ctx5, own5 = w.reads, w.cols, w.ctx, False
            else:
                # the main workload leaves HBM and host memory first
                w.close(); w.cols = None
                torch.cuda.empty_cache()
                from bamsignals_amd.device import Context, Reads
                sh5 = SharedReads("C5", a.strong_reads or c5["reads"], c5["ref_len"], a.seed + 5, False, rank, world, host_barrier)
                cols5 = sh5.cols
                with torch.cuda.stream(stream):
                    ctx5 = Context(local, stream=stream.cuda_stream)
                    reads5 = Reads(ctx5, cols5["ref_len"], cols5["ref_off"], cols5["pos"], cols5["flag"], cols5["mapq"], cols5["tlen"],
                                   end=cols5["end"])
                sh5.release(host_barrier)
                own5 = True
            strong = rg5 = None
            if want_strong or want_inproc:
                strong, rg5 = strong_block(a, rank, world, stream, ctx5, reads5, cols5, use_dist, dist, cdev, barrier, a.backend,
                                           a.strong_ranges, a.strong_steps)
            want5 = strong.pop("_final_host", None) if rank == 0 else None
            if rank == 0:
                # a seeded sample of the 1-GPU result against the oracle (all references; the checker imported above)
                _oc = oracle_c
                orc5 = _oc.OracleReads(cols5["ref_off"], cols5["pos"], cols5["end"], cols5["flag"], cols5["mapq"], cols5["tlen"])
                pick = np.sort(np.random.default_rng(a.seed + 9).choice(len(rg5["rid"]), size=min(2000, len(rg5["rid"])), replace=False))
                sub = {k: v[pick] for k, v in rg5.items()}
                wv, wo = _oc.pileup_core(orc5, sub, **c5["args"])
                from bamsignals_amd.device import layout as _layout
                off5 = _layout(rg5["len"], 1, False)
                for j, i in enumerate(pick):
                    if not np.array_equal(want5[off5[i]:off5[i + 1]], wv[wo[j]:wo[j + 1]]):
                        raise SystemExit(f"strong block: range {i} differs from the oracle")
                del orc5
                strong["checked"] += f"; {len(pick)} ranges on {len(np.unique(sub['rid']))} references identical to the oracle"
                res["strong"] = strong if want_strong else None
                log(f"strong: {strong['ms_per_step']:.3f} ms per step on {world} GPU(s), 1 GPU {strong['one_gpu_ms']:.3f} ms")
            # everything this process holds on the GPUs goes before ONE process drives them all
            if own5:
                reads5.close(); ctx5.close()
            elif a.config == "C5":
                w.close()
            torch.cuda.empty_cache()
            host_barrier()
            if want_inproc and all_agree(time.time() - t_program < a.budget_s + 60):
                if rank == 0:
                    try:
                        slots = a.in_process_slots or world
                        res["in_process"] = in_process_block(a, slots, ngpu, cols5, c5, rg5, want5)
                        log("in_process: " + "; ".join(f"{g} " + "/".join(f"{r['call_s']:.3f}" for r in res["in_process"][g])
                                                        for g in ("xgmi", "direct", "pcie", "blocks", "auto") if g in res["in_process"]))
                    except SystemExit:
                        raise
                    except Exception as exc:
                        res["in_process"] = {"error": f"{type(exc).__name__}: {exc}"}
                host_barrier()
            elif want_inproc and rank == 0:
                res["in_process"] = {"skipped": "time budget used up before the block"}
            del cols5

    # ---- N = 1 extras: the same workload from a BAM file, and config 2's small launch ---------------
    if rank == 0 and world == 1 and not use_dist and not (want_strong or want_inproc):
        cols = w.cols
        want_flat = got[:plan.cells]
        # the cold calls run in child processes WHILE this process still holds its workload: round 3 freed 8 GB of
        # HBM right before them, and the driver's bulk return of freed memory is what stalls a neighbour's hipMalloc
        if cpu is not None and not a.no_e2e:
            try:
                res["end_to_end"], cpu["with_bam_decode"] = end_to_end(cfg, cols, rg, want_flat, local, oracle_c)
            except Exception as exc:      # e.g. no room for the BAM on local disk: the metric does not depend on it
                res["end_to_end"] = {"error": f"{type(exc).__name__}: {exc}"}
        w.close()
        w.cols = None
        torch.cuda.empty_cache()
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
            _, par2 = w2.check_parity(oracle_c)
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
