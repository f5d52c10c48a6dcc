#!/usr/bin/env python3
"""Turn the rocprofv3 output of scripts/profile_round.sh (merged into gpurun_out/) into the
committed files under profiles/:  <tag>_<case>_kernel_stats.csv, <tag>_pmc_<case>.json, <tag>_summary.md."""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
os.makedirs(P, exist_ok=True)
# case -> (substring of the dominant kernel's name, the command that was profiled)
CASES = {
    "ns": ("k_profile<64, false", "python3 bench.py --steps 32 --warmup 8 --no-cpu-baseline --no-e2e --no-also"),
    "c2": ("k_profile<64, false", "python3 bench.py --config C2 --steps 64 --warmup 16 --no-cpu-baseline --no-e2e"),
    "c5": ("k_profile<64, false", "python3 bench.py --config C5 --steps 32 --warmup 8 --no-cpu-baseline --no-e2e --no-also"),
    "c3": ("k_coverage<64", "python3 scripts/profile_case.py C3"),
    "c4": ("k_profile<64, true", "python3 scripts/profile_case.py C4"),
    "count": ("k_count", "python3 scripts/profile_case.py count"),
    "bins": ("k_profile_small", "python3 scripts/profile_case.py bins"),
    "t500": ("k_profile_multi", "python3 scripts/profile_case.py t500"),
    "t1000": ("k_profile<64, false", "python3 scripts/profile_case.py t1000"),
}


def newest(pattern):
    f = glob.glob(os.path.join(G, pattern))
    return max(f, key=os.path.getmtime) if f else None      # gpurun merges runs into the same directory: newest wins


def pmc(case, kind, counter, kernel):
    f = newest(f"{tag}_{case}_pmc_{kind}/*/*_counter_collection.csv")
    if not f:
        return None, 0
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter and kernel in r["Kernel_Name"]]
    return (sum(vals) / len(vals) if vals else None), len(vals)


lines = [f"# {tag}: rocprofv3 summaries (MI355X, one GPU)\n",
         "Per case: the un-profiled run (HIP events around the K-launch train / K), `rocprofv3 --kernel-trace --stats` of the same "
         "command, and two PMC passes (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, separate runs; FETCH_SIZE doubled as "
         "MI355X_MICROARCH.md prescribes for gfx950: 128-B requests are tallied at 64 B; both in units of 1,024 B).  "
         "**The read side is an UPPER bound**: the guide establishes the factor of two for wide coalesced reads (16 B per lane) only, and "
         "these launches also issue scalar loads and narrow index loads whose requests may be tallied at their true size; ALL of "
         "FETCH_SIZE is doubled here, so traffic / B can only be overstated (reads are about a quarter of a launch's traffic).\n"]
for case, (kernel, cmd) in CASES.items():
    stats = newest(f"{tag}_{case}_trace/*/*_kernel_stats.csv")
    plain = os.path.join(G, f"{tag}_{case}_plain.json")
    if not stats or not os.path.exists(plain):
        continue
    shutil.copyfile(stats, os.path.join(P, f"{tag}_{case}_kernel_stats.csv"))
    bench = json.loads(open(plain).read().strip().splitlines()[-1])
    if "roofline" in bench:
        workload, kernel_ms, alg = bench["config"]["workload"], bench["roofline"]["kernel_ms"], bench["roofline"]["algorithmic_bytes"]
        shutil.copyfile(plain, os.path.join(P, f"{tag}_{case}_bench.json"))
    else:
        workload, kernel_ms, alg = bench["workload"], bench["kernel_ms"], bench["algorithmic_bytes"]
    row = [r for r in csv.DictReader(open(stats)) if kernel in r["Name"]]
    # launches of more than 32,768 tiles look their windows up in a launch of their own in front of the pileup kernel
    # (k_resolve_tiles): a step is then both kernels, and so are its bytes
    res_row = [r for r in csv.DictReader(open(stats)) if "k_resolve_tiles" in r["Name"]]
    fetch, nf = pmc(case, "fetch", "FETCH_SIZE", kernel)
    write, nw = pmc(case, "write", "WRITE_SIZE", kernel)
    rfetch, _ = pmc(case, "fetch", "FETCH_SIZE", "k_resolve_tiles")
    rwrite, _ = pmc(case, "write", "WRITE_SIZE", "k_resolve_tiles")
    out = {"tag": tag, "case": case, "workload": workload, "command": cmd, "kernel": kernel, "algorithmic_bytes": alg,
           "kernel_ms_hip_events": kernel_ms}
    lines.append(f"\n## {case}: {workload}\n\nCommand: `{cmd}`\n")
    if row:
        r = row[0]
        out["rocprof_avg_ns"] = float(r["AverageNs"])
        out["rocprof_calls"] = int(r["Calls"])
        lines.append("| kernel | calls | avg ns | min ns | max ns |\n|---|---|---|---|---|")
        lines.append(f"| `{r['Name'][:60]}` | {r['Calls']} | {float(r['AverageNs']):.0f} | {r['MinNs']} | {r['MaxNs']} |")
        step_ns = float(r["AverageNs"])
        if res_row and int(res_row[0]["Calls"]) >= int(r["Calls"]) // 2:
            q = res_row[0]
            out["resolve_launch_avg_ns"] = float(q["AverageNs"])
            step_ns += float(q["AverageNs"])
            lines.append(f"| `{q['Name'][:60]}` (the windows of every tile, in front of each pileup launch) | {q['Calls']} | {float(q['AverageNs']):.0f} | {q['MinNs']} | {q['MaxNs']} |")
        out["rocprof_step_ns"] = step_ns
        lines.append(f"\nun-profiled: {kernel_ms * 1e3:.1f} us per step (HIP events around the launch train, boundaries between launches "
                     f"included) -> {alg / kernel_ms / 1e6:.0f} GB/s of algorithmic bytes = "
                     f"{alg / kernel_ms / 1e6 / 8000:.3f} of 8 TB/s; by the rocprofv3 averages ({step_ns:.0f} ns of kernels per step): "
                     f"{alg / step_ns:.0f} GB/s = {alg / step_ns / 8000:.3f}")
    if fetch is not None and write is not None:
        rd, wr = 2.0 * (fetch + (rfetch or 0.0)) * 1024, (write + (rwrite or 0.0)) * 1024
        out.update(FETCH_SIZE_raw_KB=fetch, WRITE_SIZE_raw_KB=write, hbm_read_bytes=rd, hbm_write_bytes=wr,
                   step_hbm_bytes=rd + wr, launches_sampled=[nf, nw])
        lines.append(f"\nHBM traffic per launch (PMC): read {rd:.4e} B + written {wr:.4e} B = {rd + wr:.4e} B vs algorithmic {alg:.4e} B "
                     f"(ratio {(rd + wr) / alg:.3f})")
    json.dump(out, open(os.path.join(P, f"{tag}_pmc_{case}.json"), "w"), indent=1)

dec = newest(f"{tag}_decode_trace/*/*_kernel_stats.csv")
if dec:
    shutil.copyfile(dec, os.path.join(P, f"{tag}_decode_kernel_stats.csv"))
    lines.append("\n## Device-side decode (`rocprofv3 --kernel-trace --stats -- python3 scripts/decode_device_time.py`: "
                 "5e7-read BAM, 3 whole-file decodes + index-driven decodes of 100 / 1,000 / 10,000 regions)\n")
    lines.append("| kernel | calls | avg ns | min ns | max ns |\n|---|---|---|---|---|")
    for r in csv.DictReader(open(dec)):
        if any(s in r["Name"] for s in ("k_inflate", "k_crc32", "k_bam_walk", "k_bam_extract", "k_scatter", "k_span_hist", "k_build_idx")):
            lines.append(f"| `{r['Name'][:70]}` | {r['Calls']} | {float(r['AverageNs']):.0f} | {r['MinNs']} | {r['MaxNs']} |")
real = newest(f"{tag}_decodereal_trace/*/*_kernel_stats.csv")
plain = os.path.join(G, f"{tag}_decodereal_plain.txt")
if real:
    shutil.copyfile(real, os.path.join(P, f"{tag}_decode_realshaped_kernel_stats.csv"))
    lines.append("\n## Device-side decode of a REAL-SHAPED BAM (`rocprofv3 --kernel-trace --stats -- python3 "
                 "scripts/decode_realshaped_device_time.py`: 2e7 reads with names, bases, qualities; 2.1 GB file, 4.08 GB of "
                 "stream in 62,000 blocks; four decodes with the GPU inflating)\n")
    lines.append("| kernel | calls | avg ns | min ns | max ns |\n|---|---|---|---|---|")
    infl = 0.0
    for r in csv.DictReader(open(real)):
        if any(s in r["Name"] for s in ("k_inflate", "k_crc32", "k_bam_walk", "k_bam_extract")):
            lines.append(f"| `{r['Name'][:70]}` | {r['Calls']} | {float(r['AverageNs']):.0f} | {r['MinNs']} | {r['MaxNs']} |")
        if "k_inflate" in r["Name"]:
            infl = float(r["TotalDurationNs"]) / 4.0 if "TotalDurationNs" in r else float(r["AverageNs"]) * float(r["Calls"]) / 4.0
    if infl:
        lines.append(f"\n`k_inflate` per decode (all its launches): {infl / 1e6:.1f} ms for 4.08 GB of output = "
                     f"**{4.08e9 / infl:.1f} GB/s of output**")
    if os.path.exists(plain):
        shutil.copyfile(plain, os.path.join(P, f"{tag}_decode_realshaped_plain.txt"))
        lines.append("\nun-profiled stage times (last of four decodes): `" + open(plain).read().strip().splitlines()[-1] + "`")
ns = newest(f"{tag}_decodens_trace/*/*_kernel_stats.csv")
if ns:
    shutil.copyfile(ns, os.path.join(P, f"{tag}_decode_ns_kernel_stats.csv"))
    lines.append("\n## Device-side decode of the NORTH STAR's BAM (`rocprofv3 --kernel-trace --stats -- python3 "
                 "scripts/decode_ns_time.py 500000000 3`: 5e8 bare reads, 3.0 GB file, 21 GB of stream in 327,000 blocks; "
                 "three decodes, each streamed: the file into HBM once, six or seven shares of up to one round of inflate lanes)\n")
    lines.append("| kernel | calls | total ms per decode | avg ns | min ns | max ns |\n|---|---|---|---|---|---|")
    for r in csv.DictReader(open(ns)):
        if any(s in r["Name"] for s in ("k_inflate", "k_crc32", "k_bam_walk", "k_bam_extract", "k_scatter", "k_span_hist", "k_build_idx", "k_check_idx")):
            lines.append(f"| `{r['Name'][:70]}` | {r['Calls']} | {float(r['AverageNs']) * float(r['Calls']) / 3e6:.1f} | {float(r['AverageNs']):.0f} | {r['MinNs']} | {r['MaxNs']} |")
open(os.path.join(P, f"{tag}_summary.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
