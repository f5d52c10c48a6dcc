#!/usr/bin/env python3
"""Turn the rocprofv3 output of scripts/profile_round.sh (merged into gpurun_out/) into the
committed files under profiles/:  <tag>_kernel_stats.csv, <tag>_pmc.json, <tag>_summary.md."""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
os.makedirs(P, exist_ok=True)


def one(pattern):
    f = glob.glob(os.path.join(G, pattern))
    if not f:
        raise SystemExit("missing " + pattern)
    return max(f, key=os.path.getmtime)      # gpurun merges runs into the same directory: newest wins


stats = one(f"{tag}_trace/*/*_kernel_stats.csv")
shutil.copyfile(stats, os.path.join(P, f"{tag}_kernel_stats.csv"))
rows = list(csv.DictReader(open(stats)))


def pmc(kind, counter):
    f = one(f"{tag}_pmc_{kind}/*/*_counter_collection.csv")
    vals = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            k = "k_profile" if "k_profile" in r["Kernel_Name"] else "k_resolve" if "k_resolve" in r["Kernel_Name"] else None
            if k:
                vals.setdefault(k, []).append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in vals.items()}, {k: len(v) for k, v in vals.items()}


fetch, nf = pmc("fetch", "FETCH_SIZE")
write, nw = pmc("write", "WRITE_SIZE")
bench = json.loads(open(os.path.join(G, f"{tag}_bench.json")).read().strip().splitlines()[-1])
# rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB-like units of 1024 B; on gfx950 FETCH_SIZE counts
# 128-B read requests as 64 B (MI355X_MICROARCH.md, HBM section) -> doubled.  WRITE_SIZE is exact.
out = {"tag": tag, "workload": bench["config"]["workload"], "per_launch": {}}
for k in ("k_profile", "k_resolve"):
    rd = 2.0 * fetch.get(k, 0.0) * 1024
    wr = write.get(k, 0.0) * 1024
    out["per_launch"][k] = {"FETCH_SIZE_raw_KB": fetch.get(k), "WRITE_SIZE_raw_KB": write.get(k),
                            "hbm_read_bytes": rd, "hbm_write_bytes": wr, "hbm_bytes": rd + wr,
                            "launches_sampled": [nf.get(k), nw.get(k)]}
# k_resolve belongs to a step only when it is launched once per step (resolve=1); by default it
# runs once per plan (the heavy-tile probe of bsig_plan_create), outside the steps
per_step = ["k_profile"] + (["k_resolve"] if (nf.get("k_resolve") or 0) >= (nf.get("k_profile") or 1) else [])
out["step_kernels"] = per_step
out["step_hbm_bytes"] = sum(out["per_launch"][k]["hbm_bytes"] for k in per_step)
out["algorithmic_bytes"] = bench["roofline"]["algorithmic_bytes"]
json.dump(out, open(os.path.join(P, f"{tag}_pmc.json"), "w"), indent=1)
shutil.copyfile(os.path.join(G, f"{tag}_bench.json"), os.path.join(P, f"{tag}_bench.json"))

with open(os.path.join(P, f"{tag}_summary.md"), "w") as f:
    f.write(f"# {tag}: rocprofv3 summary for `python bench.py` ({bench['config']['workload']})\n\n")
    f.write("Command: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 64 --warmup 16 --no-cpu-baseline`\n\n")
    f.write("| kernel | calls | avg ns | min ns | max ns |\n|---|---|---|---|---|\n")
    for r in rows:
        if any(s in r["Name"] for s in ("k_profile", "k_resolve", "k_scatter", "k_span_hist", "k_build_idx", "k_visits")):
            f.write(f"| `{r['Name'][:70]}` | {r['Calls']} | {float(r['AverageNs']):.0f} | {r['MinNs']} | {r['MaxNs']} |\n")
    f.write("\nbench.py (un-profiled run, HIP events around the K-launch train / K): "
            f"kernel_ms = {bench['roofline']['kernel_ms']:.5f}, ms_per_step = {bench['ms_per_step']:.5f}, "
            f"value = {bench['value']:.0f} {bench['unit']}, roofline.frac = {bench['roofline']['frac']:.3f}\n\n")
    f.write("PMC passes (separate runs; FETCH_SIZE doubled as the MI355X guide prescribes for gfx950):\n\n")
    f.write("| kernel | HBM read B / launch | HBM write B / launch |\n|---|---|---|\n")
    for k, v in out["per_launch"].items():
        if not v["hbm_bytes"] or k not in per_step:
            continue        # kernel not launched per step in this configuration
        f.write(f"| {k} | {v['hbm_read_bytes']:.3e} | {v['hbm_write_bytes']:.3e} |\n")
    f.write(f"\nstep HBM traffic = {out['step_hbm_bytes']:.4e} B vs algorithmic {out['algorithmic_bytes']:.4e} B "
            f"(ratio {out['step_hbm_bytes'] / out['algorithmic_bytes']:.3f})\n")
# optional: the decode kernels' trace (scripts/decode_device_time.py under rocprofv3)
dec = glob.glob(os.path.join(G, f"{tag}_decode_trace/*/*_kernel_stats.csv"))
if dec:
    dec = max(dec, key=os.path.getmtime)
    shutil.copyfile(dec, os.path.join(P, f"{tag}_decode_kernel_stats.csv"))
    with open(os.path.join(P, f"{tag}_summary.md"), "a") as f:
        f.write("\n## Device-side decode (`rocprofv3 --kernel-trace --stats -- python3 scripts/decode_device_time.py`: "
                "5e7-read BAM, 3 whole-file decodes + index-driven decodes of 100 / 1,000 / 10,000 regions)\n\n")
        f.write("| kernel | calls | avg ns | min ns | max ns |\n|---|---|---|---|---|\n")
        for r in csv.DictReader(open(dec)):
            if any(s in r["Name"] for s in ("k_inflate", "k_crc32", "k_bam_walk", "k_bam_extract", "k_scatter", "k_span_hist", "k_build_idx")):
                f.write(f"| `{r['Name'][:70]}` | {r['Calls']} | {float(r['AverageNs']):.0f} | {r['MinNs']} | {r['MaxNs']} |\n")
print(open(os.path.join(P, f"{tag}_summary.md")).read())
