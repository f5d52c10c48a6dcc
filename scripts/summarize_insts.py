#!/usr/bin/env python3
"""The instruction counters of scripts/pmc_insts.sh as a table: per launch and per wave, and the time the vector
instructions alone occupy the chip's 1,024 SIMDs (a wave64 vector instruction issues over 4 cycles; 2.4 GHz) beside the
launch's duration from the same run's kernel trace.  Usage: scripts/summarize_insts.py r04 > profiles/r04_pmc_insts.md"""
import collections
import csv
import glob
import os
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
print("| case | kernel | launches | waves | VALU / wave | SALU / wave | LDS / wave | VMEM_RD / wave | VALU x 4 cycles / 1,024 SIMDs @ 2.4 GHz | launch (same run) | share |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
for case in ("ns", "C3", "C4", "count", "bins"):
    # (gpurun merges every run into the same directory: the newest files are the ones meant)
    f = sorted(glob.glob(os.path.join(G, f"{tag}_insts_{case}", "*", "*counter_collection.csv")), key=os.path.getmtime)
    t = sorted(glob.glob(os.path.join(G, f"{tag}_insts_{case}", "*", "*kernel_trace.csv")), key=os.path.getmtime)
    if not f:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    for row in csv.DictReader(open(f[-1])):
        k = row["Kernel_Name"]
        if not any(x in k for x in ("k_profile", "k_coverage", "k_count_multi")):
            continue
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        if row["Counter_Name"] == "SQ_WAVES":
            n[k] += 1
    dur = collections.defaultdict(list)
    if t:
        for row in csv.DictReader(open(t[-1])):
            dur[row["Kernel_Name"]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    for k, c in agg.items():
        per = {a: b / n[k] for a, b in c.items()}
        w = per["SQ_WAVES"]
        valu_us = per["SQ_INSTS_VALU"] * 4 / 1024 / 2400
        d = sorted(dur.get(k, [0]))
        med = d[len(d) // 2]
        m = re.search(r"k_\w+(<[^>]*>)?", k)
        name = m.group(0) if m else k[:40]
        print(f"| {case} | `{name}` | {n[k]} | {w:.0f} | {per['SQ_INSTS_VALU'] / w:.0f} | {per['SQ_INSTS_SALU'] / w:.0f} | "
              f"{per['SQ_INSTS_LDS'] / w:.0f} | {per['SQ_INSTS_VMEM_RD'] / w:.1f} | {valu_us:.1f} us | {med:.1f} us | {valu_us / med if med else 0:.2f} |")
