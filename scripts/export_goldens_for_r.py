#!/usr/bin/env python3
"""Write the committed golden vectors (tests/golden/*.npz, regions.json) as plain text that the R
acceptance tests (bamsignals_amd/r_package/tests/testthat/test_golden.R) can read:
    <out>/regions.tsv        chrom start width strand
    <out>/expected_grid.txt  one line per grid point:  key<TAB>v1,v2,...
    <out>/randomBam.bam(.bai)
Usage: python scripts/export_goldens_for_r.py /tmp/bsig_golden ; then
       BAMSIGNALS_GOLDEN_DIR=/tmp/bsig_golden Rscript -e 'testthat::test_dir("bamsignals_amd/r_package/tests/testthat")'
"""
import json
import os
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
out = sys.argv[1] if len(sys.argv) > 1 else "/tmp/bsig_golden"
os.makedirs(out, exist_ok=True)
reg = json.load(open(os.path.join(G, "regions.json")))
with open(os.path.join(out, "regions.tsv"), "w") as f:
    f.write("chrom\tstart\twidth\tstrand\n")
    for c, s, w, t in zip(reg["chrom"], reg["start"], reg["width"], reg["strand"]):
        f.write(f"{c}\t{s}\t{w}\t{t}\n")
z = np.load(os.path.join(G, "expected_grid.npz"))
with open(os.path.join(out, "expected_grid.txt"), "w") as f:
    for k in z.files:
        f.write(k + "\t" + ",".join(str(int(v)) for v in z[k]) + "\n")
for b in ("randomBam.bam", "randomBam.bam.bai"):
    shutil.copyfile(os.path.join(G, b), os.path.join(out, b))
print("wrote", out)
