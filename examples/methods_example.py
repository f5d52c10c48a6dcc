#!/usr/bin/env python3
"""The reference's usage example (its inst/examples/methods_example.R) with the Python host:
counts, profiles and coverage of the fixture BAM over the fixture genes.  Needs an MI355X."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from bamsignals_amd import GRanges, bamCount, bamCoverage, bamProfile  # noqa: E402

bampath = os.path.join(ROOT, "tests", "golden", "randomBam.bam")
reg = json.load(open(os.path.join(ROOT, "tests", "golden", "regions.json")))
genes = GRanges(reg["chrom"][:20], reg["start"][:20], width=reg["width"][:20], strand=reg["strand"][:20])

# reads whose 5' end falls in each gene; strand-specific: row 0 sense, row 1 antisense
print(bamCount(bampath, genes, verbose=False))
print(bamCount(bampath, genes, ss=True, verbose=False))

# promoters: 100 bp around each gene's start, on the gene's strand
proms = GRanges(genes.seqnames, [max(1, s - 50) for s in genes.start], width=100, strand=genes.strand)
prof = bamProfile(bampath, proms, binsize=1, verbose=False)
print(prof)                               # CountSignals object with 20 signals
print(prof.alignSignals().shape)          # (100, 20): all promoters have the same width
binned = bamProfile(bampath, proms, binsize=20, ss=True, verbose=False)
print(binned[0])                          # 2 x 5 matrix

# paired-end: count fragments at their midpoint, fragments of 50-300 bp only
print(bamCount(bampath, genes, paired_end="midpoint", tlenFilter=(50, 300), verbose=False))

# per-base coverage, fragments extended to the whole template
cov = bamCoverage(bampath, genes, paired_end="extend", verbose=False)
print(cov[0][:20])
