#!/usr/bin/env python3
"""Diagnostic: tests/test_methods_like_reference_gpu.py's four loops for many region seeds in one process."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_methods_like_reference_gpu as T  # noqa: E402

z = np.load(os.path.join(ROOT, "tests", "golden", "fixture_reads.npz"))
fx = {k: z[k] for k in z.files}
reads = T.reads.__wrapped__(fx)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for seed in range(1, n + 1):
    os.environ["BSIG_REGION_SEED"] = str(seed)
    regions = T.regions.__wrapped__(fx)
    T.test_bamCount_function(reads, regions)
    T.test_bamProfile_function(reads, regions)
    T.test_bamCoverage_function(reads, regions)
    T.test_filtering_on_SAMFLAGS(reads, regions)
print("all", n, "seeds agree")
