#!/usr/bin/env python3
"""Diagnostic: the cold file-level call of the north star's BAM in N fresh sessions (bench.py's
cold_call_in_fresh_process), each with the decode's stage marks; prints every session's time and, for the ones
well above the median, where the time went -- what does a session's first call catch now and then?"""
import os
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["BSIG_DIAG_DECODE"] = "1"
import numpy as np  # noqa: E402

import bench  # noqa: E402
from bamsignals_amd.bamio import write_columns_as_bam  # noqa: E402
from bamsignals_amd.synth import synth_ranges, synth_reads  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000_000
sessions = int(sys.argv[2]) if len(sys.argv) > 2 else 12
ref = [250_000_000] * 10
cols = synth_reads(n, ref, seed=9)
d = tempfile.mkdtemp(prefix="bsig_sessions_", dir="/tmp")
bam = os.path.join(d, "ns.bam")
names = ["c%d" % i for i in range(10)]
write_columns_as_bam(bam, names, cols, level=1)
del cols
bench._settle(bam)
rg = synth_ranges(100_000, 2000, ref, seed=10)
call = dict(tlen_filter=(), device=0)
runs = []
_real_stderr = sys.stderr
# (A/B: `VARIANTS="BAMSIGNALS_STAGING_AHEAD=0;BAMSIGNALS_STAGING_AHEAD=1"` alternates the sessions between environments)
variants = [dict(kv.split("=", 1) for kv in v.split(",") if kv) for v in os.environ.get("VARIANTS", "").split(";")] if os.environ.get("VARIANTS") else [{}]
for k in range(sessions):
    sys.stderr = open(os.devnull, "w")            # (cold_call_in_fresh_process echoes the child's marks)
    try:
        child, _ = bench.cold_call_in_fresh_process(d, "s", bam, names, rg, call, 0, env=variants[k % len(variants)], reps=1, want_result=False)
    finally:
        sys.stderr.close()
        sys.stderr = _real_stderr
    c = child["calls"][0]
    runs.append((c["call_s"], c["stages_s"], child.get("stderr", "")))
    st = c["stages_s"]
    print(variants[k % len(variants)] or "", "session %d: call %.3f s = decode %.3f + layout %.3f + plan %.3f + kernels %.3f + download %.3f; "
          "in the driver's allocator %.3f s (%d calls; decode+layout %.3f, run %.3f), reserved %.2f GB (waited %.3f s)" % (
              k, c["call_s"], st["decode"], st["upload_and_layout"], st.get("plan", 0), st.get("kernels", 0), st.get("download", 0),
              st.get("alloc_total", 0), st.get("alloc_calls", 0), st.get("alloc_in_decode_and_layout", 0), st.get("alloc_in_plan_run_download", 0),
              st.get("reserved_bytes", 0) / 2**30, st.get("reservation_wait", 0)), flush=True)
med = float(np.median([r[0] for r in runs]))
print("median %.3f s, min %.3f, max %.3f" % (med, min(r[0] for r in runs), max(r[0] for r in runs)))
if len(variants) > 1:
    for v in range(len(variants)):
        print(variants[v], "median %.3f s" % float(np.median([r[0] for r in runs[v::len(variants)]])))
for k, (t, st, err) in enumerate(runs):
    if t > 1.25 * med or (os.environ.get("MARKS") and k < int(os.environ["MARKS"])):      # (MARKS=n: also the first n sessions')
        print("\n== session %d (%.3f s): stage marks" % (k, t))
        print("\n".join(l for l in err.splitlines() if "[decode]" in l or l.startswith("pass ")))
shutil.rmtree(d, ignore_errors=True)
