#!/usr/bin/env python3
"""Diagnostic: where a lane of k_inflate spends its time.  Needs the profiling build
  (cd bamsignals_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-kernarg-preload-count=8 \
     -DBSIG_INFLATE_PROF -shared -o ../libbamsignals_hip_prof.so kernels.hip runtime.hip devdecode.hip collect.hip \
     bamio.cpp fileapi.cpp -lz -lpthread -ldl)
which counts per lane: turns of the symbol loop, turns that carried literals / a slice of a match, symbols that
missed the first-level table, deflate blocks (= code tables built) and the shader-clock cycles between a deflate
block's first bit and its first symbol.  Lanes of a wave run in lockstep, so a lane's cycles include what it
waited for its neighbours."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["BSIG_LIB_PATH"] = os.path.join(ROOT, "bamsignals_amd", "libbamsignals_hip_prof.so")
import numpy as np  # noqa: E402
import torch  # noqa: F401,E402

from bamsignals_amd import _lib  # noqa: E402
from bamsignals_amd.bamio import BamFile, write_columns_as_bam  # noqa: E402
from bamsignals_amd.device import Context, Reads  # noqa: E402
from bamsignals_amd.synth import synth_reads  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "real"
if kind in ("real", "real6"):
    # (real6: the same records written at zlib level 6, htslib's default: shorter literal runs, more matches)
    bam = "/tmp/dd_real.bam" if kind == "real" else "/tmp/dd_real6.bam"
    if not os.path.exists(bam):
        cols = synth_reads(20_000_000, [250_000_000], seed=12)
        write_columns_as_bam(bam, ["chr1"], cols, level=1 if kind == "real" else 6, l_seq=100, seed=3)
        del cols
else:
    bam = "/tmp/ns_small.bam"
    if not os.path.exists(bam):
        cols = synth_reads(100_000_000, [250_000_000] * 10, seed=9)
        write_columns_as_bam(bam, ["c%d" % i for i in range(10)], cols, level=1)
        del cols
ctx = Context(0)
b = BamFile(bam)
os.environ["BAMSIGNALS_DEVICE_DECODE"] = "require"
os.environ["BAMSIGNALS_INFLATE"] = "gpu"
lanes = int(os.environ.get("BAMSIGNALS_INFLATE_LANES", "32"))
for rep in range(2):
    r = Reads.from_bam(ctx, b)
    print({k: round(v, 4) for k, v in Reads.device_decode_timing().items()}, flush=True)
    r.close()
row = np.dtype([("turns", "<u4"), ("hdrs", "<u4"), ("walks", "<u4"), ("match_turns", "<u4"), ("lit_turns", "<u4"), ("lits", "<u4"),
                ("hdr_cycles", "<u8"), ("sec_cycles", "<u8", (8,)), ("sec_n", "<u4", (8,)), ("short_period", "<u4"),
                ("long_dist", "<u4"), ("pad", "<u4", (2,)), ("cycles", "<u8"), ("isize", "<u4"), ("in_len", "<u4")])
assert row.itemsize == 160
n = 1 << 17
rows = np.zeros(n, dtype=row)
lib = _lib.load()
fn = lib.bsig_debug_inflate_prof
fn.argtypes = [ctypes.c_void_p, ctypes.c_int64]
assert fn(rows.ctypes.data, n) == 0
rows = rows[rows["isize"] > 0]
print(len(rows), "lanes of the last launches")
full = rows[: len(rows) // lanes * lanes].reshape(-1, lanes)
f = lambda a: "%.4g" % float(np.mean(a))
print("per lane: isize", f(rows["isize"]), "in_len", f(rows["in_len"]), "deflate blocks", f(rows["hdrs"]), "turns", f(rows["turns"]),
      "literal turns", f(rows["lit_turns"]), "literals", f(rows["lits"]), "match-slice turns", f(rows["match_turns"]),
      "first-level misses", f(rows["walks"]), "matches at a distance below 8", f(rows["short_period"]),
      "distance codes beyond their first-level table", f(rows["long_dist"]))
print("per lane: cycles", f(rows["cycles"]), "cycles before the first symbol of its deflate blocks", f(rows["hdr_cycles"]),
      "= %.3f of the lane's time; per deflate block %.4g cycles" % (rows["hdr_cycles"].sum() / rows["cycles"].sum(), rows["hdr_cycles"].sum() / max(1, rows["hdrs"].sum())))
wc = full["cycles"].max(axis=1).astype(np.float64)
print("per wave (%d lanes): cycles" % lanes, f(wc), "; longest lane's turns", f(full["turns"].max(axis=1)), "mean lane's turns", f(full["turns"].mean(axis=1)),
      "; cycles per turn of the longest lane %.4g" % (wc.sum() / full["turns"].max(axis=1).sum()),
      "; sum over lanes of header cycles / wave cycles %.3f" % (full["hdr_cycles"].sum() / wc.sum()))
print("per wave: turns in which SOME lane missed the first-level table: at most", f(np.minimum(full["walks"].sum(axis=1), full["turns"].max(axis=1))),
      "of", f(full["turns"].max(axis=1)))
names = ["refill + first symbol", "further literals (+ symbol behind them)", "length/distance of a match", "deferred stores",
         "literals' store", "match loads / direct copy", "whole turn", "the wait for memory (between 2 and 3)"]
turn = rows["sec_cycles"][:, 6].sum() / rows["sec_n"][:, 6].sum()
for k, nm in enumerate(names):
    nk = rows["sec_n"][:, k].sum()
    ck = rows["sec_cycles"][:, k].sum()
    print("section %d %-42s entered in %.3f of a lane's turns, %.0f cycles when entered, %.0f per turn of the lane (%.2f of the turn)"
          % (k, nm, nk / rows["turns"].sum(), ck / max(1, nk), ck / rows["turns"].sum(), ck / rows["turns"].sum() / turn))
