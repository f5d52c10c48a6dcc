import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "fullsize: BASELINE configurations at full size that take minutes (opt-in: BAMSIGNALS_FULLSIZE=1)")
    # a fresh checkout has no built libraries (they are git-ignored): build them once, as
    # __graft_entry__.build() does (hipcc cross-compiles gfx950 without a GPU)
    so = os.path.join(ROOT, "bamsignals_amd", "libbamsignals_hip.so")
    orc = os.path.join(ROOT, "oracle", "liboracle.so")
    if not (os.path.exists(so) and os.path.exists(orc)):
        import subprocess
        subprocess.check_call(["make", "-j4", "-C", os.path.join(ROOT, "bamsignals_amd", "csrc")])
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"])


@pytest.fixture(scope="session")
def fixture_reads():
    """The reference's 99,000 fixture reads as columns (tests/golden/make_golden.py)."""
    z = np.load(os.path.join(GOLDEN, "fixture_reads.npz"))
    return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def fixture_regions(fixture_reads):
    """grgenes (20) + 30 seeded ranges; returns (regions json, ranges dict of int32 arrays)."""
    reg = json.load(open(os.path.join(GOLDEN, "regions.json")))
    names = [str(s) for s in fixture_reads["ref_names"]]
    ranges = dict(
        rid=np.asarray([names.index(c) for c in reg["chrom"]], dtype=np.int32),
        loc=np.asarray(reg["start"], dtype=np.int32) - 1,
        len=np.asarray(reg["width"], dtype=np.int32),
        strand=np.asarray([{"+": 1, "-": -1}.get(s, 0) for s in reg["strand"]], dtype=np.int32),
    )
    return reg, ranges


@pytest.fixture(scope="session")
def expected_grid():
    z = np.load(os.path.join(GOLDEN, "expected_grid.npz"))
    return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def expected_extra():
    z = np.load(os.path.join(GOLDEN, "expected_extra.npz"))
    return {k: z[k] for k in z.files}


def parse_key(key):
    """'profile|shift=0,mapq=0,ss=1,pe=filter,tf=50_200' -> (kind, dict)."""
    kind, rest = key.split("|")
    d = dict(kv.split("=") for kv in rest.split(","))
    out = dict(pe=d["pe"], tf=None if d["tf"] == "NULL" else (50, 200), mapq=int(d["mapq"]))
    if "shift" in d:
        out["shift"] = int(d["shift"])
    if "ss" in d:
        out["ss"] = bool(int(d["ss"]))
    return kind, out


def core_args(kind, p):
    """User-level parameters -> native arguments, as R/wrappers.R:76-173 does."""
    pe = p["pe"]
    req = 66 if pe != "ignore" else 0
    tlf = () if pe == "ignore" else ((0, 1000) if p["tf"] is None else p["tf"])
    if kind == "coverage":
        return dict(tlen_filter=tlf, mapqual=p["mapq"], requiredF=req, filteredF=-1, tspan=(pe == "extend"))
    a = dict(tlen_filter=tlf, mapqual=p["mapq"], shift=p["shift"], ss=p.get("ss", False), requiredF=req,
             filteredF=-1, pe_mid=(pe == "midpoint"))
    a["binsize"] = 1 if kind == "profile" else -1
    if kind == "ff16":
        a["filteredF"] = 16
    return a


def expected_packed(pos, end, flag, mapq, ref_off, ref_len, n_codes=512):
    """Which reads the resident layout puts into its packed class (csrc/bsig_types.h), restated: short reads
    (span <= 256) with 12-bit flags and a position inside their reference (rounded up to 64-kbp units) whose
    (flag, mapq) pair is one of the `n_codes` most frequent pairs -- most frequent first, ties by key
    flag | mapq << 12 -- among the short reads of a sample of the file (up to 1,024 evenly spaced chunks of
    2,048 reads: all of a file of up to 2 M reads).  Returns (boolean mask, number of codes)."""
    pos = np.asarray(pos, np.int64); end = np.asarray(end, np.int64)
    flag = np.asarray(flag, np.int64); mapq = np.asarray(mapq, np.int64)
    n = len(pos)
    span = end - pos + 1
    cand = (span >= 1) & (span <= 256) & (flag < 4096) & (pos >= 0)
    chunks = (n + 2047) // 2048
    stride = max(1, chunks // 1024)
    sampled = (np.arange(n) // 2048) % stride == 0
    key = flag | (mapq << 12)
    keys, counts = np.unique(key[cand & sampled], return_counts=True)
    order = np.lexsort((keys, -counts))
    table = keys[order][:n_codes]
    rid = np.searchsorted(np.asarray(ref_off)[1:], np.arange(n), side="right")
    ref_bp = ((np.asarray(ref_len, np.int64) >> 16) + 1) << 16
    return cand & (pos < ref_bp[rid]) & np.isin(key, table), len(table)


def layout_info(reads):
    """What two layouts of the same reads must agree on: the classes, their sizes, spans, bucket widths and the pair
    table's size -- not the bytes held on the device, which depend on which cached blocks a layout was given."""
    return {k: v for k, v in reads.info().items() if k != "hbm_bytes"}
