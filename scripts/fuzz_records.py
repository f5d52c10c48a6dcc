#!/usr/bin/env python3
"""Diagnostic: the device-side record walk and extraction (k_bam_walk, k_bam_extract) on DAMAGED RECORDS inside sound
BGZF blocks -- the deflate data and every CRC are right, the BAM fields are not: block_size too small / too large / negative,
l_read_name and n_cigar_op beyond the record, refID beyond the header, negative or unsorted positions, l_seq of 2^31 - 1,
a CG:B,I tag whose count runs past the record, cut-off last records.  The device decode (BAMSIGNALS_DEVICE_DECODE=1: it
declines what it cannot prove and the CPU path names the problem) must end like the CPU decode of the same file: the same
error class, or reads that are indistinguishable.  What is looked for is a launch that does not come back or that reads
where it must not.   usage: fuzz_records.py [first seed] [seeds] [files per seed]"""
import os
import struct
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def record(rng, rid, pos, damage):
    name = bytes(rng.integers(33, 127, int(rng.integers(1, 30))).astype(np.uint8)) + b"\x00"
    ncig = int(rng.integers(0, 5))
    cig = [(int(rng.integers(1, 200)) << 4) | int(rng.choice([0, 1, 2, 3, 4, 7, 8])) for _ in range(ncig)]
    lseq = int(rng.choice([0, 0, 20, 100]))
    flag = int(rng.choice([0, 16, 99, 147, 1040]))
    aux = b"" if rng.random() < 0.6 else b"NMC\x02"
    l_name, n_cig, l_seq_field, rid_f, pos_f = len(name), ncig, lseq, rid, pos
    if damage == "l_name":
        l_name = int(rng.integers(len(name) + 1, 256))
    elif damage == "n_cig":
        n_cig = int(rng.choice([ncig + 1, 1000, 65535]))
    elif damage == "rid":
        rid_f = int(rng.choice([1000, 2 ** 31 - 1, -7]))
    elif damage == "pos":
        pos_f = int(rng.choice([-5, 2 ** 31 - 1, -2 ** 31]))
    elif damage == "l_seq":
        l_seq_field = int(rng.choice([2 ** 31 - 1, -1, 10 ** 6]))
    elif damage == "cg":
        # the 2-operation placeholder CIGAR of a long-CIGAR record (kS lseq, kN) and a CG tag whose count runs past the record
        cig = [(max(lseq, 1) << 4) | 4, (100 << 4) | 3]
        n_cig = ncig = 2
        aux = b"CGBI" + struct.pack("<I", int(rng.choice([5, 10 ** 6, 2 ** 32 - 1]))) + struct.pack("<3I", 160, 162, 163)
    body = struct.pack("<iiBBHHHiiii", rid_f, pos_f, l_name & 255, int(rng.integers(0, 61)), 4681, n_cig & 65535, flag, l_seq_field,
                       -1, -1, int(rng.integers(-600, 600)))
    body += name + struct.pack("<%dI" % ncig, *cig) + bytes((lseq + 1) // 2) + bytes(lseq) + aux
    bs = len(body)
    if damage == "bs_small":
        bs = int(rng.choice([0, 8, 31, -4]))
    elif damage == "bs_large":
        bs = int(rng.choice([len(body) + 1, len(body) + 40, 70000, 2 ** 31 - 1]))
    elif damage == "bs_short":
        bs = max(32, len(body) - int(rng.integers(1, 12)))
    return struct.pack("<i", bs) + body


KINDS = ["l_name", "n_cig", "rid", "pos", "l_seq", "cg", "bs_small", "bs_large", "bs_short", "truncate", "unsorted", "none"]


def one_file(rng, path, T):
    n_ref = int(rng.integers(1, 4))
    ref_len = rng.integers(2000, 300_000, n_ref).astype(np.int64)
    text = b"".join(b"@SQ\tSN:r%d\tLN:%d\n" % (i, ref_len[i]) for i in range(n_ref))
    hdr = b"BAM\x01" + struct.pack("<i", len(text)) + text + struct.pack("<i", n_ref)
    for i in range(n_ref):
        nm = b"r%d\x00" % i
        hdr += struct.pack("<i", len(nm)) + nm + struct.pack("<i", int(ref_len[i]))
    kind = str(rng.choice(KINDS))
    recs = []
    for r in range(n_ref):
        for p in np.sort(rng.integers(0, ref_len[r], int(rng.integers(1, 600)))):
            recs.append((r, int(p)))
    hit = set(int(x) for x in rng.integers(0, len(recs), int(rng.integers(1, 4)))) if kind not in ("none", "truncate", "unsorted") else set()
    if kind == "unsorted" and len(recs) > 3:
        i = int(rng.integers(0, len(recs) - 1))
        recs[i], recs[i + 1] = recs[i + 1], recs[i]
    stream = hdr + b"".join(record(rng, r, p, kind if k in hit else "") for k, (r, p) in enumerate(recs))
    if kind == "truncate":
        stream = stream[:len(stream) - int(rng.integers(1, 60))]
    sizes = [int(x) for x in rng.integers(200, 65000, 5)] if rng.random() < 0.7 else [65000]
    with open(path, "wb") as fh:
        fh.write(T._bgzf(stream, sizes, [(int(rng.choice([1, 6])), 0)]))
    T._empty_bai(path + ".bai", n_ref)
    return kind


def outcome(ctx, path, mode, T):
    from bamsignals_amd import _lib
    from bamsignals_amd.bamio import BamFile
    from bamsignals_amd.device import Reads
    os.environ["BAMSIGNALS_DEVICE_DECODE"] = mode
    try:
        bam = BamFile(path)
    except (_lib.BsigError, ValueError, OSError) as e:
        return ("open", type(e).__name__), None
    try:
        r = Reads.from_bam(ctx, bam)
    except _lib.BsigError as e:
        bam.close()
        return ("error", str(e)[:60]), None
    info = T.layout_info(r)
    res = T._results(ctx, r, bam.ref_len.astype(np.int64), seed=1)
    r.close(); bam.close()
    return ("reads", info), res


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    files = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    import test_device_decode_gpu as T
    from bamsignals_amd.device import Context
    ctx = Context(0)
    tmp = tempfile.mkdtemp(prefix="fuzzrec_")
    path = os.path.join(tmp, "f.bam")
    os.environ["BAMSIGNALS_INFLATE"] = "gpu"
    bad = 0
    for seed in range(first, first + seeds):
        rng = np.random.default_rng(seed)
        t0 = time.time()
        tally = {}
        for k in range(files):
            kind = one_file(rng, path, T)
            os.environ["BAMSIGNALS_STREAM_MIN_MB"] = "0" if k % 3 == 1 else "100000"      # every third file: the streamed route
            dev, dres = outcome(ctx, path, "1", T)
            os.environ["BAMSIGNALS_STREAM_MIN_MB"] = "100000"
            cpu, cres = outcome(ctx, path, "0", T)
            same = dev[0] == cpu[0] and (dev[0] != "reads" or (dev[1] == cpu[1] and all(np.array_equal(a, b) for a, b in zip(dres, cres))))
            if not same:
                bad += 1
                print(f"seed {seed} file {k} ({kind}): device {dev} != cpu {cpu}", flush=True)
            tally[(kind, dev[0])] = tally.get((kind, dev[0]), 0) + 1
        print(f"seed {seed}: {files} files, device decode ended as the CPU decode did in {files - bad if bad == 0 else 'NOT all'} "
              f"({time.time() - t0:.1f} s): " + ", ".join(f"{k[0]}->{k[1]} {v}" for k, v in sorted(tally.items())), flush=True)
    for f in (path, path + ".bai"):
        if os.path.exists(f):
            os.remove(f)
    os.rmdir(tmp)
    ctx.close()
    if bad:
        raise SystemExit(f"{bad} files ended differently")


if __name__ == "__main__":
    main()
