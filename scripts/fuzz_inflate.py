#!/usr/bin/env python3
"""Diagnostic: k_inflate ALONE (bsig_debug_inflate_bench: every block's CRC32 and length checked on the device) on
seeded random BGZF blocks of every kind of content a DEFLATE stream can hold -- not BAM records: runs (matches at distance 1:
the consumer's pattern path), short periods (distances 2 ... 40: a match that overlaps its own output inside the 16-byte
accumulator), copies at random distances up to 32,768 and lengths 3 ... 258, skewed alphabets (long codes: the canonical
walk behind the first-level tables), random bytes (stored blocks), zeros, and mixtures -- at zlib levels 0 / 1 / 6 / 9 and all five
strategies (default, filtered, Huffman only, RLE, fixed codes), in blocks of 1 ... 65,280 bytes, 4 / 8 / 16 / 32 / 64 blocks per
workgroup.   usage: fuzz_inflate.py [first seed] [seeds] [blocks per seed] [damage]
With `damage`: one to four bytes of the DEFLATE data of seven blocks in ten are then overwritten at random (headers and
trailers stay: the block table is as before) and the file goes through the kernel again -- a launch over damaged blocks must
come back, with a status that says so, whatever the damage made of the code tables, the distances and the lengths."""
import ctypes
import os
import struct
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
EOF_BLOCK = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def content(rng, n):
    kind = int(rng.integers(0, 8))
    if kind == 0:                                            # runs
        out = bytearray()
        while len(out) < n:
            out += bytes([int(rng.integers(0, 256))]) * int(rng.geometric(1.0 / float(rng.choice([2, 10, 300]))))
        return bytes(out[:n])
    if kind == 1:                                            # a short period
        p = int(rng.integers(1, 41))
        pat = bytes(rng.integers(0, 256, p).astype(np.uint8))
        return (pat * (n // p + 1))[:n]
    if kind == 2:                                            # copies at random distances and lengths between literals
        out = bytearray(bytes(rng.integers(0, 256, min(n, int(rng.integers(1, 400)))).astype(np.uint8)))
        while len(out) < n:
            if rng.random() < 0.3:
                out += bytes(rng.integers(0, 256, int(rng.integers(1, 9))).astype(np.uint8))
            else:
                d = int(min(len(out), rng.choice([1, 2, 3, 7, 8, 9, 15, 16, 17, 31, 33, 255, 257, 4096, 32768, int(rng.integers(1, 32769))])))
                ln = int(rng.choice([3, 4, 10, 11, 16, 17, 18, 32, 64, 257, 258, int(rng.integers(3, 259))]))
                s = len(out) - d
                for k in range(ln):                          # (byte by byte: a copy may overlap itself)
                    out.append(out[s + k])
        return bytes(out[:n])
    if kind == 3:                                            # a skewed alphabet: long codes
        k = int(rng.choice([2, 20, 200, 256]))
        p = 1.0 / np.arange(1, k + 1) ** float(rng.choice([1.0, 2.0, 3.5]))
        return bytes(rng.choice(k, n, p=p / p.sum()).astype(np.uint8))
    if kind == 4:
        return bytes(rng.integers(0, 256, n).astype(np.uint8))     # nothing to compress
    if kind == 5:
        return bytes(n)                                      # zeros
    if kind == 6:                                            # text-like lines with repeats
        words = [bytes(rng.integers(97, 123, int(rng.integers(1, 12))).astype(np.uint8)) for _ in range(int(rng.integers(2, 300)))]
        out = bytearray()
        while len(out) < n:
            out += words[int(rng.integers(0, len(words)))] + (b"\t" if rng.random() < 0.8 else b"\n")
        return bytes(out[:n])
    a = content(rng, n // 2 + 1)                             # a mixture
    return (a + content(rng, n))[:n]


def block(rng):
    while True:
        n = int(rng.choice([1, 2, 15, 16, 17, 159, 160, 161, 1000, 65280, int(rng.integers(1, 65281))]))
        chunk = content(rng, n)
        co = zlib.compressobj(int(rng.choice([0, 1, 6, 9])), zlib.DEFLATED, -15, int(rng.choice([1, 8, 9])), int(rng.integers(0, 5)))
        dd = co.compress(chunk) + co.flush()
        if len(dd) + 25 <= 65535:
            return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(dd) + 25) + dd
                    + struct.pack("<II", zlib.crc32(chunk), len(chunk))), len(chunk)


def damaged(rng, blk):
    """`blk` (one BGZF block) with 1-4 bytes of its deflate data overwritten"""
    b = bytearray(blk)
    lo, hi = 18, len(b) - 8
    if hi <= lo or rng.random() < 0.3:
        return blk
    for _ in range(int(rng.integers(1, 5))):
        b[int(rng.integers(lo, hi))] = int(rng.integers(0, 256))
    return bytes(b)


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    n_blocks = int(sys.argv[3]) if len(sys.argv) > 3 else 1500
    damage = len(sys.argv) > 4 and sys.argv[4] == "damage"
    from bamsignals_amd import _lib
    lib = _lib.load()
    fn = lib.bsig_debug_inflate_bench
    fn.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.POINTER(ctypes.c_double),
                   ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int)]
    path = os.path.join(os.environ.get("TMPDIR", "/tmp"), "fuzz_inflate_%d.bgzf" % os.getpid())
    bad = 0
    for seed in range(first, first + seeds):
        rng = np.random.default_rng(seed)
        t0 = time.time()
        total = 0
        blocks = []
        with open(path, "wb") as fh:
            for _ in range(n_blocks):
                b, n = block(rng)
                fh.write(b)
                blocks.append(b)
                total += n
            fh.write(EOF_BLOCK)
        t1 = time.time()
        res = []
        for lanes in ("4", "8", "16", "32", "64"):
            os.environ["BAMSIGNALS_INFLATE_LANES"] = lanes
            ms, by, st = (ctypes.c_double * 2)(), (ctypes.c_int64 * 3)(), ctypes.c_int(-1)
            rc = fn(0, path.encode(), 0, n_blocks, 1, ms, by, ctypes.byref(st))
            ok = rc == 0 and st.value == 0 and by[2] == n_blocks and by[1] == total
            res.append(f"{lanes}: {'ok' if ok else f'FAILED rc {rc} status {st.value} blocks {by[2]} bytes {by[1]}'}")
            bad += not ok
        if damage:
            with open(path, "wb") as fh:
                for b in blocks:
                    fh.write(damaged(rng, b))
                fh.write(EOF_BLOCK)
            for lanes in ("8", "32", "64"):
                os.environ["BAMSIGNALS_INFLATE_LANES"] = lanes
                ms, by, st = (ctypes.c_double * 2)(), (ctypes.c_int64 * 3)(), ctypes.c_int(-1)
                rc = fn(0, path.encode(), 0, n_blocks, 1, ms, by, ctypes.byref(st))
                ok = rc == 0 and st.value != 0 and by[2] == n_blocks
                res.append(f"damaged, {lanes}: {'came back, status ' + str(st.value) if ok else f'FAILED rc {rc} status {st.value}'}")
                bad += not ok
        os.environ.pop("BAMSIGNALS_INFLATE_LANES", None)
        print(f"seed {seed}: {n_blocks} blocks, {total / 1e6:.1f} MB (made in {t1 - t0:.1f} s); blocks per workgroup " + ", ".join(res), flush=True)
    os.remove(path)
    if bad:
        raise SystemExit(f"{bad} launches FAILED")


if __name__ == "__main__":
    main()
