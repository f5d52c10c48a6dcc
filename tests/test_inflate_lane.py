"""CPU: the per-lane DEFLATE decoder behind the GPU inflate kernel (csrc/inflate_lane.h, the same
source compiled for the host) against zlib: every BGZF block of the reference's fixture BAM, data of
several kinds at every zlib level and strategy (stored, fixed and dynamic Huffman blocks, runs and
short-period matches, long codes), and damaged streams, which must fail cleanly."""
import ctypes
import gzip
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

BAM = os.path.join(GOLDEN, "randomBam.bam")


@pytest.fixture(scope="module")
def lane(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("inflate") / "libinflate_lane_host.so")
    # (BSIG_LANE_CXXFLAGS: the decoder's compile-time variants, e.g. -DBSIG_DFAST=6)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-shared", "-fPIC", *os.environ.get("BSIG_LANE_CXXFLAGS", "").split(),
                           "-o", so, os.path.join(ROOT, "tests", "inflate_lane_host.cpp")])
    lib = ctypes.CDLL(so)
    lib.inflate_lane_host.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32]

    def run(raw, n):
        out = ctypes.create_string_buffer(max(n, 1))
        rc = lib.inflate_lane_host(raw, len(raw), out, n)
        return rc, out.raw[:n]
    return run


def _blocks(path):
    d = open(path, "rb").read()
    o = 0
    while o < len(d):
        xlen = struct.unpack_from("<H", d, o + 10)[0]
        bsize = struct.unpack_from("<H", d, o + 16)[0] + 1
        yield d[o + 12 + xlen:o + bsize - 8], struct.unpack_from("<I", d, o + bsize - 4)[0]
        o += bsize


def test_fixture_bam_blocks(lane):
    n = 0
    for raw, isize in _blocks(BAM):                       # htslib's own deflate output
        rc, got = lane(raw, isize)
        assert rc == 0 and got == zlib.decompress(raw, -15)
        n += 1
    assert n == 72


def test_levels_strategies_and_data_kinds(lane):
    rng = np.random.default_rng(1)
    stream = gzip.decompress(open(BAM, "rb").read())
    for trial in range(96):
        kind = trial % 8
        sz = int(rng.integers(0, 65536))
        if kind == 0:
            data = bytes(rng.integers(0, 256, sz).astype(np.uint8))          # incompressible: stored / long codes
        elif kind == 1:
            data = bytes(rng.integers(0, 4, sz).astype(np.uint8))            # four symbols
        elif kind == 2:
            o = int(rng.integers(0, len(stream) - sz))
            data = stream[o:o + sz]                                          # BAM records
        elif kind == 3:
            data = bytes([int(rng.integers(0, 256))]) * sz                   # one run: distance 1, length 258
        elif kind == 4:
            data = (b"ACGT" * 20000)[:sz]                                    # period 4
        elif kind == 5:
            data = (bytes(rng.integers(0, 256, 7).astype(np.uint8)) * 10000)[:sz]   # period 7
        elif kind == 6:
            # periods 8..70: a match whose source overlaps its destination by less than a slice (the
            # distance-doubling path of inflate_block), the way one bare BAM record repeats the previous one
            per = int(rng.integers(8, 71))
            data = (bytes(rng.integers(0, 256, per).astype(np.uint8)) * (65536 // per + 1))[:sz]
        else:
            # ... and records that repeat with a few bytes changed each time
            per = int(rng.integers(9, 64))
            rec = rng.integers(0, 256, per).astype(np.uint8)
            rows = np.tile(rec, (65536 // per + 1, 1))
            rows[:, int(rng.integers(0, per))] = rng.integers(0, 256, len(rows))
            rows[::7, int(rng.integers(0, per))] += 1
            data = rows.tobytes()[:sz]
        for level, strategy in ((0, 0), (1, 0), (6, 0), (9, 0), (6, 4), (6, 2), (6, 3), (1, 1)):
            co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
            raw = co.compress(data) + co.flush()
            rc, got = lane(raw, len(data))
            assert rc == 0 and got == data, (kind, sz, level, strategy)
    # several deflate blocks in one stream (Z_FULL_FLUSH between them), as a BGZF block may hold
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    raw = b"".join(co.compress(stream[k:k + 9000]) + co.flush(zlib.Z_FULL_FLUSH) for k in range(0, 54000, 9000)) + co.flush()
    rc, got = lane(raw, 54000)
    assert rc == 0 and got == stream[:54000]


def test_damaged_streams_fail_cleanly(lane):
    rng = np.random.default_rng(2)
    stream = gzip.decompress(open(BAM, "rb").read())[:60000]
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    raw = co.compress(stream) + co.flush()
    rc, got = lane(raw, 60000)
    assert rc == 0 and got == stream
    bad = 0
    for _ in range(300):
        b = bytearray(raw)
        i = int(rng.integers(0, len(b)))
        b[i] ^= 1 << int(rng.integers(0, 8))
        rc, got = lane(bytes(b), 60000)                     # must return, whatever it returns
        bad += rc != 0 or got != stream
    assert bad == 300
    assert lane(raw[:len(raw) // 2], 60000)[0] != 0           # truncated input
    assert lane(raw, 59999)[0] != 0 and lane(raw, 60001)[0] != 0   # wrong uncompressed size
    assert lane(b"", 10)[0] != 0
