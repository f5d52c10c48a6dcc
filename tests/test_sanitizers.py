"""CPU: the parsers of untrusted input under AddressSanitizer + UndefinedBehaviorSanitizer.

`make -C bamsignals_amd/csrc asan` builds csrc/bamio.cpp (BGZF / BAM / BAI reader, BAM writer, SAM
parser) and the host build of csrc/inflate_lane.h with -fsanitize=address,undefined into a small
driver (tests/asan/bamio_asan_driver.cpp).  Seeded mutations of the reference's fixture BAM and
BAI -- at the compressed level, and at the record level with the stream re-compressed so that it
passes inflate and CRC -- of SAM text, and of raw DEFLATE streams go through it: every run must
end in a successful parse or a clean error, never in a sanitizer report.  (CPU build only: GPU
sanitizers are not available on the pool.)"""
import gzip
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

BAM = os.path.join(GOLDEN, "randomBam.bam")
EOF_BLOCK = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
DRIVER = os.path.join(ROOT, "tests", "asan", "bamio_asan_driver")


@pytest.fixture(scope="module")
def driver():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "bamsignals_amd", "csrc"), "asan"])

    def run(*args, env=None):
        e = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:exitcode=66", UBSAN_OPTIONS="print_stacktrace=1",
                 BAMSIGNALS_THREADS="3")
        e.update(env or {})
        p = subprocess.run([DRIVER, *args], capture_output=True, text=True, errors="replace", env=e, timeout=120)
        assert p.returncode in (0, 3), (args, p.returncode, p.stdout[-400:], p.stderr[-3000:])
        return p.returncode, p.stdout
    return run


def _bgzf(data, size=60000, level=1):
    out = b""
    for i in range(0, len(data), size):
        chunk = data[i:i + size]
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        dd = co.compress(chunk) + co.flush()
        out += (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(dd) + 25) + dd
                + struct.pack("<II", zlib.crc32(chunk), len(chunk)))
    return out + EOF_BLOCK


def _mutate(rng, raw, lo=0, n_max=6):
    b = bytearray(raw)
    kind = rng.integers(0, 5) if len(b) > lo + 80 else 0
    if kind == 0:                                        # flip a few bytes
        for _ in range(int(rng.integers(1, n_max + 1))):
            b[int(rng.integers(lo, len(b)))] = int(rng.integers(0, 256))
    elif kind == 1:                                      # truncate
        del b[int(rng.integers(lo, len(b))):]
    elif kind == 2:                                      # overwrite a 32-bit field with an extreme value
        at = int(rng.integers(lo, len(b) - 4))
        b[at:at + 4] = struct.pack("<i", int(rng.choice([-1, 0, 2**31 - 1, -2**31, 65536, 1 << 28])))
    elif kind == 3:                                      # duplicate a slice
        at = int(rng.integers(lo, len(b) - 64))
        b[at:at] = b[at:at + int(rng.integers(1, 64))]
    else:                                                # delete a slice
        at = int(rng.integers(lo, len(b) - 64))
        del b[at:at + int(rng.integers(1, 64))]
    return bytes(b)


def test_clean_inputs_parse(driver, tmp_path):
    rc, out = driver("decode", BAM)
    assert rc == 0 and out.startswith("ok")
    sam = tmp_path / "a.sam"
    sam.write_text("@HD\tVN:1.0\tSO:coordinate\n@SQ\tSN:c1\tLN:5000\n"
                   "r1\t99\tc1\t10\t30\t5S20M2D10M\t=\t100\t120\tACGTACGTACGTACGTACGTACGTACGTACGTACGTA\t*\tNM:i:3\tXB:B:c,1,-2\n"
                   "r2\t147\tc1\t100\t30\t30M\t=\t10\t-120\t*\t*\n")
    rc, out = driver("sam2bam", str(sam), str(tmp_path / "a.bam"))
    assert rc == 0 and out.startswith("ok")


def test_mutated_bam_and_bai(driver, tmp_path):
    rng = np.random.default_rng(20251003)
    raw = open(BAM, "rb").read()
    bai = open(BAM + ".bai", "rb").read()
    stream = gzip.decompress(raw)
    small = stream[:400_000]                             # header + ~8,000 records: quick to re-compress
    # cut at a record border so that the unmutated file is valid
    o = 12 + struct.unpack_from("<i", small, 4)[0]
    for _ in range(3):
        o += 8 + struct.unpack_from("<i", small, o)[0]
    first = o
    while o + 4 <= len(small) and o + 4 + struct.unpack_from("<i", small, o)[0] <= len(small):
        o += 4 + struct.unpack_from("<i", small, o)[0]
    small = small[:o]
    p = tmp_path / "m.bam"
    outcomes = {0: 0, 3: 0}
    for case in range(int(os.environ.get("BSIG_ASAN_CASES", "90"))):
        which = case % 3
        if which == 0:                                   # compressed level (CRC check on and off)
            p.write_bytes(_mutate(rng, raw[:600_000] + EOF_BLOCK, lo=0))
            (tmp_path / "m.bam.bai").write_bytes(bai)
            env = {"BAMSIGNALS_NO_CRC": "1"} if case % 2 else {}
        elif which == 1:                                 # record level: passes inflate + CRC, reaches the record parser
            p.write_bytes(_bgzf(_mutate(rng, small, lo=int(rng.choice([0, first])))))
            (tmp_path / "m.bam.bai").write_bytes(bai)
            env = {}
        else:                                            # the index
            p.write_bytes(raw)
            (tmp_path / "m.bam.bai").write_bytes(_mutate(rng, bai, lo=0, n_max=3))
            env = {}
        rc, _ = driver("decode", str(p), env=env)
        outcomes[rc] += 1
    assert outcomes[3] > 20 and outcomes[0] > 5, outcomes      # both clean errors and surviving parses were seen


def test_counts_the_file_cannot_hold_are_errors_not_allocations(driver, tmp_path):
    """Found by the 1,500-case soak of the test above: a BAI whose n_ref field says 368 million made
    bai_load allocate 94 GB before looking at the file size."""
    raw = open(BAM, "rb").read()
    bai = open(BAM + ".bai", "rb").read()
    p = tmp_path / "m.bam"
    p.write_bytes(raw)
    for n_ref in (0x15F00000, 2**31 - 1, -5):
        (tmp_path / "m.bam.bai").write_bytes(bai[:4] + struct.pack("<i", n_ref) + bai[8:])
        rc, out = driver("decode", str(p))
        assert rc == 0                      # the driver decodes the whole file and skips the region query without an index
    # the same in the BAM header: l_text and n_ref far beyond the file
    stream = gzip.decompress(raw)
    for patch in (lambda b: b[:4] + struct.pack("<i", 2**31 - 1) + b[8:],
                  lambda b: b[:8 + struct.unpack_from("<i", b, 4)[0]] + struct.pack("<i", 2**30) + b[12 + struct.unpack_from("<i", b, 4)[0]:]):
        p.write_bytes(_bgzf(patch(stream[:200_000])))
        (tmp_path / "m.bam.bai").write_bytes(bai)
        rc, _ = driver("decode", str(p))
        assert rc == 3


def test_writer_refuses_coordinates_a_bai_cannot_address(driver, tmp_path):
    """Found by the 1,200-case soak of the SAM test below: POS far below 0 (or beyond 2^29) indexed the
    writer's linear index out of bounds.  POS 0 (placed on a reference, no position) is legal."""
    head = "@HD\tVN:1.0\tSO:coordinate\n@SQ\tSN:c1\tLN:50000\n"
    for pos, ok in ((0, True), (-70000, False), (-1, False), (2**29 + 5, False), (2**29 - 10, False), (2**29 - 31, True)):
        (tmp_path / "p.sam").write_text(head + f"r\t0\tc1\t{pos}\t30\t30M\t*\t0\t0\t*\t*\n")
        rc, out = driver("sam2bam", str(tmp_path / "p.sam"), str(tmp_path / "p.bam"))
        assert (rc == 0) == ok, (pos, out)


def test_mutated_sam_text(driver, tmp_path):
    rng = np.random.default_rng(7)
    lines = ["@HD\tVN:1.0\tSO:coordinate", "@SQ\tSN:c1\tLN:50000", "@SQ\tSN:c2\tLN:900"]
    for i in range(40):
        lines.append(f"r{i}\t{int(rng.choice([0, 16, 99, 147, 4]))}\tc1\t{10 + 7 * i}\t{i % 60}\t"
                     f"{int(rng.integers(1, 40))}M{int(rng.integers(1, 9))}D{int(rng.integers(1, 40))}M\t=\t{50 + i}\t{int(rng.integers(-300, 300))}\t*\t*\tNM:i:{i}\tZZ:Z:abc\tXB:B:S,1,2,3")
    text = ("\n".join(lines) + "\n").encode()
    outcomes = {0: 0, 3: 0}
    n_sam = int(os.environ.get("BSIG_ASAN_CASES", "40"))
    for case in range(n_sam):
        (tmp_path / "m.sam").write_bytes(_mutate(rng, text, lo=0, n_max=4))
        rc, _ = driver("sam2bam", str(tmp_path / "m.sam"), str(tmp_path / "m.bam"))
        outcomes[rc] += 1
    assert outcomes[0] + outcomes[3] == n_sam


def test_inflate_lane_on_damaged_streams(driver, tmp_path):
    """inflate_lane.h stores 8 bytes at a time and reads its input 8 bytes ahead: with exactly-sized
    heap buffers any access outside them is an AddressSanitizer report."""
    rng = np.random.default_rng(99)
    data = gzip.decompress(open(BAM, "rb").read())[100_000:140_000]
    kinds = [data, bytes(30_000), bytes(rng.integers(0, 256, 20_000, dtype=np.uint8)), (b"abcde" * 9000)[:40_000], b"", b"x"]
    f = tmp_path / "d.raw"
    ok = bad = 0
    for k, d in enumerate(kinds):
        for level, strategy in ((1, 0), (6, 0), (9, 0), (0, 0), (6, 4), (6, 2)):
            co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
            dd = co.compress(d) + co.flush()
            f.write_bytes(dd)
            rc, out = driver("inflate", str(f), str(len(d)))
            assert rc == 0 and int(out.split()[2]) == sum(d), (k, level, strategy)
            ok += 1
            for _ in range(3):
                f.write_bytes(_mutate(rng, dd + b"\x00" * 8, lo=0, n_max=3)[: max(1, len(dd))])
                rc, _ = driver("inflate", str(f), str(int(len(d) * rng.choice([1, 1, 0.5, 2]))))
                bad += rc == 3
    assert ok == 36 and bad > 30
