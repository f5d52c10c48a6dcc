"""CPU: the self-contained BGZF/BAM/BAI reader and writer (csrc/bamio.cpp) against the reference's
fixture BAM (written by htslib) and against known answers built from the SAM/BAM spec."""
import gzip
import os
import struct

import numpy as np
import pytest

from conftest import GOLDEN

BAM = os.path.join(GOLDEN, "randomBam.bam")


def _cols_equal(got, fx):
    assert np.array_equal(got["pos"], fx["bam_pos"])
    assert np.array_equal(got["flag"], fx["bam_flag"])
    assert np.array_equal(got["mapq"], fx["bam_mapq"])
    assert np.array_equal(got["tlen"], fx["bam_tlen"])
    assert np.array_equal(got["cigar_off"], fx["bam_cigar_off"])
    assert np.array_equal(got["cigar"], fx["bam_cigar"])
    assert np.array_equal(got["ref_off"], fx["ref_off"])


@pytest.mark.parametrize("threads", [1, 4])
def test_decode_reference_fixture(fixture_reads, threads):
    from bamsignals_amd.bamio import BamFile
    b = BamFile(BAM)
    assert b.ref_names == ["chr1", "chr2", "chr3"]
    assert list(b.ref_len) == [10237, 10279, 10238]
    assert b.name2id("chr3") == 2 and b.name2id("chrX") == -1
    _cols_equal(b.decode(threads=threads), fixture_reads)
    b.close()


def test_decode_with_zlib_only(fixture_reads, monkeypatch):
    """the zlib fallback of the inflater gives the same columns as libdeflate"""
    import subprocess
    import sys
    code = ("import numpy as np, sys; sys.path.insert(0, %r); from bamsignals_amd.bamio import BamFile; "
            "c = BamFile(%r).decode(); print(int(c['pos'].sum()), len(c['pos']))" % (os.path.dirname(os.path.dirname(GOLDEN)), BAM))
    env = dict(os.environ, BAMSIGNALS_NO_LIBDEFLATE="1")
    out = subprocess.check_output([sys.executable, "-c", code], env=env).decode().split()
    assert int(out[0]) == int(fixture_reads["bam_pos"].astype(np.int64).sum()) and int(out[1]) == 99000


def test_decode_in_many_small_batches(fixture_reads):
    """records straddling batch boundaries (BAMSIGNALS_BATCH_BLOCKS forces 3-block batches)"""
    import subprocess
    import sys
    code = ("import numpy as np, sys; sys.path.insert(0, %r); from bamsignals_amd.bamio import BamFile; "
            "c = BamFile(%r).decode(threads=3); "
            "print(len(c['pos']), int(c['pos'].astype(np.int64).sum()), int(c['tlen'].astype(np.int64).sum()), "
            "int(c['flag'].astype(np.int64).sum()), int(c['cigar'].astype(np.int64).sum()), [int(x) for x in c['ref_off']])"
            % (os.path.dirname(os.path.dirname(GOLDEN)), BAM))
    fx = fixture_reads
    want = "%d %d %d %d %d %s" % (99000, fx["bam_pos"].astype(np.int64).sum(), fx["bam_tlen"].astype(np.int64).sum(),
                                  fx["bam_flag"].astype(np.int64).sum(), fx["bam_cigar"].astype(np.int64).sum(),
                                  [int(x) for x in fx["ref_off"]])
    for nb in ("1", "3", "7"):
        env = dict(os.environ, BAMSIGNALS_BATCH_BLOCKS=nb)
        out = subprocess.check_output([sys.executable, "-c", code], env=env).decode().strip()
        assert out == want, nb


def test_records_straddling_bgzf_blocks(tmp_path, fixture_reads):
    """The parallel boundary scan assumes that BGZF blocks start at record boundaries and must fall
    back to a serial walk where they do not: re-block the fixture stream at odd sizes."""
    import subprocess
    import sys
    import zlib
    stream = gzip.decompress(open(BAM, "rb").read())
    out = b""
    rng = np.random.default_rng(8)
    i = 0
    while i < len(stream):
        n = int(rng.integers(500, 9000))
        chunk = stream[i:i + n]
        co = zlib.compressobj(1, zlib.DEFLATED, -15)
        dd = co.compress(chunk) + co.flush()
        out += (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(dd) + 25) + dd
                + struct.pack("<II", zlib.crc32(chunk), len(chunk)))
        i += n
    out += bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    p = tmp_path / "straddle.bam"
    p.write_bytes(out)
    (tmp_path / "straddle.bam.bai").write_bytes(b"BAI\x01" + struct.pack("<i", 3) + struct.pack("<ii", 0, 0) * 3 + struct.pack("<Q", 0))
    from bamsignals_amd.bamio import BamFile
    _cols_equal(BamFile(str(p)).decode(threads=4), fixture_reads)
    code = ("import numpy as np, sys; sys.path.insert(0, %r); from bamsignals_amd.bamio import BamFile; "
            "c = BamFile(%r).decode(threads=3); print(len(c['pos']), int(c['pos'].astype(np.int64).sum()), "
            "int(c['cigar'].astype(np.int64).sum()), [int(x) for x in c['ref_off']])" % (os.path.dirname(os.path.dirname(GOLDEN)), str(p)))
    fx = fixture_reads
    want = "%d %d %d %s" % (99000, fx["bam_pos"].astype(np.int64).sum(), fx["bam_cigar"].astype(np.int64).sum(),
                            [int(x) for x in fx["ref_off"]])
    for nb in ("2", "11"):
        got = subprocess.check_output([sys.executable, "-c", code], env=dict(os.environ, BAMSIGNALS_BATCH_BLOCKS=nb)).decode().strip()
        assert got == want, nb


def test_region_decode_is_a_superset_in_file_order(fixture_reads):
    from bamsignals_amd.bamio import BamFile
    fx = fixture_reads
    b = BamFile(BAM)
    rid = np.asarray([2, 0, 0], dtype=np.int32)
    beg = np.asarray([100, 5000, 5100], dtype=np.int64)
    end = np.asarray([900, 5200, 5600], dtype=np.int64)
    got = b.decode(rid, beg, end)
    key = got["rid"].astype(np.int64) << 32 | got["pos"]
    assert np.all(np.diff(key) >= 0)
    # every overlapping record is there, exactly once
    for r, s, e in zip(rid, beg, end):
        m = (fx["bam_rid"] == r) & (fx["bam_pos"] < e) & (fx["bam_end"] + 1 > s)
        gm = (got["rid"] == r) & (got["pos"] < e)
        want = np.stack([fx["bam_pos"][m], fx["bam_flag"][m], fx["bam_tlen"][m]], 1)
        have = np.stack([got["pos"][gm], got["flag"][gm].astype(np.int32), got["tlen"][gm]], 1)
        ws = {tuple(x) for x in want}
        hs = {tuple(x) for x in have}
        assert ws <= hs
    assert got["ref_off"][1] == got["ref_off"][2]          # nothing asked on chr2
    b.close()


def test_errors_match_the_reference(tmp_path):
    from bamsignals_amd import _lib
    from bamsignals_amd.bamio import BamFile
    with pytest.raises(_lib.BsigError) as e:
        BamFile(str(tmp_path / "nope.bam"))
    assert e.value.code_name == "BSIG_ERR_IO" and "Fail to open BAM file" in str(e.value)      # ref :204
    p = tmp_path / "noidx.bam"
    p.write_bytes(open(BAM, "rb").read())
    with pytest.raises(_lib.BsigError) as e:
        BamFile(str(p))
    assert e.value.code_name == "BSIG_ERR_NOINDEX"
    assert "BAM indexing file is not available for file" in str(e.value)                       # ref :209
    q = tmp_path / "junk.bam"
    q.write_bytes(b"this is not a bam file at all, not even close.....")
    (tmp_path / "junk.bam.bai").write_bytes(open(BAM + ".bai", "rb").read())
    with pytest.raises(_lib.BsigError) as e:
        BamFile(str(q))
    assert e.value.code_name == "BSIG_ERR_FORMAT"


def _synth(n=60_000, seed=3):
    from bamsignals_amd.synth import synth_reads
    return synth_reads(n, [3_000_000, 40_000, 1_500_000], seed=seed)


def test_writer_roundtrip_and_multibin_index(tmp_path):
    """columns -> BAM+BAI -> columns; the index spans many 16-kbp windows and bins here."""
    from bamsignals_amd.bamio import BamFile, write_columns_as_bam
    cols = _synth()
    p = str(tmp_path / "synth.bam")
    write_columns_as_bam(p, ["chrA", "chrB", "chrC"], cols)
    # the file is valid BGZF (= concatenated gzip members) with the 28-byte EOF marker
    raw = open(p, "rb").read()
    assert raw[-28:] == bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    stream = gzip.decompress(raw)
    assert stream[:4] == b"BAM\x01"
    b = BamFile(p)
    got = b.decode()
    for k in ("pos", "flag", "mapq", "tlen", "cigar_off", "cigar", "ref_off"):
        assert np.array_equal(got[k], cols[k]), k
    # region queries through our own BAI: superset of the truly overlapping reads, no duplicates
    rng = np.random.default_rng(1)
    for _ in range(25):
        r = int(rng.integers(0, 3))
        s = int(rng.integers(0, cols["ref_len"][r]))
        e = s + int(rng.integers(1, 200_000))
        sub = b.decode([r], [s], [e])
        m = (cols["rid"] == r) & (cols["pos"] < e) & (cols["end"] + 1 > s)
        sm = (sub["rid"] == r) & (sub["pos"] < e)
        from oracle import oracle_np
        sub_end = oracle_np.cigar_end(sub["pos"], sub["flag"], sub["cigar_off"], sub["cigar"])
        sm &= sub_end + 1 > s
        assert int(sm.sum()) == int(m.sum())
        assert np.array_equal(sub["pos"][sm], cols["pos"][m])
        assert len(sub["pos"]) < len(cols["pos"])
    b.close()
    # BAI structure: magic, n_ref, per-ref linear index of ceil(last_end / 16384) windows
    bai = open(p + ".bai", "rb").read()
    assert bai[:4] == b"BAI\x01" and struct.unpack_from("<i", bai, 4)[0] == 3


def test_bai_of_fixture_equivalent_bam_matches_htslib_layout(tmp_path, fixture_reads):
    """Re-emit the reference fixture with our writer: the BAI has the same shape as htslib's
    (per reference: leaf bin 4681 with one chunk + pseudo-bin 37450 with the mapped counts, one
    linear-index window, trailing n_no_coor = 0)."""
    from bamsignals_amd.bamio import BamFile, write_columns_as_bam
    fx = fixture_reads
    cols = dict(ref_len=fx["ref_len"], ref_off=fx["ref_off"], pos=fx["bam_pos"], flag=fx["bam_flag"],
                mapq=fx["bam_mapq"], tlen=fx["bam_tlen"], cigar_off=fx["bam_cigar_off"], cigar=fx["bam_cigar"])
    p = str(tmp_path / "re.bam")
    write_columns_as_bam(p, ["chr1", "chr2", "chr3"], cols, level=6)
    _cols_equal(BamFile(p).decode(), fx)

    def parse(bai):
        o = 8
        out = []
        for _ in range(struct.unpack_from("<i", bai, 4)[0]):
            nb = struct.unpack_from("<i", bai, o)[0]; o += 4
            bins = {}
            for _ in range(nb):
                b, nc = struct.unpack_from("<Ii", bai, o); o += 8
                bins[b] = [struct.unpack_from("<QQ", bai, o + 16 * k) for k in range(nc)]
                o += 16 * nc
            ni = struct.unpack_from("<i", bai, o)[0]; o += 4
            lin = struct.unpack_from("<%dQ" % ni, bai, o); o += 8 * ni
            out.append((bins, lin))
        return out, struct.unpack_from("<Q", bai, o)[0]

    ours, nn = parse(open(p + ".bai", "rb").read())
    ref, rn = parse(open(BAM + ".bai", "rb").read())
    assert nn == rn == 0
    for (b1, l1), (b2, l2) in zip(ours, ref):
        assert set(b1) == set(b2) == {4681, 37450}
        assert len(b1[4681]) == len(b2[4681]) == 1 and len(l1) == len(l2) == 1
        assert b1[37450][1] == b2[37450][1]            # (n_mapped, n_unmapped)


def test_sam_to_bam(tmp_path):
    from bamsignals_amd.bamio import BamFile, writeSamAsBamAndIndex
    sam = tmp_path / "t.sam"
    sam.write_text(
        "@HD\tVN:1.0\tSO:coordinate\n@SQ\tSN:chr1\tLN:5000\n@SQ\tSN:chr2\tLN:3000\n"
        "r1\t99\tchr1\t101\t30\t20M\t=\t201\t120\t*\t*\n"
        "r2\t16\tchr1\t150\t7\t5S10M2D10M3I5M\t*\t0\t0\tACGTNACGTNACGTNACGTNACGTNACGTNACG\t*\tNM:i:3\tXS:Z:hello\tXB:B:s,1,-2,300\n"
        "r3\t147\tchr1\t201\t30\t20M\t=\t101\t-120\t*\t*\tXA:A:c\tXF:f:1.5\tXL:i:-70000\n"
        "r4\t4\tchr2\t10\t0\t*\t*\t0\t0\tACGT\tIIII\n"
        "r5\t0\tchr2\t500\t60\t10M1000N10M\t*\t0\t0\t*\t*\n")
    bam = str(tmp_path / "t.bam")
    assert writeSamAsBamAndIndex(str(sam), bam) is True
    assert os.path.exists(bam + ".bai")
    b = BamFile(bam)
    c = b.decode()
    assert list(c["pos"]) == [100, 149, 200, 9, 499]
    assert list(c["flag"]) == [99, 16, 147, 4, 0]
    assert list(c["mapq"]) == [30, 7, 30, 0, 60]
    assert list(c["tlen"]) == [120, 0, -120, 0, 0]
    assert list(c["ref_off"]) == [0, 3, 5]
    assert list(np.diff(c["cigar_off"])) == [1, 6, 1, 0, 3]
    from oracle import oracle_np
    end = oracle_np.cigar_end(c["pos"], c["flag"], c["cigar_off"], c["cigar"])
    assert list(end) == [119, 149 + 27 - 1, 219, 9, 499 + 1020 - 1]
    # byte-level check of the second record against the SAM spec (section 4.2)
    stream = gzip.decompress(open(bam, "rb").read())
    i = stream.rindex(b"r2\x00") - 36      # the last match: "chr2\0" of the header comes first
    bs, rid, pos, lrn, mq, bin_, ncig, flag, lseq, nrid, npos, tlen = struct.unpack_from("<iiiBBHHHiiii", stream, i)
    assert (rid, pos, lrn, mq, ncig, flag, lseq, nrid, npos, tlen) == (0, 149, 3, 7, 6, 16, 33, -1, -1, 0)
    assert bin_ == 4681
    seq = stream[i + 36 + 3 + 24:i + 36 + 3 + 24 + 17]
    assert seq[0] == 0x12 and seq[1] == 0x48           # A=1 C=2 | G=4 T=8
    aux = stream[i + 36 + 3 + 24 + 17 + 33:i + 4 + bs]
    assert aux.startswith(b"NMC\x03XSZhello\x00XBBs\x03\x00\x00\x00\x01\x00\xfe\xff\x2c\x01")
    # sorted check: an unsorted SAM cannot be indexed
    bad = tmp_path / "bad.sam"
    bad.write_text("@SQ\tSN:chr1\tLN:5000\nr1\t0\tchr1\t300\t30\t20M\t*\t0\t0\t*\t*\nr2\t0\tchr1\t100\t30\t20M\t*\t0\t0\t*\t*\n")
    from bamsignals_amd import _lib
    with pytest.raises(_lib.BsigError):
        writeSamAsBamAndIndex(str(bad), str(tmp_path / "bad.bam"))


def test_long_cigar_cg_tag(tmp_path):
    """> 65535 operations: kSmN placeholder + CG:B,I tag (SAM spec 4.2.2)."""
    from bamsignals_amd.bamio import BamFile
    # hand-build a BAM with one such record
    ops = [(1 << 4) | 0, (1 << 4) | 2] * 40000      # 1M1D x 40000 = 80000 ops, span 80000
    lseq = 40000
    name = b"long\x00"
    cig_placeholder = struct.pack("<II", lseq << 4 | 4, 80000 << 4 | 3)
    seq = bytes((lseq + 1) // 2)
    qual = b"\xff" * lseq
    aux = b"CGBI" + struct.pack("<I", len(ops)) + struct.pack("<%dI" % len(ops), *ops)
    body = struct.pack("<iiBBHHHiiii", 0, 1000, len(name), 50, 4681, 2, 0, lseq, -1, -1, 0) + name + cig_placeholder + seq + qual + aux
    rec = struct.pack("<i", len(body)) + body
    text = b"@SQ\tSN:c\tLN:200000\n"
    hdr = b"BAM\x01" + struct.pack("<i", len(text)) + text + struct.pack("<i", 1) + struct.pack("<i", 2) + b"c\x00" + struct.pack("<i", 200000)
    stream = hdr + rec

    def bgzf(data):
        out = b""
        for i in range(0, len(data), 60000):
            import zlib
            chunk = data[i:i + 60000]
            co = zlib.compressobj(6, zlib.DEFLATED, -15)
            d = co.compress(chunk) + co.flush()
            out += b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(d) + 25) + d + struct.pack("<II", zlib.crc32(chunk), len(chunk))
        return out + bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")

    p = tmp_path / "long.bam"
    p.write_bytes(bgzf(stream))
    # minimal BAI: no bins, no linear index
    (tmp_path / "long.bam.bai").write_bytes(b"BAI\x01" + struct.pack("<i", 1) + struct.pack("<ii", 0, 0) + struct.pack("<Q", 0))
    c = BamFile(str(p)).decode()
    assert len(c["pos"]) == 1 and c["cigar_off"][-1] == 80000
    assert np.array_equal(c["cigar"], np.asarray(ops, dtype=np.uint32))


def test_block_scan_in_segments(fixture_reads, monkeypatch, tmp_path):
    """large files have their BGZF header chain walked in segments at once (a segment starts at the
    first offset from which a chain of valid headers follows; compressed data holds the two magic
    bytes every 64 kB or so by chance); forced here on the fixture (456 kB) and on a synthetic BAM"""
    from bamsignals_amd.bamio import BamFile, write_columns_as_bam
    for kb in ("16", "40", "100"):
        monkeypatch.setenv("BAMSIGNALS_SCAN_SEGMENT_KB", kb)
        _cols_equal(BamFile(BAM).decode(threads=4), fixture_reads)
    monkeypatch.delenv("BAMSIGNALS_SCAN_SEGMENT_KB")
    cols = _synth(400_000, seed=9)
    p = str(tmp_path / "syn.bam")
    write_columns_as_bam(p, ["a", "b", "c"], cols)
    want = BamFile(p).decode(threads=2)
    assert np.array_equal(want["pos"], cols["pos"])
    for kb in ("8", "64"):
        monkeypatch.setenv("BAMSIGNALS_SCAN_SEGMENT_KB", kb)
        got = BamFile(p).decode(threads=2)
        for k in ("pos", "flag", "mapq", "tlen", "cigar", "ref_off"):
            assert np.array_equal(got[k], want[k])


def test_block_table_through_pread_equals_the_mapped_walk(tmp_path, monkeypatch):
    """The device-side decode tabulates the BGZF blocks with pread() (a block's trailer and the next block's
    header in one small read; no page of the mapping is touched) in parallel segments; the table must be the
    one the walk through the mapped file gives, whatever the segment size, and a file the segments cannot be
    proven on (too small, damaged) must fall back to that walk with its error messages."""
    import ctypes as C
    from bamsignals_amd import _lib
    from bamsignals_amd.bamio import write_columns_as_bam
    lib = _lib.load()

    def table(path, how, seg_kb=None):
        monkeypatch.setenv("BAMSIGNALS_SCAN", how)
        if seg_kb:
            monkeypatch.setenv("BAMSIGNALS_SCAN_SEGMENT_KB", seg_kb)
        else:
            monkeypatch.delenv("BAMSIGNALS_SCAN_SEGMENT_KB", raising=False)
        n, h = C.c_int64(), C.c_uint64()
        _lib.check(lib.bsig_debug_block_table(path.encode(), C.byref(n), C.byref(h)))
        return n.value, h.value

    want = table(BAM, "mmap")
    assert want[0] > 5
    for kb in ("16", "40", "100", None):
        assert table(BAM, "pread", kb) == want
    cols = _synth(900_000, seed=10)
    p = str(tmp_path / "syn.bam")
    write_columns_as_bam(p, ["a", "b", "c"], cols)
    want = table(p, "mmap")
    assert want[0] > 500
    for kb in ("8", "64", "1024", None):
        assert table(p, "pread", kb) == want
    # the table in two steps (head first, the rest in the background), segments smaller than a block included
    def progressive(path, head, seg_kb):
        monkeypatch.setenv("BAMSIGNALS_SCAN", "pread")
        monkeypatch.setenv("BAMSIGNALS_SCAN_SEGMENT_KB", seg_kb)
        nh, n, h = C.c_int64(), C.c_int64(), C.c_uint64()
        _lib.check(lib.bsig_debug_block_table_progressive(path.encode(), head, C.byref(nh), C.byref(n), C.byref(h)))
        return nh.value, (n.value, h.value)
    want_fx = table(BAM, "mmap")
    for kb in ("8", "16", "100"):
        nh, got = progressive(BAM, 300 << 10, kb)
        assert got == want_fx and 8 < nh < want_fx[0]
    for kb, head in (("8", 1 << 20), ("64", 2 << 20), ("1024", 2 << 20)):
        nh, got = progressive(p, head, kb)
        assert got == want and 8 < nh < want[0], (kb, head, nh)
    nh, got = progressive(p, 1 << 30, "64")                 # a head larger than the file: all at once
    assert got == want and nh == want[0]
    # a file cut in the middle of a block: both walks refuse it with the same message
    raw = open(p, "rb").read()
    q = str(tmp_path / "cut.bam")
    open(q, "wb").write(raw[: len(raw) // 2 + 1234])
    for how in ("mmap", "pread"):
        with pytest.raises(_lib.BsigError, match="malformed BGZF block"):
            table(q, how, "64")


def test_crc_of_bgzf_blocks_is_checked(tmp_path):
    """a block whose trailer CRC32 does not match what it inflates to is refused (as htslib does)"""
    from bamsignals_amd import _lib
    from bamsignals_amd.bamio import BamFile
    raw = bytearray(open(BAM, "rb").read())
    bsize = struct.unpack_from("<H", raw, 16)[0] + 1           # first block; damage the second one's CRC
    b2 = struct.unpack_from("<H", raw, bsize + 16)[0] + 1
    raw[bsize + b2 - 8] ^= 0x5A
    p = tmp_path / "crc.bam"
    p.write_bytes(bytes(raw))
    (tmp_path / "crc.bam.bai").write_bytes(open(BAM + ".bai", "rb").read())
    with pytest.raises(_lib.BsigError, match="CRC"):
        BamFile(str(p)).decode()


@pytest.mark.parametrize("threads", ["1", "5"])
def test_pooled_column_writer_equals_record_by_record(tmp_path, monkeypatch, threads):
    """bsig_write_columns_as_bam builds and deflates the BGZF blocks on the worker pool; the file and
    its index must be byte-identical to the record-by-record writer's (same block cuts, same virtual
    offsets), including a block that a record fills exactly to the brim."""
    import filecmp
    from bamsignals_amd.bamio import BamFile, write_columns_as_bam
    from bamsignals_amd.synth import synth_reads
    cols = synth_reads(400_000, [900_000, 70_000, 400_000], seed=3, paired=True)
    # 1,536 reads of 42 bytes + one of 46 bytes... make the first block end exactly at 0xff00:
    # 0xff00 = 65,280 = 1,554 * 42 + 12 -> use 42-byte records (1 op) and one 54-byte record (4 ops)
    n = 5000
    brim = dict(ref_len=np.asarray([100_000], np.int32), ref_off=np.asarray([0, n], np.int64),
                pos=np.arange(n, dtype=np.int32), flag=np.zeros(n, np.uint16), mapq=np.full(n, 9, np.uint8),
                tlen=np.zeros(n, np.int32))
    nops = np.ones(n, np.int64)
    nops[7] = 4                                           # 1,553 * 42 + 54 = 65,280
    brim["cigar_off"] = np.concatenate([[0], np.cumsum(nops)]).astype(np.int64)
    brim["cigar"] = np.full(int(brim["cigar_off"][-1]), 10 << 4, np.uint32)
    monkeypatch.setenv("BAMSIGNALS_THREADS", threads)
    for tag, c, names in (("syn", cols, ["a", "b", "c"]), ("brim", brim, ["z"])):
        monkeypatch.setenv("BAMSIGNALS_WRITER", "serial")
        write_columns_as_bam(str(tmp_path / f"{tag}_s.bam"), names, c)
        monkeypatch.delenv("BAMSIGNALS_WRITER")
        write_columns_as_bam(str(tmp_path / f"{tag}_p.bam"), names, c)
        assert filecmp.cmp(tmp_path / f"{tag}_s.bam", tmp_path / f"{tag}_p.bam", shallow=False), tag
        assert filecmp.cmp(tmp_path / f"{tag}_s.bam.bai", tmp_path / f"{tag}_p.bam.bai", shallow=False), tag
        b = BamFile(str(tmp_path / f"{tag}_p.bam"))
        got = b.decode(threads=2)
        assert np.array_equal(got["pos"], c["pos"]) and np.array_equal(got["cigar"], c["cigar"])
        b.close()


def test_real_shaped_synthetic_records(tmp_path):
    """write_columns_as_bam(l_seq=100): read names, 4-bit bases, qualities and an NM tag around the caller's
    alignment columns (204-byte records that compress about 2 : 1, for decode benchmarks on data shaped
    like real BAMs); the columns decode back unchanged and the file depends on the seed only."""
    import filecmp
    from bamsignals_amd.bamio import BamFile, write_columns_as_bam
    from bamsignals_amd.synth import synth_reads
    cols = synth_reads(60_000, [900_000, 70_000], seed=8, paired=True)
    a, b, c = (str(tmp_path / f"{k}.bam") for k in "abc")
    write_columns_as_bam(a, ["x", "y"], cols, l_seq=100, seed=5)
    write_columns_as_bam(b, ["x", "y"], cols, l_seq=100, seed=5)
    write_columns_as_bam(c, ["x", "y"], cols, l_seq=75, seed=6)
    assert filecmp.cmp(a, b, shallow=False) and not filecmp.cmp(a, c, shallow=False)
    for path, l_seq in ((a, 100), (c, 75)):
        f = BamFile(path)
        got = f.decode(threads=2)
        for k in ("pos", "flag", "mapq", "tlen", "cigar", "cigar_off", "ref_off"):
            assert np.array_equal(got[k], cols[k]), k
        f.close()
        raw = gzip.decompress(open(path, "rb").read())
        assert 1.5 < len(raw) / os.path.getsize(path) < 2.6
        o = 12 + struct.unpack_from("<i", raw, 4)[0]
        for _ in range(2):
            o += 8 + struct.unpack_from("<i", raw, o)[0]
        n = 0
        while o < len(raw) and n < 500:
            bs, rid, pos, l_name, mapq, bin_, n_cig, flag, ls = struct.unpack_from("<iiiBBHHHi", raw, o)
            assert ls == l_seq and l_name == 10 and raw[o + 36:o + 46] == b"q%08x\x00" % n
            sq = raw[o + 46 + 4 * n_cig:o + 46 + 4 * n_cig + (l_seq + 1) // 2]
            ql = raw[o + 46 + 4 * n_cig + (l_seq + 1) // 2:o + 46 + 4 * n_cig + (l_seq + 1) // 2 + l_seq]
            assert all((x >> 4) in (1, 2, 4, 8) for x in sq) and all(2 <= q <= 41 for q in ql)
            assert raw[o + 4 + bs - 4:o + 4 + bs - 1] == b"NMC"
            o += 4 + bs
            n += 1


def _bai_to_csi(bai_bytes, depth=5, bgzf_like=True):
    """A CSI (CSIv1) index equivalent to a BAI: same bins and chunks (for depth 6 every bin moves one level
    down under a new root: BAI level l is CSI(14, 6) level l + 1), loffset of a bin = the linear index's
    entry of the bin's first 16-kbp window (what htslib derives when it loads a BAI)."""
    assert bai_bytes[:4] == b"BAI\x01" and depth in (5, 6)
    n_ref, = struct.unpack_from("<i", bai_bytes, 4)
    o = 8
    out = [b"CSI\x01", struct.pack("<iii", 14, depth, 0), struct.pack("<i", n_ref)]
    t5 = [((1 << 3 * l) - 1) // 7 for l in range(7)]
    for _ in range(n_ref):
        n_bin, = struct.unpack_from("<i", bai_bytes, o); o += 4
        bins = []
        for _ in range(n_bin):
            b, n_chunk = struct.unpack_from("<Ii", bai_bytes, o); o += 8
            chunks = bai_bytes[o:o + 16 * n_chunk]; o += 16 * n_chunk
            bins.append((b, n_chunk, chunks))
        n_intv, = struct.unpack_from("<i", bai_bytes, o); o += 4
        linear = struct.unpack_from("<%dQ" % n_intv, bai_bytes, o); o += 8 * n_intv
        recs = []
        for b, n_chunk, chunks in bins:
            if b == 37450:                      # htslib's metadata pseudo-bin: moves to the new scheme's pseudo-bin
                nb, loff = ((1 << 3 * (depth + 1)) - 1) // 7 + 1, 0
            else:
                lvl = max(l for l in range(6) if t5[l] <= b)
                first_window = (b - t5[lvl]) << (3 * (5 - lvl))
                loff = linear[first_window] if first_window < n_intv else (linear[-1] if n_intv else 0)
                nb = b if depth == 5 else t5[lvl + 1] + (b - t5[lvl])
            recs.append(struct.pack("<IQi", nb, loff, n_chunk) + chunks)
        out.append(struct.pack("<i", len(recs)) + b"".join(recs))
    raw = b"".join(out)
    return gzip.compress(raw) if bgzf_like else raw


@pytest.mark.parametrize("depth", [5, 6])
def test_csi_index_is_read_and_queried(tmp_path, fixture_reads, depth):
    """htslib's bam_index_load (ref: src/bamsignals.cpp:207) accepts a CSI index as well; here a BAM that has
    only a .csi opens, decodes whole, and answers region queries exactly as through its BAI (same bins, same
    chunks; min_shift 14 with depth 5 and, one level deeper, depth 6)."""
    import shutil
    from bamsignals_amd import _lib
    from bamsignals_amd.bamio import BamFile
    p = tmp_path / "c.bam"
    shutil.copy(BAM, p)
    with pytest.raises(_lib.BsigError, match="BAM indexing file is not available"):
        BamFile(str(p))
    # a file of that name that is not a CSI index (BGZF-compressed or not) is no index
    (tmp_path / "c.bam.csi").write_bytes(open(BAM, "rb").read()[:200])
    with pytest.raises(_lib.BsigError, match="BAM indexing file is not available"):
        BamFile(str(p))
    (tmp_path / "c.bam.csi").write_bytes(b"not an index")
    with pytest.raises(_lib.BsigError, match="BAM indexing file is not available"):
        BamFile(str(p))
    (tmp_path / "c.bam.csi").write_bytes(_bai_to_csi(open(BAM + ".bai", "rb").read(), depth))
    b = BamFile(str(p))
    _cols_equal(b.decode(threads=2), fixture_reads)
    ref = BamFile(BAM)
    rng = np.random.default_rng(17 + depth)
    for trial in range(12):
        n = int(rng.integers(1, 9))
        rid = rng.integers(0, 3, n).astype(np.int32)
        beg = rng.integers(0, 9000, n).astype(np.int64)
        end = beg + rng.integers(1, 3000, n)
        got, want = b.decode(rid=rid, beg=beg, end=end), ref.decode(rid=rid, beg=beg, end=end)
        for k in ("pos", "flag", "mapq", "tlen", "ref_off", "cigar_off", "cigar"):
            assert np.array_equal(got[k], want[k]), (trial, k)
        assert len(got["pos"]) > 0
    b.close(); ref.close()


def test_a_csi_index_is_preferred_to_a_bai(tmp_path, fixture_reads):
    """htslib's index search looks for <bam>.csi (and <stem>.csi) before the .bai files (ref: bam_index_load,
    src/bamsignals.cpp:207): with both present the CSI is the index -- a damaged BAI next to it is never read --
    and without the CSI the damaged BAI is reported."""
    import shutil
    from bamsignals_amd import _lib
    from bamsignals_amd.bamio import BamFile
    p = tmp_path / "both.bam"
    shutil.copy(BAM, p)
    (tmp_path / "both.bam.bai").write_bytes(b"BAI\x01" + b"\xff" * 40)
    with pytest.raises(_lib.BsigError):
        BamFile(str(p))
    # the first index file that exists IS the index (htslib: a .csi that does not load fails bam_index_load, the
    # reference then reports "not available", ref: src/bamsignals.cpp:207-210): a good .bai behind it is not consulted
    shutil.copy(BAM + ".bai", tmp_path / "both.bam.bai")
    (tmp_path / "both.csi").write_bytes(b"CSI\x01 truncated")
    with pytest.raises(_lib.BsigError, match="BAM indexing file is not available"):
        BamFile(str(p))
    (tmp_path / "both.bam.bai").write_bytes(b"BAI\x01" + b"\xff" * 40)
    (tmp_path / "both.csi").write_bytes(_bai_to_csi(open(BAM + ".bai", "rb").read(), 5))      # <stem>.csi
    b = BamFile(str(p))
    got = b.decode(rid=np.asarray([1], np.int32), beg=np.asarray([100], np.int64), end=np.asarray([4000], np.int64))
    ref = BamFile(BAM)
    want = ref.decode(rid=np.asarray([1], np.int32), beg=np.asarray([100], np.int64), end=np.asarray([4000], np.int64))
    assert len(got["pos"]) > 0 and np.array_equal(got["pos"], want["pos"])
    b.close(); ref.close()


def test_csi_index_of_a_multibin_bam(tmp_path):
    """The same on a BAM whose index spans many windows and every bin level (this repo's writer)."""
    from bamsignals_amd.bamio import BamFile, write_columns_as_bam
    cols = _synth(80_000, seed=8)
    names = ["r%d" % i for i in range(len(cols["ref_len"]))]
    p = str(tmp_path / "m.bam")
    write_columns_as_bam(p, names, cols)
    ref = BamFile(p)
    q = str(tmp_path / "only_csi.bam")
    os.link(p, q)
    rng = np.random.default_rng(3)
    for depth in (5, 6):
        open(q + ".csi", "wb").write(_bai_to_csi(open(p + ".bai", "rb").read(), depth))
        b = BamFile(q)
        for trial in range(10):
            n = int(rng.integers(1, 20))
            rid = rng.integers(0, len(names), n).astype(np.int32)
            beg = (rng.random(n) * cols["ref_len"][rid]).astype(np.int64)
            end = beg + rng.integers(1, 40_000, n)
            got, want = b.decode(rid=rid, beg=beg, end=end), ref.decode(rid=rid, beg=beg, end=end)
            for k in ("pos", "flag", "mapq", "tlen", "ref_off"):
                assert np.array_equal(got[k], want[k]), (depth, trial, k)
        b.close()
        os.remove(q + ".csi")
    ref.close()


def test_the_streams_header_walk_gives_the_same_block_table(tmp_path, monkeypatch):
    """Large files are streamed into HBM in chunks and the BGZF block table is read off the chunks on their way
    (devdecode.hip: RawStream::Walk): a block's header or trailer may straddle a chunk border, a chunk may be smaller
    than a block, larger than the file, or end exactly at a block's end.  Host only: the walk over the fixture and a
    synthetic file under chunk sizes from 1 byte up must give the table of the mapped walk; files it cannot take
    (cut inside a block, odd extra fields) are declined, never mis-tabulated."""
    import ctypes as C
    from bamsignals_amd import _lib
    from bamsignals_amd.bamio import write_columns_as_bam
    lib = _lib.load()
    lib.bsig_debug_stream_walk.argtypes = [C.c_char_p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_uint64)]

    def mapped(path):
        monkeypatch.setenv("BAMSIGNALS_SCAN", "mmap")
        n, h = C.c_int64(), C.c_uint64()
        _lib.check(lib.bsig_debug_block_table(path.encode(), C.byref(n), C.byref(h)))
        return n.value, h.value

    def streamed(path, chunk):
        n, h = C.c_int64(), C.c_uint64()
        _lib.check(lib.bsig_debug_stream_walk(path.encode(), chunk, C.byref(n), C.byref(h)))
        return n.value, h.value

    want = mapped(BAM)
    size = os.path.getsize(BAM)
    first = struct.unpack_from("<H", open(BAM, "rb").read(18), 16)[0] + 1          # the first block's size
    for chunk in (1, 2, 3, 7, 17, 18, 19, 25, 26, 27, 64, 1000, 4096, first - 1, first, first + 1, 2 * first, 65536, 65537,
                  100_000, size - 1, size, size + 1, 1 << 30):
        assert streamed(BAM, chunk) == want, chunk
    cols = _synth(600_000, seed=11)
    p = str(tmp_path / "syn.bam")
    write_columns_as_bam(p, ["a", "b", "c"], cols)
    want = mapped(p)
    assert want[0] > 300
    rng = np.random.default_rng(5)
    for chunk in [5, 333, 8192, 65536, 1 << 20, 32 << 20] + [int(x) for x in rng.integers(20, 200_000, 25)]:
        assert streamed(p, chunk) == want, chunk
    raw = open(p, "rb").read()
    q = str(tmp_path / "cut.bam")
    open(q, "wb").write(raw[: len(raw) // 2 + 1234])                                # the file ends inside a block
    for chunk in (100, 65536, 1 << 30):
        with pytest.raises(_lib.BsigError, match="declines"):
            streamed(q, chunk)
    bad = bytearray(raw)
    bad[3] &= ~4 & 0xFF                                                             # no extra field: not a BGZF block
    open(q, "wb").write(bytes(bad))
    with pytest.raises(_lib.BsigError, match="declines"):
        streamed(q, 4096)
