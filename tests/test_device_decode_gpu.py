"""GPU: the device-side BAM record decode (csrc/devdecode.hip) against the CPU decode stage and the
C oracle.  The decode replaces what the reference gets record by record from htslib's
bam_itr_next (ref: src/bamsignals.cpp:271); the only htslib-written bytes available are the
reference's fixture BAM, everything else is written by this repo's writer or built by hand from
the SAM/BAM specification (marked "spec-derived")."""
import gzip
import os
import struct
import zlib

import numpy as np
import pytest

from conftest import GOLDEN, layout_info

pytestmark = pytest.mark.gpu

BAM = os.path.join(GOLDEN, "randomBam.bam")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EOF_BLOCK = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


@pytest.fixture(scope="module")
def ctx():
    from bamsignals_amd.device import Context
    c = Context(0)
    yield c
    c.close()



def _bgzf(data, sizes, levels=((1, 0),)):
    """BGZF-compress `data`, cutting it into blocks of the given sizes (cycled); `levels` = (zlib
    level, strategy) pairs, cycled too (level 0: stored blocks; strategy 4: fixed Huffman codes)."""
    out, i, k = b"", 0, 0
    while i < len(data):
        n = sizes[k % len(sizes)]
        chunk = data[i:i + n]
        lv, st = levels[k % len(levels)]
        co = zlib.compressobj(lv, zlib.DEFLATED, -15, 8, st)
        dd = co.compress(chunk) + co.flush()
        out += (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(dd) + 25) + dd
                + struct.pack("<II", zlib.crc32(chunk), len(chunk)))
        i += n
        k += 1
    return out + EOF_BLOCK


def _first_record(stream):
    """offset of the first record of an uncompressed BAM stream"""
    o = 8 + struct.unpack_from("<i", stream, 4)[0]
    n_ref = struct.unpack_from("<i", stream, o)[0]
    o += 4
    for _ in range(n_ref):
        o += 8 + struct.unpack_from("<i", stream, o)[0]
    return o


def _boundary_after(stream, target):
    """first record boundary at or behind `target`"""
    o = _first_record(stream)
    while o < target:
        o += 4 + struct.unpack_from("<i", stream, o)[0]
    return o


def _empty_bai(path, n_ref):
    with open(path, "wb") as f:
        f.write(b"BAI\x01" + struct.pack("<i", n_ref) + struct.pack("<ii", 0, 0) * n_ref + struct.pack("<Q", 0))


def _results(ctx, reads, ref_len, seed=3):
    """profile / coverage / count over ranges covering every reference, as one tuple of arrays"""
    from bamsignals_amd import _lib
    from bamsignals_amd.device import Plan, make_params
    from bamsignals_amd.synth import synth_ranges, tile_ranges
    out = []
    rg = synth_ranges(300, 900, ref_len, seed=seed, jitter=600)
    tiles = tile_ranges(ref_len, 5000)
    for mode, args, r in ((_lib.MODE_PROFILE, dict(binsize=1, ss=True, shift=15), rg),
                          (_lib.MODE_PROFILE, dict(binsize=7, requiredF=66, tlen_filter=(40, 600), pe_mid=True), rg),
                          (_lib.MODE_COVERAGE, dict(), tiles), (_lib.MODE_COUNT, dict(binsize=-1, ss=True), tiles)):
        p = Plan(ctx, reads, r["rid"], r["loc"], r["len"], r["strand"], make_params(mode, **args))
        out.append(p.run_host().copy())
        p.close()
    return out


def _both_ways(ctx, path, monkeypatch, expect_device=True):
    """decode `path` on the device and on the CPU; the resident reads must be indistinguishable"""
    from bamsignals_amd import _lib
    from bamsignals_amd.bamio import BamFile
    from bamsignals_amd.device import Reads
    bam = BamFile(path)
    monkeypatch.setenv("BAMSIGNALS_DEVICE_DECODE", "require" if expect_device else "1")
    dev = Reads.from_bam(ctx, bam)
    took_device = Reads.device_decode_timing()["total"] > 0
    assert took_device == expect_device
    monkeypatch.setenv("BAMSIGNALS_DEVICE_DECODE", "0")
    cpu = Reads.from_bam(ctx, bam)
    assert Reads.device_decode_timing()["total"] == 0
    assert layout_info(dev) == layout_info(cpu)
    for a, b in zip(_results(ctx, dev, bam.ref_len), _results(ctx, cpu, bam.ref_len)):
        assert np.array_equal(a, b)
    if not expect_device:
        monkeypatch.setenv("BAMSIGNALS_DEVICE_DECODE", "require")
        with pytest.raises(_lib.BsigError, match="CPU decode path"):
            Reads.from_bam(ctx, bam)
    cpu.close()
    return bam, dev


def test_fixture_bam(ctx, monkeypatch, fixture_reads):
    """the reference's own (htslib-written) BAM: device decode == CPU decode == the fixture columns"""
    from oracle import oracle_c
    bam, dev = _both_ways(ctx, BAM, monkeypatch)
    assert dev.n_reads == 99000
    from bamsignals_amd.device import Context
    ctx2 = Context(0)                                  # a second context: the multi-GPU replication path
    twin = dev.clone(ctx2)
    assert layout_info(twin) == layout_info(dev)
    for a, b in zip(_results(ctx2, twin, bam.ref_len), _results(ctx, dev, bam.ref_len)):
        assert np.array_equal(a, b)
    twin.close()
    ctx2.close()
    fx = fixture_reads
    orc = oracle_c.OracleReads(fx["ref_off"], fx["bam_pos"], fx["bam_end"], fx["bam_flag"], fx["bam_mapq"], fx["bam_tlen"])
    from bamsignals_amd import _lib
    from bamsignals_amd.device import Plan, make_params
    from bamsignals_amd.synth import tile_ranges
    tiles = tile_ranges(bam.ref_len, 777, strand=-1)
    p = Plan(ctx, dev, tiles["rid"], tiles["loc"], tiles["len"], tiles["strand"], make_params(_lib.MODE_PROFILE, binsize=1, ss=True))
    want, _ = oracle_c.pileup_core(orc, tiles, binsize=1, ss=True)
    assert np.array_equal(p.run_host(), want)
    dev.close()


@pytest.mark.parametrize("batch_blocks", ["", "1", "3"])
def test_synthetic_bam_gapped_cigars_empty_references(ctx, tmp_path, monkeypatch, batch_blocks):
    """our writer's BAM: D/N/S/I CIGARs, duplicates, references without reads in front of, between
    and behind the populated ones; also with the stream cut into many tiny copy batches"""
    from bamsignals_amd import write_columns_as_bam
    from bamsignals_amd.synth import synth_reads
    from oracle import oracle_c
    cols = synth_reads(200_000, [600_000, 90_000, 300_000], seed=11, paired=True)
    # references 0, 2 and 5 are empty
    ref_len = np.asarray([5000, 600_000, 1234, 90_000, 300_000, 77], dtype=np.int32)
    ro = cols["ref_off"]
    cols2 = dict(cols, ref_len=ref_len, ref_off=np.asarray([0, 0, ro[1], ro[1], ro[2], ro[3], ro[3]], dtype=np.int64))
    path = str(tmp_path / "syn.bam")
    write_columns_as_bam(path, ["e0", "a", "e2", "b", "c", "e5"], cols2)
    if batch_blocks:
        monkeypatch.setenv("BAMSIGNALS_BATCH_BLOCKS", batch_blocks)
    bam, dev = _both_ways(ctx, path, monkeypatch)
    assert dev.n_reads == 200_000
    orc = oracle_c.OracleReads(cols2["ref_off"], cols["pos"], cols["end"], cols["flag"], cols["mapq"], cols["tlen"])
    from bamsignals_amd import _lib
    from bamsignals_amd.device import Plan, make_params
    from bamsignals_amd.synth import synth_ranges
    rg = synth_ranges(500, 2000, ref_len, seed=9, jitter=1500)
    a = dict(requiredF=66, tlen_filter=(0, 1000), tspan=True)
    p = Plan(ctx, dev, rg["rid"], rg["loc"], rg["len"], rg["strand"], make_params(_lib.MODE_COVERAGE, **a))
    want, _ = oracle_c.coverage_core(orc, rg, **a)
    assert np.array_equal(p.run_host(), want)
    dev.close()


def test_unplaced_and_unmapped_records(ctx, tmp_path, monkeypatch):
    """spec-derived: records with refID -1 are skipped, 0x4 records placed at their mate count as
    1-bp reads (bam_endpos), an empty CIGAR is a 1-bp read"""
    from bamsignals_amd.bamio import writeSamAsBamAndIndex
    sam = tmp_path / "u.sam"
    lines = ["@HD\tVN:1.0\tSO:coordinate", "@SQ\tSN:c1\tLN:5000", "@SQ\tSN:c2\tLN:3000"]
    lines += ["r%d\t0\tc1\t%d\t30\t50M\t*\t0\t0\t*\t*" % (i, 10 + 7 * i) for i in range(300)]
    lines += ["m%d\t69\tc1\t%d\t0\t*\t=\t%d\t0\t*\t*" % (i, 2200 + i, 2200 + i) for i in range(40)]     # unmapped, placed
    lines += ["s%d\t16\tc2\t%d\t11\t20M5D20M\t*\t0\t0\t*\t*" % (i, 5 + 3 * i) for i in range(500)]
    lines += ["x%d\t4\t*\t0\t0\t*\t*\t0\t0\t*\t*" % i for i in range(25)]                              # unplaced
    sam.write_text("\n".join(lines) + "\n")
    path = str(tmp_path / "u.bam")
    writeSamAsBamAndIndex(str(sam), path)
    bam, dev = _both_ways(ctx, path, monkeypatch)
    assert dev.n_reads == 840
    from bamsignals_amd import _lib
    from bamsignals_amd.device import Plan, make_params
    cov = Plan(ctx, dev, [0], [2150], [200], [1], make_params(_lib.MODE_COVERAGE)).run_host()
    want = np.zeros(200, np.int32)
    # the 50M reads starting at 10 + 7i - 1 (0-based) that reach into [2150, 2350)
    for i in range(300):
        s = 9 + 7 * i
        want[max(s - 2150, 0):max(s + 50 - 2150, 0)] += 1
    for i in range(40):
        want[2200 + i - 1 - 2150] += 1                       # 1-bp reads
    assert np.array_equal(cov, want)
    dev.close()


def test_records_crossing_block_borders_and_oversized_records(ctx, tmp_path, monkeypatch, fixture_reads):
    """records crossing BGZF block borders (htsjdk-style): the lanes propose their starts and the
    host proves them; a record larger than a block, followed by more records or last in the file;
    a CG-tag CIGAR (real operations in the CG:B,I tag).  Same results every time."""
    stream = gzip.decompress(open(BAM, "rb").read())
    for k, sizes in enumerate(([4000, 9001, 517, 65000], [65536], [33, 70, 1000])):
        p = tmp_path / ("straddle%d.bam" % k)
        p.write_bytes(_bgzf(stream if k < 2 else stream[:_boundary_after(stream, 400_000)], sizes))
        _empty_bai(str(p) + ".bai", 3)
        _, dev = _both_ways(ctx, str(p), monkeypatch, expect_device=True)
        if k < 2:
            assert dev.n_reads == 99000
        dev.close()

    # spec-derived: one 150-kB record (l_seq 100,000) between ordinary ones
    text = b"@SQ\tSN:c\tLN:200000\n"
    hdr = b"BAM\x01" + struct.pack("<i", len(text)) + text + struct.pack("<i", 1) + struct.pack("<i", 2) + b"c\x00" + struct.pack("<i", 200000)

    def rec(pos, lseq, cigar, aux=b""):
        name = b"q\x00"
        body = struct.pack("<iiBBHHHiiii", 0, pos, len(name), 40, 4681, len(cigar), 0, lseq, -1, -1, 0) + name
        body += struct.pack("<%dI" % len(cigar), *cigar) + bytes((lseq + 1) // 2) + b"\xff" * lseq + aux
        return struct.pack("<i", len(body)) + body
    small = [rec(100 + 3 * i, 0, [(30 << 4) | 0]) for i in range(50)]
    big = rec(400, 100_000, [(100_000 << 4) | 0])
    tail = [rec(500 + i, 0, [(10 << 4) | 0, (5 << 4) | 2, (10 << 4) | 0]) for i in range(50)]
    first = hdr + b"".join(small)
    data = first + big + b"".join(tail)
    p = tmp_path / "big.bam"
    p.write_bytes(_bgzf(first, [len(first)])[:-len(EOF_BLOCK)] + _bgzf(big + b"".join(tail), [65000]))
    assert gzip.decompress(p.read_bytes()) == data
    _empty_bai(str(p) + ".bai", 1)
    _, dev = _both_ways(ctx, str(p), monkeypatch, expect_device=True)
    assert dev.n_reads == 101
    dev.close()

    # the same oversized record as the LAST record of the file: every later block start lies
    # inside it, which the device path recognises and accepts
    p = tmp_path / "biglast.bam"
    p.write_bytes(_bgzf(first, [len(first)])[:-len(EOF_BLOCK)] + _bgzf(big, [65000]))
    _empty_bai(str(p) + ".bai", 1)
    _, dev = _both_ways(ctx, str(p), monkeypatch, expect_device=True)
    assert dev.n_reads == 51
    dev.close()

    # CG-tag CIGAR (SAM spec 4.2.2)
    ops = [(1 << 4) | 0, (1 << 4) | 2] * 40000
    lseq = 40000
    aux = b"CGBI" + struct.pack("<I", len(ops)) + struct.pack("<%dI" % len(ops), *ops)
    cg = rec(1000, lseq, [lseq << 4 | 4, 80000 << 4 | 3], aux)
    p = tmp_path / "cg.bam"
    p.write_bytes(_bgzf(first, [len(first)])[:-len(EOF_BLOCK)] + _bgzf(cg, [60000]))
    _empty_bai(str(p) + ".bai", 1)
    _, dev = _both_ways(ctx, str(p), monkeypatch, expect_device=True)
    assert dev.info()["class_maxspan"][3] == 80000
    dev.close()


@pytest.mark.parametrize("inflate", ["cpu", "gpu", "gpu-two-views"])
def test_stream_in_chunks(ctx, tmp_path, monkeypatch, inflate):
    """the uncompressed stream passes through HBM in chunks (8 GiB by default; 1 MiB here): the record
    cut off by a chunk border is carried in front of the next chunk.  With the GPU inflating, the
    compressed bytes of pass j + 1 are packed and copied by the helper thread into the other buffer
    while pass j is worked on (every file here takes at least four passes); with BAMSIGNALS_TWO_VIEWS=1
    pass j + 1 is also inflated, into a second view, while pass j is walked (the carried record then
    moves from one view into the other)."""
    from bamsignals_amd import write_columns_as_bam
    from bamsignals_amd.synth import synth_reads
    monkeypatch.setenv("BAMSIGNALS_DEVICE_DECODE_CHUNK_MB", "1")
    if inflate == "gpu-two-views":
        monkeypatch.setenv("BAMSIGNALS_TWO_VIEWS", "1")
        inflate = "gpu"
    monkeypatch.setenv("BAMSIGNALS_INFLATE", inflate)
    monkeypatch.setenv("BAMSIGNALS_BATCH_BLOCKS", "5")
    stream = gzip.decompress(open(BAM, "rb").read())
    assert len(stream) > (4 << 20)
    _, dev = _both_ways(ctx, BAM, monkeypatch)                          # htslib's blocks: borders are record starts
    assert dev.n_reads == 99000
    dev.close()
    for k, sizes in enumerate(([4000, 9001, 517, 65000], [65536])):    # records across block AND chunk borders
        p = tmp_path / ("straddle%d.bam" % k)
        p.write_bytes(_bgzf(stream, sizes))
        _empty_bai(str(p) + ".bai", 3)
        _, dev = _both_ways(ctx, str(p), monkeypatch)
        assert dev.n_reads == 99000
        dev.close()
    # 300,000 reads, 16 MB of stream, three references
    cols = synth_reads(300_000, [900_000, 70_000, 400_000], seed=4, paired=True)
    path = str(tmp_path / "syn.bam")
    write_columns_as_bam(path, ["a", "b", "c"], cols)
    _, dev = _both_ways(ctx, path, monkeypatch)
    assert dev.n_reads == 300_000
    dev.close()
    # a 150-kB record carried over a chunk border (spec-derived)
    text = b"@SQ\tSN:c\tLN:3000000\n"
    hdr = b"BAM\x01" + struct.pack("<i", len(text)) + text + struct.pack("<i", 1) + struct.pack("<i", 2) + b"c\x00" + struct.pack("<i", 3000000)

    def rec(pos, lseq):
        name = b"q\x00"
        body = struct.pack("<iiBBHHHiiii", 0, pos, len(name), 40, 4681, 1, 0, lseq, -1, -1, 0) + name
        body += struct.pack("<I", (max(lseq, 30) << 4) | 0) + bytes((lseq + 1) // 2) + b"\xff" * lseq
        return struct.pack("<i", len(body)) + body
    recs = [rec(10 + i, 20_000 if i % 7 else 100_000) for i in range(120)]        # 30 kB and 150 kB records
    data = hdr + b"".join(recs)
    p = tmp_path / "bigs.bam"
    p.write_bytes(_bgzf(data, [65536]))
    _empty_bai(str(p) + ".bai", 1)
    _, dev = _both_ways(ctx, str(p), monkeypatch)
    assert dev.n_reads == 120
    dev.close()


def test_head_of_the_file_decoded_while_the_rest_is_tabulated(ctx, tmp_path, monkeypatch, capfd):
    """Large files are decoded in two steps (GPU inflate): the head as a share of its own while the host walks the
    rest of the block table, joined like the shares of several GPUs.  Forced here on small files: htslib's
    fixture, files whose records cross block borders (the second share must find where the chain stands) and a
    synthetic one with several references, with the stream also cut into many passes."""
    from bamsignals_amd import write_columns_as_bam
    from bamsignals_amd.synth import synth_reads
    monkeypatch.setenv("BAMSIGNALS_INFLATE", "gpu")
    monkeypatch.setenv("BAMSIGNALS_SCAN_SEGMENT_KB", "16")
    monkeypatch.setenv("BAMSIGNALS_SCAN_HEAD_KB", "400")
    monkeypatch.setenv("BAMSIGNALS_TWO_STEP_MIN_BLOCKS", "1")
    monkeypatch.setenv("BSIG_DIAG_DECODE", "1")
    stream = gzip.decompress(open(BAM, "rb").read())
    files = [BAM]
    for k, sizes in enumerate(([4000, 9001, 517, 65000], [65536], [30000])):
        p = tmp_path / ("straddle%d.bam" % k)
        p.write_bytes(_bgzf(stream, sizes))
        _empty_bai(str(p) + ".bai", 3)
        files.append(str(p))
    for chunk in (None, "1"):
        if chunk:
            monkeypatch.setenv("BAMSIGNALS_DEVICE_DECODE_CHUNK_MB", chunk)
        for f in files:
            capfd.readouterr()
            _, dev = _both_ways(ctx, f, monkeypatch)
            assert dev.n_reads == 99000
            assert "decode_share (head)" in capfd.readouterr().err, f      # the two-step route was taken
            dev.close()
    monkeypatch.delenv("BAMSIGNALS_DEVICE_DECODE_CHUNK_MB")
    cols = synth_reads(1_500_000, [900_000, 70_000, 0, 400_000], seed=14, paired=True)
    path = str(tmp_path / "syn.bam")
    write_columns_as_bam(path, ["a", "b", "empty", "c"], cols)
    for head_kb in ("300", "700", "3000"):
        monkeypatch.setenv("BAMSIGNALS_SCAN_HEAD_KB", head_kb)
        capfd.readouterr()
        _, dev = _both_ways(ctx, path, monkeypatch)
        assert dev.n_reads == 1_500_000
        assert "decode_share (head)" in capfd.readouterr().err, head_kb
        dev.close()


def test_large_files_are_streamed_into_hbm_and_decoded_round_by_round(ctx, tmp_path, monkeypatch, capfd):
    """Large files (GPU inflate) travel to HBM ONCE, front to back, into a device buffer at their file offsets; the
    block table is read off the chunks on their way, and every round of inflate lanes is decoded as a share of its own
    as soon as the stream has passed it (a share is what has been tabulated by then, a quarter of a round to a round)
    -- the shares joined like the shares of several GPUs.  Forced here on small
    files (the route starts at 256 MB) with rounds of a few blocks and chunks from smaller than a block (a block then
    spans several chunks) to larger than the file: htslib's fixture, files whose records cross block borders, a
    synthetic one with several references and an empty one between them; damaged files still get the CPU path's
    messages."""
    import re
    from bamsignals_amd import _lib
    from bamsignals_amd import write_columns_as_bam
    from bamsignals_amd.bamio import BamFile
    from bamsignals_amd.device import Reads
    from bamsignals_amd.synth import synth_reads
    monkeypatch.setenv("BAMSIGNALS_STREAM_MIN_MB", "0")
    monkeypatch.setenv("BAMSIGNALS_INFLATE", "gpu")
    monkeypatch.setenv("BSIG_DIAG_DECODE", "1")
    stream = gzip.decompress(open(BAM, "rb").read())
    files = [(BAM, 99000)]
    for k, sizes in enumerate(([4000, 9001, 517, 65000], [65536], [30000])):
        p = tmp_path / ("straddle%d.bam" % k)
        p.write_bytes(_bgzf(stream, sizes))
        _empty_bai(str(p) + ".bai", 3)
        files.append((str(p), 99000))
    cols = synth_reads(1_500_000, [900_000, 70_000, 0, 400_000], seed=16, paired=True)
    path = str(tmp_path / "syn.bam")
    write_columns_as_bam(path, ["a", "b", "empty", "c"], cols)                # ~1,200 blocks
    files.append((path, 1_500_000))
    for f, n_reads in files:
        # ... and the shares' extractions write into ONE set of columns sized by the first share's reads per byte
        # (large files by themselves; here by request): with head-room, and too small -- the shares that do not fit
        # take pieces of their own and everything is joined as before
        for rnd, chunk_kb, arena in ((7, "3", "1.05"), (40, "100", "0.5"), (64, "5000", None), (100000, None, "1.05")):
            if arena:
                monkeypatch.setenv("BAMSIGNALS_COLUMN_ARENA", arena)
            else:
                monkeypatch.delenv("BAMSIGNALS_COLUMN_ARENA", raising=False)
            monkeypatch.setenv("BAMSIGNALS_INFLATE_ROUND_BLOCKS", str(rnd))
            if chunk_kb:
                monkeypatch.setenv("BAMSIGNALS_STREAM_CHUNK_KB", chunk_kb)
            else:
                monkeypatch.delenv("BAMSIGNALS_STREAM_CHUNK_KB", raising=False)
            if f == path and rnd == 7:
                continue                                                      # (170 shares of seven blocks: slow, nothing new)
            capfd.readouterr()
            _, dev = _both_ways(ctx, f, monkeypatch)
            assert dev.n_reads == n_reads
            dev.close()
            err = capfd.readouterr().err
            assert "streamed: all shares" in err and "the ordinary route" not in err, (f, rnd, chunk_kb, err[-600:])
            if rnd < 100:
                assert ("nothing to join" in err) == (arena == "1.05"), (f, rnd, arena)
            shares = [(int(a), int(b)) for a, b in re.findall(r"streamed: share of blocks \[(\d+), (\d+)\) done", err)]
            assert shares[0][0] == 0 and all(x[1] == y[0] for x, y in zip(shares, shares[1:])), shares
            own = rnd - 4 if rnd > 8 else rnd                                 # (a share sees four blocks beyond its own: one round in all)
            assert all(0 < b - a <= own for a, b in shares[:-1]) and 0 < shares[-1][1] - shares[-1][0] <= own + 4, (rnd, shares)
    # the engine choice still holds: a few badly compressible blocks are the CPU pool's, whatever the route
    monkeypatch.delenv("BAMSIGNALS_COLUMN_ARENA", raising=False)
    monkeypatch.delenv("BAMSIGNALS_INFLATE_ROUND_BLOCKS")
    monkeypatch.delenv("BAMSIGNALS_INFLATE")
    capfd.readouterr()
    _, dev = _both_ways(ctx, BAM, monkeypatch)
    dev.close()
    assert "streamed: all shares" not in capfd.readouterr().err
    # damaged files: the stream (or a share) declines, the ordinary route and then the CPU path name the problem
    monkeypatch.setenv("BAMSIGNALS_INFLATE", "gpu")
    monkeypatch.setenv("BAMSIGNALS_INFLATE_ROUND_BLOCKS", "5")
    monkeypatch.setenv("BAMSIGNALS_DEVICE_DECODE", "1")
    p = tmp_path / "trunc.bam"
    p.write_bytes(_bgzf(stream[:-7], [60000]))
    _empty_bai(str(p) + ".bai", 3)
    with pytest.raises(_lib.BsigError, match="truncated"):
        Reads.from_bam(ctx, BamFile(str(p)))
    raw = bytearray(open(BAM, "rb").read())
    bsize = struct.unpack_from("<H", raw, 16)[0] + 1
    b2 = struct.unpack_from("<H", raw, bsize + 16)[0] + 1
    raw[bsize + b2 - 8] ^= 0x5A
    p = tmp_path / "crc.bam"
    p.write_bytes(bytes(raw))
    _empty_bai(str(p) + ".bai", 3)
    with pytest.raises(_lib.BsigError, match="CRC"):
        Reads.from_bam(ctx, BamFile(str(p)))
    cutoff = tmp_path / "cutoff.bam"                                           # the file ends inside a block
    cutoff.write_bytes(open(BAM, "rb").read()[:-40])
    _empty_bai(str(cutoff) + ".bai", 3)
    with pytest.raises(_lib.BsigError):
        Reads.from_bam(ctx, BamFile(str(cutoff)))


@pytest.mark.timeout(300)
def test_a_300_mb_file_takes_the_streamed_route_by_itself(ctx, tmp_path, monkeypatch, capfd):
    """No knobs: 5e7 bare reads are a 300-MB BAM of 33,000 blocks -- beyond the 256 MB at which whole-file decodes are
    streamed, and many enough blocks for the GPU to inflate them.  Same resident reads, same results as the CPU decode."""
    from bamsignals_amd import write_columns_as_bam
    from bamsignals_amd.synth import synth_reads
    for k in ("BAMSIGNALS_STREAM", "BAMSIGNALS_STREAM_MIN_MB", "BAMSIGNALS_INFLATE", "BAMSIGNALS_INFLATE_ROUND_BLOCKS", "BAMSIGNALS_STREAM_CHUNK_KB"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("BSIG_DIAG_DECODE", "1")
    ref_len = [120_000_000, 80_000_000, 0, 50_000_000]
    cols = synth_reads(50_000_000, ref_len, seed=21, paired=True)
    path = str(tmp_path / "mid.bam")
    write_columns_as_bam(path, ["a", "b", "empty", "c"], cols, level=1)
    del cols
    assert os.path.getsize(path) > 256 << 20
    capfd.readouterr()
    _, dev = _both_ways(ctx, path, monkeypatch)
    assert dev.n_reads == 50_000_000
    dev.close()
    err = capfd.readouterr().err
    assert "streamed: all shares" in err and "the ordinary route" not in err, err[-600:]
    # switched off, the same file takes the ordinary route to the same reads
    monkeypatch.setenv("BAMSIGNALS_STREAM", "0")
    _, dev = _both_ways(ctx, path, monkeypatch)
    assert dev.n_reads == 50_000_000
    dev.close()
    assert "streamed:" not in capfd.readouterr().err


def test_inflate_kernel_keeps_its_state_in_registers(ctx):
    """k_inflate's per-lane state -- bit buffer, code counts, construction slots, deferred match words -- lives in
    registers and LDS.  Twice in round 3 a harmless-looking change (a select chain over eight words, a local array
    indexed by a loop) made the compiler move it to scratch memory: a trip to memory per access.  And the occupancy
    the decode's passes are cut for (seven 32-lane workgroups per CU) depends on the register and LDS budget."""
    import ctypes
    from bamsignals_amd import _lib
    lib = _lib.load()
    fn = lib.bsig_debug_inflate_attrs
    fn.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int64)]
    regs, scratch, rnd = ctypes.c_int(0), ctypes.c_int(-1), ctypes.c_int64(0)
    assert fn(0, ctypes.byref(regs), ctypes.byref(scratch), ctypes.byref(rnd)) == 0
    assert scratch.value == 0, scratch.value
    # round 5: two waves per workgroup (one decodes, one copies): seven workgroups per CU are 14 waves, up to four on a SIMD
    assert 0 < regs.value <= 128, regs.value
    assert rnd.value >= 256 * 7 * 32, rnd.value                    # at least seven workgroups of 32 blocks per CU


@pytest.mark.parametrize("shape", ["bare, level 1", "real-shaped, level 6", "stored blocks"])
def test_inflate_kernel_alone_in_every_form(ctx, tmp_path, monkeypatch, shape):
    """k_inflate by itself (bsig_debug_inflate_bench: the compressed bytes resident, every block's CRC32 against its
    BGZF trailer afterwards) with 4, 8, 16, 32 and 64 blocks per workgroup -- two waves each, a decoding lane and a copying
    lane per block, a 12-byte mailbox between them -- and with padded LDS, on bare records (matches at distances below
    64 whose sources reach into the copier's 16-byte accumulator), on real-shaped records at zlib level 6 (six deflate
    blocks per BGZF block: headers met at different times by the lanes of a wave) and on stored blocks (literal tokens
    only).  Also: a launch of fewer blocks than a workgroup holds, and one block."""
    import ctypes
    from bamsignals_amd import _lib, write_columns_as_bam
    from bamsignals_amd.synth import synth_reads
    lib = _lib.load()
    fn = lib.bsig_debug_inflate_bench
    fn.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.POINTER(ctypes.c_double),
                   ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int)]
    cols = synth_reads(300_000 if shape.startswith("bare") else 60_000, [5_000_000, 900_000], seed=77)
    path = str(tmp_path / "k.bam")
    level, l_seq = {"bare, level 1": (1, 0), "real-shaped, level 6": (6, 100), "stored blocks": (0, 50)}[shape]
    write_columns_as_bam(path, ["a", "b"], cols, level=level, l_seq=l_seq, seed=5)

    def run(first, n):
        ms, by, st = (ctypes.c_double * 2)(), (ctypes.c_int64 * 3)(), ctypes.c_int(-1)
        rc = fn(0, path.encode(), first, n, 1, ms, by, ctypes.byref(st))
        return rc, st.value, by[2], by[1]

    rc, st, n_all, out_all = run(0, 1 << 20)
    assert (rc, st) == (0, 0) and n_all >= 40 and out_all > 2_000_000, (rc, st, n_all, out_all)
    for lanes, pad in (("4", "0"), ("8", "0"), ("16", "100"), ("32", "0"), ("32", "288"), ("64", "0")):
        monkeypatch.setenv("BAMSIGNALS_INFLATE_LANES", lanes)
        monkeypatch.setenv("BAMSIGNALS_INFLATE_LDS_PAD", pad)
        for first, n in ((0, 1 << 20), (3, 1), (1, 7), (2, 33)):
            rc, st, nb, _ = run(first, n)
            assert (rc, st) == (0, 0) and nb == min(n, n_all - first), (shape, lanes, pad, first, n, rc, st, nb)


def test_inflate_kernel_on_every_kind_of_content(ctx, tmp_path, monkeypatch):
    """k_inflate by itself on what BAM records never hold (scripts/fuzz_inflate.py's generator, two seeds here; DESIGN.md
    section 4 has the long campaign): runs, periods of 1-40 bytes, copies at chosen distances (1, 7, 8, 9, 15, 16, 17, ... 32,768)
    and lengths (3 ... 258), skewed alphabets, random bytes, zeros, in blocks of 1 ... 65,280 bytes at zlib levels 0 / 1 / 6 / 9
    under all five strategies; every block's CRC32 and length checked on the device."""
    import ctypes
    import importlib.util
    from bamsignals_amd import _lib
    spec = importlib.util.spec_from_file_location("fuzz_inflate", os.path.join(ROOT, "scripts", "fuzz_inflate.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    fn = _lib.load().bsig_debug_inflate_bench
    fn.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.POINTER(ctypes.c_double),
                   ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int)]
    for seed in (5, 6):
        rng = np.random.default_rng(seed)
        path, total, n_blocks = str(tmp_path / ("c%d.bgzf" % seed)), 0, 700
        with open(path, "wb") as fh:
            for _ in range(n_blocks):
                b, n = fz.block(rng)
                fh.write(b)
                total += n
            fh.write(fz.EOF_BLOCK)
        for lanes in ("8", "32", "64"):
            monkeypatch.setenv("BAMSIGNALS_INFLATE_LANES", lanes)
            ms, by, st = (ctypes.c_double * 2)(), (ctypes.c_int64 * 3)(), ctypes.c_int(-1)
            rc = fn(0, path.encode(), 0, n_blocks, 1, ms, by, ctypes.byref(st))
            assert (rc, st.value, by[2], by[1]) == (0, 0, n_blocks, total), (seed, lanes, rc, st.value, by[2], by[1])


def test_passes_are_whole_rounds_of_inflate_lanes(ctx, tmp_path, monkeypatch, capfd):
    """A k_inflate launch lasts one block's latency per round of resident lanes, so every pass of a decode that is
    not its share's last holds a whole number of rounds (57,344 blocks on an MI355X: only the bench's files are
    that large, so the round is set to a few blocks here): first pass one round -- or half the share if that is
    less --, second two, then what the chunk allows; the head share of a two-step decode too."""
    import re
    from bamsignals_amd import write_columns_as_bam
    from bamsignals_amd.synth import synth_reads
    monkeypatch.setenv("BAMSIGNALS_INFLATE", "gpu")
    monkeypatch.setenv("BSIG_DIAG_DECODE", "1")
    cols = synth_reads(2_000_000, [900_000, 70_000, 0, 400_000], seed=15, paired=True)
    path = str(tmp_path / "syn.bam")
    write_columns_as_bam(path, ["a", "b", "empty", "c"], cols)                # ~1,600 blocks
    stream = gzip.decompress(open(BAM, "rb").read())
    straddle = tmp_path / "straddle.bam"
    straddle.write_bytes(_bgzf(stream, [4000, 9001, 517, 65000]))
    _empty_bai(str(straddle) + ".bai", 3)
    for f, n_reads in ((path, 2_000_000), (BAM, 99000), (str(straddle), 99000)):
        for rnd, chunk_mb, two_step in ((7, "1", False), (64, "16", False), (5, "1", True), (100, None, True)):
            monkeypatch.setenv("BAMSIGNALS_INFLATE_ROUND_BLOCKS", str(rnd))
            if chunk_mb:
                monkeypatch.setenv("BAMSIGNALS_DEVICE_DECODE_CHUNK_MB", chunk_mb)
            else:
                monkeypatch.delenv("BAMSIGNALS_DEVICE_DECODE_CHUNK_MB", raising=False)
            if two_step:
                monkeypatch.setenv("BAMSIGNALS_SCAN_SEGMENT_KB", "16")
                monkeypatch.setenv("BAMSIGNALS_SCAN_HEAD_KB", "600")
                monkeypatch.setenv("BAMSIGNALS_TWO_STEP_MIN_BLOCKS", "1")
            else:
                for k in ("BAMSIGNALS_SCAN_SEGMENT_KB", "BAMSIGNALS_SCAN_HEAD_KB", "BAMSIGNALS_TWO_STEP_MIN_BLOCKS"):
                    monkeypatch.delenv(k, raising=False)
            capfd.readouterr()
            _, dev = _both_ways(ctx, f, monkeypatch)
            assert dev.n_reads == n_reads
            dev.close()
            err = capfd.readouterr().err
            # the device decode's launches (the CPU decode of _both_ways prints none), share by share
            shares, cur = [], []
            for line in err.splitlines():
                m = re.match(r"pass (\d+): waited .* for k_inflate of (\d+) blocks", line)
                if m:
                    if int(m.group(1)) == 0 and cur:
                        shares.append(cur); cur = []
                    cur.append(int(m.group(2)))
            if cur:
                shares.append(cur)
            assert shares, err[-400:]
            assert (len(shares) == 2) == (two_step and "decode_share (head)" in err), (f, rnd, shares)
            for sh in shares:
                for n in sh[:-1]:
                    assert n <= rnd or n % rnd == 0, (f, rnd, chunk_mb, two_step, shares)
            if f == path and not two_step:
                assert len(shares[0]) >= 3 and shares[0][0] == rnd and shares[0][1] == 2 * rnd, (rnd, shares)


def test_damaged_files_report_the_cpu_paths_errors(ctx, tmp_path, monkeypatch):
    """a truncated last record and an unsorted file: the device path declines, the CPU path names
    the problem"""
    from bamsignals_amd import _lib
    from bamsignals_amd.bamio import BamFile
    from bamsignals_amd.device import Reads
    stream = gzip.decompress(open(BAM, "rb").read())
    monkeypatch.setenv("BAMSIGNALS_DEVICE_DECODE", "1")
    p = tmp_path / "trunc.bam"
    p.write_bytes(_bgzf(stream[:-7], [60000]))
    _empty_bai(str(p) + ".bai", 3)
    with pytest.raises(_lib.BsigError, match="truncated"):
        Reads.from_bam(ctx, BamFile(str(p)))
    # swap two records: no longer sorted
    o = _first_record(stream)
    l0 = 4 + struct.unpack_from("<i", stream, o)[0]
    l1 = 4 + struct.unpack_from("<i", stream, o + l0)[0]
    l2 = 4 + struct.unpack_from("<i", stream, o + l0 + l1)[0]
    r0, r1, r2 = stream[o:o + l0], stream[o + l0:o + l0 + l1], stream[o + l0 + l1:o + l0 + l1 + l2]
    assert struct.unpack_from("<i", r2, 8)[0] > struct.unpack_from("<i", r0, 8)[0]
    bad = stream[:o] + r2 + r1 + r0 + stream[o + l0 + l1 + l2:]
    p = tmp_path / "unsorted.bam"
    p.write_bytes(_bgzf(bad, [len(bad[:o + l0 + l1 + l2]), 50000]))
    _empty_bai(str(p) + ".bai", 3)
    with pytest.raises(_lib.BsigError, match="sorted"):
        Reads.from_bam(ctx, BamFile(str(p)))


def test_file_level_calls_take_the_device_decode(monkeypatch, fixture_regions, expected_grid):
    """bamProfile(bampath, gr) with the whole-file decode goes through bsig_reads_from_bam"""
    from bamsignals_amd import GRanges, _lib, bamProfile
    from bamsignals_amd.device import Reads
    reg, _ = fixture_regions
    gr = GRanges(reg["chrom"], reg["start"], width=reg["width"], strand=reg["strand"])
    monkeypatch.setenv("BAMSIGNALS_DECODE", "all")
    monkeypatch.setenv("BAMSIGNALS_DEVICE_DECODE", "require")
    _lib.load().bsig_cache_clear()
    try:
        sig = bamProfile(BAM, gr, ss=True, shift=100, paired_end="midpoint", tlenFilter=(50, 200), verbose=False)
        got = np.concatenate([m.T.reshape(-1) for m in sig.as_list()])
        assert np.array_equal(got, expected_grid["profile|shift=100,mapq=0,ss=1,pe=midpoint,tf=50_200"])
        assert Reads.device_decode_timing()["total"] > 0
    finally:
        _lib.load().bsig_cache_clear()


def test_header_only_bam(ctx, tmp_path, monkeypatch):
    """no records at all: empty resident reads, every call returns zeros"""
    from bamsignals_amd import _lib, write_columns_as_bam
    from bamsignals_amd.device import Plan, make_params
    cols = dict(ref_len=np.asarray([1000, 2000], np.int32), ref_off=np.zeros(3, np.int64), pos=np.zeros(0, np.int32),
                flag=np.zeros(0, np.uint16), mapq=np.zeros(0, np.uint8), tlen=np.zeros(0, np.int32),
                cigar_off=np.zeros(1, np.int64), cigar=np.zeros(0, np.uint32))
    path = str(tmp_path / "empty.bam")
    write_columns_as_bam(path, ["a", "b"], cols)
    bam, dev = _both_ways(ctx, path, monkeypatch)
    assert dev.n_reads == 0
    p = Plan(ctx, dev, [0, 1], [10, 0], [100, 2000], [1, -1], make_params(_lib.MODE_COVERAGE))
    assert not p.run_host().any()
    dev.close()


def _regions_both_ways(ctx, bam, rg, monkeypatch, expect_device=True):
    """index-driven decode of the regions `rg` on the device and on the CPU: same resident reads"""
    from bamsignals_amd import _lib
    from bamsignals_amd.device import Plan, Reads, make_params
    beg = rg["loc"].astype(np.int64)
    end = beg + rg["len"]
    monkeypatch.setenv("BAMSIGNALS_DEVICE_DECODE", "require" if expect_device else "1")
    dev = Reads.from_bam_regions(ctx, bam, rg["rid"], beg, end)
    assert (Reads.device_decode_timing()["total"] > 0) == expect_device
    monkeypatch.setenv("BAMSIGNALS_DEVICE_DECODE", "0")
    cpu = Reads.from_bam_regions(ctx, bam, rg["rid"], beg, end)
    assert layout_info(dev) == layout_info(cpu)
    host = bam.decode(rg["rid"], beg, end)
    assert dev.n_reads == len(host["pos"])
    out = []
    for mode, a in ((_lib.MODE_PROFILE, dict(binsize=1, ss=True, shift=9)), (_lib.MODE_COVERAGE, dict()),
                    (_lib.MODE_COUNT, dict(binsize=-1))):
        res = []
        for r in (dev, cpu):
            p = Plan(ctx, r, rg["rid"], rg["loc"], rg["len"], rg["strand"], make_params(mode, **a))
            res.append(p.run_host().copy())
            p.close()
        assert np.array_equal(res[0], res[1])
        out.append(res[0])
    cpu.close()
    return dev, out


def test_index_driven_regions(ctx, tmp_path, monkeypatch, fixture_regions):
    """the BAI's chunks as independent islands: fixture BAM (one leaf bin per reference), a
    multi-bin synthetic BAM, regions that list nothing, one region, many overlapping regions; also in
    several passes through HBM"""
    from bamsignals_amd import write_columns_as_bam
    from bamsignals_amd.bamio import BamFile
    from bamsignals_amd.synth import synth_ranges, synth_reads
    from oracle import oracle_c
    _, rg = fixture_regions
    dev, _ = _regions_both_ways(ctx, BamFile(BAM), rg, monkeypatch)
    dev.close()

    cols = synth_reads(400_000, [3_000_000, 200_000, 1_500_000], seed=31, paired=True)
    path = str(tmp_path / "syn.bam")
    write_columns_as_bam(path, ["a", "b", "c"], cols)
    bam = BamFile(path)
    orc = oracle_c.OracleReads(cols["ref_off"], cols["pos"], cols["end"], cols["flag"], cols["mapq"], cols["tlen"])
    for n, width, seed in ((1, 500, 1), (40, 3000, 2), (600, 1500, 3)):
        rg = synth_ranges(n, width, cols["ref_len"], seed=seed, jitter=width // 2)
        for chunk_mb in ("", "1"):
            if chunk_mb:
                monkeypatch.setenv("BAMSIGNALS_DEVICE_DECODE_CHUNK_MB", chunk_mb)
            dev, out = _regions_both_ways(ctx, bam, rg, monkeypatch)
            # the decoded subset answers the query like the whole file does
            want, _ = oracle_c.pileup_core(orc, rg, binsize=1, ss=True, shift=9)
            assert np.array_equal(out[0], want)
            want, _ = oracle_c.coverage_core(orc, rg)
            assert np.array_equal(out[1], want)
            dev.close()
            monkeypatch.delenv("BAMSIGNALS_DEVICE_DECODE_CHUNK_MB", raising=False)
    # a region on a reference without reads nearby, and no regions at all
    empty = dict(rid=np.asarray([1], np.int32), loc=np.asarray([199_990], np.int32), len=np.asarray([5], np.int32),
                 strand=np.asarray([1], np.int32))
    dev, _ = _regions_both_ways(ctx, bam, empty, monkeypatch)
    dev.close()
    none = dict(rid=np.zeros(0, np.int32), loc=np.zeros(0, np.int32), len=np.zeros(0, np.int32), strand=np.zeros(0, np.int32))
    dev, _ = _regions_both_ways(ctx, bam, none, monkeypatch)
    assert dev.n_reads == 0
    dev.close()


def test_fuzz_streams(ctx, tmp_path, monkeypatch):
    """seeded random BAM streams (spec-derived records: random names, CIGARs, sequence lengths, tags,
    references without reads, unplaced tail) cut into BGZF blocks at random sizes and chunked at
    random sizes: device decode == CPU decode, whole file and index-free region lists alike"""
    from bamsignals_amd.bamio import BamFile
    from bamsignals_amd.device import Reads
    rng = np.random.default_rng(int(os.environ.get("BSIG_FUZZ_SEED", "99")))
    ops_ref = [0, 2, 3, 7, 8]
    n_sharded = 0
    n_cases = int(os.environ.get("BSIG_DECODE_FUZZ_CASES", "25"))
    for case in range(n_cases):
        n_ref = int(rng.integers(1, 5))
        ref_len = rng.integers(500, 200_000, n_ref).astype(np.int64)
        text = b"".join(b"@SQ\tSN:r%d\tLN:%d\n" % (i, ref_len[i]) for i in range(n_ref))
        hdr = b"BAM\x01" + struct.pack("<i", len(text)) + text + struct.pack("<i", n_ref)
        for i in range(n_ref):
            nm = b"r%d\x00" % i
            hdr += struct.pack("<i", len(nm)) + nm + struct.pack("<i", int(ref_len[i]))
        recs = []
        n_placed = 0
        for r in range(n_ref):
            if rng.random() < 0.2:
                continue                                            # a reference without reads
            m = int(rng.integers(1, 400))
            pos = np.sort(rng.integers(0, ref_len[r], m))
            for p in pos:
                name = bytes(rng.integers(33, 127, int(rng.integers(1, 40))).astype(np.uint8)) + b"\x00"
                ncig = int(rng.integers(0, 6))
                cig = [(int(rng.integers(1, 300)) << 4) | int(rng.choice([0, 1, 2, 3, 4, 5, 7, 8])) for _ in range(ncig)]
                lseq = int(rng.choice([0, 0, 5, 36, 151, 3000]))
                flag = int(rng.choice([0, 16, 4, 99, 147, 1024 + 16]))
                aux = b"" if rng.random() < 0.5 else b"NMC\x03" + b"XZZ" + bytes(rng.integers(65, 90, int(rng.integers(0, 50))).astype(np.uint8)) + b"\x00"
                body = struct.pack("<iiBBHHHiiii", r, int(p), len(name), int(rng.integers(0, 61)), 4681, ncig, flag, lseq,
                                   r if rng.random() < 0.5 else -1, int(rng.integers(-1, 1000)), int(rng.integers(-600, 600)))
                body += name + struct.pack("<%dI" % ncig, *cig) + bytes(rng.integers(0, 256, (lseq + 1) // 2).astype(np.uint8))
                body += bytes(rng.integers(0, 60, lseq).astype(np.uint8)) + aux
                recs.append(struct.pack("<i", len(body)) + body)
                n_placed += 1
        for _ in range(int(rng.integers(0, 5))):                     # unplaced tail
            name = b"u\x00"
            body = struct.pack("<iiBBHHHiiii", -1, -1, len(name), 0, 4680, 0, 4, 0, -1, -1, 0) + name
            recs.append(struct.pack("<i", len(body)) + body)
        stream = hdr + b"".join(recs)
        sizes = [int(x) for x in rng.integers(40, 65536, 7)] if rng.random() < 0.7 else [65536]
        path = str(tmp_path / ("f%d.bam" % case))
        levels = [(int(rng.choice([0, 1, 6, 9])), int(rng.choice([0, 0, 2, 3, 4]))) for _ in range(3)]
        with open(path, "wb") as fh:
            fh.write(_bgzf(stream, [min(s, 60000) for s in sizes], levels))     # (a block's compressed size must fit 16 bits)
        _empty_bai(path + ".bai", n_ref)
        if rng.random() < 0.5:
            monkeypatch.setenv("BAMSIGNALS_DEVICE_DECODE_CHUNK_MB", "1")
            monkeypatch.setenv("BAMSIGNALS_BATCH_BLOCKS", str(int(rng.integers(1, 9))))
        bam = BamFile(path)
        monkeypatch.setenv("BAMSIGNALS_DEVICE_DECODE", "require")
        monkeypatch.setenv("BAMSIGNALS_INFLATE", "gpu" if case % 2 else "cpu")      # both inflate engines
        # ... and, with the GPU inflating, both routes: every other such file is streamed into HBM and decoded
        # in shares of a few blocks behind the stream (chunks from smaller than a block to larger than the file)
        streamed = case % 4 == 1
        if streamed:
            monkeypatch.setenv("BAMSIGNALS_STREAM_MIN_MB", "0")
            monkeypatch.setenv("BAMSIGNALS_INFLATE_ROUND_BLOCKS", str(int(rng.integers(6, 40))))
            monkeypatch.setenv("BAMSIGNALS_STREAM_CHUNK_KB", str(int(rng.choice([2, 7, 64, 4096]))))
        dev = Reads.from_bam(ctx, bam)
        for k in ("BAMSIGNALS_STREAM_MIN_MB", "BAMSIGNALS_INFLATE_ROUND_BLOCKS", "BAMSIGNALS_STREAM_CHUNK_KB"):
            monkeypatch.delenv(k, raising=False)
        # the same file decoded in 2-4 shares (the multi-GPU route on one GPU): either the shares are
        # proven to tile the stream, or the call steps back to the single decode -- same reads either way
        monkeypatch.setenv("BAMSIGNALS_SHARD_MIN_BLOCKS", "1")
        n_slots = int(rng.integers(2, 5))
        from bamsignals_amd.device import Context
        more = [Context(0) for _ in range(n_slots - 1)]
        shares, was_sharded = Reads.from_bam_multi([ctx] + more, bam)
        n_sharded += bool(was_sharded)
        monkeypatch.setenv("BAMSIGNALS_DEVICE_DECODE", "0")
        cpu = Reads.from_bam(ctx, bam)
        assert dev.n_reads == n_placed and layout_info(dev) == layout_info(cpu), case
        want = _results(ctx, cpu, bam.ref_len.astype(np.int64), seed=case)
        for a, b in zip(_results(ctx, dev, bam.ref_len.astype(np.int64), seed=case), want):
            assert np.array_equal(a, b), case
        for k, sh in enumerate(shares):
            assert layout_info(sh) == layout_info(cpu), (case, k)
        for a, b in zip(_results(shares[-1].ctx, shares[-1], bam.ref_len.astype(np.int64), seed=case), want):
            assert np.array_equal(a, b), case
        for sh in shares:
            sh.close()
        for c in more:
            c.close()
        dev.close(); cpu.close(); bam.close()
        monkeypatch.delenv("BAMSIGNALS_DEVICE_DECODE_CHUNK_MB", raising=False)
        monkeypatch.delenv("BAMSIGNALS_BATCH_BLOCKS", raising=False)
        os.remove(path); os.remove(path + ".bai")
    print("decoded in shares:", n_sharded, "of", n_cases)
    assert n_sharded >= n_cases // 3          # the sharded route is really taken, not only its fall-back


def test_damaged_deflate_data_with_the_gpu_inflate(ctx, tmp_path, monkeypatch):
    """an invalid DEFLATE stream inside one block: k_inflate reports it, the call takes the CPU path,
    which names the problem"""
    from bamsignals_amd import _lib
    from bamsignals_amd.bamio import BamFile
    from bamsignals_amd.device import Reads
    raw = bytearray(open(BAM, "rb").read())
    # second block: make its first deflate block's type 3 (reserved) -> invalid for any decoder
    o = struct.unpack_from("<H", raw, 16)[0] + 1
    xlen = struct.unpack_from("<H", raw, o + 10)[0]
    raw[o + 12 + xlen] |= 0x06
    p = tmp_path / "bad.bam"
    p.write_bytes(bytes(raw))
    _empty_bai(str(p) + ".bai", 3)
    monkeypatch.setenv("BAMSIGNALS_INFLATE", "gpu")
    monkeypatch.setenv("BAMSIGNALS_DEVICE_DECODE", "1")
    with pytest.raises(_lib.BsigError, match="inflate"):
        Reads.from_bam(ctx, BamFile(str(p)))
    monkeypatch.setenv("BAMSIGNALS_DEVICE_DECODE", "require")
    with pytest.raises(_lib.BsigError, match="CPU decode path"):
        Reads.from_bam(ctx, BamFile(str(p)))


def test_crc_mismatch_under_both_inflate_engines(ctx, tmp_path, monkeypatch):
    """a block that inflates fine but not to what its CRC32 says: the GPU's CRC kernel (and the CPU
    pool's check) refuse it; the call reports it through the CPU path"""
    from bamsignals_amd import _lib
    from bamsignals_amd.bamio import BamFile
    from bamsignals_amd.device import Reads
    raw = bytearray(open(BAM, "rb").read())
    bsize = struct.unpack_from("<H", raw, 16)[0] + 1
    b2 = struct.unpack_from("<H", raw, bsize + 16)[0] + 1
    raw[bsize + b2 - 8] ^= 0x5A
    p = tmp_path / "crc.bam"
    p.write_bytes(bytes(raw))
    _empty_bai(str(p) + ".bai", 3)
    for eng in ("gpu", "cpu"):
        monkeypatch.setenv("BAMSIGNALS_INFLATE", eng)
        monkeypatch.setenv("BAMSIGNALS_DEVICE_DECODE", "1")
        with pytest.raises(_lib.BsigError, match="CRC"):
            Reads.from_bam(ctx, BamFile(str(p)))
        monkeypatch.setenv("BAMSIGNALS_DEVICE_DECODE", "require")
        with pytest.raises(_lib.BsigError, match="CPU decode path"):
            Reads.from_bam(ctx, BamFile(str(p)))


def test_inflate_engine_choice_is_never_far_from_the_better_engine(ctx, tmp_path, monkeypatch):
    """gpu_inflate_pays() picks the inflate engine per call from the compressed size and the number of
    blocks.  On a BAM of bare records (many blocks, few symbols each) and on a real-shaped one (names,
    bases, qualities: literal-heavy blocks), the engine it picks must not be much slower than the one it
    did not pick (timed here, generous margin: the point is the order of magnitude, not the last 20 %)."""
    from bamsignals_amd import write_columns_as_bam
    from bamsignals_amd.bamio import BamFile
    from bamsignals_amd.device import Reads
    from bamsignals_amd.synth import synth_reads
    monkeypatch.setenv("BAMSIGNALS_DEVICE_DECODE", "require")
    for tag, n, l_seq in (("bare", 6_000_000, 0), ("real", 1_500_000, 100)):
        cols = synth_reads(n, [200_000_000], seed=12)
        path = str(tmp_path / f"{tag}.bam")
        write_columns_as_bam(path, ["c"], cols, l_seq=l_seq, seed=3)
        bam = BamFile(path)
        t = {}
        for eng in ("", "gpu", "cpu", "", "gpu", "cpu"):          # second round: warm page cache and scratch
            if eng:
                monkeypatch.setenv("BAMSIGNALS_INFLATE", eng)
            else:
                monkeypatch.delenv("BAMSIGNALS_INFLATE", raising=False)
            r = Reads.from_bam(ctx, bam)
            assert r.n_reads == n
            d = Reads.device_decode_timing()
            t[eng] = d["inflate"] + d["copy_wait"]
            r.close()
        assert t[""] <= 2.0 * min(t["gpu"], t["cpu"]) + 0.01, (tag, t)
        bam.close()
