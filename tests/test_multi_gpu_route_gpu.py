"""GPU: the single-process multi-GPU route and what the file-level calls keep between calls.

* sharded decode: every listed GPU inflates and parses its share of the BGZF blocks, the column
  shares are exchanged, every GPU lays the reads out (csrc/devdecode.hip: reads_from_bam_sharded) —
  against the single-GPU decode and the CPU decode, on htslib's fixture, on files whose records
  cross block (and therefore share) borders, with several passes per share, with both inflate engines;
* result gather on the first GPU (peer copies here; RCCL with a one-rank communicator) + device-side
  reassembly, against the per-GPU PCIe route and the oracle;
* the resident-BAM cache: least recently used out under BAMSIGNALS_CACHE_GB, a rewritten file is
  decoded again, the on-disk sidecar is loaded by a "second process" (a cleared cache).

The pool's boxes have ONE GPU: it is listed several times (separate contexts, streams, resident
copies and host threads: the same code path, with peer copies standing in for RCCL, which refuses
a GPU listed twice)."""
import gzip
import os
import shutil

import numpy as np
import pytest

from conftest import GOLDEN, layout_info
from test_device_decode_gpu import _bgzf, _empty_bai, _results

pytestmark = pytest.mark.gpu

BAM = os.path.join(GOLDEN, "randomBam.bam")


def _contexts(n):
    from bamsignals_amd.device import Context
    return [Context(0) for _ in range(n)]


def _sharded_vs_single(path, n, monkeypatch, expect_sharded=True):
    """Decode `path` in n shares and on one GPU; every slot's resident reads must be indistinguishable
    from the single-GPU decode's."""
    from bamsignals_amd.bamio import BamFile
    from bamsignals_amd.device import Reads
    bam = BamFile(path)
    ctxs = _contexts(n)
    monkeypatch.setenv("BAMSIGNALS_SHARD_MIN_BLOCKS", "2")
    monkeypatch.setenv("BAMSIGNALS_SHARDED_DECODE", "require" if expect_sharded else "1")
    many, sharded = Reads.from_bam_multi(ctxs, bam)
    assert sharded == expect_sharded
    monkeypatch.setenv("BAMSIGNALS_SHARDED_DECODE", "0")
    one, s2 = Reads.from_bam_multi(ctxs[:1], bam)
    assert not s2
    want = _results(ctxs[0], one[0], bam.ref_len)
    for k in range(n):
        assert layout_info(many[k]) == layout_info(one[0]), k
        for a, b in zip(_results(ctxs[k], many[k], bam.ref_len), want):
            assert np.array_equal(a, b), k
    n_reads = one[0].n_reads
    for r in many + one:
        r.close()
    for c in ctxs:
        c.close()
    bam.close()
    return n_reads


@pytest.mark.parametrize("n", [2, 3, 8])
@pytest.mark.parametrize("inflate", ["cpu", "gpu"])
def test_sharded_decode_of_the_reference_fixture(n, inflate, monkeypatch):
    monkeypatch.setenv("BAMSIGNALS_INFLATE", inflate)
    assert _sharded_vs_single(BAM, n, monkeypatch) == 99000


@pytest.mark.parametrize("inflate", ["cpu", "gpu"])
def test_sharded_decode_with_records_across_share_borders(tmp_path, inflate, monkeypatch):
    """htsjdk-style files: records run across BGZF block borders, so a share begins in the middle of a
    record; its lanes propose the first record start and the host accepts the shares only because
    every share's chain ends exactly where the next one's begins.  Also several passes per share
    (1-MiB chunks) with the cut-off record carried."""
    monkeypatch.setenv("BAMSIGNALS_INFLATE", inflate)
    stream = gzip.decompress(open(BAM, "rb").read())
    for k, sizes in enumerate(([4000, 9001, 517, 65000], [65536], [33000, 70, 1000])):
        p = tmp_path / ("straddle%d.bam" % k)
        p.write_bytes(_bgzf(stream, sizes))
        _empty_bai(str(p) + ".bai", 3)
        for n in (2, 5):
            assert _sharded_vs_single(str(p), n, monkeypatch) == 99000
    monkeypatch.setenv("BAMSIGNALS_DEVICE_DECODE_CHUNK_MB", "1")
    monkeypatch.setenv("BAMSIGNALS_BATCH_BLOCKS", "3")
    p = tmp_path / "straddle0.bam"
    assert _sharded_vs_single(str(p), 3, monkeypatch) == 99000


def test_sharded_decode_declines_what_it_cannot_prove(tmp_path, monkeypatch):
    """A 6-MB record in the middle of the file: the border between two shares falls inside it, megabytes
    in front of its end, far beyond the few blocks a share sees behind its own.  The sharded route steps
    aside and the call returns the single-GPU result."""
    import struct
    text = b"@SQ\tSN:c\tLN:3000000\n"
    hdr = b"BAM\x01" + struct.pack("<i", len(text)) + text + struct.pack("<i", 1) + struct.pack("<i", 2) + b"c\x00" + struct.pack("<i", 3000000)

    def rec(pos, lseq):
        name = b"q\x00"
        body = struct.pack("<iiBBHHHiiii", 0, pos, len(name), 40, 4681, 1, 0, lseq, -1, -1, 0) + name
        body += struct.pack("<I", (max(lseq, 30) << 4) | 0) + bytes((lseq + 1) // 2) + b"\xff" * lseq
        return struct.pack("<i", len(body)) + body
    recs = [rec(10 + i, 0) for i in range(1000)] + [rec(2000, 4_000_000)] + [rec(3000 + i, 0) for i in range(1000)]
    p = tmp_path / "giant.bam"
    p.write_bytes(_bgzf(hdr + b"".join(recs), [65536]))
    _empty_bai(str(p) + ".bai", 1)
    assert _sharded_vs_single(str(p), 2, monkeypatch, expect_sharded=False) == 2001


@pytest.fixture(scope="module")
def synth_bam(tmp_path_factory):
    from bamsignals_amd import GRanges, write_columns_as_bam
    from bamsignals_amd.synth import synth_ranges, synth_reads
    names = ["c%d" % i for i in range(5)]
    ref_len = [700_000, 30_000, 1_200_000, 90_000, 400_000]
    cols = synth_reads(500_000, ref_len, seed=141, paired=True)
    d = tmp_path_factory.mktemp("routebam")
    bam = str(d / "s.bam")
    write_columns_as_bam(bam, names, cols)
    rg = synth_ranges(777, 1800, ref_len, seed=142, jitter=900)
    gr = GRanges([names[r] for r in rg["rid"]], rg["loc"] + 1, width=rg["len"],
                 strand=[{1: "+", -1: "-", 0: "*"}[int(s)] for s in rg["strand"]])
    return bam, names, ref_len, cols, rg, gr


def _oracle(cols):
    from oracle import oracle_c
    return oracle_c.OracleReads(cols["ref_off"], cols["pos"], cols["end"], cols["flag"], cols["mapq"], cols["tlen"])


def _three_calls(bam, gr):
    from bamsignals_amd import bamCount, bamCoverage, bamProfile
    p = bamProfile(bam, gr, ss=True, shift=33, verbose=False)
    c = bamCount(bam, gr, paired_end="midpoint", tlenFilter=(30, 700), verbose=False)
    v = bamCoverage(bam, gr, mapqual=5, verbose=False)
    return np.concatenate([m.T.reshape(-1) for m in p]), np.asarray(c), np.concatenate(v.as_list())


def _want(cols, rg):
    from oracle import oracle_c
    orc = _oracle(cols)
    return (oracle_c.pileup_core(orc, rg, binsize=1, ss=True, shift=33)[0],
            oracle_c.pileup_core(orc, rg, binsize=-1, pe_mid=True, tlen_filter=(30, 700), requiredF=66)[0],
            oracle_c.coverage_core(orc, rg, mapqual=5)[0])


ROUTE_OF = {"xgmi": "result: xgmi/direct", "copy": "result: xgmi/peer", "pcie": "result: pcie (", "blocks": "result: pcie/blocks"}


@pytest.mark.parametrize("gather", ["xgmi", "copy", "pcie", "blocks"])
@pytest.mark.parametrize("devices", ["0,0,0", "0,0,0,0,0,0,0,0"])
def test_file_level_route_over_several_slots(synth_bam, devices, gather, monkeypatch):
    from bamsignals_amd import _lib
    from bamsignals_amd.wrappers import last_call_route, last_call_timing
    bam, names, ref_len, cols, rg, gr = synth_bam
    monkeypatch.setenv("BAMSIGNALS_DEVICES", devices)
    monkeypatch.setenv("BAMSIGNALS_GATHER", gather)
    monkeypatch.setenv("BAMSIGNALS_DECODE", "all")
    monkeypatch.setenv("BAMSIGNALS_SHARD_MIN_BLOCKS", "2")
    _lib.load().bsig_cache_clear()
    try:
        want = _want(cols, rg)
        got = _three_calls(bam, gr)
        for a, b in zip(got, want):
            assert np.array_equal(a, b)
        got = _three_calls(bam, gr)                          # resident on every slot
        assert last_call_timing()["bam_was_resident"]
        route = last_call_route()
        assert route.startswith("%d GPU slot(s); reads: resident" % len(devices.split(",")))
        # same-device slots: the first GPU reads the shard buffers in place ("direct"); "copy" forces the
        # gather into a receive buffer (what RCCL does with distinct GPUs), here with peer copies
        assert ROUTE_OF[gather] in route, route
        for a, b in zip(got, want):
            assert np.array_equal(a, b)
        # ... and delivered in place, range by range (bsig_*_core_into: what the R shim binds), over the same route
        from bamsignals_amd.wrappers import coverage_core_into, pileup_core_into
        p_in = pileup_core_into(bam, gr, (), binsize=1, ss=True, shift=33)
        c_in = pileup_core_into(bam, gr, (30, 700), binsize=-1, pe_mid=True, requiredF=66)
        v_in = coverage_core_into(bam, gr, (), mapqual=5)
        assert ROUTE_OF[gather] in last_call_route()
        assert np.array_equal(np.concatenate([m.T.reshape(-1) for m in p_in]), want[0])
        assert np.array_equal(c_in[0], want[1]) and np.array_equal(np.concatenate(v_in), want[2])
        # a call on a resident BAM with shapes seen before allocates nothing for the result path
        before = _lib.load().bsig_debug_scratch_allocs()
        assert before > 0
        got = _three_calls(bam, gr)
        assert _lib.load().bsig_debug_scratch_allocs() == before
        for a, b in zip(got, want):
            assert np.array_equal(a, b)
        _lib.load().bsig_cache_clear()
        _three_calls(bam, gr[list(range(5))])
        _lib.load().bsig_cache_clear()
        from bamsignals_amd import bamProfile
        bamProfile(bam, gr, verbose=False)
        assert "sharded decode, columns over peer" in last_call_route()
    finally:
        _lib.load().bsig_cache_clear()


def test_rccl_route_with_a_one_rank_communicator(synth_bam, monkeypatch):
    """BAMSIGNALS_FORCE_SHARDED=1 takes the multi-GPU route with the box's one GPU listed once: librccl is
    loaded, ncclCommInitAll builds a communicator, the exchanges go through grouped RCCL calls."""
    from bamsignals_amd import _lib
    from bamsignals_amd.wrappers import last_call_route
    bam, names, ref_len, cols, rg, gr = synth_bam
    monkeypatch.setenv("BAMSIGNALS_DEVICES", "0")
    monkeypatch.setenv("BAMSIGNALS_FORCE_SHARDED", "1")
    monkeypatch.setenv("BAMSIGNALS_EXCHANGE", "rccl")
    monkeypatch.setenv("BAMSIGNALS_DECODE", "all")
    _lib.load().bsig_cache_clear()
    try:
        got = _three_calls(bam, gr)
        route = last_call_route()
        assert "xgmi/rccl" in route, route
        for a, b in zip(got, _want(cols, rg)):
            assert np.array_equal(a, b)
        _lib.load().bsig_cache_clear()
        from bamsignals_amd import bamProfile
        bamProfile(bam, gr, verbose=False)
        assert "columns over rccl" in last_call_route(), last_call_route()
    finally:
        _lib.load().bsig_cache_clear()


def test_decode_on_second_context_after_first(synth_bam):
    """Two contexts in one process, used one after the other for a decode and a pageable download of
    more than 8 MiB: the staging buffers and events belong to the device they are used on."""
    from bamsignals_amd import _lib
    from bamsignals_amd.bamio import BamFile
    from bamsignals_amd.device import Context, Plan, Reads, make_params
    from bamsignals_amd.synth import tile_ranges
    bam, names, ref_len, cols, rg, gr = synth_bam
    b = BamFile(bam)
    tiles = tile_ranges(ref_len, 4000)
    outs = []
    for k in range(2):
        ctx = Context(0)
        r = Reads.from_bam(ctx, b)
        p = Plan(ctx, r, tiles["rid"], tiles["loc"], tiles["len"], tiles["strand"], make_params(_lib.MODE_COVERAGE))
        assert p.cells * 4 > (8 << 20)
        outs.append(p.run_host().copy())
        p.close(); r.close(); ctx.close()
    assert np.array_equal(outs[0], outs[1])
    from oracle import oracle_c
    assert np.array_equal(outs[0], oracle_c.coverage_core(_oracle(cols), tiles)[0])


def test_cache_lru_rewrite_and_sidecar(synth_bam, tmp_path, monkeypatch):
    from bamsignals_amd import _lib, bamCount, write_columns_as_bam
    from bamsignals_amd.synth import synth_reads
    from bamsignals_amd.wrappers import last_call_route, last_call_timing
    from oracle import oracle_c
    bam0, names, ref_len, cols, rg, gr = synth_bam
    monkeypatch.setenv("BAMSIGNALS_DEVICES", "0")
    monkeypatch.setenv("BAMSIGNALS_DECODE", "all")
    lib = _lib.load()
    lib.bsig_cache_clear()
    try:
        # three BAMs (copies with different content), a budget that holds two of them
        paths, wants = [], []
        for k in range(3):
            c = synth_reads(300_000, ref_len, seed=500 + k, paired=True)
            p = str(tmp_path / f"b{k}.bam")
            write_columns_as_bam(p, names, c)
            paths.append(p)
            wants.append(oracle_c.pileup_core(_oracle(c), rg, binsize=-1)[0])
        from bamsignals_amd.bamio import BamFile
        from bamsignals_amd.device import Context, Reads
        ctx = Context(0)
        bf = BamFile(paths[0])
        one = Reads.from_bam(ctx, bf).info()["hbm_bytes"]
        bf.close(); ctx.close()
        monkeypatch.setenv("BAMSIGNALS_CACHE_GB", repr(2.5 * one / 2**30))
        def call(k):
            got = bamCount(paths[k], gr, verbose=False)
            assert np.array_equal(got, wants[k]), k
            return last_call_timing()["bam_was_resident"]
        assert [call(0), call(1), call(0), call(1)] == [False, False, True, True]      # both stay resident
        assert call(2) is False                                                        # evicts the least recently used: 0
        assert call(1) is True
        assert call(0) is False                                                        # 0 was evicted (and now evicts 2)
        assert call(2) is False
        # a rewritten file (same path, new content) is decoded again and gives the new counts
        c = synth_reads(300_000, ref_len, seed=900, paired=True)
        write_columns_as_bam(paths[1], names, c)
        wants[1] = oracle_c.pileup_core(_oracle(c), rg, binsize=-1)[0]
        assert call(1) is False
        assert call(1) is True
        # another spelling of the same path is the same file (one resident copy)
        other = os.path.join(os.path.dirname(paths[1]), ".", os.path.basename(paths[1]))
        got = bamCount(other, gr, verbose=False)
        assert np.array_equal(got, wants[1]) and last_call_timing()["bam_was_resident"]
        # the sidecar: written by the first cold call, loaded by a "second process" (cleared cache)
        lib.bsig_cache_clear()
        side = tmp_path / "sidecars"
        side.mkdir()
        monkeypatch.setenv("BAMSIGNALS_SIDECAR_DIR", str(side))
        assert call(0) is False and last_call_route().split("; ")[1] == "reads: decode"
        files = os.listdir(side)
        assert len(files) == 1 and files[0].endswith(".bsig")
        lib.bsig_cache_clear()
        assert call(0) is False and last_call_route().split("; ")[1] == "reads: sidecar"
        # a sidecar made from another version of the BAM is ignored (and replaced)
        shutil.copy(paths[2], paths[0]); shutil.copy(paths[2] + ".bai", paths[0] + ".bai")
        wants[0] = wants[2]
        lib.bsig_cache_clear()
        assert call(0) is False and last_call_route().split("; ")[1] == "reads: decode"
        lib.bsig_cache_clear()
        assert call(0) is False and last_call_route().split("; ")[1] == "reads: sidecar"
        # a damaged sidecar is ignored too
        f = side / os.listdir(side)[0]
        raw = bytearray(f.read_bytes())
        f.write_bytes(bytes(raw[: len(raw) // 2]))
        lib.bsig_cache_clear()
        assert call(0) is False and last_call_route().split("; ")[1] == "reads: decode"
    finally:
        lib.bsig_cache_clear()


def test_reads_save_and_load_round_trip(synth_bam, tmp_path):
    from bamsignals_amd import _lib
    from bamsignals_amd.bamio import BamFile
    from bamsignals_amd.device import Context, Reads
    bam, names, ref_len, cols, rg, gr = synth_bam
    ctx = Context(0)
    b = BamFile(bam)
    r = Reads.from_bam(ctx, b)
    f = str(tmp_path / "r.bsig")
    r.save(f, "stamp-1")
    back = Reads.load(ctx, f, "stamp-1")
    assert layout_info(back) == layout_info(r)
    for x, y in zip(_results(ctx, back, ref_len), _results(ctx, r, ref_len)):
        assert np.array_equal(x, y)
    with pytest.raises(_lib.BsigError, match="another version"):
        Reads.load(ctx, f, "stamp-2")
    with pytest.raises(_lib.BsigError):
        Reads.load(ctx, str(tmp_path / "absent.bsig"), "stamp-1")
    back.close(); r.close(); b.close(); ctx.close()


def test_concurrent_host_threads(synth_bam, tmp_path, monkeypatch):
    """The file-level calls hold the cache lock for look-ups only: calls from several host threads -- on
    resident BAMs side by side, cold decodes taking turns -- give the single-threaded results."""
    import threading
    from bamsignals_amd import _lib, bamCount, bamProfile, write_columns_as_bam
    from bamsignals_amd.synth import synth_reads
    from oracle import oracle_c
    bam0, names, ref_len, cols0, rg, gr = synth_bam
    monkeypatch.setenv("BAMSIGNALS_DEVICES", "0")
    monkeypatch.setenv("BAMSIGNALS_DECODE", "all")
    _lib.load().bsig_cache_clear()
    paths, want_p, want_c = [bam0], [], []
    cols_all = [cols0]
    for k in range(2):
        c = synth_reads(200_000, ref_len, seed=700 + k, paired=True)
        p = str(tmp_path / f"t{k}.bam")
        write_columns_as_bam(p, names, c)
        paths.append(p)
        cols_all.append(c)
    for c in cols_all:
        want_p.append(oracle_c.pileup_core(_oracle(c), rg, binsize=1, ss=True, shift=12)[0])
        want_c.append(oracle_c.pileup_core(_oracle(c), rg, binsize=-1)[0])
    errors = []

    def worker(t):
        try:
            for it in range(6):
                k = (t + it) % len(paths)
                if (t + it) % 2:
                    got = np.concatenate([m.T.reshape(-1) for m in bamProfile(paths[k], gr, ss=True, shift=12, verbose=False)])
                    assert np.array_equal(got, want_p[k]), (t, it, k)
                else:
                    assert np.array_equal(bamCount(paths[k], gr, verbose=False), want_c[k]), (t, it, k)
        except Exception as exc:            # noqa: BLE001 - reported below
            errors.append(repr(exc))
    try:
        th = [threading.Thread(target=worker, args=(t,)) for t in range(5)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        assert not errors, errors
    finally:
        _lib.load().bsig_cache_clear()


def test_sorted_ranges_leave_in_contiguous_slices(synth_bam, monkeypatch):
    """The (rid, loc)-sorted ranges are dealt to the GPUs in blocks of consecutive ranges (ref: each range owns its
    output, src/bamsignals.cpp:164,181,186; sorted as :246 sorts them).  A caller whose ranges ARE in that order -- a
    tiling, sorted peaks -- owns, per block, one contiguous slice of the result: with a large result the library then
    lets every GPU download its slices over its own PCIe link straight into place ("pcie/blocks": no gather, no
    reassembly) without being asked; ranges in any other order keep the gather on the first GPU."""
    from bamsignals_amd import GRanges, _lib, bamCoverage, bamProfile
    from bamsignals_amd.wrappers import last_call_route, pileup_core_into
    bam, names, ref_len, cols, rg, gr = synth_bam
    # 2-kb tiles over all five references, every 500 bp: 4,840 ranges, 77 MB strand-split
    chrom, start = [], []
    for nm, ln in zip(names, ref_len):
        s0 = np.arange(1, ln - 2000, 500)
        chrom += [nm] * len(s0); start += s0.tolist()
    tiles = GRanges(chrom, start, width=2000, strand=["+", "-", "*", "+"] * (len(start) // 4) + ["+"] * (len(start) % 4))
    monkeypatch.setenv("BAMSIGNALS_DECODE", "all")
    monkeypatch.setenv("BAMSIGNALS_DEVICES", "0")
    _lib.load().bsig_cache_clear()
    try:
        want = np.concatenate([m.T.reshape(-1) for m in bamProfile(bam, tiles, ss=True, shift=-20, verbose=False)])
        want_cov = np.concatenate(bamCoverage(bam, tiles[list(range(1000))], verbose=False).as_list())
        monkeypatch.setenv("BAMSIGNALS_DEVICES", "0,0,0,0,0")
        monkeypatch.setenv("BAMSIGNALS_SHARD_MIN_BLOCKS", "2")
        got = np.concatenate([m.T.reshape(-1) for m in bamProfile(bam, tiles, ss=True, shift=-20, verbose=False)])
        route = last_call_route()
        assert "result: pcie/blocks" in route and np.array_equal(got, want), route
        n_slices = int(route.split("download in ")[1].split(" slices")[0])
        assert n_slices <= len(tiles) // 64                      # blocks of consecutive ranges, not single ranges
        got = np.concatenate([m.T.reshape(-1) for m in pileup_core_into(bam, tiles, (), binsize=1, ss=True, shift=-20)])
        assert "result: pcie/blocks" in last_call_route() and np.array_equal(got, want)
        # a small result, or ranges out of order: the gather on the first GPU
        got = np.concatenate(bamCoverage(bam, tiles[list(range(1000))], verbose=False).as_list())
        assert "result: xgmi/" in last_call_route() and np.array_equal(got, want_cov)
        perm = np.random.default_rng(4).permutation(len(tiles))
        got = bamProfile(bam, tiles[perm.tolist()], ss=True, shift=-20, verbose=False)
        assert "result: xgmi/" in last_call_route()
        back = np.empty_like(want).reshape(len(tiles), -1)
        for k, i in enumerate(perm):
            back[i] = got[k].T.reshape(-1)
        assert np.array_equal(back.reshape(-1), want)
        # range by range (the round-robin of rounds 1-3) on request
        monkeypatch.setenv("BAMSIGNALS_SHARD_BLOCK", "1")
        got = np.concatenate([m.T.reshape(-1) for m in bamProfile(bam, tiles, ss=True, shift=-20, verbose=False)])
        assert "result: xgmi/" in last_call_route() and np.array_equal(got, want)
    finally:
        _lib.load().bsig_cache_clear()


@pytest.mark.parametrize("gather", ["xgmi", "pcie", "blocks"])
def test_fewer_ranges_than_slots(synth_bam, gather, monkeypatch):
    """No range at all, one range, and one zero-width range over four slots: empty shards everywhere."""
    from bamsignals_amd import GRanges, _lib, bamCount, bamCoverage, bamProfile
    bam, names, ref_len, cols, rg, gr = synth_bam
    monkeypatch.setenv("BAMSIGNALS_DEVICES", "0,0,0,0")
    monkeypatch.setenv("BAMSIGNALS_GATHER", gather)
    monkeypatch.setenv("BAMSIGNALS_DECODE", "all")
    monkeypatch.setenv("BAMSIGNALS_SHARD_MIN_BLOCKS", "2")
    _lib.load().bsig_cache_clear()
    try:
        none = GRanges([], [], width=[], strand=[])
        assert len(bamProfile(bam, none, verbose=False)) == 0
        assert bamCount(bam, none, verbose=False).shape == (0,)
        assert bamCount(bam, none, ss=True, verbose=False).shape == (2, 0)
        one = gr[[5]]
        monkeypatch.setenv("BAMSIGNALS_DEVICES", "0")
        want_p = bamProfile(bam, one, ss=True, verbose=False)[0]
        want_c = bamCount(bam, one, verbose=False)
        want_v = bamCoverage(bam, one, verbose=False)[0]
        monkeypatch.setenv("BAMSIGNALS_DEVICES", "0,0,0,0")
        assert np.array_equal(bamProfile(bam, one, ss=True, verbose=False)[0], want_p)
        assert np.array_equal(bamCount(bam, one, verbose=False), want_c)
        assert np.array_equal(bamCoverage(bam, one, verbose=False)[0], want_v)
        zero = GRanges([names[0]], [100], width=[0], strand=["-"])
        assert bamProfile(bam, zero, verbose=False)[0].shape == (0,)
        assert bamCoverage(bam, zero, verbose=False)[0].shape == (0,)
        assert bamCount(bam, zero, verbose=False).tolist() == [0]
    finally:
        _lib.load().bsig_cache_clear()


def test_concurrent_host_threads_on_the_multi_gpu_route(synth_bam, monkeypatch):
    """Several host threads inside the multi-GPU route at once (slots "0,0" and the forced one-rank RCCL
    exchange): the runs take turns on the slots' cached buffers and on the exchange (RCCL forbids
    interleaved group calls on one communicator), and the cache may be cleared while calls are in flight."""
    import threading
    from bamsignals_amd import _lib, bamCount, bamProfile
    from oracle import oracle_c
    bam, names, ref_len, cols, rg, gr = synth_bam
    monkeypatch.setenv("BAMSIGNALS_DECODE", "all")
    monkeypatch.setenv("BAMSIGNALS_SHARD_MIN_BLOCKS", "2")
    want_p = oracle_c.pileup_core(_oracle(cols), rg, binsize=1, ss=True, shift=12)[0]
    want_c = oracle_c.pileup_core(_oracle(cols), rg, binsize=-1)[0]
    for devices, force in (("0,0", "0"), ("0", "1")):
        monkeypatch.setenv("BAMSIGNALS_DEVICES", devices)
        monkeypatch.setenv("BAMSIGNALS_FORCE_SHARDED", force)
        _lib.load().bsig_cache_clear()
        errors = []

        def worker(t):
            try:
                for it in range(5):
                    if (t + it) % 2:
                        got = np.concatenate([m.T.reshape(-1) for m in bamProfile(bam, gr, ss=True, shift=12, verbose=False)])
                        assert np.array_equal(got, want_p), (t, it)
                    else:
                        assert np.array_equal(bamCount(bam, gr, verbose=False), want_c), (t, it)
                    if t == 0 and it == 2:
                        _lib.load().bsig_cache_clear()          # while the other threads are inside their calls
            except Exception as exc:            # noqa: BLE001 - reported below
                errors.append(repr(exc))
        try:
            th = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
            for x in th:
                x.start()
            for x in th:
                x.join()
            assert not errors, errors
        finally:
            _lib.load().bsig_cache_clear()


def test_alternating_device_lists_keep_their_resident_copies(synth_bam, monkeypatch):
    """A session that alternates between two device lists (here "0" and "0,0") keeps both resident sets."""
    from bamsignals_amd import _lib, bamCount
    from bamsignals_amd.wrappers import last_call_timing
    from oracle import oracle_c
    bam, names, ref_len, cols, rg, gr = synth_bam
    monkeypatch.setenv("BAMSIGNALS_DECODE", "all")
    monkeypatch.setenv("BAMSIGNALS_SHARD_MIN_BLOCKS", "2")
    want = oracle_c.pileup_core(_oracle(cols), rg, binsize=-1)[0]
    _lib.load().bsig_cache_clear()
    try:
        seen = []
        for devices in ("0", "0,0", "0", "0,0", "0"):
            monkeypatch.setenv("BAMSIGNALS_DEVICES", devices)
            assert np.array_equal(bamCount(bam, gr, verbose=False), want)
            seen.append(bool(last_call_timing()["bam_was_resident"]))
        assert seen == [False, False, True, True, True]
    finally:
        _lib.load().bsig_cache_clear()


def test_region_decodes_are_remembered(synth_bam, monkeypatch):
    """Index-driven decodes are kept (a few, LRU): a repeated call -- same ranges, or ranges inside an
    earlier call's -- finds its reads resident; BAMSIGNALS_REGION_CACHE=0 restores the reference's behaviour."""
    from bamsignals_amd import _lib, bamCount, bamProfile
    from bamsignals_amd.wrappers import last_call_route, last_call_timing
    from oracle import oracle_c
    bam, names, ref_len, cols, rg, gr = synth_bam
    monkeypatch.setenv("BAMSIGNALS_DEVICES", "0")
    monkeypatch.setenv("BAMSIGNALS_DECODE", "regions")
    pick = list(range(0, 50))
    sub = {k: v[pick] for k, v in rg.items()}
    want_c = oracle_c.pileup_core(_oracle(cols), sub, binsize=-1)[0]
    want_p = oracle_c.pileup_core(_oracle(cols), sub, binsize=1, shift=20)[0]
    _lib.load().bsig_cache_clear()
    try:
        # the wider query first (shift 20 -> ext 20), then the narrower ones inside it
        got = np.concatenate(bamProfile(bam, gr[pick], shift=20, verbose=False).as_list())
        assert np.array_equal(got, want_p) and not last_call_timing()["bam_was_resident"]
        assert "index-driven decode" in last_call_route()
        for rep in range(3):
            assert np.array_equal(bamCount(bam, gr[pick], verbose=False), want_c)
            t = last_call_timing()
            assert t["bam_was_resident"] and "resident (index-driven" in last_call_route(), last_call_route()
        assert t["total"] < 0.01, t                       # a resident call on 50 ranges: no decode, no I/O
        assert np.array_equal(bamCount(bam, gr[pick[:7]], verbose=False), want_c[:7]) and last_call_timing()["bam_was_resident"]
        # ranges outside what was decoded: decoded afresh
        other = list(range(60, 90))
        w2 = oracle_c.pileup_core(_oracle(cols), {k: v[other] for k, v in rg.items()}, binsize=-1)[0]
        assert np.array_equal(bamCount(bam, gr[other], verbose=False), w2) and not last_call_timing()["bam_was_resident"]
        monkeypatch.setenv("BAMSIGNALS_REGION_CACHE", "0")
        _lib.load().bsig_cache_clear()
        for rep in range(2):
            assert np.array_equal(bamCount(bam, gr[pick], verbose=False), want_c)
            assert not last_call_timing()["bam_was_resident"]
        # the index-driven decodes count against the same byte budget as whole files (BAMSIGNALS_CACHE_GB): with room
        # for one of them the older one goes when a second is decoded, and a budget of 0 keeps none
        monkeypatch.delenv("BAMSIGNALS_REGION_CACHE")
        from bamsignals_amd.bamio import BamFile
        from bamsignals_amd.device import Context, Reads
        ctx = Context(0)
        bf = BamFile(bam)
        sizes = []
        for idx in (pick, other):
            beg = rg["loc"][idx].astype(np.int64)
            r = Reads.from_bam_regions(ctx, bf, rg["rid"][idx], beg, beg + rg["len"][idx])
            sizes.append(r.info()["hbm_bytes"])
            r.close()
        bf.close(); ctx.close()
        assert min(sizes) > 0.2 * max(sizes)
        monkeypatch.setenv("BAMSIGNALS_CACHE_GB", repr(1.1 * max(sizes) / 2**30))
        _lib.load().bsig_cache_clear()
        def count(idx):
            bamCount(bam, gr[idx], verbose=False)
            return last_call_timing()["bam_was_resident"]
        assert [count(pick), count(pick), count(other), count(other)] == [False, True, False, True]
        assert count(pick) is False                       # evicted by `other`'s decode: the budget holds one of them
        monkeypatch.setenv("BAMSIGNALS_CACHE_GB", "0")
        _lib.load().bsig_cache_clear()
        assert [count(pick), count(pick)] == [False, False]
    finally:
        _lib.load().bsig_cache_clear()


def test_damaged_reads_files_are_refused(synth_bam, tmp_path):
    """bsig_reads_load trusts nothing in the file: a flipped byte in a column or an index (checksum / index
    check on the device), shapes that do not follow from the counts, foreign unit tables."""
    import struct
    from bamsignals_amd import _lib
    from bamsignals_amd.bamio import BamFile
    from bamsignals_amd.device import Context, Reads
    bam, names, ref_len, cols, rg, gr = synth_bam
    ctx = Context(0)
    b = BamFile(bam)
    r = Reads.from_bam(ctx, b)
    f = tmp_path / "r.bsig"
    r.save(str(f), "s")
    raw = f.read_bytes()
    Reads.load(ctx, str(f), "s").close()
    # header: magic 8, version 4, n_ref 4, n_reads 8, stamp_len 4, n_classes 4, file_bytes 8, then 5 x (n 8, maxspan 4, kshift 4,
    # col_cap 8, idx_entries 8), checksum 8, n_codes 4, reserved 4; the entry of the packed class (the last) is patched: it holds
    # nearly all reads of this file
    cls0 = 8 + 4 + 4 + 8 + 4 + 4 + 8 + 4 * 32
    n_codes_at = 8 + 4 + 4 + 8 + 4 + 4 + 8 + 5 * 32 + 8
    assert struct.unpack_from("<q", raw, cls0)[0] > 0 and 0 < struct.unpack_from("<I", raw, n_codes_at)[0] <= 512
    def patched(off, fmt, value):
        x = bytearray(raw)
        x[off:off + struct.calcsize(fmt)] = struct.pack(fmt, value)
        return bytes(x)
    n0, = struct.unpack_from("<q", raw, cls0)
    kshift0, = struct.unpack_from("<i", raw, cls0 + 12)
    cases = {
        "a byte of a column": bytes(raw[:len(raw) // 2]) + bytes([raw[len(raw) // 2] ^ 0x40]) + bytes(raw[len(raw) // 2 + 1:]),
        "a byte near the end (index)": bytes(raw[:-200]) + bytes([raw[-200] ^ 0x01]) + bytes(raw[-199:]),
        "kshift": patched(cls0 + 12, "<i", kshift0 + 1),
        "kshift huge": patched(cls0 + 12, "<i", 40),
        "class count": patched(cls0, "<q", n0 - 1),
        "read count": patched(16, "<q", struct.unpack_from("<q", raw, 16)[0] + 5),
        "idx entries": patched(cls0 + 24, "<Q", struct.unpack_from("<Q", raw, cls0 + 24)[0] - 1),
        "maxspan": patched(cls0 + 8, "<i", 0),
        "maxspan beyond a packed word's span bits": patched(cls0 + 8, "<i", 300),
        "number of codes": patched(n_codes_at, "<I", 600),
        "no codes but packed reads": patched(n_codes_at, "<I", 0),
    }
    for what, blob in cases.items():
        g = tmp_path / "bad.bsig"
        g.write_bytes(blob)
        with pytest.raises(_lib.BsigError):
            Reads.load(ctx, str(g), "s")
    r.close(); b.close(); ctx.close()


def test_segment_map_on_the_device(synth_bam):
    """bsig_segmap_*: the device-side twin of bsig_scatter_segments, and its argument checks."""
    import torch
    from bamsignals_amd import _lib
    from bamsignals_amd.device import Context, SegmentMap
    rng = np.random.default_rng(5)
    lens = rng.integers(0, 3000, 4000).astype(np.int64)
    dst_off = np.concatenate([[0], np.cumsum(lens)])
    which = rng.permutation(len(lens)).astype(np.int64)
    src_off = np.concatenate([[0], np.cumsum(lens[which])])
    src = rng.integers(-5, 1 << 20, int(src_off[-1])).astype(np.int32)
    want = np.zeros(int(dst_off[-1]), np.int32)
    lib = _lib.load()
    _lib.check(lib.bsig_scatter_segments(len(which), src.ctypes.data, src_off.ctypes.data, want.ctypes.data, dst_off.ctypes.data, which.ctypes.data))
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        ctx = Context(0, stream=stream.cuda_stream)
        m = SegmentMap(ctx, src_off, dst_off, which)
        d_src = torch.from_numpy(src).cuda()
        d_dst = torch.zeros(len(want), dtype=torch.int32, device="cuda")
        m.run(d_src.data_ptr(), d_dst.data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(d_dst.cpu().numpy(), want)
        m.close()
        bad = src_off.copy(); bad[5] += 1
        with pytest.raises(_lib.BsigError):
            SegmentMap(ctx, bad, dst_off, which)
        w2 = which.copy(); w2[0] = len(lens)
        with pytest.raises(_lib.BsigError):
            SegmentMap(ctx, src_off, dst_off, w2)
        ctx.close()


def _narrow_decode(msg, n_cells, cap):
    """The narrow message as include/bamsignals_abi.h lays it out, read back with numpy: dword 0 = number of exceptions,
    dwords 1-3 zero, then ceil(n / 16) words of two-bit codes (cell c in bits 2 (c % 16) .. +1 of word c // 16; 3 = see
    the list), then `cap` (cell, value) pairs of which the first `number of exceptions` are in use, in any order."""
    m = msg.view(np.uint32)
    n_exc, n_words = int(m[0]), (n_cells + 15) // 16
    assert not m[1:4].any() and len(m) == 4 + n_words + 2 * cap and n_exc <= cap
    codes = (m[4:4 + n_words, None] >> (2 * np.arange(16, dtype=np.uint32))[None, :]) & 3
    out = codes.reshape(-1)[:n_cells].astype(np.int32)
    pairs = m[4 + n_words:4 + n_words + 2 * n_exc].reshape(-1, 2)
    assert len(np.unique(pairs[:, 0])) == n_exc and np.array_equal(np.sort(pairs[:, 0]), np.flatnonzero(out == 3))
    out[pairs[:, 0]] = pairs[:, 1].view(np.int32)
    return out


def test_narrow_wire_is_lossless(synth_bam):
    """bsig_narrow_pack + bsig_segmap_run_narrow against bsig_segmap_run on the same shard: two bits a cell and a list of
    exceptions must give back every int32 (negative coverage differences, big counts, INT32_MIN), whatever the
    alignment of a segment's destination, with cells of the shard's padding (in no segment) carrying exceptions of
    their own; a list that is too short is reported, never silently wrong."""
    import torch
    from bamsignals_amd import _lib
    from bamsignals_amd.device import Context, SegmentMap, narrow_bytes, narrow_count, narrow_pack
    rng = np.random.default_rng(11)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        ctx = Context(0, stream=stream.cuda_stream)
        for n_seg, max_len, tail, dense in ((4000, 3000, 37, 0.02), (1, 70_001, 0, 0.3), (300, 5, 16, 1.0), (64, 1 << 14, 5, 0.0)):
            lens = rng.integers(0, max_len + 1, n_seg).astype(np.int64)
            dst_off = np.concatenate([[0], np.cumsum(lens)])
            which = rng.permutation(n_seg).astype(np.int64)
            src_off = np.concatenate([[0], np.cumsum(lens[which])])
            n_src = int(src_off[-1]) + tail                     # (`tail` cells of padding behind the last segment)
            src = rng.integers(0, 3, n_src).astype(np.int32)
            odd = rng.random(n_src) < dense
            src[odd] = rng.integers(-(1 << 31), 1 << 31, int(odd.sum()), dtype=np.int64).astype(np.int32)
            if n_src > 40:
                src[[0, 15, 16, n_src - 1]] = [-1, 3, np.iinfo(np.int32).min, np.iinfo(np.int32).max]
            n_exc = int(((src < 0) | (src > 2)).sum())
            want = np.full(int(dst_off[-1]) + 3, -7, np.int32)
            d_src = torch.from_numpy(src).cuda()
            m = SegmentMap(ctx, src_off, dst_off, which)
            for shift in (0, 1, 3):                             # destinations that are not 16-B aligned
                d_plain = torch.full((len(want),), -7, dtype=torch.int32, device="cuda")
                m.run(d_src.data_ptr(), d_plain.data_ptr() + 4 * shift)
                for cap in (n_exc, n_exc + 100):
                    msg = torch.full((narrow_bytes(n_src, cap) // 4,), 0x5a5a5a5a, dtype=torch.int32, device="cuda")
                    narrow_pack(ctx, d_src.data_ptr(), n_src, msg.data_ptr(), cap)
                    assert narrow_count(ctx, msg.data_ptr()) == n_exc
                    if shift == 0:
                        assert np.array_equal(_narrow_decode(msg.cpu().numpy(), n_src, cap), src)
                    d_dst = torch.full((len(want),), -7, dtype=torch.int32, device="cuda")
                    m.run_narrow(msg.data_ptr(), n_src, cap, d_dst.data_ptr() + 4 * shift)
                    torch.cuda.synchronize()
                    assert torch.equal(d_dst, d_plain), (n_seg, shift, cap)
                    assert not m.narrow_overflowed()
            assert narrow_bytes(n_src, n_exc) == 4 * (4 + (n_src + 15) // 16 + 2 * n_exc)
            if n_exc > 1:                                       # a list one short: told, and nothing written out of place
                msg = torch.zeros(narrow_bytes(n_src, n_exc - 1) // 4, dtype=torch.int32, device="cuda")
                narrow_pack(ctx, d_src.data_ptr(), n_src, msg.data_ptr(), n_exc - 1)
                assert narrow_count(ctx, msg.data_ptr()) == n_exc
                d_dst = torch.full((len(want),), -7, dtype=torch.int32, device="cuda")
                m.run_narrow(msg.data_ptr(), n_src, n_exc - 1, d_dst.data_ptr())
                assert m.narrow_overflowed()
                assert int((d_dst[int(dst_off[-1]):] != -7).sum()) == 0
            with pytest.raises(_lib.BsigError):
                m.run_narrow(d_src.data_ptr(), int(src_off[-1]) - 1, 0, d_src.data_ptr())      # fewer cells than the segments span
            m.close()
        with pytest.raises(_lib.BsigError):
            narrow_pack(ctx, d_src.data_ptr() + 4, 16, d_src.data_ptr(), 0)                    # not 16-B aligned
        ctx.close()


def test_arena_reserved_with_the_context(synth_bam, monkeypatch):
    """BAMSIGNALS_ARENA_GB: one allocation made when a context comes up; scratch, resident reads and result
    buffers are carved out of it (and come back to it), what does not fit takes the ordinary route; the results
    are the same, and bsig_cache_clear() hands an idle arena back."""
    from bamsignals_amd import _lib
    from bamsignals_amd.device import Context
    bam, names, ref_len, cols, rg, gr = synth_bam
    want = _want(cols, rg)
    for gb, devices in (("1", "0"), ("1", "0,0,0")):
        monkeypatch.setenv("BAMSIGNALS_ARENA_GB", gb)
        monkeypatch.setenv("BAMSIGNALS_DEVICES", devices)
        monkeypatch.setenv("BAMSIGNALS_DECODE", "all")
        monkeypatch.setenv("BAMSIGNALS_SHARD_MIN_BLOCKS", "2")
        _lib.load().bsig_cache_clear()
        try:
            Context(0).close()                               # (the arena comes with the first context)
            for rep in range(3):
                for a, b in zip(_three_calls(bam, gr), want):
                    assert np.array_equal(a, b)
        finally:
            _lib.load().bsig_cache_clear()
    monkeypatch.delenv("BAMSIGNALS_ARENA_GB")
    _lib.load().bsig_cache_clear()
