"""GPU: coordinates the small parity inputs never reach.

The resident layout addresses reads by a cumulative genome coordinate
``g = (ref_unit0 << 16) + pos`` and 32-bit bucket numbers (csrc/bsig_types.h); above 2^31 / 2^32
cumulative bp a 32-bit slip would leave every small test green.  Here the HIP path is compared with
the C oracle, bit for bit, on

* an hg38-shaped genome (24 references, 3.1 Gbp -- BASELINE config 5's shape) with ranges drawn
  over ALL references including the last three, and
* the north star's own shape (10 x 250 Mbp, 100k x 2 kb ranges) with ranges on references 9-10,

and the sharded routes (8 GPU slots inside one process; one process per rank over gloo) are
compared with the single-slot result (ranges own their outputs, ref: src/bamsignals.cpp:164,181,186;
sort order ref: :222-226,246).
"""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HG38 = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636,
        138394717, 133797422, 135086622, 133275309, 114364328, 107043718, 101991189, 90338345,
        83257441, 80373285, 58617616, 64444167, 46709983, 50818468, 156040895, 57227415]


@pytest.fixture(scope="module")
def ctx():
    from bamsignals_amd.device import Context
    c = Context(0)
    yield c
    c.close()


def _run(ctx, reads, rg, mode, **a):
    from bamsignals_amd.device import Plan, make_params
    plan = Plan(ctx, reads, rg["rid"], rg["loc"], rg["len"], rg["strand"], make_params(mode, **a))
    out = plan.run_host()
    plan.close()
    return out


def _oracle(cols):
    from oracle import oracle_c
    return oracle_c.OracleReads(cols["ref_off"], cols["pos"], cols["end"], cols["flag"], cols["mapq"], cols["tlen"])


def _per_reference_mismatch(got, want, off, rid):
    """Names the references whose ranges differ (so that a 32-bit slip points at its coordinate)."""
    bad = [int(rid[i]) for i in range(len(rid)) if not np.array_equal(got[off[i]:off[i + 1]], want[off[i]:off[i + 1]])]
    return sorted(set(bad))


def test_hg38_shape_all_references(ctx):
    """24 references / 3.1 Gbp: the last reference starts at cumulative 3.03e9 bp (> 2^31, and the last
    ones' reads sit above 2^31 + 2^30).  2e7 sparse single-end reads, 125,000 x 1 kb ranges over all
    references (config 5's per-GPU share), all three modes + the strand-split / shifted call."""
    from bamsignals_amd import _lib
    from bamsignals_amd.device import Reads
    from bamsignals_amd.synth import synth_ranges, synth_reads
    from oracle import oracle_c
    assert sum(HG38[:21]) > 2**31 and sum(HG38) < 2**32
    cols = synth_reads(20_000_000, HG38, seed=55, with_cigar=False)
    rg = synth_ranges(125_000, 1000, HG38, seed=56)
    # ranges on every reference, the last three included, and one touching the very last base
    extra = dict(rid=np.asarray([21, 22, 23, 23], np.int32),
                 loc=np.asarray([HG38[21] - 1000, 5, 0, HG38[23] - 700], np.int32),
                 len=np.asarray([1000, 1000, 1000, 1000], np.int32), strand=np.asarray([1, -1, 0, -1], np.int32))
    rg = {k: np.concatenate([rg[k], extra[k]]) for k in rg}
    assert set(np.unique(rg["rid"]).tolist()) == set(range(24))
    assert (rg["rid"] >= 21).sum() > 1000
    reads = Reads(ctx, cols["ref_len"], cols["ref_off"], cols["pos"], cols["flag"], cols["mapq"], cols["tlen"], end=cols["end"])
    orc = _oracle(cols)
    for mode, fn, a in (
        (_lib.MODE_PROFILE, oracle_c.pileup_core, dict(binsize=1)),
        (_lib.MODE_PROFILE, oracle_c.pileup_core, dict(binsize=1, ss=True, shift=-40, mapqual=7)),
        (_lib.MODE_PROFILE, oracle_c.pileup_core, dict(binsize=50, shift=13)),
        (_lib.MODE_COUNT, oracle_c.pileup_core, dict(binsize=-1, ss=True)),
        (_lib.MODE_COVERAGE, oracle_c.coverage_core, dict()),
    ):
        got = _run(ctx, reads, rg, mode, **a)
        want, off = fn(orc, rg, **a)
        assert np.array_equal(got, want), (a, _per_reference_mismatch(got, want, off, rg["rid"]))
    # the reads of the last reference are really counted (not an all-zero agreement)
    last = dict(rid=np.asarray([23], np.int32), loc=np.asarray([0], np.int32), len=np.asarray([HG38[23]], np.int32),
                strand=np.asarray([0], np.int32))
    a, b = int(cols["ref_off"][23]), int(cols["ref_off"][24])
    p5 = np.where(cols["flag"][a:b] & 16, cols["end"][a:b], cols["pos"][a:b])      # ref: src/bamsignals.cpp:340-344
    n_last = int((p5 < HG38[23]).sum())
    assert n_last > 100_000
    assert int(_run(ctx, reads, last, _lib.MODE_COUNT, binsize=-1)[0]) == n_last
    reads.close()


def test_hg38_shape_whole_genome_binning(ctx):
    """bamProfile over whole chromosomes in 10-kb bins (the wide-bin kernels) and chromosome-wide
    counts on the 3.1-Gbp layout: every read counted exactly once on its own reference."""
    from bamsignals_amd import _lib
    from bamsignals_amd.device import Reads
    from bamsignals_amd.synth import synth_reads
    from oracle import oracle_c
    cols = synth_reads(8_000_000, HG38, seed=77, with_cigar=False)
    reads = Reads(ctx, cols["ref_len"], cols["ref_off"], cols["pos"], cols["flag"], cols["mapq"], cols["tlen"], end=cols["end"])
    whole = dict(rid=np.arange(24, dtype=np.int32), loc=np.zeros(24, np.int32), len=np.asarray(HG38, np.int32),
                 strand=np.asarray([1, -1, 0] * 8, np.int32))
    cnt = _run(ctx, reads, whole, _lib.MODE_COUNT, binsize=-1)
    # every read is counted once on its own reference, except '-' reads whose 5' end (= end) lies
    # beyond the reference's last base
    p5 = np.where(cols["flag"] & 16, cols["end"], cols["pos"]).astype(np.int64)
    inside = p5 < np.asarray(HG38, np.int64)[cols["rid"]]
    per_ref = np.bincount(cols["rid"][inside], minlength=24)
    assert np.array_equal(cnt.astype(np.int64), per_ref)
    got = _run(ctx, reads, whole, _lib.MODE_PROFILE, binsize=10_000, ss=True)
    want, off = oracle_c.pileup_core(_oracle(cols), whole, binsize=10_000, ss=True)
    assert np.array_equal(got, want), _per_reference_mismatch(got, want, off, whole["rid"])
    reads.close()


def test_north_star_shape_high_references(ctx):
    """10 x 250 Mbp, 100k x 2 kb (the shape BASELINE.json's north star quotes), 5e7 reads: full parity
    on all ranges, and a second batch drawn ONLY on references 9-10 (cumulative 2.0-2.5e9 bp)."""
    from bamsignals_amd import _lib
    from bamsignals_amd.device import Reads
    from bamsignals_amd.synth import synth_ranges, synth_reads
    from oracle import oracle_c
    ref_len = [250_000_000] * 10
    cols = synth_reads(50_000_000, ref_len, seed=0xBA51, with_cigar=False)
    reads = Reads(ctx, cols["ref_len"], cols["ref_off"], cols["pos"], cols["flag"], cols["mapq"], cols["tlen"], end=cols["end"])
    orc = _oracle(cols)
    rg = synth_ranges(100_000, 2000, ref_len, seed=0xBA51 + 1)
    assert (rg["rid"] >= 8).sum() > 15_000
    got = _run(ctx, reads, rg, _lib.MODE_PROFILE, binsize=1)
    want, off = oracle_c.pileup_core(orc, rg, binsize=1)
    assert np.array_equal(got, want), _per_reference_mismatch(got, want, off, rg["rid"])
    hi = synth_ranges(40_000, 2000, [250_000_000] * 2, seed=99)
    hi["rid"] = (hi["rid"] + 8).astype(np.int32)
    for mode, fn, a in ((_lib.MODE_PROFILE, oracle_c.pileup_core, dict(binsize=1, ss=True, shift=75)),
                        (_lib.MODE_COVERAGE, oracle_c.coverage_core, dict()),
                        (_lib.MODE_COUNT, oracle_c.pileup_core, dict(binsize=-1))):
        got = _run(ctx, reads, hi, mode, **a)
        want, off = fn(orc, hi, **a)
        assert got.any()
        assert np.array_equal(got, want), (a, _per_reference_mismatch(got, want, off, hi["rid"]))
    reads.close()


# ------------------------------------------------------------------------------------------------
# sharded routes against the single-slot result
# ------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def synth_bam(tmp_path_factory):
    """A 6-reference BAM written by this repo's writer (gapped CIGARs, pairs, duplicates) + ranges."""
    from bamsignals_amd import GRanges, write_columns_as_bam
    from bamsignals_amd.synth import synth_ranges, synth_reads
    names = ["c%d" % i for i in range(6)]
    ref_len = [700_000, 30_000, 1_200_000, 90_000, 400_000, 650_000]
    cols = synth_reads(600_000, ref_len, seed=41, paired=True)
    d = tmp_path_factory.mktemp("shardbam")
    bam = str(d / "s.bam")
    write_columns_as_bam(bam, names, cols)
    rg = synth_ranges(1003, 1800, ref_len, seed=42, jitter=900)
    rg["len"][17] = 0
    gr = GRanges([names[r] for r in rg["rid"]], rg["loc"] + 1, width=rg["len"],
                 strand=[{1: "+", -1: "-", 0: "*"}[int(s)] for s in rg["strand"]])
    return bam, cols, rg, gr


def _flat(sig, ss):
    return np.concatenate([m.T.reshape(-1) if ss else np.asarray(m) for m in sig])


@pytest.mark.parametrize("decode", ["all", "regions"])
def test_eight_slots_in_one_process(synth_bam, decode, monkeypatch):
    """BAMSIGNALS_DEVICES with 8 slots (the box's one GPU listed eight times: eight contexts, streams,
    resident copies and plans) against the 1-slot result and the oracle."""
    from bamsignals_amd import _lib, bamCount, bamCoverage, bamProfile
    from bamsignals_amd.wrappers import last_call_route
    from oracle import oracle_c
    bam, cols, rg, gr = synth_bam
    orc = _oracle(cols)
    monkeypatch.setenv("BAMSIGNALS_DECODE", decode)
    monkeypatch.setenv("BAMSIGNALS_SHARD_MIN_BLOCKS", "2")
    res = {}
    try:
        for devs in ("0", "0,0,0,0,0,0,0,0"):
            monkeypatch.setenv("BAMSIGNALS_DEVICES", devs)
            _lib.load().bsig_cache_clear()
            for rep in range(2):                          # second call: resident on every slot
                p = _flat(bamProfile(bam, gr, ss=True, shift=60, paired_end="midpoint", tlenFilter=(40, 600), verbose=False), True)
                if devs != "0" and rep == 0:
                    # every slot decodes its share -- of the BGZF blocks, or of the BAI islands of the ranges
                    assert "sharded" in last_call_route(), last_call_route()
                c = bamCount(bam, gr, mapqual=20, verbose=False)
                v = _flat(bamCoverage(bam, gr, paired_end="extend", verbose=False), False)
                res[(devs, rep)] = (p, c, v)
        want_p, _ = oracle_c.pileup_core(orc, rg, binsize=1, ss=True, shift=60, pe_mid=True, tlen_filter=(40, 600), requiredF=66)
        want_c, _ = oracle_c.pileup_core(orc, rg, binsize=-1, mapqual=20)
        want_v, _ = oracle_c.coverage_core(orc, rg, tspan=True, tlen_filter=(0, 1000), requiredF=66)
        for key, (p, c, v) in res.items():
            assert np.array_equal(p, want_p), key
            assert np.array_equal(c, want_c), key
            assert np.array_equal(v, want_v), key
    finally:
        _lib.load().bsig_cache_clear()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_worker(rank, world, port, bam, rgd, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["BAMSIGNALS_DEVICE"] = "0"
    os.environ.pop("BAMSIGNALS_DEVICES", None)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from bamsignals_amd import GRanges
        from bamsignals_amd.dist import bamCount_sharded, bamCoverage_sharded, bamProfile_sharded
        gr = GRanges(rgd["chrom"], rgd["start"], width=rgd["width"], strand=rgd["strand"])
        p = bamProfile_sharded(bam, gr, ss=True, shift=60, paired_end="midpoint", tlenFilter=(40, 600))
        c = bamCount_sharded(bam, gr, mapqual=20)
        v = bamCoverage_sharded(bam, gr, paired_end="extend")
        if rank == 0:
            q.put((np.concatenate([m.T.reshape(-1) for m in p]), np.asarray(c), np.concatenate(v.as_list())))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_one_process_per_rank_over_gloo(synth_bam):
    """dist.bam*_sharded with the HIP path as the per-rank compute: 4 ranks over gloo sharing the
    box's GPU (the pool allows at most 6 processes on one card, so the 8-rank run of the sharding
    logic is the CPU test tests/test_dist_gloo.py::test_eight_rank_gather_over_gloo)."""
    import torch.multiprocessing as mp
    from oracle import oracle_c
    bam, cols, rg, gr = synth_bam
    world = 4
    names = ["c%d" % i for i in range(6)]
    rgd = dict(chrom=[names[r] for r in rg["rid"]], start=(rg["loc"] + 1).tolist(), width=rg["len"].tolist(),
               strand=[{1: "+", -1: "-", 0: "*"}[int(s)] for s in rg["strand"]])
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_rank_worker, args=(r, world, port, bam, rgd, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        got_p, got_c, got_v = q.get(timeout=480)
    finally:
        for p in procs:
            p.join(timeout=120)
    assert all(p.exitcode == 0 for p in procs)
    orc = _oracle(cols)
    want_p, _ = oracle_c.pileup_core(orc, rg, binsize=1, ss=True, shift=60, pe_mid=True, tlen_filter=(40, 600), requiredF=66)
    want_c, _ = oracle_c.pileup_core(orc, rg, binsize=-1, mapqual=20)
    want_v, _ = oracle_c.coverage_core(orc, rg, tspan=True, tlen_filter=(0, 1000), requiredF=66)
    assert np.array_equal(got_p, want_p) and np.array_equal(got_c, want_c) and np.array_equal(got_v, want_v)


def test_hg38_shaped_bam_file_level_both_decode_modes(tmp_path, monkeypatch):
    """The same genome as a BAM FILE (written by this repo's pooled writer: 24 references, bins of every
    BAI level, hundreds of linear-index windows per reference) through the drop-in entry points: the
    index-driven decode (BAI chunks -> islands, ref: one bam_itr_queryi per chunk of ranges,
    src/bamsignals.cpp:252-271) and the whole-file decode, against the oracle on the generator's columns."""
    from bamsignals_amd import GRanges, _lib, bamCount, bamCoverage, bamProfile, write_columns_as_bam
    from bamsignals_amd.synth import synth_ranges, synth_reads
    from bamsignals_amd.wrappers import last_call_route
    from oracle import oracle_c
    names = ["chr%d" % (i + 1) for i in range(22)] + ["chrX", "chrY"]
    cols = synth_reads(12_000_000, HG38, seed=91, paired=True)
    bam = str(tmp_path / "hg38like.bam")
    write_columns_as_bam(bam, names, cols)
    rg = synth_ranges(3000, 1200, HG38, seed=92, jitter=600)
    extra = dict(rid=np.asarray([23, 23, 0, 21], np.int32), loc=np.asarray([HG38[23] - 900, 0, HG38[0] - 1200, 16384 * 3 - 10], np.int32),
                 len=np.asarray([900, 1000, 1200, 40], np.int32), strand=np.asarray([-1, 1, 0, 1], np.int32))
    rg = {k: np.concatenate([rg[k], extra[k]]) for k in rg}
    # level order of the GRanges differs from the BAM's
    gr = GRanges([names[r] for r in rg["rid"]], rg["loc"] + 1, width=rg["len"], strand=[{1: "+", -1: "-", 0: "*"}[int(s)] for s in rg["strand"]])
    orc = _oracle(cols)
    want_p, _ = oracle_c.pileup_core(orc, rg, binsize=1, ss=True, shift=-20, pe_mid=True, tlen_filter=(30, 800), requiredF=66)
    want_c, _ = oracle_c.pileup_core(orc, rg, binsize=-1)
    want_v, _ = oracle_c.coverage_core(orc, rg, tspan=True, tlen_filter=(0, 1000), requiredF=66)
    try:
        for mode in ("regions", "all"):
            monkeypatch.setenv("BAMSIGNALS_DECODE", mode)
            _lib.load().bsig_cache_clear()
            p = bamProfile(bam, gr, ss=True, shift=-20, paired_end="midpoint", tlenFilter=(30, 800), verbose=False)
            assert ("index-driven" in last_call_route()) == (mode == "regions")
            assert np.array_equal(np.concatenate([m.T.reshape(-1) for m in p]), want_p), mode
            assert np.array_equal(bamCount(bam, gr, verbose=False), want_c), mode
            v = bamCoverage(bam, gr, paired_end="extend", verbose=False)
            assert np.array_equal(np.concatenate(v.as_list()), want_v), mode
    finally:
        _lib.load().bsig_cache_clear()


def test_references_longer_than_a_bai_can_address(ctx):
    """Two references of 1.6 and 0.9 Gbp (wheat- or axolotl-sized chromosomes, beyond the 2^29 bp a BAI
    index reaches: such BAMs carry a CSI index and are decoded whole): positions up to 1.6e9 in the
    int32 columns, ranges at both ends of both references."""
    from bamsignals_amd import _lib
    from bamsignals_amd.device import Reads
    from bamsignals_amd.synth import synth_ranges, synth_reads
    from oracle import oracle_c
    refs = [1_600_000_000, 900_000_000]
    cols = synth_reads(12_000_000, refs, seed=131, paired=True, with_cigar=False)
    rg = synth_ranges(60_000, 1500, refs, seed=132, jitter=500)
    extra = dict(rid=np.asarray([0, 0, 1, 1], np.int32), loc=np.asarray([refs[0] - 1500, 2**30 - 700, 0, refs[1] - 800], np.int32),
                 len=np.asarray([1500, 1500, 900, 800], np.int32), strand=np.asarray([1, -1, 0, -1], np.int32))
    rg = {k: np.concatenate([rg[k], extra[k]]) for k in rg}
    assert int((rg["loc"].astype(np.int64) > 2**30).sum()) > 5000
    reads = Reads(ctx, cols["ref_len"], cols["ref_off"], cols["pos"], cols["flag"], cols["mapq"], cols["tlen"], end=cols["end"])
    orc = _oracle(cols)
    for mode, fn, a in ((_lib.MODE_PROFILE, oracle_c.pileup_core, dict(binsize=1, ss=True, shift=40, pe_mid=True, tlen_filter=(0, 900), requiredF=66)),
                        (_lib.MODE_PROFILE, oracle_c.pileup_core, dict(binsize=25)),
                        (_lib.MODE_COUNT, oracle_c.pileup_core, dict(binsize=-1)),
                        (_lib.MODE_COVERAGE, oracle_c.coverage_core, dict(tspan=True, tlen_filter=(0, 1000), requiredF=66))):
        got = _run(ctx, reads, rg, mode, **a)
        want, off = fn(orc, rg, **a)
        assert got.any()
        assert np.array_equal(got, want), (a, _per_reference_mismatch(got, want, off, rg["rid"]))
    reads.close()
