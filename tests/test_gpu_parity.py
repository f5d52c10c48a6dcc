"""GPU: the HIP path (through the C ABI) against the oracle and the committed golden vectors.
Bit-exact: all arithmetic is 32-bit integer."""
import numpy as np
import pytest

from conftest import core_args, expected_packed, parse_key

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from bamsignals_amd.device import Context
    c = Context(0)
    yield c
    c.close()


def _gpu(ctx, reads, ranges, kind, tile_cells=0, threads=0, **a):
    from bamsignals_amd import _lib
    from bamsignals_amd.device import Plan, make_params
    if kind == "coverage":
        p = make_params(_lib.MODE_COVERAGE, tile_cells=tile_cells, threads=threads, **a)
    else:
        bs = a.pop("binsize", 1)
        mode = _lib.MODE_COUNT if bs <= 0 else _lib.MODE_PROFILE
        p = make_params(mode, binsize=bs, tile_cells=tile_cells, threads=threads, **a)
    plan = Plan(ctx, reads, ranges["rid"], ranges["loc"], ranges["len"], ranges["strand"], p)
    out = plan.run_host()
    off = plan.offsets
    plan.close()
    return out, off


@pytest.fixture(scope="module")
def fx_reads(ctx, fixture_reads):
    from bamsignals_amd.device import Reads
    fx = fixture_reads
    r = Reads(ctx, fx["ref_len"], fx["ref_off"], fx["bam_pos"], fx["bam_flag"], fx["bam_mapq"], fx["bam_tlen"],
              end=fx["bam_end"])
    yield r
    r.close()


def test_golden_grid(ctx, fx_reads, fixture_regions, expected_grid):
    """Full grid of tests/testthat/test_methods.R:33-104 on the reference's fixture reads."""
    _, ranges = fixture_regions
    plus = dict(ranges, strand=np.ones_like(ranges["strand"]))
    for key, want in expected_grid.items():
        kind, p = parse_key(key)
        a = core_args(kind, p)
        got, _ = _gpu(ctx, fx_reads, plus if kind == "ff16" else ranges, "coverage" if kind == "coverage" else "pileup", **a)
        assert np.array_equal(got, want), key


def test_golden_extra(ctx, fx_reads, fixture_regions, expected_extra):
    _, ranges = fixture_regions
    star = dict(ranges, strand=np.asarray([(1, -1, 0)[i % 3] for i in range(len(ranges["rid"]))], dtype=np.int32))
    cases = {
        "profile_star_bs1": ("pileup", star, dict(binsize=1)),
        "profile_star_bs7_ss": ("pileup", star, dict(binsize=7, ss=True)),
        "profile_bs50_shift-30": ("pileup", ranges, dict(binsize=50, shift=-30)),
        "profile_ff0": ("pileup", ranges, dict(binsize=1, filteredF=0)),
        "profile_ff1024": ("pileup", ranges, dict(binsize=1, filteredF=1024)),
        "profile_ff1040": ("pileup", ranges, dict(binsize=1, filteredF=1040)),
        "count_star_ss": ("pileup", star, dict(binsize=-1, ss=True)),
        "profile_mid_bs3_ss": ("pileup", star, dict(binsize=3, ss=True, requiredF=66, tlen_filter=(30, 300),
                                                    pe_mid=True, shift=5)),
        "coverage_star": ("coverage", star, dict()),
        "coverage_star_extend": ("coverage", star, dict(requiredF=66, tlen_filter=(0, 1000), tspan=True)),
    }
    assert set(cases) == set(expected_extra)
    for name, (kind, rg, a) in cases.items():
        got, _ = _gpu(ctx, fx_reads, rg, kind, **a)
        assert np.array_equal(got, expected_extra[name]), name


def test_end_from_cigar_on_gpu(ctx, fixture_reads, fixture_regions, expected_grid):
    """bam_endpos computed by k_cigar_end instead of being supplied."""
    from bamsignals_amd.device import Reads
    fx = fixture_reads
    r = Reads(ctx, fx["ref_len"], fx["ref_off"], fx["bam_pos"], fx["bam_flag"], fx["bam_mapq"], fx["bam_tlen"],
              cigar_off=fx["bam_cigar_off"], cigar=fx["bam_cigar"])
    _, ranges = fixture_regions
    key = "profile|shift=100,mapq=0,ss=1,pe=midpoint,tf=50_200"
    kind, p = parse_key(key)
    got, _ = _gpu(ctx, r, ranges, "pileup", **core_args(kind, p))
    assert np.array_equal(got, expected_grid[key])
    r.close()


@pytest.fixture(scope="module")
def synth(ctx):
    """Synthetic multi-reference reads with D/N/S/I CIGARs (several span classes) + oracle handle."""
    from bamsignals_amd.device import Reads
    from bamsignals_amd.synth import synth_reads
    from oracle import oracle_c
    out = {}
    for name, paired in (("se", False), ("pe", True)):
        cols = synth_reads(400_000, [700_000, 150_000, 65_536, 300_001], seed=11 + paired, paired=paired)
        # a few very long reads -> classes 2 and 3; unmapped-placed reads; a zero-op read
        cig = cols["cigar"].copy()
        first = cols["cigar_off"][:-1]
        sel = np.arange(0, len(first), 40_001)
        cig[first[sel]] = (70_000 << 4) | 3          # first op becomes a 70 kb N skip
        sel2 = np.arange(7, len(first), 30_011)
        cig[first[sel2]] = (5_000 << 4) | 2          # 5 kb deletion
        flag = cols["flag"].copy()
        flag[np.arange(3, len(flag), 9_973)] |= 4      # unmapped but placed: 1-bp reads
        end = oracle_c.cigar_end(cols["pos"], flag, cols["cigar_off"], cig)
        gpu = Reads(ctx, cols["ref_len"], cols["ref_off"], cols["pos"], flag, cols["mapq"], cols["tlen"],
                    cigar_off=cols["cigar_off"], cigar=cig)
        orc = oracle_c.OracleReads(cols["ref_off"], cols["pos"], end, flag, cols["mapq"], cols["tlen"])
        out[name] = (gpu, orc, cols, end, flag)
    yield out
    for gpu, *_ in out.values():
        gpu.close()


def test_span_classes(synth):
    gpu, _, cols, end, flag = synth["se"]
    info = gpu.info()
    span = end - cols["pos"] + 1
    packed, n_codes = expected_packed(cols["pos"], end, flag, cols["mapq"], cols["ref_off"], cols["ref_len"])
    want = [int(np.sum((span <= 256) & ~packed)), int(np.sum((span > 256) & (span <= 4096))),
            int(np.sum((span > 4096) & (span <= 65536))), int(np.sum(span > 65536)), int(np.sum(packed))]
    assert info["class_n"] == want and info["n_codes"] == n_codes
    assert all(w > 0 for w in want[1:])
    for c, (lo, hi) in enumerate(((0, 256), (256, 4096), (4096, 65536), (65536, 1 << 31))):
        m = (span > lo) & (span <= hi) & ~packed
        assert info["class_maxspan"][c] == (int(span[m].max()) if m.any() else 0)
    assert info["class_maxspan"][4] == int(span[packed].max()) and info["class_bucket_shift"][4] <= 15


def _rand_ranges(rng, ref_len, n, maxw):
    ref_len = np.asarray(ref_len)
    rid = rng.integers(0, len(ref_len), n).astype(np.int32)
    ln = rng.integers(0, maxw, n).astype(np.int32)
    ln[rng.random(n) < 0.05] = 0                                   # zero-width ranges
    loc = (rng.random(n) * (ref_len[rid] + 400) - 200).astype(np.int32)   # some hang over both ends
    strand = rng.integers(-1, 2, n).astype(np.int32)
    rg = dict(rid=rid, loc=loc, len=ln, strand=strand)
    # duplicates and nested ranges
    for k in ("rid", "loc", "len", "strand"):
        rg[k][1] = rg[k][0]
    rg["loc"][3] = rg["loc"][2] + 5
    rg["rid"][3] = rg["rid"][2]
    return rg


PILEUP_CASES = [
    dict(binsize=1),
    dict(binsize=1, ss=True, shift=37),
    dict(binsize=1, mapqual=30, filteredF=1024),
    dict(binsize=1, shift=-80, ss=True, filteredF=16),
    dict(binsize=2, ss=True),
    dict(binsize=13, shift=5),
    dict(binsize=200, ss=True, mapqual=10),
    dict(binsize=100_000),
    dict(binsize=-1),
    dict(binsize=-1, ss=True, shift=-20, mapqual=17),
]
PE_CASES = [
    dict(binsize=1, requiredF=66, tlen_filter=(0, 1000)),
    dict(binsize=1, requiredF=66, tlen_filter=(120, 180), pe_mid=True, ss=True),
    dict(binsize=9, requiredF=66, tlen_filter=(0, 1000), pe_mid=True, shift=75, ss=True),
    dict(binsize=-1, requiredF=66, tlen_filter=(50, 500), pe_mid=True, ss=True, shift=75),
    dict(binsize=1, tlen_filter=(50, 500), shift=75, ss=True, requiredF=66),
]


def test_flag_bits_above_the_sam_specification(ctx, synth):
    """Span classes 0 and 1 pack span - 1 next to the flag; class 1 has room for the 12 flag bits SAM defines.
    Reads that set a higher bit (a uint16 can) and span more than 256 bp must land in class 2 and still be
    filtered on all 16 bits, exactly as the reference filters them (ref: src/bamsignals.cpp:328-333)."""
    from bamsignals_amd.device import Reads
    from bamsignals_amd.synth import synth_ranges
    from oracle import oracle_c
    _, _, cols, end, _ = synth["pe"]
    flag = cols["flag"].copy()
    span = end - cols["pos"] + 1
    rng = np.random.default_rng(77)
    hit = rng.random(len(flag)) < 0.2
    flag[hit] |= rng.choice(np.asarray([0x1000, 0x2000, 0x4000, 0x8000, 0x9000], np.uint16), int(hit.sum()))
    reads = Reads(ctx, cols["ref_len"], cols["ref_off"], cols["pos"], flag, cols["mapq"], cols["tlen"], end=end)
    info = reads.info()
    c1 = (span > 256) & (span <= 4096)
    assert info["class_n"][1] == int(np.sum(c1 & (flag < 4096))) and info["class_n"][1] > 0
    assert info["class_n"][2] == int(np.sum((span > 4096) & (span <= 65536))) + int(np.sum(c1 & (flag >= 4096)))
    assert int(np.sum(c1 & (flag >= 4096))) > 0
    orc = oracle_c.OracleReads(cols["ref_off"], cols["pos"], end, flag, cols["mapq"], cols["tlen"])
    rg = synth_ranges(400, 3000, cols["ref_len"], seed=78, jitter=2000)
    for a in (dict(binsize=1, ss=True, requiredF=0x1000), dict(binsize=1, filteredF=0x8010), dict(binsize=-1, requiredF=0x9000, ss=True),
              dict(binsize=13, requiredF=66, filteredF=0x4000, tlen_filter=(0, 900), pe_mid=True, shift=7)):
        got, _ = _gpu(ctx, reads, rg, "pileup", **dict(a))
        want, _ = oracle_c.pileup_core(orc, rg, **a)
        assert got.any() and np.array_equal(got, want), a
    for a in (dict(requiredF=0x2000), dict(filteredF=0x1400, requiredF=66, tlen_filter=(0, 1000), tspan=True)):
        got, _ = _gpu(ctx, reads, rg, "coverage", **dict(a))
        want, _ = oracle_c.coverage_core(orc, rg, **a)
        assert np.array_equal(got, want), a
    reads.close()


def test_packed_class_and_what_stays_outside_it(ctx, synth, monkeypatch):
    """The packed class holds short reads whose (flag, mapq) pair has one of the file's 512 codes, one 32-bit
    word per read: 15 position bits, span - 1, the code.  Everything else a short read can be stays in class 0
    and must give the same counts: more than 512 pairs in the file (the rare ones), flag bits above the SAM
    specification, a position beyond the reference.  A window wider than the 32,768 bases the position bits span
    (a shift or a template length filter of tens of kilobases) is walked in chunks.  BAMSIGNALS_PACK=0 lays
    the same reads out without a packed class (every short read in class 0): same results again."""
    from bamsignals_amd.device import Reads
    from bamsignals_amd.synth import synth_ranges
    from oracle import oracle_c
    _, _, cols, end, flag0 = synth["pe"]
    rng = np.random.default_rng(404)
    n = len(cols["pos"])
    flag = flag0.copy()
    mapq = rng.integers(0, 256, n).astype(np.uint8)                   # 256 mapq values x 8+ flags: far more than 512 pairs
    mapq[rng.random(n) < 0.5] = 60                                    # ... half of the reads on a few frequent ones
    odd = rng.random(n) < 0.02
    flag[odd] |= rng.choice(np.asarray([0x1000, 0x8000], np.uint16), int(odd.sum()))
    pos = cols["pos"].copy()
    end = end.copy()
    # the last reads of the last reference hang over its end by more than a 64-kbp unit (invalid but
    # representable): their position is not one the bucket index can vouch for
    last = np.arange(n - 6, n)
    over = (int(cols["ref_len"][-1]) >> 16 << 16) + 65_536 + np.arange(6, dtype=np.int32) * 7
    pos[last] = over
    end[last] = over + 49
    for pack in ("1", "0"):
        monkeypatch.setenv("BAMSIGNALS_PACK", pack)
        reads = Reads(ctx, cols["ref_len"], cols["ref_off"], pos, flag, mapq, cols["tlen"], end=end)
        info = reads.info()
        span = end.astype(np.int64) - pos + 1
        if pack == "1":
            packed, n_codes = expected_packed(pos, end, flag, mapq, cols["ref_off"], cols["ref_len"])
            assert n_codes == 512 == info["n_codes"]
            assert not packed[last].any() and not packed[flag >= 4096].any()
            assert info["class_n"][4] == int(packed.sum()) > 0
            assert info["class_n"][0] == int(np.sum((span <= 256) & ~packed)) > 0
        else:
            assert info["class_n"][4] == 0 and info["n_codes"] == 0 and info["class_n"][0] == int(np.sum(span <= 256))
        orc = oracle_c.OracleReads(cols["ref_off"], pos, end, flag, mapq, cols["tlen"])
        rg = synth_ranges(300, 3000, cols["ref_len"], seed=405, jitter=2500)
        big = _rand_ranges(rng, cols["ref_len"], 10, 150_000)
        for ranges in (rg, big):
            for a in (dict(binsize=1, ss=True, mapqual=61), dict(binsize=1, filteredF=0x8010), dict(binsize=-1, ss=True, requiredF=66),
                      dict(binsize=25, requiredF=66, tlen_filter=(0, 900), pe_mid=True, shift=7, mapqual=3),
                      # windows of 2 x 40 kb + the range: two or three chunks of position bits per tile
                      dict(binsize=1, shift=40_000, ss=True), dict(binsize=-1, shift=-70_000),
                      dict(binsize=300, requiredF=66, tlen_filter=(0, 50_000), pe_mid=True, ss=True)):
                want, _ = oracle_c.pileup_core(orc, ranges, **a)
                for threads in (64, 256):
                    got, _ = _gpu(ctx, reads, ranges, "pileup", threads=threads, **dict(a))
                    assert np.array_equal(got, want), (pack, a, threads)
            for a in (dict(mapqual=200), dict(filteredF=0x1400, requiredF=66, tlen_filter=(0, 1000), tspan=True),
                      dict(requiredF=66, tlen_filter=(0, 45_000), tspan=True)):
                want, _ = oracle_c.coverage_core(orc, ranges, **a)
                got, _ = _gpu(ctx, reads, ranges, "coverage", **dict(a))
                assert np.array_equal(got, want), (pack, a)
        # ... and through the slices of heavy tiles (fixed read ranges clip the packed class's chunks)
        monkeypatch.setenv("BAMSIGNALS_HEAVY_READS", "64")
        for a in (dict(binsize=1, shift=40_000, ss=True), dict(binsize=-1, mapqual=20), dict(binsize=7)):
            want, _ = oracle_c.pileup_core(orc, rg, **a)
            got, _ = _gpu(ctx, reads, rg, "pileup", **dict(a))
            assert np.array_equal(got, want), (pack, "heavy", a)
        want, _ = oracle_c.coverage_core(orc, rg, requiredF=66, tlen_filter=(0, 45_000), tspan=True)
        got, _ = _gpu(ctx, reads, rg, "coverage", requiredF=66, tlen_filter=(0, 45_000), tspan=True)
        assert np.array_equal(got, want), (pack, "heavy coverage")
        monkeypatch.delenv("BAMSIGNALS_HEAVY_READS")
        reads.close()
    monkeypatch.delenv("BAMSIGNALS_PACK")


@pytest.mark.parametrize("width,what", [(600, "four tiles per wave, 8 waves per SIMD"), (1000, "one tile per workgroup, 8 waves per SIMD"),
                                        (2000, "one tile per workgroup")])
def test_large_launches_look_their_windows_up_in_a_launch_of_their_own(ctx, synth, width, what):
    """From 32,768 tiles on, a step is two launches: k_resolve_tiles (the index lookup of every tile, one lane per tile)
    and the pileup kernel, which comes in forms of its own for that case (no lookup code; narrow tiles: 8 waves per SIMD;
    very narrow ones: four consecutive tiles per wave through one image).  40,000 ranges of each kind against the oracle --
    ragged widths, all strands, ranges hanging over reference ends, duplicates, zero widths."""
    from bamsignals_amd.synth import synth_ranges
    from oracle import oracle_c
    for which, a in (("se", dict(binsize=1)), ("pe", dict(binsize=1, ss=True, shift=-30, requiredF=66, tlen_filter=(50, 500))),
                     ("se", dict(binsize=3, ss=True, mapqual=20))):
        gpu, orc, cols, _, _ = synth[which]
        rg = synth_ranges(40_000, width, cols["ref_len"], seed=width, jitter=width // 4)
        rg["len"][::97] = 0
        rg["loc"][::41] -= width // 2
        for k in ("rid", "loc", "len", "strand"):
            rg[k][5::1000] = rg[k][4::1000][:len(rg[k][5::1000])]
        want, woff = oracle_c.pileup_core(orc, rg, **a)
        got, off = _gpu(ctx, gpu, rg, "pileup", **dict(a))
        assert np.array_equal(off, woff) and np.array_equal(got, want), (what, a)
    gpu, orc, cols, _, _ = synth["pe"]
    rg = synth_ranges(40_000, width, cols["ref_len"], seed=width + 1, jitter=width // 4)
    for a in (dict(), dict(requiredF=66, tlen_filter=(0, 1000), tspan=True)):
        want, _ = oracle_c.coverage_core(orc, rg, **a)
        got, _ = _gpu(ctx, gpu, rg, "coverage", **dict(a))
        assert np.array_equal(got, want), (what, "coverage", a)
    want, _ = oracle_c.pileup_core(orc, rg, binsize=-1, ss=True)
    got, _ = _gpu(ctx, gpu, rg, "pileup", binsize=-1, ss=True)
    assert np.array_equal(got, want), (what, "count")
    # bamCount's forms: 1, 2, 4 or 8 consecutive tiles per wave, 2-4 packed passes in flight (launches of 65,536 tiles
    # and more take 8 x 4 by themselves, smaller ones 4 x 2), also under a template-length rule
    from bamsignals_amd import _lib
    knob = _lib.load().bsig_debug_set_knob
    try:
        for tiles, pre in ((1, 2), (2, 3), (4, 2), (8, 4)):
            assert knob(1, tiles) == 0 and knob(2, pre) == 0
            for a in (dict(binsize=-1, ss=True), dict(binsize=-1, shift=40, requiredF=66, tlen_filter=(50, 500), pe_mid=True)):
                want, _ = oracle_c.pileup_core(orc, rg, **a)
                got, _ = _gpu(ctx, gpu, rg, "pileup", **dict(a))
                assert np.array_equal(got, want), (what, "count", tiles, pre, a)
    finally:
        knob(1, 0); knob(2, 0)


def test_a_plans_windows_are_kept_after_its_first_run_and_die_with_the_layout(ctx, synth, monkeypatch):
    """A large launch looks its tiles' windows up (k_resolve_tiles, bam_itr_queryi's counterpart, ref:
    src/bamsignals.cpp:267) in the plan's FIRST run and keeps them: a plan and a layout of the reads are immutable, so the
    windows are a function of the two.  Runs 1, 2 and 3 of one plan are identical to the oracle with the windows kept and
    with BAMSIGNALS_CACHE_WINDOWS=0 (looked up in every run); results written into a second buffer do not depend on
    what the first run left in the first; and a plan refuses to run once the reads have been laid out again -- its kept
    windows, its heavy-tile slices and its filter table were read off the old layout."""
    import ctypes
    from bamsignals_amd import _lib
    from bamsignals_amd.device import Plan, Reads, make_params
    from bamsignals_amd.synth import synth_ranges
    from oracle import oracle_c
    gpu, orc, cols, _, _ = synth["pe"]
    rg = synth_ranges(40_000, 900, cols["ref_len"], seed=4321, jitter=200)
    for mode, kind, a in ((_lib.MODE_PROFILE, "pileup", dict(binsize=1, ss=True, shift=15)), (_lib.MODE_COVERAGE, "coverage", dict(mapqual=5))):
        want, _ = (oracle_c.coverage_core if kind == "coverage" else oracle_c.pileup_core)(orc, rg, **a)
        for keep in ("1", "0"):
            monkeypatch.setenv("BAMSIGNALS_CACHE_WINDOWS", keep)
            plan = Plan(ctx, gpu, rg["rid"], rg["loc"], rg["len"], rg["strand"], make_params(mode, **a))
            st = plan.stats()
            for run in range(3):
                got = plan.run_host()
                assert np.array_equal(got, want), (kind, keep, run)
            # the step's algorithmic bytes say which form ran: kept windows are read back (48 B per tile), not looked up
            per_tile = (st["algorithmic_bytes"] - 4 * st["cells"]) / st["n_items"]
            plan.close()
            if keep == "1":
                kept_per_tile = per_tile
            else:
                assert per_tile > kept_per_tile, (kind, per_tile, kept_per_tile)
    monkeypatch.delenv("BAMSIGNALS_CACHE_WINDOWS")
    # a plan does not outlive the layout it was made on
    small = Reads(ctx, cols["ref_len"], cols["ref_off"], cols["pos"], cols["flag"], cols["mapq"], cols["tlen"], end=cols["end"])
    plan = Plan(ctx, small, rg["rid"], rg["loc"], rg["len"], rg["strand"], make_params(_lib.MODE_PROFILE, binsize=1))
    plan.run_host()
    fn = _lib.load().bsig_debug_new_layout_gen
    fn.argtypes = [ctypes.c_void_p]
    assert fn(small._h) == 0
    with pytest.raises(_lib.BsigError, match="laid out again"):
        plan.run_host()
    plan.close()
    small.close()


def test_a_plan_that_is_run_again_takes_the_resolved_form(ctx, synth, monkeypatch):
    """Below 32,768 tiles a plan's first run looks its windows up inside the pileup kernel (a file-level call runs its plan
    once and must not pay a launch for nothing); a plan that is run AGAIN is a resident one: its second run looks the
    windows up in a launch of their own and keeps them, the third reads them back (bamCount included).  Every run is
    identical to the oracle; which form the next run takes shows in the plan's algorithmic bytes (the figure is
    computed once, when it is first asked for: one plan per question)."""
    from bamsignals_amd import _lib
    from bamsignals_amd.device import Plan, make_params
    from bamsignals_amd.synth import synth_ranges
    from oracle import oracle_c
    gpu, orc, cols, _, _ = synth["se"]
    rg = synth_ranges(3000, 700, cols["ref_len"], seed=99, jitter=300)

    def per_tile(plan):
        st = plan.stats()
        assert st["n_items"] < 32768
        return (st["algorithmic_bytes"] - 4 * st["cells"]) / st["n_items"]

    for mode, a in ((_lib.MODE_PROFILE, dict(binsize=1, ss=True)), (_lib.MODE_COUNT, dict(binsize=-1, ss=True)),
                    (_lib.MODE_PROFILE, dict(binsize=40)), (_lib.MODE_COVERAGE, dict())):
        want, _ = (oracle_c.coverage_core if mode == _lib.MODE_COVERAGE else oracle_c.pileup_core)(orc, rg, **a)
        asked = {}
        for runs_before_asking in (0, 1, 3):
            plan = Plan(ctx, gpu, rg["rid"], rg["loc"], rg["len"], rg["strand"], make_params(mode, **a))
            for run in range(runs_before_asking):
                assert np.array_equal(plan.run_host(), want), (mode, a, run)
            asked[runs_before_asking] = per_tile(plan)
            for run in range(3):
                assert np.array_equal(plan.run_host(), want), (mode, a, "after asking", run)
            plan.close()
        # before any run: the fused form's bytes (the index entries); once it has run: the kept windows' (48 B a tile)
        assert asked[1] == asked[3] != asked[0], (mode, a, asked)
    monkeypatch.setenv("BAMSIGNALS_CACHE_WINDOWS", "0")                  # nothing is kept: nothing to run again for
    plan = Plan(ctx, gpu, rg["rid"], rg["loc"], rg["len"], rg["strand"], make_params(_lib.MODE_PROFILE, binsize=1, ss=True))
    for run in range(2):
        plan.run_host()
    assert per_tile(plan) == asked_fused(ctx, gpu, rg)
    plan.close()


def asked_fused(ctx, gpu, rg):
    """per-tile algorithmic bytes (without the cells) of a fresh plan of bamProfile binsize=1 ss=TRUE: the fused form's"""
    from bamsignals_amd import _lib
    from bamsignals_amd.device import Plan, make_params
    plan = Plan(ctx, gpu, rg["rid"], rg["loc"], rg["len"], rg["strand"], make_params(_lib.MODE_PROFILE, binsize=1, ss=True))
    st = plan.stats()
    plan.close()
    return (st["algorithmic_bytes"] - 4 * st["cells"]) / st["n_items"]


@pytest.mark.parametrize("which,cases", [("se", PILEUP_CASES), ("pe", PE_CASES)])
def test_pileup_vs_oracle(ctx, synth, which, cases):
    from oracle import oracle_c
    gpu, orc, cols, _, _ = synth[which]
    rng = np.random.default_rng(21)
    small = _rand_ranges(rng, cols["ref_len"], 300, 3000)
    big = _rand_ranges(rng, cols["ref_len"], 12, 120_000)          # many tiles / split counts
    for rg in (small, big):
        for a in cases:
            want, woff = oracle_c.pileup_core(orc, rg, **a)
            for threads, tile in ((64, 0), (256, 512), (128, 4096)):
                got, off = _gpu(ctx, gpu, rg, "pileup", tile_cells=tile, threads=threads, **dict(a))
                assert np.array_equal(off, woff)
                assert np.array_equal(got, want), (which, a, threads, tile)


@pytest.mark.parametrize("which", ["se", "pe"])
def test_coverage_vs_oracle(ctx, synth, which):
    from oracle import oracle_c
    gpu, orc, cols, _, _ = synth[which]
    rng = np.random.default_rng(22)
    small = _rand_ranges(rng, cols["ref_len"], 300, 3000)
    big = _rand_ranges(rng, cols["ref_len"], 12, 120_000)
    cases = [dict(), dict(mapqual=25, filteredF=1024)]
    if which == "pe":
        cases += [dict(requiredF=66, tlen_filter=(0, 1000), tspan=True),
                  dict(requiredF=66, tlen_filter=(100, 200), tspan=True, mapqual=5)]
    for rg in (small, big):
        for a in cases:
            want, woff = oracle_c.coverage_core(orc, rg, **a)
            for threads, tile in ((64, 0), (256, 512), (128, 4096)):
                got, off = _gpu(ctx, gpu, rg, "coverage", tile_cells=tile, threads=threads, **dict(a))
                assert np.array_equal(off, woff)
                assert np.array_equal(got, want), (which, a, threads, tile)


def test_whole_reference_tiling_properties(ctx, synth):
    """Size-independent checks on a 1-bp tiling of every reference (BASELINE config 3 shape)."""
    from bamsignals_amd.synth import tile_ranges
    from oracle import oracle_c
    gpu, orc, cols, end, _ = synth["se"]
    tiles = tile_ranges(cols["ref_len"], 2000, strand=0)
    prof, _ = _gpu(ctx, gpu, tiles, "pileup", binsize=1)
    # every read whose 5' end is inside its reference is counted exactly once
    neg = (cols["flag"] & 16) != 0
    p5 = np.where(neg, end, cols["pos"])
    inside = (p5 >= 0) & (p5 < cols["ref_len"][cols["rid"]])
    assert int(prof.sum()) == int(inside.sum())
    # strand-split sums to the unsplit profile (vignettes/bamsignals.Rmd:148)
    ss, _ = _gpu(ctx, gpu, tiles, "pileup", binsize=1, ss=True)
    assert np.array_equal(ss.reshape(-1, 2).sum(axis=1), prof)
    # binned = sums of the per-base signal (vignettes/bamsignals.Rmd:235); 2000 % 50 == 0
    b50, off50 = _gpu(ctx, gpu, tiles, "pileup", binsize=50)
    off1 = np.concatenate([[0], np.cumsum(tiles["len"])])
    for i in (0, len(tiles["len"]) // 2, len(tiles["len"]) - 1):
        v = prof[off1[i]:off1[i + 1]]
        pad = (-len(v)) % 50
        assert np.array_equal(np.concatenate([v, np.zeros(pad, dtype=v.dtype)]).reshape(-1, 50).sum(axis=1),
                              b50[off50[i]:off50[i + 1]])
    # coverage integrates to the clipped total span of the reads
    cov, _ = _gpu(ctx, gpu, tiles, "coverage")
    L = cols["ref_len"][cols["rid"]].astype(np.int64)
    span = np.minimum(end.astype(np.int64), L - 1) - cols["pos"] + 1
    assert int(cov.astype(np.int64).sum()) == int(span.sum())
    want, _ = oracle_c.coverage_core(orc, tiles)
    assert np.array_equal(cov, want)
    # bamCount == sum of the profile
    cnt, _ = _gpu(ctx, gpu, tiles, "pileup", binsize=-1)
    sums = np.add.reduceat(prof, off1[:-1])
    assert np.array_equal(cnt, sums)


def test_empty_inputs(ctx, synth):
    from bamsignals_amd.device import Reads
    gpu, _, cols, _, _ = synth["se"]
    empty = dict(rid=np.zeros(0, np.int32), loc=np.zeros(0, np.int32), len=np.zeros(0, np.int32),
                 strand=np.zeros(0, np.int32))
    for kind, a in (("pileup", dict(binsize=1)), ("pileup", dict(binsize=-1)), ("coverage", dict())):
        got, off = _gpu(ctx, gpu, empty, kind, **a)
        assert len(got) == 0 and list(off) == [0]
    noreads = Reads(ctx, [1000, 2000], [0, 0, 0], [], [], [], [], end=[])
    rg = dict(rid=np.asarray([0, 1], np.int32), loc=np.asarray([10, 0], np.int32), len=np.asarray([100, 50], np.int32),
              strand=np.asarray([1, -1], np.int32))
    for kind, a, n in (("pileup", dict(binsize=1, ss=True), 300), ("pileup", dict(binsize=-1), 2), ("coverage", dict(), 150)):
        got, _ = _gpu(ctx, noreads, rg, kind, **a)
        assert len(got) == n and not got.any()
    noreads.close()


def test_error_paths(ctx, synth):
    from bamsignals_amd import _lib
    gpu, _, cols, _, _ = synth["se"]
    rg = dict(rid=np.asarray([0], np.int32), loc=np.asarray([10], np.int32), len=np.asarray([100], np.int32),
              strand=np.asarray([1], np.int32))
    with pytest.raises(_lib.BsigError) as e:
        _gpu(ctx, gpu, rg, "pileup", binsize=1, pe_mid=True)               # tlen_filter[1] of an empty vector
    assert e.value.code_name == "BSIG_ERR_ARG"
    with pytest.raises(_lib.BsigError) as e:
        _gpu(ctx, gpu, rg, "coverage", tspan=True, tlen_filter=(0, -5))    # ref: src/bamsignals.cpp:243
    assert e.value.code_name == "BSIG_ERR_EXT" and "negative 'ext'" in str(e.value)
    with pytest.raises(_lib.BsigError) as e:
        _gpu(ctx, gpu, dict(rg, rid=np.asarray([9], np.int32)), "pileup", binsize=1)
    assert e.value.code_name == "BSIG_ERR_CHROM"
    from bamsignals_amd.device import Plan, Reads, make_params
    with pytest.raises(_lib.BsigError) as e:
        Plan(ctx, gpu, rg["rid"], rg["loc"], rg["len"], rg["strand"], make_params(_lib.MODE_PROFILE, binsize=0))
    assert "binsize greater or equal to 1" in str(e.value)
    # unsorted columns are refused (a decrease is only allowed where the next reference starts)
    ok = Reads(ctx, [100, 100], [0, 2, 4], [50, 60, 10, 20], [0] * 4, [9] * 4, [0] * 4, end=[55, 65, 15, 25])
    ok.close()
    with pytest.raises(_lib.BsigError, match="sorted"):
        Reads(ctx, [100, 100], [0, 2, 4], [60, 50, 10, 20], [0] * 4, [9] * 4, [0] * 4, end=[65, 55, 15, 25])


def test_fuzz_small_inputs(ctx):
    """Seeded random inputs at the edges: tiny references, reads at position 0 and over the end,
    extreme spans, every parameter drawn at random.  HIP vs the C oracle, bit for bit."""
    from bamsignals_amd.device import Reads
    from oracle import oracle_c
    import os
    rng = np.random.default_rng(int(os.environ.get("BSIG_FUZZ_SEED", "2024")))
    for case in range(int(os.environ.get("BSIG_FUZZ_CASES", "60"))):
        n_ref = int(rng.integers(1, 4))
        ref_len = rng.integers(1, 70_000, n_ref).astype(np.int32)
        n = int(rng.integers(0, 4000))
        rid = np.sort(rng.integers(0, n_ref, n)).astype(np.int32)
        pos = (rng.random(n) * ref_len[rid]).astype(np.int32)
        order = np.lexsort((pos, rid))
        rid, pos = rid[order], pos[order]
        span = np.where(rng.random(n) < 0.8, rng.integers(1, 200, n),
                        np.where(rng.random(n) < 0.7, rng.integers(200, 5000, n), rng.integers(5000, 90_000, n)))
        end = (pos + span - 1).astype(np.int32)
        flag = (rng.integers(0, 2, n) * 16 + rng.integers(0, 2, n) * 1024 + rng.integers(0, 2, n) * 64
                + rng.integers(0, 2, n) * 2 + (rng.random(n) < 0.02) * 4).astype(np.uint16)
        mapq = rng.integers(0, 256, n).astype(np.uint8)
        tlen = rng.integers(-700, 700, n).astype(np.int32)
        ref_off = np.searchsorted(rid, np.arange(n_ref + 1)).astype(np.int64)
        gpu = Reads(ctx, ref_len, ref_off, pos, flag, mapq, tlen, end=end)
        orc = oracle_c.OracleReads(ref_off, pos, end, flag, mapq, tlen)
        m = int(rng.integers(1, 40))
        rg = dict(rid=rng.integers(0, n_ref, m).astype(np.int32), loc=rng.integers(-300, 70_300, m).astype(np.int32),
                  len=np.where(rng.random(m) < 0.1, 0, rng.integers(1, 9000, m)).astype(np.int32),
                  strand=rng.integers(-1, 2, m).astype(np.int32))
        pe = rng.random() < 0.5
        tf = tuple(sorted(int(x) for x in rng.integers(0, 700, 2))) if pe else ()
        common = dict(mapqual=int(rng.integers(0, 80)), requiredF=int(rng.choice([0, 2, 66])),
                      filteredF=int(rng.choice([-1, 0, 16, 1024, 1040])), tlen_filter=tf)
        pile = dict(common, binsize=int(rng.choice([-1, 1, 1, 2, 3, 10, 147, 5000])), shift=int(rng.integers(-150, 150)),
                    ss=bool(rng.integers(0, 2)), pe_mid=bool(pe and rng.integers(0, 2)))
        want, woff = oracle_c.pileup_core(orc, rg, **pile)
        got, off = _gpu(ctx, gpu, rg, "pileup", tile_cells=int(rng.choice([0, 64, 256, 1000])),
                        threads=int(rng.choice([0, 64, 128, 256])), **dict(pile))
        assert np.array_equal(off, woff) and np.array_equal(got, want), (case, pile)
        cov = dict(common, tspan=bool(pe and rng.integers(0, 2)))
        want, woff = oracle_c.coverage_core(orc, rg, **cov)
        got, off = _gpu(ctx, gpu, rg, "coverage", tile_cells=int(rng.choice([0, 64, 256, 1000])),
                        threads=int(rng.choice([0, 64, 128, 256])), **dict(cov))
        assert np.array_equal(off, woff) and np.array_equal(got, want), (case, cov)
        gpu.close()


def test_very_wide_bins(ctx, synth):
    """binsize > 16384: bins are executed as bamCount sub-intervals with global atomics; strand
    mirror, partial last bin, strand-split and paired-end midpoint must survive that path."""
    from oracle import oracle_c
    rng = np.random.default_rng(77)
    for which, extra in (("se", dict()), ("pe", dict(requiredF=66, tlen_filter=(0, 800), pe_mid=True))):
        gpu, orc, cols, _, _ = synth[which]
        m = 30
        rid = rng.integers(0, len(cols["ref_len"]), m).astype(np.int32)
        rg = dict(rid=rid, loc=(rng.random(m) * cols["ref_len"][rid] * 0.5).astype(np.int32) - 100,
                  len=rng.integers(1, 400_000, m).astype(np.int32), strand=rng.integers(-1, 2, m).astype(np.int32))
        for bs, ss, shift in ((16385, True, 0), (50_000, False, -40), (50_000, True, 120), (1_000_000, True, 7)):
            a = dict(binsize=bs, ss=ss, shift=shift, **extra)
            want, woff = oracle_c.pileup_core(orc, rg, **a)
            got, off = _gpu(ctx, gpu, rg, "pileup", **dict(a))
            assert np.array_equal(off, woff) and np.array_equal(got, want), (which, a)


def test_heavy_tile_slices(ctx, synth, monkeypatch):
    """Tiles on read hotspots are cut into slices that a second launch adds up with atomics.  With
    the threshold forced down (BAMSIGNALS_HEAVY_READS) nearly every tile takes that path; results
    must not change.  Then a real hotspot at the default threshold."""
    from bamsignals_amd.device import Reads
    from oracle import oracle_c
    gpu, orc, cols, _, _ = synth["pe"]
    rng = np.random.default_rng(5)
    rg = _rand_ranges(rng, cols["ref_len"], 120, 6000)
    for thr in ("8", "300"):
        monkeypatch.setenv("BAMSIGNALS_HEAVY_READS", thr)
        for a in (dict(binsize=1, ss=True, shift=20), dict(binsize=40, requiredF=66, tlen_filter=(0, 900), pe_mid=True),
                  dict(binsize=-1, ss=True), dict(binsize=3000), dict(binsize=70_000, ss=True)):
            want, _ = oracle_c.pileup_core(orc, rg, **a)
            got, _ = _gpu(ctx, gpu, rg, "pileup", **dict(a))
            assert np.array_equal(got, want), (thr, a)
        for a in (dict(), dict(requiredF=66, tlen_filter=(0, 900), tspan=True)):
            want, _ = oracle_c.coverage_core(orc, rg, **a)
            got, _ = _gpu(ctx, gpu, rg, "coverage", threads=128, **dict(a))
            assert np.array_equal(got, want), (thr, a)
        monkeypatch.setenv("BSIG_FUZZ_CASES", "25")
        test_fuzz_small_inputs(ctx)
    monkeypatch.delenv("BAMSIGNALS_HEAVY_READS")
    monkeypatch.delenv("BSIG_FUZZ_CASES")
    # 300,000 reads piled on 1.5 kb of a small reference, default threshold (32,768 reads)
    n = 300_000
    pos = np.sort(rng.integers(2000, 3500, n)).astype(np.int32)
    end = (pos + rng.integers(30, 150, n) - 1).astype(np.int32)
    flag = (rng.integers(0, 2, n) * 16).astype(np.uint16)
    mapq = rng.integers(0, 60, n).astype(np.uint8)
    tlen = np.zeros(n, np.int32)
    hot = Reads(ctx, [10_000], [0, n], pos, flag, mapq, tlen, end=end)
    horc = oracle_c.OracleReads([0, n], pos, end, flag, mapq, tlen)
    hr = dict(rid=np.zeros(4, np.int32), loc=np.asarray([0, 1900, 2500, 2600], np.int32),
              len=np.asarray([10_000, 2000, 300, 0], np.int32), strand=np.asarray([1, -1, 0, 1], np.int32))
    for kind, a in (("pileup", dict(binsize=1, ss=True)), ("pileup", dict(binsize=-1)), ("pileup", dict(binsize=250)),
                    ("coverage", dict(mapqual=10))):
        fn = oracle_c.coverage_core if kind == "coverage" else oracle_c.pileup_core
        want, _ = fn(horc, hr, **a)
        got, _ = _gpu(ctx, hot, hr, kind, **dict(a))
        assert np.array_equal(got, want), (kind, a)
    hot.close()


def test_pileup_kernels_keep_their_register_budget(ctx):
    """The pileup launches are bound by their vector instructions and by how many waves a SIMD holds.  Twice in round 4
    a harmless-looking change moved that: a four-read prefetch of the rare classes held across the packed class's work
    cost 19 VGPRs (a wave per SIMD), and with one more path on top the 8-wave builds spilled 22-52 registers to
    scratch (config 5's share 0.16 -> 0.22 ms).  No scratch, and at most 64 VGPRs -- eight waves per SIMD -- for the
    kernels the BASELINE configurations run."""
    import ctypes
    from bamsignals_amd import _lib
    fn = _lib.load().bsig_debug_pileup_attrs
    fn.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    names = ["k_profile resolved", "k_profile resolved, 8 waves", "k_profile fused", "k_profile_multi", "k_coverage resolved",
             "k_count_multi<4, 2>", "k_profile resolved, strands"]
    for which, name in enumerate(names):
        regs, scratch = ctypes.c_int(0), ctypes.c_int(-1)
        assert fn(which, ctypes.byref(regs), ctypes.byref(scratch)) == 0, name
        assert scratch.value == 0, (name, scratch.value)
        assert 0 < regs.value <= 64, (name, regs.value)

