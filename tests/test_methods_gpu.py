"""GPU: bamCount / bamProfile / bamCoverage on the reference's fixture BAM, written to read like
the reference's tests/testthat/test_methods.R:33-104.  The expected values are the outputs of the
reference's own R test oracle (tests/golden/make_golden.py)."""
import itertools
import os
import warnings

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

bampath = os.path.join(GOLDEN, "randomBam.bam")


@pytest.fixture(scope="module")
def regions(fixture_regions):
    from bamsignals_amd import GRanges
    reg, _ = fixture_regions
    return GRanges(reg["chrom"], reg["start"], width=reg["width"], strand=reg["strand"])


def _key(**kw):
    tf = kw.pop("tlenFilter")
    parts = [f"{k}={int(v) if isinstance(v, bool) else v}" for k, v in kw.items()]
    return ",".join(parts + ["tf=" + ("NULL" if tf is None else "50_200")])


@pytest.fixture(params=["all", "regions"], scope="module")
def decode_mode(request):
    """whole-file decode (cached in HBM) and index-driven region decode must agree"""
    from bamsignals_amd import _lib
    old = os.environ.get("BAMSIGNALS_DECODE")
    os.environ["BAMSIGNALS_DECODE"] = request.param
    _lib.load().bsig_cache_clear()
    yield request.param
    if old is None:
        os.environ.pop("BAMSIGNALS_DECODE", None)
    else:
        os.environ["BAMSIGNALS_DECODE"] = old
    _lib.load().bsig_cache_clear()


def test_bamCount_function(regions, expected_grid, decode_mode):
    from bamsignals_amd import bamCount
    for shift, mapq, ss, pe, tlenFilter in itertools.product((0, 100), (0, 100), (False, True),
                                                             ("ignore", "filter", "midpoint"), (None, (50, 200))):
        want = expected_grid["count|" + _key(shift=shift, mapq=mapq, ss=ss, pe=pe, tlenFilter=tlenFilter)]
        got = bamCount(bampath, regions, ss=ss, shift=shift, paired_end=pe, mapqual=mapq,
                       tlenFilter=tlenFilter, verbose=False)
        if ss:
            assert got.shape == (2, len(regions))
            got = got.T.reshape(-1)
        assert np.array_equal(got, want), (shift, mapq, ss, pe, tlenFilter)


def test_bamProfile_function(regions, expected_grid, decode_mode):
    from bamsignals_amd import bamProfile
    for shift, mapq, ss, pe, tlenFilter in itertools.product((0, 100), (0, 100), (False, True),
                                                             ("ignore", "filter", "midpoint"), (None, (50, 200))):
        want = expected_grid["profile|" + _key(shift=shift, mapq=mapq, ss=ss, pe=pe, tlenFilter=tlenFilter)]
        sig = bamProfile(bampath, regions, ss=ss, shift=shift, paired_end=pe, mapqual=mapq,
                         tlenFilter=tlenFilter, verbose=False)
        assert list(sig.width()) == list(regions.width)              # vignette :134
        got = np.concatenate([(m.T.reshape(-1) if ss else m) for m in sig.as_list()])
        assert np.array_equal(got, want), (shift, mapq, ss, pe, tlenFilter)


def test_bamCoverage_function(regions, expected_grid, decode_mode):
    from bamsignals_amd import bamCoverage
    for mapq, pe, tlenFilter in itertools.product((0, 100), ("ignore", "extend"), (None, (50, 200))):
        want = expected_grid["coverage|" + _key(mapq=mapq, pe=pe, tlenFilter=tlenFilter)]
        sig = bamCoverage(bampath, regions, paired_end=pe, mapqual=mapq, tlenFilter=tlenFilter, verbose=False)
        assert np.array_equal(np.concatenate(sig.as_list()), want), (mapq, pe, tlenFilter)


def test_filtering_on_SAMFLAGS(regions, expected_grid):
    from bamsignals_amd import GRanges, bamCount
    plus = GRanges(regions.seqnames, regions.start, width=regions.width, strand="+")
    for shift, mapq, pe, tlenFilter in itertools.product((0, 100), (0, 100), ("ignore", "filter", "midpoint"),
                                                         (None, (50, 200))):
        want = expected_grid["ff16|" + _key(shift=shift, mapq=mapq, pe=pe, tlenFilter=tlenFilter)]
        got = bamCount(bampath, plus, ss=False, shift=shift, paired_end=pe, mapqual=mapq, tlenFilter=tlenFilter,
                       filteredFlag=16, verbose=False)
        assert np.array_equal(got, want)


def test_binsize_warning_and_vignette_invariants(regions):
    from bamsignals_amd import bamProfile
    with pytest.warns(UserWarning, match="not a multiple of the selected"):
        binned = bamProfile(bampath, regions, binsize=20, verbose=False)
    perbase = bamProfile(bampath, regions, verbose=False)
    assert list(binned.width()) == [int(np.ceil(w / 20)) for w in regions.width]      # vignette :224
    for b, p in zip(binned, perbase):                                                    # vignette :235
        pad = (-len(p)) % 20
        assert np.array_equal(b, np.concatenate([p, np.zeros(pad, np.int32)]).reshape(-1, 20).sum(axis=1))
    ss = bamProfile(bampath, regions, ss=True, verbose=False)
    for s, p in zip(ss, perbase):                                                        # vignette :148
        assert np.array_equal(s.sum(axis=0), p)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        bamProfile(bampath, regions[[0]], binsize=int(regions.width[0]), verbose=False)    # exact multiple: no warning


def test_errors(regions, tmp_path, capsys):
    from bamsignals_amd import GRanges, _lib, bamCount, bamProfile
    with pytest.raises(_lib.BsigError, match="chromosome chrZ not present in the bam file"):     # ref :119
        bamCount(bampath, GRanges(["chr1", "chrZ"], [1, 1], width=[5, 5]), verbose=False)
    with pytest.raises(_lib.BsigError, match="Fail to open BAM file"):                           # ref :204
        bamProfile(str(tmp_path / "none.bam"), regions, verbose=False)
    bamCount(bampath, regions, verbose=True)
    assert "Processing " + bampath in capsys.readouterr().err                                    # R/wrappers.R:182-184
    from bamsignals_amd.wrappers import last_call_timing
    t = last_call_timing()
    assert t["total"] > 0 and t["total"] >= t["plan_run_download"] > 0


def test_file_level_on_synthetic_bam_with_gapped_cigars(tmp_path):
    """BAM written by our writer (D/N/S/I CIGARs, three references, 0x400 duplicates) -> the user
    API end to end (decode, CIGAR->end on the GPU, span classes, kernels) vs the C oracle on the
    generator's columns; both decode modes."""
    from bamsignals_amd import GRanges, _lib, bamCount, bamCoverage, bamProfile, write_columns_as_bam
    from bamsignals_amd.synth import synth_ranges, synth_reads
    from oracle import oracle_c
    names = ["chrA", "chrB", "chrC"]
    cols = synth_reads(300_000, [900_000, 70_000, 400_000], seed=21, paired=True)
    bam = str(tmp_path / "syn.bam")
    write_columns_as_bam(bam, names, cols)
    rg = synth_ranges(400, 1500, cols["ref_len"], seed=5, jitter=700)
    gr = GRanges([names[r] for r in rg["rid"]], rg["loc"] + 1, width=rg["len"],
                 strand=[{1: "+", -1: "-", 0: "*"}[int(s)] for s in rg["strand"]])
    orc = oracle_c.OracleReads(cols["ref_off"], cols["pos"], cols["end"], cols["flag"], cols["mapq"], cols["tlen"])
    old = os.environ.get("BAMSIGNALS_DECODE")
    try:
        for mode in ("regions", "all"):
            os.environ["BAMSIGNALS_DECODE"] = mode
            _lib.load().bsig_cache_clear()
            sig = bamProfile(bam, gr, binsize=1, ss=True, shift=30, paired_end="midpoint", tlenFilter=(60, 400),
                             mapqual=10, verbose=False)
            want, _ = oracle_c.pileup_core(orc, rg, binsize=1, ss=True, shift=30, pe_mid=True, tlen_filter=(60, 400),
                                           requiredF=66, mapqual=10)
            assert np.array_equal(np.concatenate([m.T.reshape(-1) for m in sig]), want), mode
            cov = bamCoverage(bam, gr, paired_end="extend", filteredFlag=1024, verbose=False)
            want, _ = oracle_c.coverage_core(orc, rg, tspan=True, tlen_filter=(0, 1000), requiredF=66, filteredF=1024)
            assert np.array_equal(np.concatenate(cov.as_list()), want), mode
            cnt = bamCount(bam, gr, shift=-25, verbose=False)
            want, _ = oracle_c.pileup_core(orc, rg, binsize=-1, shift=-25)
            assert np.array_equal(cnt, want), mode
    finally:
        if old is None:
            os.environ.pop("BAMSIGNALS_DECODE", None)
        else:
            os.environ["BAMSIGNALS_DECODE"] = old
        _lib.load().bsig_cache_clear()


@pytest.mark.parametrize("devices", ["0,0", "0,0,0"])
def test_single_process_multi_gpu_sharding(regions, expected_grid, devices, monkeypatch):
    """BAMSIGNALS_DEVICES: one process deals the sorted ranges round-robin to several GPU contexts
    and reassembles the shards.  This pool has one GPU per box, so the same GPU is listed several
    times: separate contexts, streams and resident copies, same code path."""
    from bamsignals_amd import _lib, bamCount, bamCoverage, bamProfile
    monkeypatch.setenv("BAMSIGNALS_DEVICES", devices)
    _lib.load().bsig_cache_clear()
    try:
        for decode in ("all", "regions"):
            monkeypatch.setenv("BAMSIGNALS_DECODE", decode)
            _lib.load().bsig_cache_clear()
            for rep in range(2):                      # second round: reads cached on every slot
                sig = bamProfile(bampath, regions, ss=True, shift=100, paired_end="midpoint", tlenFilter=(50, 200), verbose=False)
                got = np.concatenate([m.T.reshape(-1) for m in sig.as_list()])
                assert np.array_equal(got, expected_grid["profile|shift=100,mapq=0,ss=1,pe=midpoint,tf=50_200"])
                cnt = bamCount(bampath, regions, verbose=False)
                assert np.array_equal(cnt, expected_grid["count|shift=0,mapq=0,ss=0,pe=ignore,tf=NULL"])
                cov = bamCoverage(bampath, regions, paired_end="extend", verbose=False)
                assert np.array_equal(np.concatenate(cov.as_list()), expected_grid["coverage|mapq=0,pe=extend,tf=NULL"])
    finally:
        _lib.load().bsig_cache_clear()
