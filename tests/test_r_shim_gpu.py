"""GPU: the R shim's .Call routines executed end to end (stand-in R runtime tests/r_stub/r_mock.c +
the real libbamsignals_hip.so) on the reference's fixture BAM: an S4 GRanges whose seqnames levels
are in the fixture's order (chr1, chr3, chr2 -- not the BAM's, so names must be mapped by name,
ref: src/bamsignals.cpp:113-120) goes in, the R list the reference builds (allocateList, ref:
:139-192) comes out, and is compared with the golden grid (tests/golden/expected_grid.npz, the
reference's own R test oracle on the grid of its tests/testthat/test_methods.R:33-104)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, core_args, parse_key

pytestmark = pytest.mark.gpu
BAM = os.path.join(GOLDEN, "randomBam.bam")


@pytest.fixture(scope="module")
def R(tmp_path_factory):
    import r_mock
    return r_mock.MockR(r_mock.build(tmp_path_factory.mktemp("shim")))


@pytest.fixture(scope="module")
def gr(fixture_regions):
    from bamsignals_amd import GRanges
    reg, _ = fixture_regions
    return GRanges(reg["chrom"], reg["start"], width=reg["width"], strand=reg["strand"])


def _pileup_args(R, g, a):
    tf = a["tlen_filter"]
    return (R.str(BAM), g, R.real([float(x) for x in tf]) if len(tf) else R.int([]), R.int([a["mapqual"]]), R.int([a["binsize"]]),
            R.int([a["shift"]]), R.lgl(a["ss"]), R.int([a["requiredF"]]), R.int([a["filteredF"]]), R.lgl(a["pe_mid"]), R.int([16385]))


def test_golden_grid_through_the_shim(R, gr, expected_grid):
    g = R.granges(gr, seq_levels=["chr1", "chr3", "chr2"])
    n = len(gr)
    done = 0
    for key in sorted(expected_grid):
        kind, p = parse_key(key)
        if p["mapq"] == 100:
            continue                                          # (all zeros; the other two mapq values carry the signal)
        a = core_args(kind, p)
        want = expected_grid[key]
        if kind == "coverage":
            res = R.call("bamsignals_coverage_core", R.str(BAM), g, R.real([float(x) for x in a["tlen_filter"]]) if len(a["tlen_filter"]) else R.int([]),
                         R.int([a["mapqual"]]), R.int([a["requiredF"]]), R.int([a["filteredF"]]), R.lgl(a["tspan"]), R.int([16385]))
            out = R.to_py(res)
            assert len(out) == n and all(v.ndim == 1 for v in out)
            assert [len(v) for v in out] == gr.width.tolist()
            assert np.array_equal(np.concatenate(out), want), key
        elif kind == "profile":
            res = R.call("bamsignals_pileup_core", *_pileup_args(R, g, a))
            out = R.to_py(res)
            assert len(out) == n
            if a["ss"]:                                       # 2 x width matrices, dimnames list(c("sense","antisense"), NULL) (ref :145,178-179)
                assert all(v.shape == (2, w) for v, w in zip(out, gr.width))
                assert R.dimnames(R.L.mock_list_get(res, 0)) == [["sense", "antisense"], None]
                flat = np.concatenate([v.T.reshape(-1) for v in out])
            else:
                flat = np.concatenate(out)
            assert np.array_equal(flat, want), key
        elif kind == "count":
            res = R.call("bamsignals_pileup_core", *_pileup_args(R, g, a))
            out = R.to_py(res)
            assert len(out) == 1                              # list of length one (ref :148-169)
            v = out[0]
            if a["ss"]:
                assert v.shape == (2, n) and R.dimnames(R.L.mock_list_get(res, 0)) == [["sense", "antisense"], None]
                v = v.T.reshape(-1)
            assert np.array_equal(v, want), key
        else:
            continue
        done += 1
    assert done >= 50
    assert R.L.mock_protect_depth() == 0


def test_rle_runs_zero_width_and_binned(R, gr):
    """Long runs in the Rle (sorted ranges), a zero-width range, binsize > 1: against the Python host."""
    from bamsignals_amd import GRanges, bamProfile
    order = np.lexsort((gr.start, np.asarray(gr.seqnames)))
    g2 = gr[order]
    g2 = GRanges(g2.seqnames, g2.start, width=np.where(np.arange(len(g2)) == 3, 0, g2.width), strand=["-"] * 10 + ["+"] * (len(g2) - 10))
    res = R.call("bamsignals_pileup_core", R.str(BAM), R.granges(g2, seq_levels=["chrUnused", "chr2", "chr1", "chr3"]), R.int([]), R.int([0]),
                 R.int([7]), R.int([0]), R.lgl(True), R.int([0]), R.int([-1]), R.lgl(False), R.int([16385]))
    out = R.to_py(res)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = bamProfile(BAM, g2, binsize=7, ss=True, verbose=False)
    assert len(out) == len(g2) and out[3].shape == (2, 0)
    for a, b in zip(out, want):
        assert np.array_equal(a, b)


def test_the_result_is_held_once(R, gr, expected_grid):
    """The reference allocates the R vectors first (allocateList, ref: src/bamsignals.cpp:139-192) and counts straight
    into them: one copy of the result in host memory.  So does the shim (bsig_pileup_core_into /
    bsig_coverage_core_into): the vectors of a call add up to the result, and what the call takes from R_alloc is
    the offsets and the destination pointers -- no flat staging buffer the size of the result."""
    from bamsignals_amd import GRanges
    g = R.granges(gr, seq_levels=["chr1", "chr3", "chr2"])
    n = len(gr)
    width = int(np.sum(gr.width))
    for ss in (False, True):
        key = f"profile|shift=0,mapq=0,ss={int(ss)},pe=ignore,tf=NULL"
        a = core_args("profile", parse_key(key)[1])
        res = R.call("bamsignals_pileup_core", *_pileup_args(R, g, a))
        vec, ralloc = R.last_call_bytes()
        cells = width * (2 if ss else 1)
        assert 4 * cells <= vec <= 4 * cells + 16 * n + 4096, (ss, vec, cells)       # (+ dims, dimnames, the coerced tlen filter)
        assert ralloc <= 64 * (n + 2) and ralloc < vec // 20, (ss, ralloc, vec)
        out = R.to_py(res)
        flat = np.concatenate([v.T.reshape(-1) for v in out]) if ss else np.concatenate(out)
        assert np.array_equal(flat, expected_grid[key])
    res = R.call("bamsignals_coverage_core", R.str(BAM), g, R.int([]), R.int([0]), R.int([0]), R.int([-1]), R.lgl(False), R.int([16385]))
    vec, ralloc = R.last_call_bytes()
    assert 4 * width <= vec <= 4 * width + 4096 and ralloc <= 64 * (n + 2)
    assert np.array_equal(np.concatenate(R.to_py(res)), expected_grid["coverage|mapq=0,pe=ignore,tf=NULL"])
    # a large result takes the staged route (page-locked halves, several threads moving them on range by range):
    # 6,000 x 3 kb ranges tiled over the fixture's three references = 72 MB, compared with the Python host's call
    from bamsignals_amd import bamProfile
    rng = np.random.default_rng(8)
    m = 6000
    chrom = rng.choice(["chr1", "chr2", "chr3"], m)
    big = GRanges(list(chrom), rng.integers(1, 7000, m), width=np.where(np.arange(m) % 97 == 0, 0, 3000), strand=list(rng.choice(["+", "-", "*"], m)))
    res = R.call("bamsignals_pileup_core", R.str(BAM), R.granges(big, seq_levels=["chr1", "chr3", "chr2"]), R.int([]), R.int([0]), R.int([1]),
                 R.int([0]), R.lgl(True), R.int([0]), R.int([-1]), R.lgl(False), R.int([16385]))
    vec, ralloc = R.last_call_bytes()
    cells = 2 * int(np.sum(big.width))
    assert 4 * cells <= vec <= 4 * cells + 16 * m + 4096 and ralloc <= 64 * (m + 2) and 4 * cells > (64 << 20)
    want = bamProfile(BAM, big, ss=True, verbose=False)
    for a_, b_ in zip(R.to_py(res), want):
        assert np.array_equal(a_, b_)
    assert R.L.mock_protect_depth() == 0
