"""CPU: argument normalisation of the user API (R/wrappers.R:76-98,136-143) — everything that
happens before the native call."""
import numpy as np
import pytest

from bamsignals_amd import GRanges, bamCount, bamCoverage, bamProfile
from bamsignals_amd.wrappers import _match_arg, flagMask, tlenFilter


def test_flag_mask():
    assert flagMask("ignore") == 0 and flagMask("filter") == 66 and flagMask("midpoint") == 66 and flagMask("extend") == 66


def test_tlen_filter():
    assert tlenFilter(None, "ignore") == () and tlenFilter((5, 9), "ignore") == ()
    assert tlenFilter(None, "filter") == (0, 1000)
    assert tlenFilter((50, 200), "midpoint") == (50, 200)
    for bad in ((1,), (1, 2, 3), (-1, 5), (5, -1)):
        with pytest.raises(ValueError, match="tlenFilter must be NULL or vector of 2 positive integers"):
            tlenFilter(bad, "filter")
    with pytest.raises(ValueError, match=r"tlenFilter\[1\] must be smaller or equal to tlenFilter\[2\]"):
        tlenFilter((9, 5), "filter")


def test_match_arg():
    ch = ("ignore", "filter", "midpoint")
    assert _match_arg(ch, ch, "paired.end") == "ignore"
    assert _match_arg("mid", ch, "paired.end") == "midpoint"
    assert _match_arg("filter", ch, "paired.end") == "filter"
    with pytest.raises(ValueError, match="should be one of"):
        _match_arg("extend", ch, "paired.end")


def test_argument_errors_before_any_io():
    gr = GRanges("chr1", [1, 50], width=[10, 13])
    with pytest.raises(ValueError, match="provide a binsize greater or equal to 1"):
        bamProfile("nope.bam", gr, binsize=0, verbose=False)
    with pytest.raises(TypeError, match="must provide a GRanges object"):
        bamCount("nope.bam", {"chr1": 1}, verbose=False)
    with pytest.raises(ValueError, match="should be one of"):
        bamCoverage("nope.bam", gr, paired_end="midpoint", verbose=False)
    with pytest.raises(ValueError, match="tlenFilter must be NULL"):
        bamCount("nope.bam", gr, paired_end="filter", tlenFilter=(1, 2, 3), verbose=False)


def test_granges():
    gr = GRanges(["chr2", "chr1", "chr2"], [5, 1, 9], end=[10, 4, 9], strand=["+", "-", "*"])
    assert list(gr.width) == [6, 4, 1] and list(gr.end) == [10, 4, 9] and len(gr) == 3
    levels, codes, start, width, strand = gr.flatten()
    assert levels == ["chr2", "chr1"] and list(codes) == [0, 1, 0] and list(strand) == [1, -1, 0]
    assert len(gr[1:]) == 2 and gr[1:].seqnames == ("chr1", "chr2")
    # immutable: the factor encodings are made once, so an edit must fail instead of being ignored
    with pytest.raises(AttributeError):
        gr.strand = ["+", "+", "+"]
    with pytest.raises(AttributeError):
        gr.seqnames = ["chr1"] * 3
    with pytest.raises(ValueError):
        gr.start[0] = 7
    with pytest.raises(ValueError):
        gr.width[0] = 7
    with pytest.raises(ValueError):
        GRanges("chr1", [1], width=[-2])
    with pytest.raises(ValueError):
        GRanges("chr1", [1], width=[2], strand=["x"])
