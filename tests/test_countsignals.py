"""CPU: the CountSignals container, mirroring tests/testthat/test_CountSignals.R:1-64."""
import numpy as np
import pytest

from bamsignals_amd import CountSignals


def get_sig(n, ss):
    # getSig, test_CountSignals.R:3-7 (0-based here)
    if ss:
        return np.arange(1, 2 * n + 1, dtype=np.int32).reshape(n, 2).T
    return np.arange(1, n + 1, dtype=np.int32)


@pytest.mark.parametrize("ss", [True, False])
def test_countsignals(ss):
    widths = [5, 7, 3, 10]
    sigs = [get_sig(w, ss) for w in widths]
    n = CountSignals(sigs, ss)
    assert len(n) == 4                                            # :24
    assert list(n.width()) == widths                              # :26
    repr(n)                                                       # show runs, :28
    repr(n[[]])                                                   # n[c()], :30
    assert len(n[[]]) == 0
    with pytest.raises(IndexError):                               # n[5] errors, :40
        n[4]
    for i in range(4):                                            # accessor, :43-46
        assert np.array_equal(n[i], sigs[i])
    sub = n[[0, 2]]                                               # subsetting, :48-52
    assert isinstance(sub, CountSignals) and len(sub) == 2
    assert np.array_equal(sub[1], sigs[2])
    assert all(np.array_equal(a, b) for a, b in zip(n.as_list(), sigs))   # as.list, :54
    with pytest.raises(ValueError, match="all signals must have the same length"):
        n.alignSignals()
    same = CountSignals([get_sig(6, ss) for _ in range(3)], ss)   # alignSignals, :56-63
    arr = same.alignSignals()
    assert arr.shape == ((2, 6, 3) if ss else (6, 3))
    assert np.array_equal(arr[..., 1], get_sig(6, ss))


def test_validity():
    with pytest.raises(ValueError, match="invalid list"):
        CountSignals([np.arange(3, dtype=np.float64)], False)      # not integer: checkList
    with pytest.raises(ValueError, match="invalid list"):
        CountSignals([np.arange(6, dtype=np.int32)], True)         # ss needs 2-row matrices
    with pytest.raises(ValueError, match="invalid list"):
        CountSignals([np.zeros((3, 2), dtype=np.int32)], True)
    with pytest.raises(ValueError, match="invalid ss slot"):
        CountSignals([], None)
    c = CountSignals([np.zeros((2, 0), dtype=np.int32), np.ones((2, 3), dtype=np.int32)], True)
    assert list(c.width()) == [0, 3]
    with pytest.raises(ValueError):
        c[1][0, 0] = 7                                             # a CountSignals object is read-only
