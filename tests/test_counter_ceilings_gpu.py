"""GPU: the exact edges of the 16-bit tile images.

The reference counts into plain ``int`` cells (``++range.array[...]``, ref: src/bamsignals.cpp:361-362; the +/-1
coverage image, ref: :423-436) which have no ceiling.  The HIP kernels keep a tile's image in LDS as 16-bit
counters, two per dword (k_profile / k_profile_multi: unsigned; k_coverage: signed; bamCount: two 16-bit halves
per lane), which is only sound because a tile whose read windows hold more than 32,768 reads (32,767 for
coverage) is cut into slices that a second launch adds into the int32 result with integer atomics
(``heavy_reads`` in runtime.hip).  A silent carry into the neighbouring cell of a dword is exactly the bug a
bit-exact port must not have, so these tests put exactly 32,766 / 32,767 / 32,768 / 32,769 reads on ONE cell
of one tile, at the default threshold, in every kernel form, and check through ``plan.stats()["heavy_tiles"]``
that the last count under the ceiling really took the one-launch path and the first above it the slices."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

COUNTS = (32766, 32767, 32768, 32769)
P = 40_000                      # where the pile sits (0-based) on a 100-kb reference
REF = 100_000


@pytest.fixture(scope="module")
def ctx():
    from bamsignals_amd.device import Context
    c = Context(0)
    yield c
    c.close()


def _pile(n, reverse=False, span=50, extra=(), tlen=0):
    """n identical reads [P, P + span) (all forward or all reverse) plus `extra` = [(pos, span, reverse, how many)],
    as sorted columns."""
    rows = [(P, span, reverse, n)] + list(extra)
    pos = np.concatenate([np.full(k, p, np.int32) for p, _, _, k in rows])
    end = np.concatenate([np.full(k, p + s - 1, np.int32) for p, s, _, k in rows])
    flag = np.concatenate([np.full(k, 16 if r else 0, np.uint16) for _, _, r, k in rows])
    o = np.argsort(pos, kind="stable")
    pos, end, flag = pos[o], end[o], flag[o]
    m = len(pos)
    return dict(pos=pos, end=end, flag=flag, mapq=np.full(m, 30, np.uint8), tlen=np.full(m, tlen, np.int32),
                ref_off=np.asarray([0, m], np.int64))


def _both(ctx, cols):
    from bamsignals_amd.device import Reads
    from oracle import oracle_c
    gpu = Reads(ctx, [REF], cols["ref_off"], cols["pos"], cols["flag"], cols["mapq"], cols["tlen"], end=cols["end"])
    orc = oracle_c.OracleReads(cols["ref_off"], cols["pos"], cols["end"], cols["flag"], cols["mapq"], cols["tlen"])
    return gpu, orc


def _run(ctx, gpu, rg, kind, **a):
    """(result, heavy tiles of the plan)"""
    from bamsignals_amd import _lib
    from bamsignals_amd.device import Plan, make_params
    if kind == "coverage":
        prm = make_params(_lib.MODE_COVERAGE, **a)
    else:
        bs = a.pop("binsize", 1)
        prm = make_params(_lib.MODE_COUNT if bs <= 0 else _lib.MODE_PROFILE, binsize=bs, **a)
    plan = Plan(ctx, gpu, rg["rid"], rg["loc"], rg["len"], rg["strand"], prm)
    heavy = plan.stats()["heavy_tiles"]
    out = plan.run_host()
    plan.close()
    return out, heavy


def _ranges(loc, length, strand):
    loc, length, strand = np.atleast_1d(loc), np.atleast_1d(length), np.atleast_1d(strand)
    return dict(rid=np.zeros(len(loc), np.int32), loc=loc.astype(np.int32), len=length.astype(np.int32),
                strand=strand.astype(np.int32))


@pytest.fixture()
def resolved_small_launches():
    """Small launches take the fused kernels; the forms for resolved windows (k_resolve_tiles in front, and
    k_profile_multi for narrow tiles) are what launches of 32,768 tiles and more run: forced here."""
    from bamsignals_amd import _lib
    fn = _lib.load().bsig_debug_set_resolve_min
    fn.argtypes = [ctypes.c_longlong]
    fn(1)
    yield
    fn(-1)


def _check_profile_forms(ctx, n, pack_note):
    """k_profile (2-kb tile, strands merged and split), forward and reverse piles, +/- ranges."""
    from oracle import oracle_c
    ceiling = 32768
    for reverse in (False, True):
        gpu, orc = _both(ctx, _pile(n, reverse))
        # the pile's 5' end is P (forward) or P + 49 (reverse): ranges that hold it, on either strand, and one whose
        # LAST cell it is (the cell next to the image's padding)
        five = P + 49 if reverse else P
        rg = _ranges([P - 700, P - 1300, five - 1999, five], [2000, 2000, 2000, 2000], [1, -1, 0, -1])
        for ss in (False, True):
            want, _ = oracle_c.pileup_core(orc, rg, binsize=1, ss=ss)
            got, heavy = _run(ctx, gpu, rg, "pileup", binsize=1, ss=ss)
            assert want.max() == n, "the pile is not on one cell"
            assert np.array_equal(got, want), (pack_note, "profile", n, reverse, ss, np.flatnonzero(got != want)[:8])
            assert (heavy == 0) == (n <= ceiling), (pack_note, n, heavy)
        gpu.close()


@pytest.mark.parametrize("n", COUNTS)
def test_profile_pile_on_one_cell(ctx, n):
    _check_profile_forms(ctx, n, "packed class")


@pytest.mark.parametrize("n", COUNTS)
def test_profile_pile_on_one_cell_resolved_windows(ctx, n, resolved_small_launches):
    _check_profile_forms(ctx, n, "packed class, resolved windows")


@pytest.mark.parametrize("n", COUNTS)
def test_profile_pile_class0_alone(ctx, n, monkeypatch):
    """BAMSIGNALS_PACK=0: no packed class, the same reads walk class 0's read-by-read form."""
    monkeypatch.setenv("BAMSIGNALS_PACK", "0")
    _check_profile_forms(ctx, n, "class 0 alone")


@pytest.mark.parametrize("n", COUNTS)
def test_profile_multi_narrow_tiles(ctx, n, resolved_small_launches):
    """k_profile_multi: tiles of at most ~760 cells, four consecutive tiles per wave through ONE image that a
    tile's store loop clears for the next -- the pile sits in the second of four, its dword partner and the
    neighbouring tiles carry a few reads of their own."""
    from oracle import oracle_c
    near = [(P + 1, 50, False, 2), (P - 1, 50, False, 1)]
    # two data sets: all n on one cell; n - 3 on the cell with 2 + 1 reads on its two neighbours (n in the window)
    for cols, top in ((_pile(n, False), n), (_pile(n - 3, False, extra=near), n - 3)):
        gpu, orc = _both(ctx, cols)
        # 500-cell tiles fit the 2-KB image with the strands merged, 350-cell tiles with the strands split
        for width in (500, 350):
            loc = [P - 2 * width - 100, P - width + 200, P - 300, P + width - 300, P - 5000, P - width + 1, P, P - 1]
            rg = _ranges(loc, [width] * len(loc), [1, 1, -1, 1, 0, -1, 1, -1])
            for ss in (False, True):
                want, _ = oracle_c.pileup_core(orc, rg, binsize=1, ss=ss)
                got, heavy = _run(ctx, gpu, rg, "pileup", binsize=1, ss=ss)
                assert want.max() == top
                assert np.array_equal(got, want), ("multi", n, width, ss, np.flatnonzero(got != want)[:8])
                assert (heavy == 0) == (n <= 32768), (n, heavy)
        gpu.close()


@pytest.mark.parametrize("n", COUNTS)
@pytest.mark.parametrize("resolved", [False, True])
def test_coverage_piles_of_plus_and_minus_one(ctx, n, resolved):
    """k_coverage's cells are SIGNED 16-bit halves of a dword, so its ceiling is 32,767: n identical reads that
    start left of the tile pile +1 on cell 0 (the reference clamps the start, ref: src/bamsignals.cpp:423-426) and
    -1 on the one cell behind their common end; on a '-' range the two piles swap sides (ref: :431-436)."""
    from bamsignals_amd import _lib
    from oracle import oracle_c
    fn = _lib.load().bsig_debug_set_resolve_min
    fn.argtypes = [ctypes.c_longlong]
    fn(1 if resolved else -1)
    try:
        gpu, orc = _both(ctx, _pile(n, False, span=50, extra=[(P + 49, 30, True, 3), (P + 51, 30, False, 2)]))
        # range 0: starts inside the reads (clamp: +n on cell 0, -n on cell 40); range 1: holds them whole (+n at cell
        # 700, -n at 750); range 2: '-' strand, the reads overhang its END (mirror: clamp on cell 0 again); range 3:
        # the pile's -1 falls exactly one past the range's last cell (p == len: not written, ref: :426)
        rg = _ranges([P + 10, P - 700, P - 1980, P - 1950], [2000, 2000, 2000, 2000], [1, 0, -1, 1])
        want, _ = oracle_c.coverage_core(orc, rg)
        got, heavy = _run(ctx, gpu, rg, "coverage")
        assert want.max() >= n
        assert np.array_equal(got, want), ("coverage", n, resolved, np.flatnonzero(got != want)[:8])
        assert (heavy == 0) == (n + 5 <= 32767), (n, heavy)
        gpu.close()
        # the pile alone, so that the window holds exactly n reads: 32,767 is the last count on the one-launch path
        gpu, orc = _both(ctx, _pile(n, True, span=50))
        want, _ = oracle_c.coverage_core(orc, rg)
        got, heavy = _run(ctx, gpu, rg, "coverage")
        assert np.array_equal(got, want), ("coverage, pile alone", n, resolved)
        assert (heavy == 0) == (n <= 32767), (n, heavy)
        gpu.close()
    finally:
        fn(-1)


@pytest.mark.parametrize("n", COUNTS)
def test_count_both_packed_halves(ctx, n):
    """bamCount keeps ONE packed counter per lane: all reads in the low half, reverse-strand ones in the high half.
    All reads reverse: both halves fill at once; strands split and merged, '+' and '-' ranges, the one-tile
    kernel (128 threads) and the several-tiles-per-wave kernel."""
    from oracle import oracle_c
    for reverse in (True, False):
        gpu, orc = _both(ctx, _pile(n, reverse))
        rg = _ranges([P - 700, P - 1300, P + 49, P + 50, P - 20_000], [2000, 2000, 1, 300, 30_000], [1, -1, 0, 1, -1])
        for ss in (False, True):
            want, _ = oracle_c.pileup_core(orc, rg, binsize=-1, ss=ss)
            for threads in (0, 128):
                got, heavy = _run(ctx, gpu, rg, "pileup", binsize=-1, ss=ss, threads=threads)
                assert want.max() == n
                assert np.array_equal(got, want), ("count", n, reverse, ss, threads)
                assert (heavy == 0) == (n <= 32768), (n, heavy)
        gpu.close()


@pytest.mark.parametrize("n", COUNTS)
def test_pile_split_over_two_classes(ctx, n):
    """Half the pile in the packed class (span 50), half in class 1 (span 300): the ceiling is on the SUM of a
    tile's windows over the classes, and one counter is fed from two classes' loops."""
    from oracle import oracle_c
    a, b = n // 2, n - n // 2
    cols = _pile(a, False, span=50, extra=[(P, 300, False, b)])
    gpu, orc = _both(ctx, cols)
    rg = _ranges([P - 700, P - 1300], [2000, 2000], [1, -1])
    for ss in (False, True):
        want, _ = oracle_c.pileup_core(orc, rg, binsize=1, ss=ss)
        got, heavy = _run(ctx, gpu, rg, "pileup", binsize=1, ss=ss)
        assert want.max() == n and np.array_equal(got, want), ("two classes", n, ss)
        assert (heavy == 0) == (n <= 32768), (n, heavy)
    want, _ = oracle_c.coverage_core(orc, rg)
    got, heavy = _run(ctx, gpu, rg, "coverage")
    assert np.array_equal(got, want), ("two classes, coverage", n)
    assert (heavy == 0) == (n <= 32767), (n, heavy)
    gpu.close()
