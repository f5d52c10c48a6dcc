"""CPU: the oracle (C and numpy restatements) against the committed golden vectors.

The goldens in tests/golden/expected_grid.npz are outputs of the reference's own TEST oracle
(tests/testthat/utils.R:178-311) on the reference's fixture reads, over the full grid of
tests/testthat/test_methods.R:33-104.
"""
import numpy as np
import pytest

from conftest import core_args, parse_key
from oracle import oracle_c, oracle_np


def _reads(fx):
    np_reads = dict(rid=fx["bam_rid"], pos=fx["bam_pos"], end=fx["bam_end"], flag=fx["bam_flag"],
                    mapq=fx["bam_mapq"], tlen=fx["bam_tlen"])
    c_reads = oracle_c.OracleReads(fx["ref_off"], fx["bam_pos"], fx["bam_end"], fx["bam_flag"],
                                   fx["bam_mapq"], fx["bam_tlen"])
    return np_reads, c_reads


def test_fixture_facts(fixture_reads):
    fx = fixture_reads
    assert len(fx["bam_pos"]) == 99000
    assert list(fx["ref_names"]) == ["chr1", "chr2", "chr3"]
    assert list(fx["ref_len"]) == [10237, 10279, 10238]
    assert list(np.diff(fx["ref_off"])) == [33290, 32850, 32860]
    assert (fx["bam_pos"][0], fx["bam_mapq"][0], fx["bam_flag"][0], fx["bam_tlen"][0]) == (40, 35, 163, 136)
    assert int(fx["bam_mapq"].min()) == 1 and int(fx["bam_mapq"].max()) == 52
    assert np.array_equal(oracle_c.cigar_end(fx["bam_pos"], fx["bam_flag"], fx["bam_cigar_off"], fx["bam_cigar"]),
                          fx["bam_end"])


def test_c_oracle_full_grid(fixture_reads, fixture_regions, expected_grid):
    _, c_reads = _reads(fixture_reads)
    _, ranges = fixture_regions
    plus = dict(ranges, strand=np.ones_like(ranges["strand"]))
    for key, want in expected_grid.items():
        kind, p = parse_key(key)
        a = core_args(kind, p)
        if kind == "coverage":
            got, _ = oracle_c.coverage_core(c_reads, ranges, **a)
        else:
            got, _ = oracle_c.pileup_core(c_reads, plus if kind == "ff16" else ranges, **a)
        assert np.array_equal(got, want), key


def test_np_oracle_grid_sample(fixture_reads, fixture_regions, expected_grid):
    np_reads, _ = _reads(fixture_reads)
    _, ranges = fixture_regions
    plus = dict(ranges, strand=np.ones_like(ranges["strand"]))
    for key in sorted(expected_grid)[::5]:
        kind, p = parse_key(key)
        a = core_args(kind, p)
        if kind == "coverage":
            got, _ = oracle_np.coverage_core(np_reads, ranges, **a)
        else:
            got, _ = oracle_np.pileup_core(np_reads, plus if kind == "ff16" else ranges, **a)
        assert np.array_equal(got, expected_grid[key]), key


def test_maxgap_chunking_is_invisible(fixture_reads, fixture_regions):
    """overlapAndPileup's chunking (src/bamsignals.cpp:252-265) must not change results."""
    _, c_reads = _reads(fixture_reads)
    _, ranges = fixture_regions
    ref, _ = oracle_c.pileup_core(c_reads, ranges, binsize=1, ss=True, shift=33)
    for mg in (-5, 0, 7, 1 << 30):
        got, _ = oracle_c.pileup_core(c_reads, ranges, binsize=1, ss=True, shift=33, maxgap=mg)
        assert np.array_equal(got, ref)


def test_synthetic_cigars_np_vs_c():
    """D/N/I/S/=/X, unmapped-placed and zero-span reads: the two restatements must agree."""
    rng = np.random.default_rng(7)
    n = 5000
    pos = np.sort(rng.integers(0, 20000, n)).astype(np.int32)
    flag = np.where(rng.random(n) < 0.5, 16, 0).astype(np.uint16)
    flag[rng.random(n) < 0.03] |= 4
    cig, off = [], [0]
    menu = [[(100, 0)], [(50, 0), (10, 2), (50, 0)], [(40, 0), (2000, 3), (60, 0)], [(5, 4), (95, 0)],
            [(48, 0), (4, 1), (48, 0)], [(30, 7), (1, 8), (20, 7)], [(10, 4)], [(3, 5), (20, 0), (2, 6)]]
    for _ in range(n):
        for ln, op in menu[rng.integers(len(menu))]:
            cig.append(ln << 4 | op)
        off.append(len(cig))
    e_np = oracle_np.cigar_end(pos, flag, off, cig)
    e_c = oracle_c.cigar_end(pos, flag, off, cig)
    assert np.array_equal(e_np, e_c)
    assert np.all(e_np >= pos)
    mapq = rng.integers(0, 61, n).astype(np.uint8)
    tlen = rng.integers(-500, 500, n).astype(np.int32)
    rid = np.zeros(n, dtype=np.int32)
    rd_np = dict(rid=rid, pos=pos, end=e_np, flag=flag, mapq=mapq, tlen=tlen)
    rd_c = oracle_c.OracleReads([0, n], pos, e_np, flag, mapq, tlen)
    m = 40
    ranges = dict(rid=np.zeros(m, dtype=np.int32), loc=rng.integers(-50, 21000, m).astype(np.int32),
                  len=rng.integers(0, 3000, m).astype(np.int32),
                  strand=rng.integers(-1, 2, m).astype(np.int32))
    for a in (dict(binsize=1), dict(binsize=13, ss=True, shift=-40), dict(binsize=-1, ss=True, mapqual=20),
              dict(binsize=5, tlen_filter=(100, 400), pe_mid=True, shift=10)):
        x, _ = oracle_np.pileup_core(rd_np, ranges, **a)
        y, _ = oracle_c.pileup_core(rd_c, ranges, **a)
        assert np.array_equal(x, y), a
    for a in (dict(), dict(tlen_filter=(0, 450), tspan=True), dict(mapqual=30, filteredF=16)):
        x, _ = oracle_np.coverage_core(rd_np, ranges, **a)
        y, _ = oracle_c.coverage_core(rd_c, ranges, **a)
        assert np.array_equal(x, y), a


def test_error_paths():
    rd = oracle_c.OracleReads([0, 1], [5], [10], [0], [30], [0])
    rg = dict(rid=[0], loc=[0], len=[20], strand=[1])
    with pytest.raises(ValueError):
        oracle_c.pileup_core(rd, rg, pe_mid=True)          # tlen_filter[1] of an empty vector
    with pytest.raises(ValueError):
        oracle_c.coverage_core(rd, rg, tspan=True)
    with pytest.raises(ValueError):
        oracle_c.coverage_core(rd, rg, tspan=True, tlen_filter=(0, -3))   # negative ext (:243)
