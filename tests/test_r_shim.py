"""CPU: the plain-C R shim (bamsignals_amd/r_package/src/shim.c).  R is absent from the image, so
 (1) the shim is type-checked against test-only declarations of the R C API (tests/r_stub/);
 (2) it is EXECUTED against a small stand-in runtime (tests/r_stub/r_mock.c: tagged vectors,
     attributes and S4 slots, a protect stack checked for balance, a gctorture-like reachability
     check at every allocation, R_alloc arenas, Rf_error as an unwind) linked with the real
     libbamsignals_hip.so: registration table (ref: src/bamsignals_init.c:12-24), checkList /
     fastWidth (ref: src/CountSignals.cpp:4-29; the reference's tests/testthat/
     test_CountSignals.R:20-28 shapes), the GRanges walk and its error messages (ref:
     src/bamsignals.cpp:55-135), NA arguments, writeSamAsBamAndIndex (ref: :496-534).
The calls that need a GPU are in tests/test_r_shim_gpu.py."""
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

SHIM = os.path.join(ROOT, "bamsignals_amd", "r_package", "src", "shim.c")
BAM = os.path.join(GOLDEN, "randomBam.bam")


def test_shim_compiles_against_api_declarations(tmp_path):
    out = tmp_path / "shim.o"
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-Wno-unused-parameter", "-Wno-cast-function-type", "-fsyntax-only",
           "-I", os.path.join(ROOT, "tests", "r_stub"), "-I", os.path.join(ROOT, "include"), SHIM]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert not out.exists()


@pytest.fixture(scope="module")
def R(tmp_path_factory):
    import r_mock
    return r_mock.MockR(r_mock.build(tmp_path_factory.mktemp("shim")))


def test_registration_table_matches_reference(R):
    assert R.registered() == {"bamsignals_checkList": 2, "bamsignals_fastWidth": 2, "bamsignals_pileup_core": 11,
                              "bamsignals_coverage_core": 8, "bamsignals_writeSamAsBamAndIndex": 2}
    assert R.L.mock_dynamic_symbols() == 0                       # R_useDynamicSymbols(info, FALSE)


def _sig(R, i, ss):
    """getSig of the reference's tests/testthat/test_CountSignals.R:3-7"""
    nums = np.arange((i - 1) * 4 + 1, i * 4 + 1, dtype=np.int32)
    return R.matrix(nums.reshape(2, 2, order="F"), 2) if ss else R.int(nums)


@pytest.mark.parametrize("ss", [True, False])
def test_checklist_and_fastwidth(R, ss):
    sigs = R.list([_sig(R, i, ss) for i in range(1, 5)])
    assert R.to_py(R.call("bamsignals_checkList", sigs, R.lgl(ss))).tolist() == [True]
    assert R.to_py(R.call("bamsignals_fastWidth", sigs, R.lgl(ss))).tolist() == [2 if ss else 4] * 4    # test_CountSignals.R:27
    # invalid lists (ref: src/CountSignals.cpp:8,11-12)
    assert R.to_py(R.call("bamsignals_checkList", R.list([_sig(R, 1, ss), R.real([1.0, 2.0])]), R.lgl(ss))).tolist() == [False]
    assert R.to_py(R.call("bamsignals_checkList", R.list([R.str(["a"])]), R.lgl(ss))).tolist() == [False]
    plain = R.list([R.int([1, 2, 3, 4])])
    three_rows = R.list([R.matrix(np.arange(6, dtype=np.int32).reshape(3, 2), 3)])
    cube = R.list([R.attr(R.int(np.arange(8)), "dim", R.int([2, 2, 2]))])
    for bad in (plain, three_rows, cube):
        assert R.to_py(R.call("bamsignals_checkList", bad, R.lgl(True))).tolist() == [False]
        assert R.to_py(R.call("bamsignals_checkList", bad, R.lgl(False))).tolist() == [True]
    assert R.to_py(R.call("bamsignals_checkList", R.list([]), R.lgl(ss))).tolist() == [True]
    assert R.to_py(R.call("bamsignals_fastWidth", R.list([]), R.lgl(ss))).tolist() == []
    # the Python host's CountSignals goes through the same native routines
    from bamsignals_amd.countsignals import CountSignals
    py = [np.arange(4, dtype=np.int32).reshape(2, 2) if ss else np.arange(4, dtype=np.int32) for _ in range(3)]
    assert CountSignals(py, ss).width().tolist() == [2 if ss else 4] * 3
    with pytest.raises(ValueError, match="invalid list"):
        CountSignals([np.arange(4, dtype=np.float64)], ss)


def test_na_arguments_are_errors(R):
    import r_mock
    with pytest.raises(r_mock.RError, match="'ss' must be TRUE or FALSE"):
        R.call("bamsignals_checkList", R.list([]), R.lgl(None))
    with pytest.raises(r_mock.RError, match="'ss' must be TRUE or FALSE"):
        R.call("bamsignals_fastWidth", R.list([]), R.lgl([]))
    with pytest.raises(r_mock.RError, match="must be a list"):
        R.call("bamsignals_checkList", R.int([1]), R.lgl(True))


def _args(R, gr_sexp, bam=BAM, tlen=(), mapqual=0, binsize=1, shift=0, ss=False, requiredF=0, filteredF=-1, pe_mid=False, maxgap=16385):
    return (R.str(bam), gr_sexp, R.real(list(tlen)) if len(tlen) else R.int([]), R.int([mapqual]), R.int([binsize]), R.int([shift]),
            R.lgl(ss), R.int([requiredF]), R.int([filteredF]), R.lgl(pe_mid), R.int([maxgap]))


def test_granges_walk_and_error_messages(R, tmp_path):
    """Everything that fails before a GPU is needed: the messages are the reference's."""
    import r_mock
    from bamsignals_amd import GRanges
    gr = GRanges(["chr1", "chr1", "chr3", "chr2"], [10, 500, 20, 30], width=[100, 50, 60, 70], strand=["+", "+", "-", "*"])
    with pytest.raises(r_mock.RError, match="must provide a GRanges object"):                              # ref :93-94
        R.call("bamsignals_pileup_core", *_args(R, R.granges(gr, klass=("IRanges",))))
    with pytest.raises(r_mock.RError, match="must provide a GRanges object"):
        R.call("bamsignals_coverage_core", R.str(BAM), R.int([1, 2]), R.int([]), R.int([0]), R.int([0]), R.int([-1]), R.lgl(False), R.int([16385]))
    bad = GRanges(["chr1", "chrZ"], [1, 1], width=[5, 5])
    with pytest.raises(r_mock.RError, match="chromosome chrZ not present in the bam file"):                # ref :119
        R.call("bamsignals_pileup_core", *_args(R, R.granges(bad)))
    with pytest.raises(r_mock.RError, match="Fail to open BAM file"):                                       # ref :204
        R.call("bamsignals_pileup_core", *_args(R, R.granges(gr), bam=str(tmp_path / "none.bam")))
    noidx = tmp_path / "noidx.bam"
    noidx.write_bytes(open(BAM, "rb").read())
    with pytest.raises(r_mock.RError, match="BAM indexing file is not available for file"):                # ref :209
        R.call("bamsignals_pileup_core", *_args(R, R.granges(gr), bam=str(noidx)))
    with pytest.raises(r_mock.RError, match="negative 'ext' values don't make sense"):                     # ref :243
        R.call("bamsignals_pileup_core", *_args(R, R.granges(gr), tlen=(0, -5), pe_mid=True))
    for kw, msg in ((dict(ss=None), "'ss' must be TRUE or FALSE"), (dict(pe_mid=None), "'pe_mid' must be TRUE or FALSE")):
        with pytest.raises(r_mock.RError, match=msg):
            R.call("bamsignals_pileup_core", *_args(R, R.granges(gr), **kw))
    a = list(_args(R, R.granges(gr)))
    a[4] = R.int([r_mock.NA_INTEGER])
    with pytest.raises(r_mock.RError, match="'binsize' must be a single integer"):
        R.call("bamsignals_pileup_core", *a)
    a = list(_args(R, R.granges(gr)))
    a[0] = R.str(["a.bam", "b.bam"])
    with pytest.raises(r_mock.RError, match="'bampath' must be a single string"):
        R.call("bamsignals_pileup_core", *a)
    # malformed Rle objects: character values instead of a factor, codes outside the levels, run
    # lengths that do not add up to the number of ranges
    g = R.granges(gr)
    R.attr(g, "seqnames", R.rle(R.str(["chr1"]), [4]))
    with pytest.raises(r_mock.RError, match="malformed Rle"):
        R.call("bamsignals_pileup_core", *_args(R, g))
    g = R.granges(gr)
    R.attr(g, "seqnames", R.rle(R.factor([7], ["chr1"]), [4]))
    with pytest.raises(r_mock.RError, match="malformed Rle"):
        R.call("bamsignals_pileup_core", *_args(R, g))
    g = R.granges(gr)
    R.attr(g, "strand", R.rle(R.factor([1], ["+", "-", "*"]), [3]))
    with pytest.raises(r_mock.RError, match="malformed Rle"):
        R.call("bamsignals_pileup_core", *_args(R, g))
    assert R.L.mock_protect_depth() == 0


def test_write_sam_as_bam_and_index(R, tmp_path):
    import r_mock
    sam = tmp_path / "a.sam"
    sam.write_text("@HD\tVN:1.0\tSO:coordinate\n@SQ\tSN:c1\tLN:5000\nr1\t0\tc1\t10\t30\t20M\t*\t0\t0\t*\t*\nr2\t16\tc1\t100\t30\t30M\t*\t0\t0\t*\t*\n")
    out = tmp_path / "a.bam"
    assert R.to_py(R.call("bamsignals_writeSamAsBamAndIndex", R.str(str(sam)), R.str(str(out)))).tolist() == [True]
    from bamsignals_amd.bamio import BamFile
    b = BamFile(str(out))
    assert b.decode()["pos"].tolist() == [9, 99]
    b.close()
    with pytest.raises(r_mock.RError, match="Fail to open SAM file"):
        R.call("bamsignals_writeSamAsBamAndIndex", R.str(str(tmp_path / "none.sam")), R.str(str(out)))


def test_the_stand_in_runtime_catches_what_it_is_there_for(tmp_path):
    """The checks that make (2) meaningful are themselves tested: a routine that forgets to PROTECT, and
    one that leaves the protect stack unbalanced, are reported."""
    import r_mock
    src = tmp_path / "bad.c"
    src.write_text('''
#include <R.h>
#include <Rinternals.h>
#include <R_ext/Rdynload.h>
SEXP bad_unprotected(SEXP a, SEXP b) { SEXP x = Rf_allocVector(INTSXP, 4); SEXP y = Rf_allocVector(INTSXP, 4); INTEGER(x)[0] = 1; return y; }
SEXP bad_imbalance(SEXP a, SEXP b) { SEXP x = PROTECT(Rf_allocVector(INTSXP, 4)); return x; }
SEXP good(SEXP a, SEXP b) { SEXP x = PROTECT(Rf_allocVector(VECSXP, 1)); SET_VECTOR_ELT(x, 0, Rf_allocVector(INTSXP, 2)); UNPROTECT(1); return x; }
static const R_CallMethodDef m[] = {{"bad_unprotected", (DL_FUNC)&bad_unprotected, 2}, {"bad_imbalance", (DL_FUNC)&bad_imbalance, 2}, {"good", (DL_FUNC)&good, 2}, {NULL, NULL, 0}};
void R_init_bamsignals(DllInfo *info) { R_registerRoutines(info, NULL, m, NULL, NULL); R_useDynamicSymbols(info, FALSE); }
''')
    so = tmp_path / "libbad.so"
    subprocess.check_call(["gcc", "-std=gnu99", "-O1", "-shared", "-fPIC", "-Wno-cast-function-type", "-I", os.path.join(ROOT, "tests", "r_stub"),
                           "-o", str(so), str(src), os.path.join(ROOT, "tests", "r_stub", "r_mock.c")])
    Rb = r_mock.MockR(str(so))
    with pytest.raises(r_mock.RViolation, match="not protected"):
        Rb.call("bad_unprotected", Rb.int([1]), Rb.int([2]))
    with pytest.raises(r_mock.RViolation, match="imbalance"):
        Rb.call("bad_imbalance", Rb.int([1]), Rb.int([2]))
    assert Rb.to_py(Rb.call("good", Rb.int([1]), Rb.int([2])))[0].tolist() == [0, 0]


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="the reference checkout is only present in the build container")
def test_graft_script_swaps_only_the_native_half(tmp_path):
    """r_package/graft_into_reference.sh on a scratch copy of the reference: R/, man/, tests/ untouched,
    src/ = shim.c + Makevars, Rcpp/Rhtslib gone from DESCRIPTION/NAMESPACE, and the grafted shim registers
    the names R/RcppExports.R calls."""
    import filecmp
    import re
    import shutil
    ref = tmp_path / "bamsignals"
    shutil.copytree("/root/reference", ref)
    subprocess.check_call(["sh", os.path.join(ROOT, "bamsignals_amd", "r_package", "graft_into_reference.sh"), str(ref), ROOT])
    assert sorted(os.listdir(ref / "src")) == ["Makevars", "shim.c"]
    for sub in ("R", "man", "vignettes", "inst"):
        cmp = filecmp.dircmp("/root/reference/" + sub, ref / sub)
        assert not cmp.diff_files and not cmp.left_only and not cmp.right_only, sub
    assert "Rcpp" not in (ref / "NAMESPACE").read_text() and "Rhtslib" not in (ref / "DESCRIPTION").read_text()
    called = set(re.findall(r"\.Call\('(bamsignals_\w+)'", (ref / "R" / "RcppExports.R").read_text()))
    registered = set(re.findall(r'\{"(bamsignals_\w+)", \(DL_FUNC\)', (ref / "src" / "shim.c").read_text()))
    assert called == registered and len(called) == 5
    assert {"test_methods.R", "test_CountSignals.R", "test_golden.R", "test_vignette_invariants.R"} <= set(os.listdir(ref / "tests" / "testthat"))
