"""CPU: the plain-C R shim is type-checked against test-only declarations of the R C API
(tests/r_stub/, R itself is absent from the image) and its registration table is compared with the
reference's (name, arity) list (ref: src/bamsignals_init.c:12-19)."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "bamsignals_amd", "r_package", "src", "shim.c")


def test_shim_compiles_against_api_declarations(tmp_path):
    out = tmp_path / "shim.o"
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-Wno-unused-parameter", "-Wno-cast-function-type", "-fsyntax-only",
           "-I", os.path.join(ROOT, "tests", "r_stub"), "-I", os.path.join(ROOT, "include"), SHIM]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert not out.exists()


def test_registration_table_matches_reference():
    src = open(SHIM).read()
    table = dict((m.group(1), int(m.group(2))) for m in re.finditer(r'\{"(bamsignals_\w+)", \(DL_FUNC\)&\w+, (\d+)\}', src))
    assert table == {"bamsignals_checkList": 2, "bamsignals_fastWidth": 2, "bamsignals_pileup_core": 11,
                     "bamsignals_coverage_core": 8, "bamsignals_writeSamAsBamAndIndex": 2}
    assert "R_useDynamicSymbols(info, FALSE)" in src and "void R_init_bamsignals(DllInfo *info)" in src
    # the R stubs call the routines by those names, with the reference's argument order
    native = open(os.path.join(ROOT, "bamsignals_amd", "r_package", "R", "native.R")).read()
    for name in table:
        assert f'"{name}"' in native
    assert re.search(r'"bamsignals_pileup_core", PACKAGE = "bamsignals", bampath, gr, tlen_filter, mapqual,\s+binsize, shift, ss, requiredF, filteredF, pe_mid, maxgap', native)
