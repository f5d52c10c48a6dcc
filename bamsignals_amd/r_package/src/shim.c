/* Plain-C .Call shim: the R side of the drop-in boundary.
 *
 * Replaces the reference's generated Rcpp glue (src/RcppExports.cpp:10-82) and registration
 * (src/bamsignals_init.c:1-24) with hand-written C that only (1) pulls flat arrays out of the
 * R objects, (2) calls the C ABI of libbamsignals_hip.so (include/bamsignals_abi.h) and
 * (3) wraps the flat int32 result into the R list the reference returns
 * (allocateList, src/bamsignals.cpp:139-192).  All R API calls happen on the calling thread.
 *
 * Not compiled in the build image (R is absent there); compiled by R CMD INSTALL on a box
 * that has R, see INTEGRATION.md.
 */
#include <R.h>
#include <Rinternals.h>
#include <R_ext/Rdynload.h>
#include <string.h>

#include "bamsignals_abi.h"

/* flat view of a GRanges (ref: parseRegions, src/bamsignals.cpp:92-135) */
typedef struct {
    R_xlen_t n;
    int *start, *width;          /* ranges@start, ranges@width (borrowed from R) */
    int *seq_code;               /* per range, 0-based into levels (R_alloc)     */
    int *strand;                 /* per range, +1 / -1 / 0          (R_alloc)     */
    int n_levels;
    const char **levels;         /* seqnames levels                  (R_alloc)    */
} flat_ranges;

static SEXP slot(SEXP obj, const char *name) { return R_do_slot(obj, Rf_install(name)); }

/* expands a factor-Rle (values = integer codes with "levels", lengths) to one code per range */
static void expand_rle(SEXP rle, R_xlen_t n, int *out, SEXP *levels_out)
{
    SEXP values = slot(rle, "values"), lengths = slot(rle, "lengths");
    *levels_out = Rf_getAttrib(values, R_LevelsSymbol);
    const int *v = INTEGER(values), *l = INTEGER(lengths);
    R_xlen_t k = 0, nrun = XLENGTH(values);
    for (R_xlen_t r = 0; r < nrun; ++r)
        for (int j = 0; j < l[r] && k < n; ++j) out[k++] = v[r] - 1;
    if (k != n) Rf_error("malformed Rle in the GRanges object");
}

static void flatten(SEXP gr, flat_ranges *f)
{
    if (!Rf_inherits(gr, "GRanges")) Rf_error("must provide a GRanges object");   /* ref :93-94 */
    SEXP ranges = slot(gr, "ranges");
    SEXP start = slot(ranges, "start"), width = slot(ranges, "width");
    f->n = XLENGTH(start);
    f->start = INTEGER(start);
    f->width = INTEGER(width);
    f->seq_code = (int *)R_alloc((size_t)f->n + 1, sizeof(int));
    f->strand = (int *)R_alloc((size_t)f->n + 1, sizeof(int));
    SEXP seq_levels, strand_levels;
    expand_rle(slot(gr, "seqnames"), f->n, f->seq_code, &seq_levels);
    expand_rle(slot(gr, "strand"), f->n, f->strand, &strand_levels);
    f->n_levels = (int)XLENGTH(seq_levels);
    f->levels = (const char **)R_alloc((size_t)f->n_levels + 1, sizeof(char *));
    for (int k = 0; k < f->n_levels; ++k) f->levels[k] = CHAR(STRING_ELT(seq_levels, k));
    /* strand codes -> +1 / -1 / 0 by level NAME (ref :123-129) */
    int n_sl = (int)XLENGTH(strand_levels);
    int *map = (int *)R_alloc((size_t)n_sl + 1, sizeof(int));
    for (int k = 0; k < n_sl; ++k) {
        const char *s = CHAR(STRING_ELT(strand_levels, k));
        map[k] = strcmp(s, "-") == 0 ? -1 : strcmp(s, "+") == 0 ? 1 : 0;
    }
    for (R_xlen_t i = 0; i < f->n; ++i) f->strand[i] = map[f->strand[i]];
}

static SEXP sense_dimnames(void)
{
    SEXP dn = PROTECT(Rf_allocVector(VECSXP, 2));
    SEXP rn = PROTECT(Rf_allocVector(STRSXP, 2));
    SET_STRING_ELT(rn, 0, Rf_mkChar("sense"));
    SET_STRING_ELT(rn, 1, Rf_mkChar("antisense"));
    SET_VECTOR_ELT(dn, 0, rn);
    UNPROTECT(2);
    return dn;
}

static SEXP int_matrix2(R_xlen_t ncol, SEXP dimnames)
{
    SEXP m = PROTECT(Rf_allocMatrix(INTSXP, 2, (int)ncol));
    Rf_setAttrib(m, R_DimNamesSymbol, dimnames);
    UNPROTECT(1);
    return m;
}

/* flat result -> list of vectors / 2 x w matrices, element i <-> range i (ref :172-190) */
static SEXP wrap_signals(const int32_t *flat, const int64_t *off, R_xlen_t n, int ss)
{
    SEXP res = PROTECT(Rf_allocVector(VECSXP, n));
    SEXP dn = PROTECT(ss ? sense_dimnames() : R_NilValue);
    for (R_xlen_t i = 0; i < n; ++i) {
        const int64_t cells = off[i + 1] - off[i];
        SEXP v = ss ? int_matrix2(cells / 2, dn) : Rf_allocVector(INTSXP, cells);
        SET_VECTOR_ELT(res, i, v);
        if (cells) memcpy(INTEGER(v), flat + off[i], (size_t)cells * sizeof(int32_t));
    }
    UNPROTECT(2);
    return res;
}

static int32_t *tlen_vec(SEXP x, int *n, SEXP *keep)
{
    *keep = PROTECT(Rf_coerceVector(x, INTSXP));      /* doubles such as c(0,1000) -> int */
    *n = (int)XLENGTH(*keep);
    return INTEGER(*keep);
}

SEXP bamsignals_pileup_core(SEXP bampath, SEXP gr, SEXP tlen_filter, SEXP mapqual, SEXP binsize,
                            SEXP shift, SEXP ss, SEXP requiredF, SEXP filteredF, SEXP pe_mid, SEXP maxgap)
{
    flat_ranges f;
    flatten(gr, &f);
    const int bs = Rf_asInteger(binsize), strand_specific = Rf_asLogical(ss) == TRUE;
    SEXP keep;
    int ntf;
    int32_t *tf = tlen_vec(tlen_filter, &ntf, &keep);
    int64_t *off = (int64_t *)R_alloc((size_t)f.n + 1, sizeof(int64_t));
    const int64_t cells = bsig_layout(f.n, f.width, bs, strand_specific, off);
    /* R_alloc'ed: released by R when the .Call returns, also on an R error (no leak on longjmp) */
    int32_t *flat = (int32_t *)R_alloc((size_t)(cells > 0 ? cells : 1), sizeof(int32_t));
    const int rc = bsig_pileup_core(CHAR(STRING_ELT(bampath, 0)), f.n, f.seq_code, f.n_levels, f.levels,
                                    f.start, f.width, f.strand, tf, ntf, Rf_asInteger(mapqual), bs,
                                    Rf_asInteger(shift), strand_specific, Rf_asInteger(requiredF),
                                    Rf_asInteger(filteredF), Rf_asLogical(pe_mid) == TRUE,
                                    Rf_asInteger(maxgap), -1, flat, off);
    if (rc != BSIG_OK) Rf_error("%s", bsig_last_error());
    SEXP res;
    if (bs <= 0) {                                    /* bamCount: list(vector) or list(2 x n) (ref :148-169) */
        res = PROTECT(Rf_allocVector(VECSXP, 1));
        SEXP v = strand_specific ? int_matrix2(f.n, PROTECT(sense_dimnames())) : Rf_allocVector(INTSXP, f.n);
        SET_VECTOR_ELT(res, 0, v);
        if (strand_specific) UNPROTECT(1);
        if (cells) memcpy(INTEGER(v), flat, (size_t)cells * sizeof(int32_t));
    } else {
        res = PROTECT(wrap_signals(flat, off, f.n, strand_specific));
    }
    UNPROTECT(2);
    return res;
}

SEXP bamsignals_coverage_core(SEXP bampath, SEXP gr, SEXP tlen_filter, SEXP mapqual, SEXP requiredF,
                              SEXP filteredF, SEXP tspan, SEXP maxgap)
{
    flat_ranges f;
    flatten(gr, &f);
    SEXP keep;
    int ntf;
    int32_t *tf = tlen_vec(tlen_filter, &ntf, &keep);
    int64_t *off = (int64_t *)R_alloc((size_t)f.n + 1, sizeof(int64_t));
    const int64_t cells = bsig_layout(f.n, f.width, 1, 0, off);
    /* R_alloc'ed: released by R when the .Call returns, also on an R error (no leak on longjmp) */
    int32_t *flat = (int32_t *)R_alloc((size_t)(cells > 0 ? cells : 1), sizeof(int32_t));
    const int rc = bsig_coverage_core(CHAR(STRING_ELT(bampath, 0)), f.n, f.seq_code, f.n_levels, f.levels,
                                      f.start, f.width, f.strand, tf, ntf, Rf_asInteger(mapqual),
                                      Rf_asInteger(requiredF), Rf_asInteger(filteredF),
                                      Rf_asLogical(tspan) == TRUE, Rf_asInteger(maxgap), -1, flat, off);
    if (rc != BSIG_OK) Rf_error("%s", bsig_last_error());
    SEXP res = PROTECT(wrap_signals(flat, off, f.n, 0));
    UNPROTECT(2);
    return res;
}

/* checkList (ref: src/CountSignals.cpp:4-16) */
SEXP bamsignals_checkList(SEXP l, SEXP ss)
{
    const int strand_specific = Rf_asLogical(ss) == TRUE;
    const R_xlen_t n = XLENGTH(l);
    for (R_xlen_t i = 0; i < n; ++i) {
        SEXP el = VECTOR_ELT(l, i);
        if (TYPEOF(el) != INTSXP) return Rf_ScalarLogical(FALSE);
        if (strand_specific) {
            SEXP d = Rf_getAttrib(el, R_DimSymbol);
            if (TYPEOF(d) != INTSXP || XLENGTH(d) != 2 || INTEGER(d)[0] != 2) return Rf_ScalarLogical(FALSE);
        }
    }
    return Rf_ScalarLogical(TRUE);
}

/* fastWidth (ref: src/CountSignals.cpp:19-29) */
SEXP bamsignals_fastWidth(SEXP l, SEXP ss)
{
    const int div = Rf_asLogical(ss) == TRUE ? 2 : 1;
    const R_xlen_t n = XLENGTH(l);
    SEXP w = PROTECT(Rf_allocVector(INTSXP, n));
    for (R_xlen_t i = 0; i < n; ++i) INTEGER(w)[i] = (int)(XLENGTH(VECTOR_ELT(l, i)) / div);
    UNPROTECT(1);
    return w;
}

/* writeSamAsBamAndIndex (ref: src/bamsignals.cpp:496-534) */
SEXP bamsignals_writeSamAsBamAndIndex(SEXP sampath, SEXP bampath)
{
    if (bsig_write_sam_as_bam_and_index(CHAR(STRING_ELT(sampath, 0)), CHAR(STRING_ELT(bampath, 0))) != BSIG_OK)
        Rf_error("%s", bsig_last_error());
    return Rf_ScalarLogical(TRUE);
}

static const R_CallMethodDef call_methods[] = {
    {"bamsignals_checkList", (DL_FUNC)&bamsignals_checkList, 2},
    {"bamsignals_fastWidth", (DL_FUNC)&bamsignals_fastWidth, 2},
    {"bamsignals_pileup_core", (DL_FUNC)&bamsignals_pileup_core, 11},
    {"bamsignals_coverage_core", (DL_FUNC)&bamsignals_coverage_core, 8},
    {"bamsignals_writeSamAsBamAndIndex", (DL_FUNC)&bamsignals_writeSamAsBamAndIndex, 2},
    {NULL, NULL, 0}
};

void R_init_bamsignals(DllInfo *info)
{
    R_registerRoutines(info, NULL, call_methods, NULL, NULL);
    R_useDynamicSymbols(info, FALSE);
}
