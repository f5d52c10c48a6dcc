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

/* a logical scalar argument: TRUE or FALSE.  (Rcpp's as<bool> turns NA into TRUE, an accident of
 * NA_LOGICAL being a non-zero int; here NA is an argument error.) */
static int flag_arg(SEXP x, const char *name)
{
    const int v = Rf_asLogical(x);
    if (v != TRUE && v != FALSE) Rf_error("'%s' must be TRUE or FALSE", name);
    return v == TRUE;
}

/* an integer scalar argument (NA is an argument error) */
static int int_arg(SEXP x, const char *name)
{
    const int v = Rf_asInteger(x);
    if (v == NA_INTEGER) Rf_error("'%s' must be a single integer", name);
    return v;
}

/* expands a factor-Rle (values = integer codes with "levels", lengths) to one code per range
 * (ref: RleIter, src/bamsignals.cpp:55-89) */
static void expand_rle(SEXP rle, R_xlen_t n, int *out, SEXP *levels_out)
{
    SEXP values = slot(rle, "values"), lengths = slot(rle, "lengths");
    SEXP levels = Rf_getAttrib(values, R_LevelsSymbol);
    if (TYPEOF(values) != INTSXP || TYPEOF(lengths) != INTSXP || TYPEOF(levels) != STRSXP ||
        XLENGTH(values) != XLENGTH(lengths))
        Rf_error("malformed Rle in the GRanges object (seqnames and strand must be factor-Rle)");
    *levels_out = levels;
    const int *v = INTEGER(values), *l = INTEGER(lengths);
    const R_xlen_t nrun = XLENGTH(values), n_lev = XLENGTH(levels);
    R_xlen_t k = 0;
    for (R_xlen_t r = 0; r < nrun; ++r) {
        if (v[r] < 1 || v[r] > n_lev || l[r] < 0) Rf_error("malformed Rle in the GRanges object");
        for (int j = 0; j < l[r] && k < n; ++j) out[k++] = v[r] - 1;
    }
    if (k != n) Rf_error("malformed Rle in the GRanges object");
}

static void flatten(SEXP gr, flat_ranges *f)
{
    if (!Rf_inherits(gr, "GRanges")) Rf_error("must provide a GRanges object");   /* ref :93-94 */
    SEXP ranges = slot(gr, "ranges");
    SEXP start = slot(ranges, "start"), width = slot(ranges, "width");
    if (TYPEOF(start) != INTSXP || TYPEOF(width) != INTSXP || XLENGTH(start) != XLENGTH(width))
        Rf_error("malformed ranges in the GRanges object");
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

/* allocateList (ref: src/bamsignals.cpp:139-192): the list of per-range vectors / 2 x w matrices, made BEFORE
 * the counting, element i <-> range i; dst[i] = the payload of element i (where the native side counts into, as
 * the reference's GArray.array does, ref :164,181,186).  R zero-initialises nothing: the native side writes every
 * cell of every element. */
static SEXP alloc_signals(const int64_t *off, R_xlen_t n, int ss, int32_t **dst)
{
    SEXP res = PROTECT(Rf_allocVector(VECSXP, n));
    SEXP dn = PROTECT(ss ? sense_dimnames() : R_NilValue);
    for (R_xlen_t i = 0; i < n; ++i) {
        const int64_t cells = off[i + 1] - off[i];
        SEXP v = ss ? int_matrix2(cells / 2, dn) : Rf_allocVector(INTSXP, cells);
        SET_VECTOR_ELT(res, i, v);
        dst[i] = cells ? INTEGER(v) : NULL;
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

static const char *path_arg(SEXP x, const char *name)
{
    if (TYPEOF(x) != STRSXP || XLENGTH(x) != 1) Rf_error("'%s' must be a single string", name);
    return CHAR(STRING_ELT(x, 0));
}

SEXP bamsignals_pileup_core(SEXP bampath, SEXP gr, SEXP tlen_filter, SEXP mapqual, SEXP binsize,
                            SEXP shift, SEXP ss, SEXP requiredF, SEXP filteredF, SEXP pe_mid, SEXP maxgap)
{
    flat_ranges f;
    flatten(gr, &f);
    const int bs = int_arg(binsize, "binsize"), strand_specific = flag_arg(ss, "ss");
    const int mid = flag_arg(pe_mid, "pe_mid");
    const char *path = path_arg(bampath, "bampath");
    SEXP keep;
    int ntf;
    int32_t *tf = tlen_vec(tlen_filter, &ntf, &keep);
    int64_t *off = (int64_t *)R_alloc((size_t)f.n + 1, sizeof(int64_t));
    bsig_layout(f.n, f.width, bs, strand_specific, off);
    /* the result is allocated first and counted into in place (ref: allocateList, then the pileup, :451-459): one
     * copy of the result in host memory, no flat staging buffer.  The list stays protected across the native call;
     * on an error Rf_error unwinds and R collects it. */
    SEXP res;
    int32_t **dst;
    if (bs <= 0) {                                    /* bamCount: list(vector) or list(2 x n) (ref :148-169) */
        res = PROTECT(Rf_allocVector(VECSXP, 1));
        SEXP v = strand_specific ? int_matrix2(f.n, PROTECT(sense_dimnames())) : Rf_allocVector(INTSXP, f.n);
        SET_VECTOR_ELT(res, 0, v);
        if (strand_specific) UNPROTECT(1);
        dst = (int32_t **)R_alloc(1, sizeof(int32_t *));
        dst[0] = INTEGER(v);
    } else {
        dst = (int32_t **)R_alloc((size_t)f.n + 1, sizeof(int32_t *));
        res = PROTECT(alloc_signals(off, f.n, strand_specific, dst));
    }
    const int rc = bsig_pileup_core_into(path, f.n, f.seq_code, f.n_levels, f.levels,
                                         f.start, f.width, f.strand, tf, ntf, int_arg(mapqual, "mapqual"), bs,
                                         int_arg(shift, "shift"), strand_specific, int_arg(requiredF, "requiredF"),
                                         int_arg(filteredF, "filteredF"), mid,
                                         int_arg(maxgap, "maxgap"), -1, dst);
    if (rc != BSIG_OK) Rf_error("%s", bsig_last_error());
    UNPROTECT(2);
    return res;
}

SEXP bamsignals_coverage_core(SEXP bampath, SEXP gr, SEXP tlen_filter, SEXP mapqual, SEXP requiredF,
                              SEXP filteredF, SEXP tspan, SEXP maxgap)
{
    flat_ranges f;
    flatten(gr, &f);
    const int span = flag_arg(tspan, "tspan");
    const char *path = path_arg(bampath, "bampath");
    SEXP keep;
    int ntf;
    int32_t *tf = tlen_vec(tlen_filter, &ntf, &keep);
    int64_t *off = (int64_t *)R_alloc((size_t)f.n + 1, sizeof(int64_t));
    bsig_layout(f.n, f.width, 1, 0, off);
    int32_t **dst = (int32_t **)R_alloc((size_t)f.n + 1, sizeof(int32_t *));
    SEXP res = PROTECT(alloc_signals(off, f.n, 0, dst));      /* allocated first, counted into in place (ref :481-492) */
    const int rc = bsig_coverage_core_into(path, f.n, f.seq_code, f.n_levels, f.levels,
                                           f.start, f.width, f.strand, tf, ntf, int_arg(mapqual, "mapqual"),
                                           int_arg(requiredF, "requiredF"), int_arg(filteredF, "filteredF"),
                                           span, int_arg(maxgap, "maxgap"), -1, dst);
    if (rc != BSIG_OK) Rf_error("%s", bsig_last_error());
    UNPROTECT(2);
    return res;
}

/* checkList (ref: src/CountSignals.cpp:4-16): what R knows about every element goes to the native
 * check as flat arrays */
SEXP bamsignals_checkList(SEXP l, SEXP ss)
{
    const int strand_specific = flag_arg(ss, "ss");
    if (TYPEOF(l) != VECSXP) Rf_error("'signals' must be a list");
    const R_xlen_t n = XLENGTH(l);
    int32_t *is_int = (int32_t *)R_alloc((size_t)n + 1, sizeof(int32_t));
    int32_t *n_dim = (int32_t *)R_alloc((size_t)n + 1, sizeof(int32_t));
    int32_t *dim0 = (int32_t *)R_alloc((size_t)n + 1, sizeof(int32_t));
    for (R_xlen_t i = 0; i < n; ++i) {
        SEXP el = VECTOR_ELT(l, i);
        SEXP d = Rf_getAttrib(el, R_DimSymbol);
        is_int[i] = TYPEOF(el) == INTSXP;
        n_dim[i] = TYPEOF(d) == INTSXP ? (int32_t)XLENGTH(d) : 0;
        dim0[i] = n_dim[i] > 0 ? INTEGER(d)[0] : 0;
    }
    return Rf_ScalarLogical(bsig_check_list(n, is_int, n_dim, dim0, strand_specific) ? TRUE : FALSE);
}

/* fastWidth (ref: src/CountSignals.cpp:19-29) */
SEXP bamsignals_fastWidth(SEXP l, SEXP ss)
{
    const int strand_specific = flag_arg(ss, "ss");
    if (TYPEOF(l) != VECSXP) Rf_error("'signals' must be a list");
    const R_xlen_t n = XLENGTH(l);
    int64_t *len = (int64_t *)R_alloc((size_t)n + 1, sizeof(int64_t));
    for (R_xlen_t i = 0; i < n; ++i) len[i] = (int64_t)XLENGTH(VECTOR_ELT(l, i));
    SEXP w = PROTECT(Rf_allocVector(INTSXP, n));
    bsig_fast_width(n, len, strand_specific, INTEGER(w));
    UNPROTECT(1);
    return w;
}

/* writeSamAsBamAndIndex (ref: src/bamsignals.cpp:496-534) */
SEXP bamsignals_writeSamAsBamAndIndex(SEXP sampath, SEXP bampath)
{
    if (bsig_write_sam_as_bam_and_index(path_arg(sampath, "sampath"), path_arg(bampath, "bampath")) != BSIG_OK)
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
