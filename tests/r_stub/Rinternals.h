/* TEST-ONLY declarations of the handful of R C-API names r_package/src/shim.c uses, so that the
 * shim can be syntax- and type-checked with gcc in an image that has no R (tests/test_r_shim.py).
 * This is not R, it is never linked, and nothing is built against it for use: the real build is
 * `R CMD INSTALL` against the real <Rinternals.h>. */
#ifndef BSIG_TEST_RINTERNALS_STUB_H
#define BSIG_TEST_RINTERNALS_STUB_H
#include <stddef.h>
typedef struct SEXPREC *SEXP;
typedef ptrdiff_t R_xlen_t;
typedef int Rboolean;
#ifndef TRUE
#define TRUE 1
#define FALSE 0
#endif
enum { INTSXP = 13, STRSXP = 16, VECSXP = 19 };
extern SEXP R_NilValue, R_DimSymbol, R_DimNamesSymbol, R_LevelsSymbol;
SEXP R_do_slot(SEXP, SEXP);
SEXP Rf_install(const char *);
SEXP Rf_getAttrib(SEXP, SEXP);
SEXP Rf_setAttrib(SEXP, SEXP, SEXP);
int *INTEGER(SEXP);
R_xlen_t XLENGTH(SEXP);
int TYPEOF(SEXP);
SEXP VECTOR_ELT(SEXP, R_xlen_t);
SEXP SET_VECTOR_ELT(SEXP, R_xlen_t, SEXP);
SEXP STRING_ELT(SEXP, R_xlen_t);
void SET_STRING_ELT(SEXP, R_xlen_t, SEXP);
const char *CHAR(SEXP);
SEXP Rf_mkChar(const char *);
SEXP Rf_allocVector(unsigned int, R_xlen_t);
SEXP Rf_allocMatrix(unsigned int, int, int);
SEXP Rf_coerceVector(SEXP, unsigned int);
SEXP Rf_ScalarLogical(int);
int Rf_asInteger(SEXP);
int Rf_asLogical(SEXP);
Rboolean Rf_inherits(SEXP, const char *);
SEXP Rf_protect(SEXP);
void Rf_unprotect(int);
#define PROTECT(x) Rf_protect(x)
#define UNPROTECT(n) Rf_unprotect(n)
char *R_alloc(size_t, int);
void Rf_error(const char *, ...) __attribute__((noreturn));
#endif
