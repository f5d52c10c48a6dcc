/* TEST-ONLY declarations of the handful of R C-API names r_package/src/shim.c uses, so that the
 * shim can be compiled in an image that has no R.  Two uses (tests/test_r_shim.py):
 *   - a type check of shim.c with gcc -fsyntax-only -Werror;
 *   - tests/r_stub/r_mock.c implements these names as a SMALL STAND-IN RUNTIME (tagged vectors,
 *     attributes, S4 slots, a protect stack with an allocation-time reachability check in the
 *     spirit of gctorture, R_alloc arenas, Rf_error as a longjmp), so that the shim's .Call routines
 *     can be EXECUTED against the real libbamsignals_hip.so from Python.
 * This is not R and nothing shipped is built against it: the real build is `R CMD INSTALL` against
 * the real <Rinternals.h>. */
#ifndef BSIG_TEST_RINTERNALS_STUB_H
#define BSIG_TEST_RINTERNALS_STUB_H
#include <limits.h>
#include <stddef.h>
typedef struct SEXPREC *SEXP;
typedef ptrdiff_t R_xlen_t;
typedef int Rboolean;
#ifndef TRUE
#define TRUE 1
#define FALSE 0
#endif
#define NA_INTEGER INT_MIN
#define NA_LOGICAL INT_MIN
enum { NILSXP = 0, SYMSXP = 1, CHARSXP = 9, LGLSXP = 10, INTSXP = 13, REALSXP = 14, STRSXP = 16, VECSXP = 19, S4SXP = 25 };
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
