/* TEST-ONLY: see ../Rinternals.h. */
#ifndef BSIG_TEST_RDYNLOAD_STUB_H
#define BSIG_TEST_RDYNLOAD_STUB_H
typedef void *(*DL_FUNC)(void);
typedef struct { const char *name; DL_FUNC fun; int numArgs; } R_CallMethodDef;
typedef struct _DllInfo DllInfo;
int R_registerRoutines(DllInfo *, const void *, const R_CallMethodDef *, const void *, const void *);
Rboolean R_useDynamicSymbols(DllInfo *, Rboolean);
#endif
