/* TEST-ONLY stand-in for the part of the R runtime that r_package/src/shim.c uses (see
 * Rinternals.h in this directory).  Built together with shim.c into a shared object that
 * tests/r_mock.py drives through ctypes, so that the shim's .Call routines really run -- slot
 * extraction, the Rle walk, NA handling, PROTECT discipline, result wrapping -- against the real
 * libbamsignals_hip.so.
 *
 * Faithful where it matters for the shim:
 *   - vectors carry type, length, data and an attribute list; S4 objects keep slots as attributes;
 *   - Rf_error() unwinds to the caller of the .Call (longjmp), after which the R_alloc arena of
 *     that call is released, as R does;
 *   - the protect stack is checked for balance at the end of every call;
 *   - at EVERY allocation inside a call, objects allocated during the call that are not reachable
 *     from the protect stack, from the call's arguments or from an object that is, are poisoned
 *     (gctorture-like); touching a poisoned object later aborts the call with a diagnostic. */
#include <setjmp.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "Rinternals.h"
#include "R_ext/Rdynload.h"

struct attr { SEXP tag, val; struct attr *next; };
struct SEXPREC {
    int type;
    R_xlen_t len;
    void *data;              /* int / double / SEXP array, char bytes */
    struct attr *attrs;
    int epoch;               /* the call during which it was allocated (0: by the harness) */
    int mark, dead;
    struct SEXPREC *all_next;
};

static struct SEXPREC nil_rec = {NILSXP, 0, NULL, NULL, 0, 0, 0, NULL};
SEXP R_NilValue = &nil_rec;
SEXP R_DimSymbol, R_DimNamesSymbol, R_LevelsSymbol;

static SEXP all_objs = NULL;
static int cur_epoch = 0;
static SEXP pstack[4096];
static int ptop = 0;
static SEXP call_args[16];
static int n_call_args = 0;
static jmp_buf jb;
static int in_call = 0;
static char err_msg[2048];
static void *arena[65536];
static int n_arena = 0;
static int n_poisoned_total = 0;
/* payload bytes of the vectors allocated during the current / last .Call, and the bytes it took from R_alloc:
 * tests check that a result is held ONCE (the vectors) and not a second time in a flat staging buffer */
static long long call_vec_bytes = 0, call_ralloc_bytes = 0;

static void die(const char *why)
{
    snprintf(err_msg, sizeof err_msg, "MOCK-R VIOLATION: %s", why);
    if (in_call) longjmp(jb, 2);
    fprintf(stderr, "%s\n", err_msg);
    abort();
}

static SEXP chk(SEXP x)
{
    if (!x) die("NULL SEXP dereferenced");
    if (x->dead) die("use of an object that was not protected when an allocation happened (would be garbage-collected in R)");
    return x;
}

static SEXP new_obj(int type, R_xlen_t len, size_t elt);
static void mark(SEXP x)
{
    if (!x || x == R_NilValue || x->mark) return;
    x->mark = 1;
    for (struct attr *a = x->attrs; a; a = a->next) { mark(a->tag); mark(a->val); }
    if (x->type == VECSXP || x->type == STRSXP)
        for (R_xlen_t i = 0; i < x->len; ++i) mark(((SEXP *)x->data)[i]);
}

/* what a garbage collection at this point would keep: everything older than the current call, the
 * call's arguments, the protect stack, and what those reach */
static void torture(void)
{
    if (!in_call) return;
    for (SEXP o = all_objs; o; o = o->all_next) o->mark = 0;
    for (int i = 0; i < ptop; ++i) mark(pstack[i]);
    for (int i = 0; i < n_call_args; ++i) mark(call_args[i]);
    for (SEXP o = all_objs; o; o = o->all_next)
        if (o->epoch == cur_epoch && !o->mark && !o->dead && o->type != SYMSXP) {
            o->dead = 1;
            ++n_poisoned_total;
            if (o->data && o->type != VECSXP && o->type != STRSXP) memset(o->data, 0xAB, 1);
        }
}

static SEXP new_obj(int type, R_xlen_t len, size_t elt)
{
    torture();
    SEXP o = (SEXP)calloc(1, sizeof *o);
    o->type = type;
    o->len = len;
    o->data = calloc((size_t)(len > 0 ? len : 1), elt ? elt : 1);
    if (in_call && (type == INTSXP || type == REALSXP || type == LGLSXP)) call_vec_bytes += (long long)(len > 0 ? len : 0) * (long long)elt;
    o->epoch = in_call ? cur_epoch : 0;
    o->all_next = all_objs;
    all_objs = o;
    if (type == VECSXP || type == STRSXP)
        for (R_xlen_t i = 0; i < len; ++i) ((SEXP *)o->data)[i] = R_NilValue;
    return o;
}

/* ---- the API shim.c uses ------------------------------------------------------------------- */
SEXP Rf_install(const char *name)
{
    for (SEXP o = all_objs; o; o = o->all_next)
        if (o->type == SYMSXP && !strcmp((const char *)o->data, name)) return o;
    const int was = in_call;
    in_call = 0;                                   /* symbols are never collected */
    SEXP s = new_obj(SYMSXP, (R_xlen_t)strlen(name) + 1, 1);
    in_call = was;
    strcpy((char *)s->data, name);
    s->epoch = 0;
    return s;
}

SEXP Rf_getAttrib(SEXP x, SEXP tag)
{
    chk(x);
    for (struct attr *a = x->attrs; a; a = a->next)
        if (a->tag == tag) return chk(a->val);
    return R_NilValue;
}

SEXP Rf_setAttrib(SEXP x, SEXP tag, SEXP val)
{
    chk(x); chk(val);
    for (struct attr *a = x->attrs; a; a = a->next)
        if (a->tag == tag) { a->val = val; return val; }
    struct attr *a = (struct attr *)calloc(1, sizeof *a);
    a->tag = tag; a->val = val; a->next = x->attrs;
    x->attrs = a;
    return val;
}

SEXP R_do_slot(SEXP obj, SEXP name)
{
    chk(obj);
    for (struct attr *a = obj->attrs; a; a = a->next)
        if (a->tag == name) return chk(a->val);
    Rf_error("no slot of name \"%s\" for this object", (const char *)name->data);
}

int TYPEOF(SEXP x) { return chk(x)->type; }
R_xlen_t XLENGTH(SEXP x) { return chk(x)->len; }
int *INTEGER(SEXP x)
{
    chk(x);
    if (x->type != INTSXP && x->type != LGLSXP) die("INTEGER() of a non-integer vector");
    return (int *)x->data;
}
SEXP VECTOR_ELT(SEXP x, R_xlen_t i)
{
    chk(x);
    if (x->type != VECSXP) die("VECTOR_ELT() of a non-list");
    if (i < 0 || i >= x->len) die("VECTOR_ELT() index out of range");
    return chk(((SEXP *)x->data)[i]);
}
SEXP SET_VECTOR_ELT(SEXP x, R_xlen_t i, SEXP v)
{
    chk(x); chk(v);
    if (x->type != VECSXP) die("SET_VECTOR_ELT() of a non-list");
    if (i < 0 || i >= x->len) die("SET_VECTOR_ELT() index out of range");
    ((SEXP *)x->data)[i] = v;
    return v;
}
SEXP STRING_ELT(SEXP x, R_xlen_t i)
{
    chk(x);
    if (x->type != STRSXP) die("STRING_ELT() of a non-character vector");
    if (i < 0 || i >= x->len) die("STRING_ELT() index out of range");
    return chk(((SEXP *)x->data)[i]);
}
void SET_STRING_ELT(SEXP x, R_xlen_t i, SEXP v)
{
    chk(x); chk(v);
    if (x->type != STRSXP || v->type != CHARSXP) die("SET_STRING_ELT() type error");
    if (i < 0 || i >= x->len) die("SET_STRING_ELT() index out of range");
    ((SEXP *)x->data)[i] = v;
}
const char *CHAR(SEXP x)
{
    chk(x);
    if (x->type != CHARSXP) die("CHAR() of a non-CHARSXP");
    return (const char *)x->data;
}
SEXP Rf_mkChar(const char *s)
{
    SEXP c = new_obj(CHARSXP, (R_xlen_t)strlen(s), 1);
    free(c->data);
    c->data = strdup(s);
    return c;
}
SEXP Rf_allocVector(unsigned int type, R_xlen_t n)
{
    if (n < 0) Rf_error("negative length vectors are not allowed");
    switch (type) {
    case INTSXP: case LGLSXP: return new_obj((int)type, n, sizeof(int));
    case REALSXP: return new_obj(REALSXP, n, sizeof(double));
    case STRSXP: case VECSXP: return new_obj((int)type, n, sizeof(SEXP));
    default: die("Rf_allocVector: type not modelled");
    }
    return R_NilValue;
}
SEXP Rf_allocMatrix(unsigned int type, int nrow, int ncol)
{
    SEXP m = Rf_protect(Rf_allocVector(type, (R_xlen_t)nrow * ncol));
    SEXP d = Rf_allocVector(INTSXP, 2);
    INTEGER(d)[0] = nrow; INTEGER(d)[1] = ncol;
    Rf_setAttrib(m, R_DimSymbol, d);
    Rf_unprotect(1);
    return m;
}
SEXP Rf_coerceVector(SEXP x, unsigned int type)
{
    chk(x);
    if ((unsigned)x->type == type) return x;
    if (type != INTSXP) die("Rf_coerceVector: only -> INTSXP is modelled");
    SEXP out = Rf_allocVector(INTSXP, x->len);
    for (R_xlen_t i = 0; i < x->len; ++i) {
        if (x->type == REALSXP) {
            const double v = ((double *)x->data)[i];
            ((int *)out->data)[i] = (v != v || v >= 2147483648.0 || v <= -2147483649.0) ? NA_INTEGER : (int)v;   /* truncation, as R */
        } else if (x->type == LGLSXP) {
            ((int *)out->data)[i] = ((int *)x->data)[i];
        } else {
            Rf_error("cannot coerce this type to integer");
        }
    }
    return out;
}
SEXP Rf_ScalarLogical(int v)
{
    SEXP x = new_obj(LGLSXP, 1, sizeof(int));
    ((int *)x->data)[0] = v == NA_LOGICAL ? NA_LOGICAL : (v != 0);
    return x;
}
int Rf_asInteger(SEXP x)
{
    chk(x);
    if (x->len < 1) return NA_INTEGER;
    if (x->type == INTSXP || x->type == LGLSXP) return ((int *)x->data)[0];
    if (x->type == REALSXP) {
        const double v = ((double *)x->data)[0];
        return (v != v || v >= 2147483648.0 || v <= -2147483649.0) ? NA_INTEGER : (int)v;
    }
    return NA_INTEGER;
}
int Rf_asLogical(SEXP x)
{
    chk(x);
    if (x->len < 1) return NA_LOGICAL;
    if (x->type == LGLSXP || x->type == INTSXP) {
        const int v = ((int *)x->data)[0];
        return v == NA_INTEGER ? NA_LOGICAL : (v != 0);
    }
    if (x->type == REALSXP) {
        const double v = ((double *)x->data)[0];
        return v != v ? NA_LOGICAL : (v != 0);
    }
    return NA_LOGICAL;
}
Rboolean Rf_inherits(SEXP x, const char *name)
{
    chk(x);
    SEXP k = Rf_getAttrib(x, Rf_install("class"));
    if (k == R_NilValue || k->type != STRSXP) return FALSE;
    for (R_xlen_t i = 0; i < k->len; ++i)
        if (!strcmp(CHAR(STRING_ELT(k, i)), name)) return TRUE;
    return FALSE;
}
SEXP Rf_protect(SEXP x)
{
    chk(x);
    if (ptop >= 4096) die("protect stack overflow");
    pstack[ptop++] = x;
    return x;
}
void Rf_unprotect(int n)
{
    if (n < 0 || n > ptop) die("UNPROTECT of more than is protected");
    ptop -= n;
}
char *R_alloc(size_t n, int size)
{
    if (n_arena >= 65536) die("too many R_alloc blocks");
    void *p = calloc(n ? n : 1, (size_t)(size > 0 ? size : 1));
    call_ralloc_bytes += (long long)n * (long long)(size > 0 ? size : 1);
    arena[n_arena++] = p;
    return (char *)p;
}
void Rf_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_msg, sizeof err_msg, fmt, ap);
    va_end(ap);
    if (!in_call) { fprintf(stderr, "Rf_error outside a call: %s\n", err_msg); abort(); }
    longjmp(jb, 1);
}
static const R_CallMethodDef *registered = NULL;
static int dyn_symbols = -1;
int R_registerRoutines(DllInfo *info, const void *c, const R_CallMethodDef *call, const void *f, const void *e)
{
    (void)info; (void)c; (void)f; (void)e;
    registered = call;
    return 1;
}
Rboolean R_useDynamicSymbols(DllInfo *info, Rboolean v) { (void)info; dyn_symbols = v; return TRUE; }

/* ---- harness API (tests/r_mock.py) ---------------------------------------------------------- */
void R_init_bamsignals(DllInfo *);
void mock_init(void)
{
    if (R_DimSymbol) return;
    R_DimSymbol = Rf_install("dim");
    R_DimNamesSymbol = Rf_install("dimnames");
    R_LevelsSymbol = Rf_install("levels");
    R_init_bamsignals(NULL);
}
int mock_n_registered(void) { int n = 0; while (registered && registered[n].name) ++n; return n; }
const char *mock_registered_name(int i) { return registered[i].name; }
int mock_registered_arity(int i) { return registered[i].numArgs; }
int mock_dynamic_symbols(void) { return dyn_symbols; }

SEXP mock_int(const int *v, R_xlen_t n) { SEXP x = Rf_allocVector(INTSXP, n); if (n) memcpy(x->data, v, (size_t)n * sizeof(int)); return x; }
SEXP mock_lgl(const int *v, R_xlen_t n) { SEXP x = Rf_allocVector(LGLSXP, n); if (n) memcpy(x->data, v, (size_t)n * sizeof(int)); return x; }
SEXP mock_real(const double *v, R_xlen_t n) { SEXP x = Rf_allocVector(REALSXP, n); if (n) memcpy(x->data, v, (size_t)n * sizeof(double)); return x; }
SEXP mock_str(const char *const *v, R_xlen_t n)
{
    SEXP x = Rf_allocVector(STRSXP, n);
    for (R_xlen_t i = 0; i < n; ++i) SET_STRING_ELT(x, i, Rf_mkChar(v[i]));
    return x;
}
SEXP mock_list(R_xlen_t n) { return Rf_allocVector(VECSXP, n); }
void mock_list_set(SEXP l, R_xlen_t i, SEXP v) { SET_VECTOR_ELT(l, i, v); }
SEXP mock_s4(void) { return new_obj(S4SXP, 0, 1); }
void mock_set_attr(SEXP x, const char *name, SEXP v) { Rf_setAttrib(x, Rf_install(name), v); }
SEXP mock_get_attr(SEXP x, const char *name) { return Rf_getAttrib(x, Rf_install(name)); }
SEXP mock_nil(void) { return R_NilValue; }
int mock_type(SEXP x) { return x->type; }
long long mock_len(SEXP x) { return (long long)x->len; }
const int *mock_int_data(SEXP x) { return (const int *)x->data; }
SEXP mock_list_get(SEXP x, R_xlen_t i) { return ((SEXP *)x->data)[i]; }
const char *mock_string(SEXP x, R_xlen_t i) { return (const char *)((SEXP *)x->data)[i]->data; }
const char *mock_error(void) { return err_msg; }
int mock_protect_depth(void) { return ptop; }
void mock_last_call_bytes(long long *vectors, long long *r_alloc) { *vectors = call_vec_bytes; *r_alloc = call_ralloc_bytes; }
int mock_poisoned(void) { return n_poisoned_total; }

/* Runs the registered routine `name` on args; returns its value, or NULL after an R error (1) or a
 * violation of the API's rules (2: *status), with the message in mock_error(). */
typedef SEXP (*fn2)(SEXP, SEXP);
typedef SEXP (*fn8)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
typedef SEXP (*fn11)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
SEXP mock_call(const char *name, SEXP *args, int n_args, int *status)
{
    *status = 0;
    err_msg[0] = 0;
    call_vec_bytes = call_ralloc_bytes = 0;
    const R_CallMethodDef *m = NULL;
    for (int i = 0; registered && registered[i].name; ++i)
        if (!strcmp(registered[i].name, name)) m = &registered[i];
    if (!m) { snprintf(err_msg, sizeof err_msg, "no registered routine %s", name); *status = 3; return NULL; }
    if (m->numArgs != n_args) { snprintf(err_msg, sizeof err_msg, "%s takes %d arguments", name, m->numArgs); *status = 3; return NULL; }
    ++cur_epoch;
    n_call_args = n_args;
    for (int i = 0; i < n_args; ++i) call_args[i] = args[i];
    const int depth = ptop;
    SEXP res = NULL;
    in_call = 1;
    const int j = setjmp(jb);
    if (j == 0) {
        DL_FUNC f = m->fun;
        if (n_args == 2) res = ((fn2)f)(args[0], args[1]);
        else if (n_args == 8) res = ((fn8)f)(args[0], args[1], args[2], args[3], args[4], args[5], args[6], args[7]);
        else if (n_args == 11) res = ((fn11)f)(args[0], args[1], args[2], args[3], args[4], args[5], args[6], args[7], args[8], args[9], args[10]);
        if (res) {
            /* the value is handed to R: it must have survived every allocation of the call */
            for (SEXP o = all_objs; o; o = o->all_next) o->mark = 0;
            mark(res);
            for (SEXP o = all_objs; o; o = o->all_next)
                if (o->mark && o->dead) { snprintf(err_msg, sizeof err_msg, "MOCK-R VIOLATION: the returned value holds an object that was unprotected during an allocation"); *status = 2; res = NULL; break; }
        }
        if (res && ptop != depth) {
            snprintf(err_msg, sizeof err_msg, "MOCK-R VIOLATION: protect stack imbalance (%d left)", ptop - depth);
            *status = 2;
            res = NULL;
        }
    } else {
        *status = j;          /* 1: Rf_error (R unwinds the protect stack itself), 2: violation */
        res = NULL;
    }
    in_call = 0;
    ptop = depth;
    for (int i = 0; i < n_arena; ++i) free(arena[i]);      /* R releases R_alloc memory at the end of .Call */
    n_arena = 0;
    n_call_args = 0;
    /* results are kept alive for the harness: move them to the "old" generation */
    if (res)
        for (SEXP o = all_objs; o; o = o->all_next)
            if (o->epoch == cur_epoch && !o->dead) o->epoch = 0;
    return res;
}
