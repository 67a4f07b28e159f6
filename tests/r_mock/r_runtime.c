/* tests/r_mock/r_runtime.c — NOT R.  A small RUNNABLE stand-in for the part of R's C API that OUR glue file
 * (integration/gficf_hip_glue.c) uses, so that the `.Call` entry points can be executed — through the routine table they
 * register, by name and arity, as R's `.Call` would (reference: src/RcppExports.cpp:85-97) — in an image without R.
 * Test infrastructure only: never shipped, never linked into the product, nothing of the reference is built with it and it
 * pins nothing about R's behaviour.  What it does model, because the glue's correctness depends on it:
 *   * SEXP = a tagged heap object (type, length, payload, attribute list); vectors of INTSXP / REALSXP / RAWSXP / STRSXP /
 *     VECSXP, CHARSXP strings, interned symbols; dim / names / arbitrary attributes; S4-style slots read with R_do_slot
 *     (kept in the attribute list, as R does);
 *   * the PROTECT stack, with a *torture collector*: at EVERY allocation every object that is neither on the protect stack,
 *     nor an argument of the running call, nor reachable from one of those, is "collected" — its payload is poisoned and the
 *     object is marked dead; a dead object that the glue then touches or returns is reported.  This is gctorture(TRUE) for
 *     the glue's PROTECT discipline;
 *   * Rf_error: formats the message and longjmps back to the `.Call` trampoline (the protect stack is reset there, as R's
 *     context unwinding does); Rprintf: captured; R_alloc: released when the call ends;
 *   * R_registerRoutines: a DllInfo owns a COPY of the `.Call` table handed in, and a second call on the same DllInfo
 *     REPLACES it (R's src/main/Rdynload.c: R_registerRoutines allocates a new CallSymbols array and overwrites numCallSymbols —
 *     it does not append); R_useDynamicSymbols is recorded.  rmock_call() looks an entry up by name, checks the arity, calls it.
 *   * allocation failure on request (rmock_fail_allocation_after): the n-th allocation from now raises the R error R raises when
 *     memory runs out ("cannot allocate vector of size"), to test what a call leaves behind when it unwinds half-way.
 * The package's init function R_init_gficf (the glue defines it) is what rmock_init() runs, with stand-ins for the three Rcpp
 * wrappers that stay in the package (tests/r_mock/package_rows.c).
 * The Python side (tests/test_glue_run.py) builds arguments with rmock_new_* and reads results with rmock_* accessors. */
#include <setjmp.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <R.h>
#include <Rinternals.h>
#include <R_ext/Rdynload.h>

#define SYMSXP 1
#define CHARSXP 9
#define NILSXP 0
#define LGLSXP 10
#define S4SXP 25

struct attr_node { SEXP tag, val; struct attr_node* next; };
struct SEXPREC {
  int type;
  R_xlen_t len;
  void* data;               /* payload: int / double / Rbyte / SEXP elements, or the characters of a CHARSXP / symbol */
  struct attr_node* attrib;
  int dead;                 /* collected by the torture collector */
  int mark;
  int permanent;            /* symbols, R_NilValue, objects made by the test driver */
  struct SEXPREC* next_all; /* every object ever allocated (freed by rmock_reset) */
};

static struct SEXPREC nil_obj = {NILSXP, 0, NULL, NULL, 0, 0, 1, NULL};
SEXP R_NilValue = &nil_obj, R_DimSymbol = NULL, R_NamesSymbol = NULL;

static SEXP all_objects = NULL;
static SEXP protect_stack[10000];
static int protect_top = 0, protect_max = 0;
static SEXP call_args[16];
static int n_call_args = 0;
static int in_call = 0;
static int torture = 1;
static long n_collected = 0, n_dead_touched = 0, n_alloc = 0;
static jmp_buf error_jmp;
static char error_msg[2048];
static char print_buf[8192];
static size_t print_len = 0;
static void* ralloc_list[256];
static int n_ralloc = 0;
struct _DllInfo { R_CallMethodDef* call_symbols; int n_call_symbols; int use_dynamic_symbols; int n_registrations; };
static struct _DllInfo the_dll = {NULL, 0, -1, 0};
#define call_table (the_dll.call_symbols)
static long fail_alloc_in = 0;     /* > 0: the allocation that brings it to 0 fails */
static char driver_msg[512];

static size_t elt_size(int type) {
  switch (type) {
    case INTSXP: case LGLSXP: return sizeof(int);
    case REALSXP: return sizeof(double);
    case RAWSXP: return 1;
    case STRSXP: case VECSXP: return sizeof(SEXP);
    default: return 1;
  }
}

static void touch(SEXP s, const char* who) {
  if (s && s->dead) {
    ++n_dead_touched;
    snprintf(driver_msg, sizeof(driver_msg), "%s on an object the collector had already taken (missing PROTECT)", who);
  }
}

/* mark everything reachable from s */
static void mark_from(SEXP s) {
  if (!s || s->mark) return;
  s->mark = 1;
  for (struct attr_node* a = s->attrib; a; a = a->next) { mark_from(a->tag); mark_from(a->val); }
  if ((s->type == VECSXP || s->type == STRSXP) && s->data)
    for (R_xlen_t i = 0; i < s->len; ++i) mark_from(((SEXP*)s->data)[i]);
}

/* gctorture: everything a real collection could free at this point is poisoned */
static void collect(void) {
  if (!torture || !in_call) return;
  for (SEXP o = all_objects; o; o = o->next_all) o->mark = 0;
  for (int i = 0; i < protect_top; ++i) mark_from(protect_stack[i]);
  for (int i = 0; i < n_call_args; ++i) mark_from(call_args[i]);
  for (SEXP o = all_objects; o; o = o->next_all) {
    if (o->mark || o->permanent || o->dead) continue;
    o->dead = 1;
    ++n_collected;
    if (o->data && o->type != SYMSXP) memset(o->data, 0xAB, (size_t)o->len * elt_size(o->type) + (o->type == CHARSXP ? 1 : 0));
  }
}

static SEXP new_obj(int type, R_xlen_t len) {
  if (in_call && fail_alloc_in > 0 && --fail_alloc_in == 0) Rf_error("cannot allocate vector of size %.1f Mb", (double)len * (double)elt_size(type) / 1048576.0);
  collect();
  SEXP s = (SEXP)calloc(1, sizeof(struct SEXPREC));
  s->type = type;
  s->len = len;
  const size_t bytes = (size_t)(len > 0 ? len : 0) * elt_size(type) + (type == CHARSXP || type == SYMSXP ? 1 : 0);
  s->data = bytes ? calloc(1, bytes) : NULL;
  if ((type == VECSXP || type == STRSXP) && s->data)
    for (R_xlen_t i = 0; i < len; ++i) ((SEXP*)s->data)[i] = R_NilValue;
  s->permanent = !in_call;          /* objects the test driver makes outside a call are the caller's (R would hold them) */
  s->next_all = all_objects;
  all_objects = s;
  ++n_alloc;
  return s;
}

/* ------------------------------------------------------------------------------------------------ the API the glue uses */
int TYPEOF(SEXP s) { touch(s, "TYPEOF"); return s->type; }
int* INTEGER(SEXP s) { touch(s, "INTEGER"); return (int*)s->data; }
double* REAL(SEXP s) { touch(s, "REAL"); return (double*)s->data; }
Rbyte* RAW(SEXP s) { touch(s, "RAW"); return (Rbyte*)s->data; }
const char* CHAR(SEXP s) { touch(s, "CHAR"); return (const char*)s->data; }
SEXP STRING_ELT(SEXP s, R_xlen_t i) { touch(s, "STRING_ELT"); return ((SEXP*)s->data)[i]; }
R_xlen_t XLENGTH(SEXP s) { touch(s, "XLENGTH"); return s->len; }
int Rf_length(SEXP s) { touch(s, "Rf_length"); return (int)s->len; }
Rboolean Rf_isNull(SEXP s) { return s == R_NilValue || s->type == NILSXP; }

SEXP Rf_install(const char* name) {
  for (SEXP o = all_objects; o; o = o->next_all)
    if (o->type == SYMSXP && strcmp((const char*)o->data, name) == 0) return o;
  const int was = in_call;
  in_call = 0;                       /* symbols are permanent; installing one does not collect */
  SEXP s = new_obj(SYMSXP, (R_xlen_t)strlen(name));
  in_call = was;
  strcpy((char*)s->data, name);
  s->permanent = 1;
  return s;
}

SEXP Rf_getAttrib(SEXP s, SEXP tag) {
  touch(s, "Rf_getAttrib");
  for (struct attr_node* a = s->attrib; a; a = a->next)
    if (a->tag == tag) return a->val;
  return R_NilValue;
}
SEXP Rf_setAttrib(SEXP s, SEXP tag, SEXP val) {
  touch(s, "Rf_setAttrib");
  touch(val, "Rf_setAttrib (value)");
  for (struct attr_node* a = s->attrib; a; a = a->next)
    if (a->tag == tag) { a->val = val; return val; }
  struct attr_node* a = (struct attr_node*)calloc(1, sizeof(*a));
  a->tag = tag; a->val = val; a->next = s->attrib;
  s->attrib = a;
  return val;
}
SEXP R_do_slot(SEXP s, SEXP name) { return Rf_getAttrib(s, name); }

Rboolean Rf_isMatrix(SEXP s) {
  SEXP d = Rf_getAttrib(s, R_DimSymbol);
  return (d != R_NilValue && d->type == INTSXP && d->len == 2) ? TRUE : FALSE;
}

SEXP Rf_allocVector(unsigned int type, R_xlen_t n) {
  if (n < 0) Rf_error("negative length vectors are not allowed");
  return new_obj((int)type, n);
}
SEXP Rf_allocMatrix(unsigned int type, int nrow, int ncol) {
  if (nrow < 0 || ncol < 0) Rf_error("negative extents to matrix");
  SEXP m = Rf_protect(new_obj((int)type, (R_xlen_t)nrow * ncol));
  SEXP d = new_obj(INTSXP, 2);
  ((int*)d->data)[0] = nrow; ((int*)d->data)[1] = ncol;
  Rf_setAttrib(m, R_DimSymbol, d);
  Rf_unprotect(1);
  return m;
}
SEXP Rf_protect(SEXP s) {
  touch(s, "PROTECT");
  if (protect_top >= 10000) Rf_error("protect(): protection stack overflow");
  protect_stack[protect_top++] = s;
  if (protect_top > protect_max) protect_max = protect_top;
  return s;
}
void Rf_unprotect(int n) {
  if (n > protect_top) {
    snprintf(driver_msg, sizeof(driver_msg), "unprotect(%d) with %d on the stack: stack imbalance", n, protect_top);
    protect_top = 0;
    return;
  }
  protect_top -= n;
}
int Rf_asLogical(SEXP s) {
  touch(s, "Rf_asLogical");
  if (s->len < 1) return INT32_MIN;
  if (s->type == LGLSXP || s->type == INTSXP) return ((int*)s->data)[0] != 0;
  if (s->type == REALSXP) return ((double*)s->data)[0] != 0.0;
  return INT32_MIN;
}
int Rf_asInteger(SEXP s) {
  touch(s, "Rf_asInteger");
  if (s->len < 1) return INT32_MIN;
  if (s->type == LGLSXP || s->type == INTSXP) return ((int*)s->data)[0];
  if (s->type == REALSXP) return (int)((double*)s->data)[0];
  return INT32_MIN;
}
double Rf_asReal(SEXP s) {
  touch(s, "Rf_asReal");
  if (s->len < 1) return 0.0 / 0.0;
  if (s->type == LGLSXP || s->type == INTSXP) return (double)((int*)s->data)[0];
  if (s->type == REALSXP) return ((double*)s->data)[0];
  return 0.0 / 0.0;
}
SEXP Rf_ScalarReal(double v) {
  SEXP s = new_obj(REALSXP, 1);
  ((double*)s->data)[0] = v;
  return s;
}
SEXP Rf_mkChar(const char* str) {
  SEXP s = new_obj(CHARSXP, (R_xlen_t)strlen(str));
  strcpy((char*)s->data, str);
  return s;
}
SEXP SET_VECTOR_ELT(SEXP v, R_xlen_t i, SEXP x) {
  touch(v, "SET_VECTOR_ELT");
  touch(x, "SET_VECTOR_ELT (element)");
  ((SEXP*)v->data)[i] = x;
  return x;
}
void SET_STRING_ELT(SEXP v, R_xlen_t i, SEXP x) {
  touch(v, "SET_STRING_ELT");
  touch(x, "SET_STRING_ELT (element)");
  ((SEXP*)v->data)[i] = x;
}
void Rprintf(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  if (print_len < sizeof(print_buf) - 1) {
    const int n = vsnprintf(print_buf + print_len, sizeof(print_buf) - print_len, fmt, ap);
    if (n > 0) print_len += (size_t)n < sizeof(print_buf) - print_len ? (size_t)n : sizeof(print_buf) - print_len - 1;
  }
  va_end(ap);
}
char* R_alloc(size_t n, int size) {
  if (n_ralloc >= 256) Rf_error("R_alloc: too many allocations in one call (mock limit)");
  void* p = calloc(n ? n : 1, (size_t)(size > 0 ? size : 1));
  ralloc_list[n_ralloc++] = p;
  return (char*)p;
}
void Rf_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(error_msg, sizeof(error_msg), fmt, ap);
  va_end(ap);
  if (!in_call) { fprintf(stderr, "r_mock: Rf_error outside a call: %s\n", error_msg); abort(); }
  longjmp(error_jmp, 1);
}
int R_registerRoutines(DllInfo* dll, const R_CMethodDef* c, const R_CallMethodDef* call, const R_FortranMethodDef* f,
                       const R_ExternalMethodDef* e) {
  (void)c; (void)f; (void)e;
  if (!dll) return 0;
  int n = 0;
  while (call && call[n].name) ++n;
  /* a NEW array replaces whatever was registered before (what R does; the old one is simply dropped there) */
  R_CallMethodDef* t = (R_CallMethodDef*)calloc((size_t)n + 1, sizeof(*t));
  for (int i = 0; i < n; ++i) t[i] = call[i];
  free(dll->call_symbols);
  dll->call_symbols = t;
  dll->n_call_symbols = n;
  ++dll->n_registrations;
  return 1;
}
int R_useDynamicSymbols(DllInfo* dll, int v) {
  if (!dll) return 0;
  const int old = dll->use_dynamic_symbols;
  dll->use_dynamic_symbols = v;
  return old;
}

/* --------------------------------------------------------------------------------------------- the test driver's side */
void R_init_gficf(DllInfo* dll);              /* the package's init function: defined by the glue */
void R_unload_gficf(DllInfo* dll);

/* what dyn.load() + the package's init do: the DLL's own R_init_<pkg> runs once on a fresh DllInfo */
void rmock_init(void) {
  if (!R_DimSymbol) { R_DimSymbol = Rf_install("dim"); R_NamesSymbol = Rf_install("names"); }
  free(the_dll.call_symbols);
  the_dll.call_symbols = NULL; the_dll.n_call_symbols = 0; the_dll.use_dynamic_symbols = -1; the_dll.n_registrations = 0;
  R_init_gficf(&the_dll);
}
void rmock_unload(void) { R_unload_gficf(&the_dll); }
int rmock_use_dynamic_symbols(void) { return the_dll.use_dynamic_symbols; }
int rmock_n_registrations(void) { return the_dll.n_registrations; }
void rmock_fail_allocation_after(long n) { fail_alloc_in = n; }

/* Self-test of the replace semantics: on a scratch DllInfo, registering table A (2 rows) and then table B (1 row) leaves B alone —
 * the recipe "call R_registerRoutines once more next to Rcpp's" loses a table.  Returns the number of rows that resolve at the end
 * (1), or -1 when a row of A still does. */
static SEXP selftest_fn(void) { return R_NilValue; }
int rmock_selftest_second_registration_replaces(void) {
  struct _DllInfo d = {NULL, 0, -1, 0};
  const R_CallMethodDef A[] = {{"a1", (DL_FUNC)&selftest_fn, 0}, {"a2", (DL_FUNC)&selftest_fn, 0}, {NULL, NULL, 0}};
  const R_CallMethodDef B[] = {{"b1", (DL_FUNC)&selftest_fn, 0}, {NULL, NULL, 0}};
  R_registerRoutines(&d, NULL, A, NULL, NULL);
  R_registerRoutines(&d, NULL, B, NULL, NULL);
  int n = 0, a_left = 0;
  for (int i = 0; i < d.n_call_symbols; ++i) { ++n; a_left |= d.call_symbols[i].name[0] == 'a'; }
  free(d.call_symbols);
  return a_left ? -1 : n;
}
void rmock_set_torture(int on) { torture = on; }
int rmock_n_routines(void) {
  int n = 0;
  while (call_table && call_table[n].name) ++n;
  return n;
}
const char* rmock_routine_name(int i) { return call_table[i].name; }
int rmock_routine_nargs(int i) { return call_table[i].numArgs; }

SEXP rmock_new_vector(int type, long len) { return new_obj(type, (R_xlen_t)len); }
SEXP rmock_new_matrix(int type, int nrow, int ncol) {
  SEXP m = new_obj(type, (R_xlen_t)nrow * ncol);
  SEXP d = new_obj(INTSXP, 2);
  ((int*)d->data)[0] = nrow; ((int*)d->data)[1] = ncol;
  Rf_setAttrib(m, R_DimSymbol, d);
  return m;
}
SEXP rmock_nil(void) { return R_NilValue; }
SEXP rmock_new_string(const char* s) {
  SEXP v = new_obj(STRSXP, 1);
  ((SEXP*)v->data)[0] = Rf_mkChar(s);
  return v;
}
void rmock_set_slot(SEXP obj, const char* name, SEXP val) { Rf_setAttrib(obj, Rf_install(name), val); }
SEXP rmock_get_attr(SEXP obj, const char* name) { return Rf_getAttrib(obj, Rf_install(name)); }
void* rmock_data(SEXP s) { return s->data; }
long rmock_length(SEXP s) { return (long)s->len; }
int rmock_type(SEXP s) { return s->type; }
int rmock_is_dead(SEXP s) { return s->dead; }
SEXP rmock_elt(SEXP s, long i) { return ((SEXP*)s->data)[i]; }
const char* rmock_chars(SEXP s) { return (const char*)s->data; }
int rmock_dim(SEXP s, int which) {
  SEXP d = Rf_getAttrib(s, R_DimSymbol);
  return d == R_NilValue ? -1 : ((int*)d->data)[which];
}
const char* rmock_error_message(void) { return error_msg; }
const char* rmock_printed(void) { print_buf[print_len] = 0; return print_buf; }
const char* rmock_driver_message(void) { return driver_msg; }
int rmock_protect_depth(void) { return protect_top; }
int rmock_protect_max(void) { return protect_max; }
long rmock_collected(void) { return n_collected; }
long rmock_dead_touched(void) { return n_dead_touched; }
long rmock_allocations(void) { return n_alloc; }

/* `.Call(name, args...)` through the registered table.  Returns 0 = the routine returned (result in *out),
 * 1 = it raised an R error (message: rmock_error_message), 2 = no such routine, 3 = wrong number of arguments,
 * 4 = the routine returned with the protect stack unbalanced, 5 = it returned a collected object. */
int rmock_call(const char* name, int nargs, SEXP* args, SEXP* out, int* protect_depth_at_exit) {
  *out = R_NilValue;
  error_msg[0] = 0; print_len = 0; driver_msg[0] = 0; protect_max = 0;
  const R_CallMethodDef* volatile ev = call_table;
  while (ev && ev->name && strcmp(ev->name, name) != 0) ++ev;
  const R_CallMethodDef* const e = ev;
  if (!e || !e->name) return 2;
  if (e->numArgs != nargs || nargs > 9) return 3;
  typedef SEXP (*f0)(void); typedef SEXP (*f1)(SEXP); typedef SEXP (*f2)(SEXP, SEXP); typedef SEXP (*f3)(SEXP, SEXP, SEXP);
  typedef SEXP (*f4)(SEXP, SEXP, SEXP, SEXP); typedef SEXP (*f5)(SEXP, SEXP, SEXP, SEXP, SEXP);
  typedef SEXP (*f6)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP); typedef SEXP (*f7)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
  typedef SEXP (*f8)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
  typedef SEXP (*f9)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
  SEXP* a = args;
  n_call_args = nargs;
  for (int i = 0; i < nargs; ++i) call_args[i] = args[i];
  const int base = protect_top;
  volatile int status = 0;
  SEXP volatile res = R_NilValue;
  in_call = 1;
  if (setjmp(error_jmp) == 0) {
    switch (nargs) {
      case 0: res = ((f0)e->fun)(); break;
      case 1: res = ((f1)e->fun)(a[0]); break;
      case 2: res = ((f2)e->fun)(a[0], a[1]); break;
      case 3: res = ((f3)e->fun)(a[0], a[1], a[2]); break;
      case 4: res = ((f4)e->fun)(a[0], a[1], a[2], a[3]); break;
      case 5: res = ((f5)e->fun)(a[0], a[1], a[2], a[3], a[4]); break;
      case 6: res = ((f6)e->fun)(a[0], a[1], a[2], a[3], a[4], a[5]); break;
      case 7: res = ((f7)e->fun)(a[0], a[1], a[2], a[3], a[4], a[5], a[6]); break;
      case 8: res = ((f8)e->fun)(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7]); break;
      default: res = ((f9)e->fun)(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8]); break;
    }
  } else {
    status = 1;                                   /* Rf_error: R unwinds to the caller's context */
  }
  in_call = 0;
  *protect_depth_at_exit = protect_top - base;
  if (status == 0 && protect_top != base) status = 4;
  protect_top = base;                             /* (what R's context restore does after an error) */
  for (int i = 0; i < n_ralloc; ++i) free(ralloc_list[i]);
  n_ralloc = 0;
  n_call_args = 0;
  if (status == 0) {
    SEXP r = res;
    if (r && r->dead) status = 5;
    *out = r;
    /* the result now belongs to the caller: it and what hangs off it survive later calls */
    for (SEXP o = all_objects; o; o = o->next_all) o->mark = 0;
    mark_from(r);
    for (SEXP o = all_objects; o; o = o->next_all) if (o->mark) o->permanent = 1;
  }
  return status;
}

/* Self-test of the torture collector: a routine that forgets a PROTECT must be caught.  Allocates a vector without protecting
 * it, allocates a second one (a collection point), then reads the first.  Returns 1 when the first was collected, poisoned
 * and the read was reported. */
int rmock_selftest_missing_protect(void) {
  const long before = n_dead_touched;
  in_call = 1;
  SEXP a = Rf_allocVector(INTSXP, 4);
  INTEGER(a)[0] = 7;
  SEXP b = Rf_protect(Rf_allocVector(INTSXP, 4));
  const int poisoned = a->dead && (unsigned)INTEGER(a)[0] == 0xABABABABu;
  Rf_unprotect(1);
  in_call = 0;
  (void)b;
  driver_msg[0] = 0;
  return poisoned && n_dead_touched > before;
}

/* frees every object (results included); the registration and the symbols stay */
void rmock_reset(void) {
  SEXP keep = NULL;
  for (SEXP o = all_objects; o;) {
    SEXP nx = o->next_all;
    if (o->type == SYMSXP) { o->next_all = keep; keep = o; }
    else {
      for (struct attr_node* a = o->attrib; a;) { struct attr_node* an = a->next; free(a); a = an; }
      free(o->data);
      free(o);
    }
    o = nx;
  }
  all_objects = keep;
  n_collected = n_dead_touched = n_alloc = 0;
}
