/* tests/r_mock/package_rows.c — NOT the reference's routines.  Stand-ins for the three Rcpp wrappers that STAY in the gficf package when
 * the glue is dropped in (reference src/RcppExports.cpp:17,48,72: `_gficf_RunModularityClusteringCpp` 9 arguments, `_gficf_rcpp_WMU_test` 3,
 * `_gficf_rcpp_parallel_WMU_test` 3), so that the glue's R_init_gficf — which declares them `extern` and registers them in the ONE table
 * next to its own rows — links and runs in tests/test_glue_run.py.  Each returns an integer vector holding its own arity: the test calls
 * them by name through the registered table and checks that the RIGHT function sits behind every name. */
#include <R.h>
#include <Rinternals.h>

static SEXP arity(int n) {
  SEXP s = PROTECT(Rf_allocVector(INTSXP, 1));
  INTEGER(s)[0] = n;
  UNPROTECT(1);
  return s;
}
SEXP _gficf_RunModularityClusteringCpp(SEXP a, SEXP b, SEXP c, SEXP d, SEXP e, SEXP f, SEXP g, SEXP h, SEXP i) {
  (void)a; (void)b; (void)c; (void)d; (void)e; (void)f; (void)g; (void)h; (void)i;
  return arity(9009);
}
SEXP _gficf_rcpp_WMU_test(SEXP a, SEXP b, SEXP c) { (void)a; (void)b; (void)c; return arity(3003); }
SEXP _gficf_rcpp_parallel_WMU_test(SEXP a, SEXP b, SEXP c) { (void)a; (void)b; (void)c; return arity(3103); }
