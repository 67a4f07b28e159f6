/* gficf_hip_glue.c — R `.Call` glue over libgficf_hip.so (C ABI: include/gficf_hip.h).
 *
 * R and <Rinternals.h> are absent from the build image, so this file is never linked here; it IS syntax- and
 * type-checked against clearly-labelled mock declarations (tests/r_mock/, tests/test_glue_compile.py), which pins
 * nothing about R's behaviour.  A maintainer of the gficf R package drops this file into src/, removes
 * src/rcpp_parallel_jaccard_coeff.cpp, deletes the `_gficf_rcpp_parallel_jaccard_coef` wrapper and
 * its CallEntries row from the generated src/RcppExports.cpp (reference :59-70, :89) and adds
 *   PKG_CPPFLAGS += -I$(GFICF_HIP_HOME)/include
 *   PKG_LIBS     += -L$(GFICF_HIP_HOME)/gficf_amd -lgficf_hip -Wl,-rpath,$(GFICF_HIP_HOME)/gficf_amd
 * to src/Makevars (reference src/Makevars:5-12).  Plain C, no Rcpp dependency.
 *
 * Symbols
 *   _gficf_rcpp_parallel_jaccard_coef(mat, printOutput)  — SAME name/arity as the reference entry
 *       (src/RcppExports.cpp:61): R/RcppExports.R:16-18 and clustcells() (R/clustCells.R:65) stay as is.
 *   _gficf_jaccard_coeff(idx, printOutput)               — SAME name/arity as the reference's serial entry
 *       (src/RcppExports.cpp:36): set intersection, rows with u > 0 packed from the top.
 *   _gficf_gficf_csc(i, p, x, dim, w, min, max)          — NEW entry for the GF-ICF chain of
 *       R/gficf.R:38-105 (the reference has no native entry on that path).
 *   _gficf_gficf_csc_raw(i, p, x, dim, w, min, max)      — the same + the values of M[keep, ] ($rawCounts, R/gficf.R:40,22).
 *   _gficf_find_nn(X, k, metric)                         — NEW, optional: exact kNN in place of the
 *       uwot:::find_nn(..., method = "annoy") call of clustcells() (R/clustCells.R:57,60).
 *   _gficf_jaccard_adjacency(from, to, weight, n)        — NEW, optional: igraph::as_adjacency_matrix (R/clustCells.R:80,86).
 *   _gficf_cluster_signatures(i, p, x, dim, cluster, C)  — NEW, optional: data$cluster.gene.rnk (R/clustCells.R:121-123).
 *   _gficf_transpose_csc(i, p, x, dim)                   — NEW, optional: t(data$gficf) (R/dimensinalityReduction.R:33,100).
 *   _gficf_RunModularityClusteringHip(SNN, ...9 args)    — OPTIONAL, same arguments as the reference's
 *       _gficf_RunModularityClusteringCpp (src/RcppExports.cpp:17): the deterministic parallel Louvain.  NOT registered
 *       under the reference's name: its results differ from the seeded sequential optimiser (same objective, see
 *       include/gficf_hip.h), so the switch is an explicit choice in R/clustCells.R:80,86, not a silent replacement.
 *   _gficf_phenograph(X, k, metric, resolution, algorithm, n.start, n.iter, seed) — OPTIONAL: lines 57-86 of clustcells()
 *       (search, Jaccard, filter, adjacency, Louvain) chained on the device in one call.
 */
#include <R.h>
#include <Rinternals.h>
#include <R_ext/Rdynload.h>
#include <limits.h>
#include <stdint.h>
#include <string.h>

#include "gficf_hip.h"

static gficf_ctx* g_ctx = NULL;
static gficf_multi* g_multi = NULL;      /* GFICF_HIP_DEVICES names more than one GPU */
static int g_multi_tried = 0;

static void print_line(const char* line) { Rprintf("%s", line); }   /* the reference's banners go through Rprintf */

static gficf_ctx* ctx_get(void) {
  if (!g_ctx) {
    int dev = 0;
    const char* e = getenv("GFICF_HIP_DEVICE");
    if (e) dev = atoi(e);
    if (gficf_ctx_create(dev, NULL, &g_ctx) != GFICF_OK) Rf_error("gficf_hip: %s", gficf_last_error());
    gficf_ctx_set_print(g_ctx, print_line);
    /* GFICF_HIP_TRUNCATE_IDS=1: non-integer double ids are truncated as the reference does (`int k = mat(i,j) - 1`,
     * src/rcpp_parallel_jaccard_coeff.cpp:28) instead of being rejected */
    e = getenv("GFICF_HIP_TRUNCATE_IDS");
    if (e && atoi(e) != 0) gficf_ctx_set_jaccard_options(g_ctx, 1);
  }
  return g_ctx;
}

/* GFICF_HIP_DEVICES="0,1,2,3": the Jaccard build and gficf() shard their cells over these GPUs (single process,
 * gficf_multi_* of the C ABI).  Unset, or one device: the single-device entries. */
static gficf_multi* multi_get(void) {
  if (!g_multi_tried) {
    const char* e = getenv("GFICF_HIP_DEVICES");
    int devs[64], n = 0;
    while (e && *e && n < 64) {
      char* end = NULL;
      const long v = strtol(e, &end, 10);
      if (end == e) break;
      devs[n++] = (int)v;
      e = end;
      while (*e == ',' || *e == ';' || *e == ' ') ++e;
    }
    if (n > 1) {
      /* on failure (say, an ordinal that does not exist) the error repeats at every call: g_multi_tried stays 0, so
       * a session that asked for several GPUs never runs single-device in silence (Rf_error does not return) */
      if (gficf_multi_create(devs, n, &g_multi) != GFICF_OK) Rf_error("gficf_hip: GFICF_HIP_DEVICES: %s", gficf_last_error());
      gficf_multi_set_print(g_multi, print_line);
    }
    g_multi_tried = 1;
  }
  return g_multi;
}

/* rows of the (N*k) x 3 result: R matrices hold at most INT_MAX rows (Rf_allocMatrix takes int) */
static int edge_rows(int64_t N, int k) {
  if (k > 0 && N > (int64_t)INT_MAX / k) Rf_error("gficf_hip: N * k = %.0f rows exceed what an R matrix holds", (double)N * (double)k);
  return (int)(N * k);
}

/* replaces src/RcppExports.cpp:61-70 + src/rcpp_parallel_jaccard_coeff.cpp:59-80 */
SEXP _gficf_rcpp_parallel_jaccard_coef(SEXP matSEXP, SEXP printOutputSEXP) {
  if (!Rf_isMatrix(matSEXP) || !(TYPEOF(matSEXP) == INTSXP || TYPEOF(matSEXP) == REALSXP))
    Rf_error("mat must be an integer or numeric matrix");
  SEXP dim = Rf_getAttrib(matSEXP, R_DimSymbol);
  const int64_t N = INTEGER(dim)[0];
  const int k = INTEGER(dim)[1];
  const int is_f64 = TYPEOF(matSEXP) == REALSXP;          /* uwot returns INTSXP; no coercion copy needed */
  const void* idx = is_f64 ? (const void*)REAL(matSEXP) : (const void*)INTEGER(matSEXP);
  SEXP rmat = PROTECT(Rf_allocMatrix(REALSXP, edge_rows(N, k), 3));   /* reference :67 */
  gficf_multi* m = multi_get();
  int rc = m ? gficf_jaccard_host_multi(m, idx, is_f64, N, k, N, REAL(rmat), Rf_asLogical(printOutputSEXP))
             : gficf_jaccard_host(ctx_get(), idx, is_f64, N, k, N, REAL(rmat), Rf_asLogical(printOutputSEXP));
  if (rc != GFICF_OK) {
    UNPROTECT(1);
    Rf_error("gficf_hip: %s", gficf_last_error());         /* BEGIN_RCPP/END_RCPP equivalent */
  }
  UNPROTECT(1);
  return rmat;
}

/* replaces src/RcppExports.cpp:36-45 + src/jaccard_coeff.cpp:19-44 (the serial entry; same name and arity) */
SEXP _gficf_jaccard_coeff(SEXP idxSEXP, SEXP printOutputSEXP) {
  if (!Rf_isMatrix(idxSEXP) || !(TYPEOF(idxSEXP) == INTSXP || TYPEOF(idxSEXP) == REALSXP))
    Rf_error("idx must be an integer or numeric matrix");
  SEXP dim = Rf_getAttrib(idxSEXP, R_DimSymbol);
  const int64_t N = INTEGER(dim)[0];
  const int k = INTEGER(dim)[1];
  const int is_f64 = TYPEOF(idxSEXP) == REALSXP;
  const void* idx = is_f64 ? (const void*)REAL(idxSEXP) : (const void*)INTEGER(idxSEXP);
  SEXP weights = PROTECT(Rf_allocMatrix(REALSXP, edge_rows(N, k), 3));
  if (gficf_jaccard_coeff_host(ctx_get(), idx, is_f64, N, k, N, REAL(weights), Rf_asLogical(printOutputSEXP)) != GFICF_OK) {
    UNPROTECT(1);
    Rf_error("gficf_hip: %s", gficf_last_error());
  }
  UNPROTECT(1);
  return weights;
}

/* list(i, p, x, keep, nt, w) for gficf() / embedNewCells(); w = NULL -> compute ICF weights */
static SEXP gficf_csc_call(SEXP iS, SEXP pS, SEXP xS, SEXP dimS, SEXP wS, SEXP minS, SEXP maxS, int raw) {
  const int64_t G = INTEGER(dimS)[0], N = INTEGER(dimS)[1];
  const double* w_in = Rf_isNull(wS) ? NULL : REAL(wS);
  int64_t gk = 0, nk = 0;
  gficf_multi* m = multi_get();
  const int prc = m ? gficf_normalize_csc_host_multi_plan(m, G, N, INTEGER(pS), 0, INTEGER(iS), REAL(xS), Rf_asReal(minS), Rf_asReal(maxS), w_in, &gk, &nk)
                    : gficf_normalize_csc_host_plan(ctx_get(), G, N, INTEGER(pS), 0, INTEGER(iS), REAL(xS), Rf_asReal(minS), Rf_asReal(maxS), w_in, &gk, &nk);
  if (prc != GFICF_OK) Rf_error("gficf_hip: %s", gficf_last_error());
  if (nk > (int64_t)INT_MAX) Rf_error("gficf_hip: %.0f kept entries exceed a dgCMatrix (its @p is integer)", (double)nk);
  SEXP out = PROTECT(Rf_allocVector(VECSXP, raw ? 7 : 6));
  SEXP oi = PROTECT(Rf_allocVector(INTSXP, nk)), op = PROTECT(Rf_allocVector(INTSXP, N + 1));
  SEXP ox = PROTECT(Rf_allocVector(REALSXP, nk)), keep = PROTECT(Rf_allocVector(RAWSXP, G));
  SEXP w = PROTECT(Rf_allocVector(REALSXP, G));
  SEXP rx = PROTECT(raw ? Rf_allocVector(REALSXP, nk) : R_NilValue);      /* the values of M[keep, ]: $rawCounts shares @i and @p with $gficf */
  int64_t* nt = (int64_t*)R_alloc((size_t)G, sizeof(int64_t));
  int frc;
  if (m) {
    frc = gficf_normalize_csc_host_multi_finish(m, RAW(keep), nt, REAL(w), INTEGER(op), INTEGER(oi), REAL(ox));
    if (frc == GFICF_OK && raw)
      frc = gficf_csc_kept_values_host(G, N, INTEGER(pS), 0, INTEGER(iS), REAL(xS), RAW(keep), INTEGER(op), NULL, REAL(rx));
  } else if (raw) {
    frc = gficf_normalize_csc_host_finish_raw(ctx_get(), RAW(keep), nt, REAL(w), INTEGER(op), INTEGER(oi), REAL(ox), INTEGER(iS), REAL(xS), NULL, REAL(rx));
  } else {
    frc = gficf_normalize_csc_host_finish(ctx_get(), RAW(keep), nt, REAL(w), INTEGER(op), INTEGER(oi), REAL(ox));
  }
  if (frc != GFICF_OK) {
    UNPROTECT(7);
    Rf_error("gficf_hip: %s", gficf_last_error());
  }
  SEXP ntS = PROTECT(Rf_allocVector(REALSXP, G));
  for (int64_t g = 0; g < G; ++g) REAL(ntS)[g] = (double)nt[g];
  SET_VECTOR_ELT(out, 0, oi); SET_VECTOR_ELT(out, 1, op); SET_VECTOR_ELT(out, 2, ox);
  SET_VECTOR_ELT(out, 3, keep); SET_VECTOR_ELT(out, 4, ntS); SET_VECTOR_ELT(out, 5, w);
  if (raw) SET_VECTOR_ELT(out, 6, rx);
  UNPROTECT(8);
  return out;
}

SEXP _gficf_gficf_csc(SEXP iS, SEXP pS, SEXP xS, SEXP dimS, SEXP wS, SEXP minS, SEXP maxS) {
  return gficf_csc_call(iS, pS, xS, dimS, wS, minS, maxS, 0);
}

/* The same with a seventh list element: the x of the kept entries, i.e. the values of normCounts' `M[keep, ]` (reference R/gficf.R:40) in the
 * structure of the result — gficf(storeRaw = TRUE) builds $rawCounts from (r[[1]], r[[2]], r[[7]]) instead of subsetting M in R. */
SEXP _gficf_gficf_csc_raw(SEXP iS, SEXP pS, SEXP xS, SEXP dimS, SEXP wS, SEXP minS, SEXP maxS) {
  return gficf_csc_call(iS, pS, xS, dimS, wS, minS, maxS, 1);
}

/* Optional: exact neighbour search in place of the approximate uwot:::find_nn(..., method = "annoy") call of
 * clustcells() (reference R/clustCells.R:57,60).  X: numeric N x d matrix; returns list(idx = N x k integer
 * matrix (1-based, column 1 = the cell itself), dist = N x k numeric) like uwot's result.
 * metric: 0 manhattan, 1 euclidean, 2 cosine, 3 correlation. */
SEXP _gficf_find_nn(SEXP XS, SEXP kS, SEXP metricS) {
  if (!Rf_isMatrix(XS) || TYPEOF(XS) != REALSXP) Rf_error("X must be a numeric matrix");
  SEXP dim = Rf_getAttrib(XS, R_DimSymbol);
  const int64_t N = INTEGER(dim)[0];
  const int d = INTEGER(dim)[1], k = Rf_asInteger(kS);
  SEXP idx = PROTECT(Rf_allocMatrix(INTSXP, (int)N, k)), dist = PROTECT(Rf_allocMatrix(REALSXP, (int)N, k));
  if (gficf_knn_host(ctx_get(), REAL(XS), N, d, N, k, Rf_asInteger(metricS), INTEGER(idx), REAL(dist)) != GFICF_OK) {
    UNPROTECT(2);
    Rf_error("gficf_hip: %s", gficf_last_error());
  }
  SEXP out = PROTECT(Rf_allocVector(VECSXP, 2)), nm = PROTECT(Rf_allocVector(STRSXP, 2));
  SET_VECTOR_ELT(out, 0, idx); SET_VECTOR_ELT(out, 1, dist);
  SET_STRING_ELT(nm, 0, Rf_mkChar("idx")); SET_STRING_ELT(nm, 1, Rf_mkChar("dist"));
  Rf_setAttrib(out, R_NamesSymbol, nm);
  UNPROTECT(4);
  return out;
}

/* Optional: the symmetric weighted adjacency matrix of the undirected graph, in place of
 *   igraph::as_adjacency_matrix(igraph::graph.data.frame(relations, directed = FALSE), attr = "weight", sparse = T)
 * (reference R/clustCells.R:69,80,86).  Returns list(i, p, x) for Matrix::sparseMatrix(i=, p=, x=, index1 = FALSE). */
SEXP _gficf_jaccard_adjacency(SEXP fromS, SEXP toS, SEXP weightS, SEXP nS) {
  const int64_t E = XLENGTH(fromS), N = (int64_t)Rf_asReal(nS);
  int64_t nnz = 0;
  if (gficf_adjacency_host_plan(ctx_get(), N, E, REAL(fromS), REAL(toS), REAL(weightS), &nnz) != GFICF_OK)
    Rf_error("gficf_hip: %s", gficf_last_error());
  SEXP out = PROTECT(Rf_allocVector(VECSXP, 3));
  SEXP oi = PROTECT(Rf_allocVector(INTSXP, nnz)), op = PROTECT(Rf_allocVector(INTSXP, N + 1)), ox = PROTECT(Rf_allocVector(REALSXP, nnz));
  if (gficf_adjacency_host_finish(ctx_get(), INTEGER(op), 0, INTEGER(oi), REAL(ox)) != GFICF_OK) {
    UNPROTECT(4);
    Rf_error("gficf_hip: %s", gficf_last_error());
  }
  SET_VECTOR_ELT(out, 0, oi); SET_VECTOR_ELT(out, 1, op); SET_VECTOR_ELT(out, 2, ox);
  UNPROTECT(4);
  return out;
}

/* Optional: data$cluster.gene.rnk = sapply(unique(cluster), function(x) rowSums(gficf[, cluster %in% x]))
 * (reference R/clustCells.R:121-123).  cluster: integer ids 0..C-1 numbered in base::unique order by the R stub.
 * Returns the G x C numeric matrix. */
SEXP _gficf_cluster_signatures(SEXP iS, SEXP pS, SEXP xS, SEXP dimS, SEXP clS, SEXP CS) {
  const int64_t G = INTEGER(dimS)[0], N = INTEGER(dimS)[1];
  const int C = Rf_asInteger(CS);
  SEXP out = PROTECT(Rf_allocMatrix(REALSXP, (int)G, C));
  if (gficf_cluster_signatures_host(ctx_get(), G, N, INTEGER(pS), 0, INTEGER(iS), REAL(xS), INTEGER(clS), C, REAL(out)) != GFICF_OK) {
    UNPROTECT(1);
    Rf_error("gficf_hip: %s", gficf_last_error());
  }
  UNPROTECT(1);
  return out;
}

/* Optional: data$pca$cells = t(data$gficf) (reference R/dimensinalityReduction.R:33, :100).
 * Returns list(i, p, x) of the cells x genes dgCMatrix. */
SEXP _gficf_transpose_csc(SEXP iS, SEXP pS, SEXP xS, SEXP dimS) {
  const int64_t G = INTEGER(dimS)[0], N = INTEGER(dimS)[1];
  const int64_t nnz = INTEGER(pS)[N];
  SEXP out = PROTECT(Rf_allocVector(VECSXP, 3));
  SEXP oi = PROTECT(Rf_allocVector(INTSXP, nnz)), op = PROTECT(Rf_allocVector(INTSXP, G + 1)), ox = PROTECT(Rf_allocVector(REALSXP, nnz));
  int64_t* ptr = (int64_t*)R_alloc((size_t)G + 1, sizeof(int64_t));
  if (gficf_csc_transpose_host(ctx_get(), G, N, INTEGER(pS), 0, INTEGER(iS), REAL(xS), ptr, INTEGER(oi), REAL(ox)) != GFICF_OK) {
    UNPROTECT(4);
    Rf_error("gficf_hip: %s", gficf_last_error());
  }
  for (int64_t g = 0; g <= G; ++g) INTEGER(op)[g] = (int)ptr[g];
  SET_VECTOR_ELT(out, 0, oi); SET_VECTOR_ELT(out, 1, op); SET_VECTOR_ELT(out, 2, ox);
  UNPROTECT(4);
  return out;
}

/* Optional: community detection on the adjacency matrix, argument list of RunModularityClusteringCpp
 * (reference src/RModularityOptimizer.cpp:25-33).  SNN: a dgCMatrix (symmetric).  nRandomStarts starts are run and the best kept; randomSeed seeds the
 * one arbitrary choice there is (see include/gficf_hip.h); an edge file is not read.  Returns the 0-based cluster of every vertex, clusters by
 * decreasing size, as the reference does (:171-173). */
SEXP _gficf_RunModularityClusteringHip(SEXP SNN, SEXP modularityFunctionS, SEXP resolutionS, SEXP algorithmS, SEXP nRandomStartsS,
                                       SEXP nIterationsS, SEXP randomSeedS, SEXP printOutputS, SEXP edgefilenameS) {
  const int modularityFunction = Rf_asInteger(modularityFunctionS);
  if (modularityFunction != 1 && modularityFunction != 2) Rf_error("Modularity parameter must be equal to 1 or 2.");
  if (modularityFunction == 2 && Rf_asReal(resolutionS) > 1.0) Rf_error("error: resolution<1 for alternative modularity");
  const int algorithm = Rf_asInteger(algorithmS);
  if (algorithm != 1 && algorithm != 2) Rf_error("Algorithm for modularity optimization must be 1 or 2 on this path");
  if (Rf_asInteger(nRandomStartsS) < 1) Rf_error("Have to have at least one start");
  if (Rf_asInteger(nIterationsS) < 1) Rf_error("Need at least one interation");
  if (Rf_length(edgefilenameS) > 0 && CHAR(STRING_ELT(edgefilenameS, 0))[0] != 0) Rf_error("edge files are not read on this path");
  SEXP iS = R_do_slot(SNN, Rf_install("i")), pS = R_do_slot(SNN, Rf_install("p")), xS = R_do_slot(SNN, Rf_install("x"));
  SEXP dimS = R_do_slot(SNN, Rf_install("Dim"));
  const int64_t N = INTEGER(dimS)[0];
  if (INTEGER(dimS)[1] != N) Rf_error("SNN must be square");
  SEXP out = PROTECT(Rf_allocVector(INTSXP, N));
  int64_t n_clusters = 0;
  double q = 0.0;
  gficf_ctx_set_louvain_options(ctx_get(), modularityFunction);
  const int lrc = gficf_louvain_host(ctx_get(), N, INTEGER(pS), 0, INTEGER(iS), REAL(xS), Rf_asReal(resolutionS), algorithm, Rf_asInteger(nRandomStartsS),
                                     Rf_asInteger(nIterationsS), Rf_asInteger(randomSeedS) & 0x7FFFFFFF, INTEGER(out), &n_clusters, &q);
  gficf_ctx_set_louvain_options(ctx_get(), 1);
  if (lrc != GFICF_OK) {
    UNPROTECT(1);
    Rf_error("gficf_hip: %s", gficf_last_error());
  }
  if (Rf_asLogical(printOutputS)) {
    Rprintf("Number of nodes: %d\n", (int)N);
    Rprintf("Modularity: %.4f\n", q);
    Rprintf("Number of communities: %d\n", (int)n_clusters);
  }
  UNPROTECT(1);
  return out;
}

/* Optional: lines 57-86 of clustcells() in one call, nothing crossing PCIe between the steps.  X: numeric N x d matrix
 * (data$pca$cells); metric as in _gficf_find_nn.  Returns the 0-based cluster of every cell (clusters by decreasing size),
 * with attributes "modularity" and "n.edges". */
SEXP _gficf_phenograph(SEXP XS, SEXP kS, SEXP metricS, SEXP resolutionS, SEXP algorithmS, SEXP nRandomStartsS, SEXP nIterationsS,
                       SEXP randomSeedS) {
  if (!Rf_isMatrix(XS) || TYPEOF(XS) != REALSXP) Rf_error("X must be a numeric matrix");
  SEXP dim = Rf_getAttrib(XS, R_DimSymbol);
  const int64_t N = INTEGER(dim)[0];
  const int d = INTEGER(dim)[1];
  SEXP out = PROTECT(Rf_allocVector(INTSXP, N));
  int64_t n_clusters = 0, n_edges = 0;
  double q = 0.0;
  if (gficf_phenograph_host(ctx_get(), REAL(XS), N, d, N, Rf_asInteger(kS), Rf_asInteger(metricS), Rf_asReal(resolutionS),
                            Rf_asInteger(algorithmS), Rf_asInteger(nRandomStartsS), Rf_asInteger(nIterationsS),
                            Rf_asInteger(randomSeedS) & 0x7FFFFFFF, INTEGER(out), &n_clusters, &q, &n_edges) != GFICF_OK) {
    UNPROTECT(1);
    Rf_error("gficf_hip: %s", gficf_last_error());
  }
  /* symbols first (Rf_install may allocate), the fresh scalars protected while the attribute is set */
  SEXP sym_q = Rf_install("modularity"), sym_e = Rf_install("n.edges");
  SEXP vq = PROTECT(Rf_ScalarReal(q));
  Rf_setAttrib(out, sym_q, vq);
  SEXP ve = PROTECT(Rf_ScalarReal((double)n_edges));
  Rf_setAttrib(out, sym_e, ve);
  UNPROTECT(3);
  return out;
}

static const R_CallMethodDef HipCallEntries[] = {
    {"_gficf_rcpp_parallel_jaccard_coef", (DL_FUNC)&_gficf_rcpp_parallel_jaccard_coef, 2},
    {"_gficf_jaccard_coeff", (DL_FUNC)&_gficf_jaccard_coeff, 2},
    {"_gficf_gficf_csc", (DL_FUNC)&_gficf_gficf_csc, 7},
    {"_gficf_gficf_csc_raw", (DL_FUNC)&_gficf_gficf_csc_raw, 7},
    {"_gficf_find_nn", (DL_FUNC)&_gficf_find_nn, 3},
    {"_gficf_jaccard_adjacency", (DL_FUNC)&_gficf_jaccard_adjacency, 4},
    {"_gficf_cluster_signatures", (DL_FUNC)&_gficf_cluster_signatures, 6},
    {"_gficf_transpose_csc", (DL_FUNC)&_gficf_transpose_csc, 4},
    {"_gficf_RunModularityClusteringHip", (DL_FUNC)&_gficf_RunModularityClusteringHip, 9},
    {"_gficf_phenograph", (DL_FUNC)&_gficf_phenograph, 8},
    {NULL, NULL, 0}};

/* Called from the package's R_init_gficf (reference src/RcppExports.cpp:94-97) next to the Rcpp entries:
 *   R_registerRoutines(dll, NULL, HipCallEntries, NULL, NULL);                                     */
void gficf_hip_register(DllInfo* dll) { R_registerRoutines(dll, NULL, HipCallEntries, NULL, NULL); }

void R_unload_gficf(DllInfo* dll) {
  (void)dll;
  if (g_ctx) { gficf_ctx_destroy(g_ctx); g_ctx = NULL; }
  if (g_multi) { gficf_multi_destroy(g_multi); g_multi = NULL; }
  g_multi_tried = 0;
}
