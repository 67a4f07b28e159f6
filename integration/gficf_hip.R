# gficf_hip.R — R side of the drop-in (replaces the bodies of tf/getIdfW/idf/l.norm in gficf();
# reference R/gficf.R:17-33).  Signature, defaults and the returned list fields are unchanged.

# The R stubs of the two replaced entries.  In the reference they are generated into R/RcppExports.R:8-10,16-18 from the `// [[Rcpp::export]]`
# tags of src/jaccard_coeff.cpp and src/rcpp_parallel_jaccard_coeff.cpp; those files are deleted, compileAttributes() stops generating
# the stubs, and clustcells() (R/clustCells.R:65) still calls the second — so they live here, with the same names and arguments.
rcpp_parallel_jaccard_coef <- function(mat, printOutput) .Call(`_gficf_rcpp_parallel_jaccard_coef`, mat, printOutput)
jaccard_coeff <- function(idx, printOutput) .Call(`_gficf_jaccard_coeff`, idx, printOutput)

# A dgCMatrix whatever sparse or dense class came in: `as(M, "CsparseMatrix")` alone keeps a pattern matrix (ngCMatrix: no @x slot at all)
# or a logical / integer one (lgCMatrix, igCMatrix: @x of another type) as it is, and the entry reads @x as double.  The three-step
# coercion is the one Matrix >= 1.5 asks for in place of the deprecated as(M, "dgCMatrix"); symmetric / triangular inputs become general.
as_dgCMatrix_hip = function(M)
{
  if (!inherits(M, "dgCMatrix")) M = methods::as(methods::as(methods::as(M, "dMatrix"), "generalMatrix"), "CsparseMatrix")
  M
}

gficf = function(M, cell_proportion_max = 1, cell_proportion_min = 0.05, storeRaw = TRUE, normalize = TRUE, verbose = TRUE)
{
  data = list()
  M = as_dgCMatrix_hip(M)
  # The reference's five progress lines (tsmessage, R/util.R:30-39), same text: "Normalize counts.." R/gficf.R:45 (only when normalize),
  # "Apply GF transformation.." :58, "Compute ICF weigth.." :87, "Applay ICF.." :68, "Apply l2" :99.  The four GF-ICF steps are ONE device
  # call here, so their lines precede it together; "Normalize counts.." is printed where the edgeR call actually runs (after the call:
  # it needs the filtered matrix, and only $rawCounts is affected) — the one line that moves relative to the reference's log.
  tsmessage("Apply GF transformation..", verbose = verbose)
  tsmessage("Compute ICF weigth..", verbose = verbose)
  tsmessage("Applay ICF..", verbose = verbose)
  tsmessage(paste("Apply", "l2"), verbose = verbose)
  r = if (storeRaw) .Call(`_gficf_gficf_csc_raw`, M@i, M@p, M@x, M@Dim, NULL, cell_proportion_min, cell_proportion_max)
      else .Call(`_gficf_gficf_csc`, M@i, M@p, M@x, M@Dim, NULL, cell_proportion_min, cell_proportion_max)
  keep = as.logical(r[[4]])
  dn = list(rownames(M)[keep], colnames(M))
  data$gficf = Matrix::sparseMatrix(i = r[[1]], p = r[[2]], x = r[[3]], index1 = FALSE, dims = c(sum(keep), ncol(M)), dimnames = dn)
  if (storeRaw) {
    # normCounts' M[keep, ] (reference R/gficf.R:40): the structure of the result with the counts as values — r[[7]], gathered by the
    # library while the result came back; @i and @p are the SAME vectors as $gficf's (R copies on modify), no subsetting in R
    raw = Matrix::sparseMatrix(i = r[[1]], p = r[[2]], x = r[[7]], index1 = FALSE, dims = c(sum(keep), ncol(M)), dimnames = dn)
    # the edgeR branch (reference R/gficf.R:43-47) only rescales what is stored here: a per-cell scale
    # cancels in x/colSums(x), so $gficf is the same with or without it
    if (normalize) {
      tsmessage("Normalize counts..", verbose = verbose)
      raw = Matrix::Matrix(edgeR::cpm(edgeR::calcNormFactors(edgeR::DGEList(counts = raw), normalized.lib.sizes = T)), sparse = TRUE)
    }
    data$rawCounts = raw
  }
  data$w = stats::setNames(r[[6]][keep], rownames(M)[keep])
  data$param <- list(cell_proportion_max = cell_proportion_max, cell_proportion_min = cell_proportion_min, normalized = normalize)
  return(data)
}

# embedNewCells() keeps its name matching (reference R/gficf.R:69-78) in R and calls
#   x = as_dgCMatrix_hip(x); .Call(`_gficf_gficf_csc`, x@i, x@p, x@x, x@Dim, as.numeric(data$w[rownames(x)]), 0, 2)
# in place of tf() / idf() / l.norm() (reference R/cellClassifier.R:50-53).

# Optional ("next" row N2): exact neighbour search for clustcells().  Replaces the two lines
#   neigh = uwot:::find_nn(data$pca$cells, k = k+1, include_self = T, n_threads = nt, verbose = verbose,
#                          method = "annoy", metric = dist.method)$idx          (reference R/clustCells.R:57,60)
# with
#   neigh = find_nn_hip(data$pca$cells, k = k+1, metric = dist.method)$idx
# Same result shape (idx: N x (k+1) integer matrix, 1-based, first column the cell itself; dist).  Exact, not
# approximate: ties broken by the smaller index; f32 arithmetic like Annoy's.
find_nn_hip = function(X, k, metric = "manhattan")
{
  m = match(metric, c("manhattan", "euclidean", "cosine", "correlation")) - 1L
  if (is.na(m)) stop("metric must be manhattan, euclidean, cosine or correlation")
  .Call(`_gficf_find_nn`, as.matrix(X) + 0, as.integer(k), m)
}

# Optional (second half of N1): the adjacency matrix for RunModularityClustering ("louvian 2" / "louvian 3").  Replaces
#   igraph::as_adjacency_matrix(g, attr = "weight", sparse = T)                       (reference R/clustCells.R:80,86)
# with
#   jaccard_adjacency_hip(relations, nrow(data$pca$cells))
jaccard_adjacency_hip = function(relations, n)
{
  r = .Call(`_gficf_jaccard_adjacency`, as.numeric(relations$from), as.numeric(relations$to), as.numeric(relations$weight), n)
  Matrix::sparseMatrix(i = r[[1]], p = r[[2]], x = r[[3]], index1 = FALSE, dims = c(n, n))
}

# Optional (N3): cluster signatures.  Replaces
#   u = base::unique(cluster.map)
#   data$cluster.gene.rnk = base::sapply(u, function(x,y=data$gficf,z=cluster.map) Matrix::rowSums(y[,z%in%x]))
#                                                                               (reference R/clustCells.R:121-123)
# with
#   data$cluster.gene.rnk = cluster_signatures_hip(data$gficf, cluster.map)
cluster_signatures_hip = function(M, cluster.map)
{
  u = base::unique(cluster.map)
  M = as_dgCMatrix_hip(M)
  r = .Call(`_gficf_cluster_signatures`, M@i, M@p, M@x, M@Dim, match(cluster.map, u) - 1L, length(u))
  rownames(r) = rownames(M)
  colnames(r) = u          # sapply(u, ...) names the columns by cluster label (reference R/clustCells.R:122-123)
  r
}

# Optional (N3): the PCA input.  Replaces
#   data$pca$cells = t(data$gficf)                                   (reference R/dimensinalityReduction.R:33, :100)
# with
#   data$pca$cells = transpose_hip(data$gficf)
transpose_hip = function(M)
{
  M = as_dgCMatrix_hip(M)
  r = .Call(`_gficf_transpose_csc`, M@i, M@p, M@x, M@Dim)
  Matrix::sparseMatrix(i = r[[1]], p = r[[2]], x = r[[3]], index1 = FALSE, dims = rev(M@Dim), dimnames = rev(M@Dimnames))
}

# Optional (N4): community detection.  Same arguments as the reference's RunModularityClustering (R/clustCells.R:145-149); in
# clustcells() the calls at R/clustCells.R:80 and :86 become
#   community <- RunModularityClusteringHip(igraph::as_adjacency_matrix(g, attr = "weight", sparse = T), 1, resolution, 1, n.start, n.iter, seed, verbose)
# (or with jaccard_adjacency_hip(relations, n) as the matrix).  A deterministic parallel Louvain on the same modularity
# (resolution as in the reference, diagonal ignored): labels are NOT those of the seeded sequential optimiser, the
# modularity is (tested) within 0.01 of it — the spread the reference itself shows between seeds.  n.start starts are run and the best kept; random.seed seeds the one arbitrary
# choice of the device algorithm (how the vertices are split into sub-rounds), so equal arguments give equal results.
RunModularityClusteringHip <- function(SNN = matrix(), modularity = 1, resolution = 0.8, algorithm = 1, n.start = 10, n.iter = 10,
                                       random.seed = 0, print.output = TRUE, temp.file.location = NULL, edge.file.name = "")
{
  SNN = as_dgCMatrix_hip(SNN)
  .Call(`_gficf_RunModularityClusteringHip`, SNN, as.integer(modularity), resolution, as.integer(algorithm), as.integer(n.start),
        as.integer(n.iter), as.integer(random.seed), print.output, edge.file.name)
}

# Optional: lines 57-86 of clustcells() in one call (search, neigh[,-1], Jaccard edges, weight > 0, adjacency matrix, Louvain)
# with nothing crossing PCIe between the steps; for store.graph = FALSE, when only the communities are wanted:
#   community = phenograph_hip(data$pca$cells, k, dist.method, resolution, 1, n.start, n.iter, seed) + 1
phenograph_hip = function(X, k = 15, metric = "manhattan", resolution = 0.8, algorithm = 1, n.start = 10, n.iter = 10, seed = 0)
{
  m = match(metric, c("manhattan", "euclidean", "cosine", "correlation")) - 1L
  if (is.na(m)) stop("metric must be manhattan, euclidean, cosine or correlation")
  .Call(`_gficf_phenograph`, as.matrix(X) + 0, as.integer(k), m, resolution, as.integer(algorithm), as.integer(n.start),
        as.integer(n.iter), as.integer(seed))
}

# Optional: clustcells() with every step on the device.  Same arguments and the same fields set on `data` as the reference's
# clustcells() (R/clustCells.R:46-126); only the Seurat-optimiser choices of community.algo are served ("louvian" is run
# as plain Louvain, resolution 1: the reference calls igraph::cluster_louvain there), the igraph / leidenalg ones are not.
#   store.graph = TRUE : find_nn_hip -> rcpp_parallel_jaccard_coef (the replaced entry) -> weight > 0 -> jaccard_adjacency_hip
#                        -> RunModularityClusteringHip; data$cell.graph is built with igraph as before
#   store.graph = FALSE: phenograph_hip, one call, nothing crossing PCIe between the steps
clustcells_hip = function(data, from.embedded = F, k = 15, dist.method = "manhattan", nt = 2, community.algo = "louvian 2",
                          store.graph = T, seed = 180582, verbose = TRUE, resolution = 0.8, n.start = 10, n.iter = 10)
{
  community.algo = base::match.arg(arg = community.algo, choices = c("louvian", "louvian 2", "louvian 3"), several.ok = F)
  if (from.embedded) {
    if (is.null(data$embedded)) {stop("First run runReduction to embed your cells")}
    X = as.matrix(data$embedded[, c(1, 2)])
  } else {
    if (is.null(data$pca)) {stop("First run runPCA or runLSA to reduce dimensionality")}
    X = as.matrix(data$pca$cells)
  }
  algorithm = if (community.algo == "louvian 3") 2L else 1L
  if (community.algo == "louvian") {resolution = 1; n.start = 1; seed = 0}
  if (store.graph) {
    neigh = find_nn_hip(X, k + 1, dist.method)$idx[, -1]
    relations = rcpp_parallel_jaccard_coef(neigh, verbose)
    relations = as.data.frame(relations[relations[, 3] > 0, ])
    colnames(relations) = c("from", "to", "weight")
    adjacency = jaccard_adjacency_hip(relations, nrow(X))
    community = RunModularityClusteringHip(adjacency, 1, resolution, algorithm, n.start, n.iter, seed, verbose) + 1
    data$cell.graph = igraph::graph.data.frame(relations, directed = FALSE)
  } else {
    community = as.integer(phenograph_hip(X, k, dist.method, resolution, algorithm, n.start, n.iter, seed)) + 1L
  }
  data$community = community
  data$embedded$cluster = as.character(community)
  data$cluster.gene.rnk = cluster_signatures_hip(data$gficf, data$embedded$cluster)
  tsmessage(paste("Detected Clusters:", length(unique(data$embedded$cluster))), verbose = verbose)
  return(data)
}
