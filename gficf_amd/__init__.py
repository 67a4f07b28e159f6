"""gficf_amd — MI355X (gfx950) GF-ICF normalisation and Phenograph kNN -> Jaccard edge build.

Drop-in for one hot path of the dibbelab/gficf R package: ``gficf()`` (sparse GF / ICF /
L2 scaling of a CSC genes x cells matrix) and the Jaccard step of ``clustcells()``
(``rcpp_parallel_jaccard_coef``), plus the exact neighbour search in front of it (``find_nn``).  All compute runs in hand-written HIP kernels behind the C
ABI of ``libgficf_hip.so`` (include/gficf_hip.h); there is no CPU fallback.
"""
from ._lib import GficfError, LIB_PATH  # noqa: F401
from .api import (  # noqa: F401
    Context,
    MultiContext,
    cluster_signatures,
    clustcells,
    clustcells_graph,
    find_nn,
    HipOps,
    default_context,
    device_count,
    gficf,
    gficf_with_weights,
    jaccard_adjacency,
    jaccard_coeff,
    jaccard_counts,
    jaccard_edges,
    jaccard_expand,
    phenograph,
    rcpp_parallel_jaccard_coef,
    run_modularity_clustering,
    transpose_gficf,
)

__version__ = "0.2.0"
